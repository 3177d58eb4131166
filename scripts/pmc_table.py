"""Per-dispatch counter table from one rocprofv3 --pmc pass (counter_collection.csv): one line per GEMM dispatch,
one column per counter (summed over the rows rocprofv3 writes per dimension).
usage: pmc_table.py <dir> [kernel-name substring, default "gemm"]"""
import collections, csv, glob, sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
names = []
for r in csv.DictReader(open(f)):
    if (sys.argv[2] if len(sys.argv) > 2 else "gemm") not in r["Kernel_Name"]:
        continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].split("<")[0].split("(")[0].split("::")[-1],
                                                "wgs": int(r["Grid_Size"]) // int(r["Workgroup_Size"])})
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    d[c] = d.get(c, 0.0) + float(r["Counter_Value"])
print("dispatch kernel wgs " + " ".join(names))
for i, d in rows.items():
    print(f"{i:4d} {d['k']:28s} {d['wgs']:6d} " + " ".join(f"{d.get(c, 0):.4g}" for c in names))
