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
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    d = rows.setdefault(int(r["Dispatch_Id"]), {"k": name.split("(")[0].split("::")[-1][:28],
                                                "wgs": int(r["Grid_Size"]) // int(r["Workgroup_Size"])})
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    d[c] = d.get(c, 0.0) + float(r["Counter_Value"])
print("dispatch kernel wgs " + " ".join(names))
for i, d in rows.items():
    print(f"{i:4d} {d['k']:28s} {d['wgs']:6d} " + " ".join(f"{d.get(c, 0):.4g}" for c in names))

# per kernel (median dispatch): shares of wave lifetime, and the matrix pipes' share of the launch
if "SQ_WAVE_CYCLES" in names and "GRBM_GUI_ACTIVE" in names:
    by = collections.OrderedDict()
    for d in rows.values():
        by.setdefault((d["k"], d["wgs"]), []).append(d)
    print()
    print("kernel, workgroups: us at 2.1 GHz | instructions issued / VALU / waiting, as shares of wave lifetime | matrix pipes busy, share of the launch (1024 SIMDs)")
    for (k, w), ds in by.items():
        d = sorted(ds, key=lambda x: x["GRBM_GUI_ACTIVE"])[len(ds) // 2]
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        wc = d["SQ_WAVE_CYCLES"]
        print(f"{k:28s} {w:6d}: {cyc / 2100:7.1f} us | {d.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.2f} {d.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.2f} "
              f"{d.get('SQ_WAIT_ANY', 0) / wc:5.2f} | {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):5.2f}")
