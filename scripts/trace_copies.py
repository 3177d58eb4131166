"""Where do the runtime's copy kernels of a rocprofv3 kernel trace come from?  Count by (grid, previous kernel, next kernel)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in csv.DictReader(open(f))))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:38]
agg = collections.Counter()
for i, (s, e, n, gx) in enumerate(rows):
    if "copyBuffer" in n:
        prev = short(rows[i - 1][2]) if i else "-"
        nxt = short(rows[i + 1][2]) if i + 1 < len(rows) else "-"
        agg[(gx, prev, nxt, (e - s) // 1000)] += 1
for (gx, prev, nxt, us), c in agg.most_common(25):
    print(f"n={c:5d} grid={gx:9d} {us:4d} us  after [{prev}]  before [{nxt}]")
