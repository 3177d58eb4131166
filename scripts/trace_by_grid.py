"""Aggregate a rocprofv3 kernel trace by (kernel, grid): total / count / average duration."""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
only = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if only and only not in n:
        continue
    short = n.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")[:64]
    key = (short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t/1e3:9.1f} ms  n={c:6d} avg {t/c:8.1f} us  grid={k[1:]}  {k[0]}")
