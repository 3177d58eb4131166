"""Idle time between consecutive kernels of a rocprofv3 kernel trace: total, and by the kernel that follows the gap."""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
tot_gap = small = 0
hist = collections.Counter()
prev_end = rows[0][1]
for s, e, n in rows[1:]:
    g = s - prev_end
    if 0 < g < 200_000:  # longer pauses are host-side (between steps), not launch gaps
        tot_gap += g
        small += 1
        hist[min(g // 1000, 20)] += 1
        k = n.replace("(anonymous namespace)::", "")[:50]
        gaps[k][0] += 1
        gaps[k][1] += g
    prev_end = max(prev_end, e)
print(f"span {span/1e6:.1f} ms, kernel time {busy/1e6:.1f} ms, gaps < 200 us: {small} totalling {tot_gap/1e6:.1f} ms ({100*tot_gap/span:.1f} % of span)")
print("gap histogram (us: count):", " ".join(f"{k}:{v}" for k, v in sorted(hist.items())))
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t/1e6:7.2f} ms  n={c:6d} avg {t/c/1e3:6.2f} us  before {k}")
