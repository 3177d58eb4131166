"""Concurrency of a rocprofv3 kernel trace: how much of the span has 0, 1, 2+ kernels in flight, the longest idle gaps and
what follows them, and the time with exactly one kernel in flight by that kernel's name.  usage: trace_overlap.py <dir> [t0_frac t1_frac]"""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f)))
lo, hi = rows[0][0], max(r[1] for r in rows)
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
w0, w1 = lo + (hi - lo) * f0, lo + (hi - lo) * f1
ev = []
for s, e, n, q in rows:
    if e <= w0 or s >= w1:
        continue
    ev.append((max(s, w0), 1, n)); ev.append((min(e, w1), -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
depth = 0; t_prev = w0; by_depth = collections.Counter(); alone = collections.Counter(); live = collections.Counter()
gaps = []
for t, d, n in ev:
    dt = t - t_prev
    by_depth[min(depth, 3)] += dt
    if depth == 1:
        alone[next(iter(k for k, v in live.items() if v > 0))[:60]] += dt
    if depth == 0 and dt > 0:
        gaps.append((dt, n))
    depth += d; live[n] += d; t_prev = t
span = w1 - w0
print(f"window {span/1e6:.1f} ms: " + ", ".join(f"{k}{'+' if k == 3 else ''} in flight {100*v/span:.1f} %" for k, v in sorted(by_depth.items())))
gaps.sort(reverse=True)
print("idle total %.1f ms; 12 longest gaps (ms, next kernel):" % (by_depth[0] / 1e6))
for dt, n in gaps[:12]:
    print(f"  {dt/1e6:7.3f}  {n[:70]}")
print("gaps > 50 us: %d totalling %.1f ms; gaps <= 50 us: %d totalling %.1f ms" % (sum(1 for g in gaps if g[0] > 50e3), sum(g[0] for g in gaps if g[0] > 50e3) / 1e6, sum(1 for g in gaps if g[0] <= 50e3), sum(g[0] for g in gaps if g[0] <= 50e3) / 1e6))
print("alone on the chip, by kernel (ms):")
for k, v in alone.most_common(14):
    print(f"  {v/1e6:8.1f}  {k}")
queues = collections.Counter(q for _, _, _, q in rows)
print("launches per queue/stream:", dict(queues))
