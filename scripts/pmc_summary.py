"""Reads the counter_collection.csv files of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
scripts/pmc_gemm.py and the shape list that program printed; writes per shape the fabric-side bytes per launch.
FETCH_SIZE is reported in KiB and, on gfx950, counts a 128-byte read request as 64 bytes: it is doubled here.
WRITE_SIZE (KiB) is taken as reported.  usage: pmc_summary.py <fetch dir> <write dir> <shapes.json> <out.json>"""
import collections, csv, glob, hashlib, json, os, sys


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "gemm" not in r["Kernel_Name"]:
            continue
        rows.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], int(r["Grid_Size"]) // int(r["Workgroup_Size"]),
                                                0.0])[2] += float(r["Counter_Value"])
    return list(rows.values())


fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
shapes = json.load(open(sys.argv[3]))
assert len(fetch) == len(write), (len(fetch), len(write))
# launches in program order; a shape whose last tile round is split off makes two launches per call
out, i = [], 0
for s in shapes:
    per_call = []
    for _ in range(4):
        tot_f = tot_w = 0.0
        kernels = []
        rows_left = s["M"]
        while rows_left > 0:
            name, wgs, v = fetch[i]
            tile = 256 if "glds256" in name else 128
            gy = wgs // ((s["N"] + tile - 1) // tile)
            kernels.append(f"{name.split('<')[0].split('::')[-1]} {tile}x{tile} tiles, {gy} tile rows")
            tot_f += 2 * v * 1024
            tot_w += write[i][2] * 1024
            rows_left -= gy * tile
            i += 1
        per_call.append((tot_f, tot_w, kernels))
    f = sorted(c[0] for c in per_call)
    w = sorted(c[1] for c in per_call)
    s.update({"launches": per_call[-1][2], "fetch_bytes_median": (f[1] + f[2]) / 2, "fetch_bytes_first": per_call[0][0],
              "write_bytes_median": (w[1] + w[2]) / 2,
              "fetch_over_operands": round((f[1] + f[2]) / 2 / s["read_bytes"], 2),
              "write_over_output": round((w[1] + w[2]) / 2 / s["write_bytes"], 2)})
    out.append(s)
def gemm_source_digest():
    """Digest of the GEMM kernel sources the capture was taken on (bench.py quotes a capture only while it matches)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "protgen-gan_amd", "csrc", "gemm*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


json.dump({"gemm_source_digest": gemm_source_digest(), "note": "fabric-side bytes per call (all launches of the call), rocprofv3 --pmc FETCH_SIZE x2 (gfx950 counts a "
                   "128-byte request as 64) and WRITE_SIZE, separate passes", "shapes": out}, open(sys.argv[4], "w"), indent=1)
for s in out:
    print(f"{s['shape']:10s} M{s['M']} N{s['N']} K{s['K']}: fetch {s['fetch_bytes_median']/1e6:8.1f} MB = {s['fetch_over_operands']:.2f} x operands "
          f"({s['read_bytes']/1e6:.1f} MB); write {s['write_bytes_median']/1e6:8.1f} MB = {s['write_over_output']:.2f} x output; {s['launches']}")
