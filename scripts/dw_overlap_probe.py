"""Does a layer group's weight-gradient launch hide in the idle CUs of the backward chain's half-empty products?
Emulates the first-order backward of a critic update over T token rows: per layer the four input-gradient products
(d gelu-out = df W2 [N 4096, K 1024]; dh = du W1 + df [N 1024, K 4096]; dctx = da Wo [N 1024, K 1024];
dx = dqkv Wqkv + da [N 1024, K 3072]) on the main stream, and per four layers the grouped weight-gradient launch
(16 products, fp32 accumulate) either behind them on the main stream or on a side stream behind an event.
usage: dw_overlap_probe.py [T ...]"""
import sys
import torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
DEV = "cuda"
bf = torch.bfloat16
H, I = 1024, 4096
g = torch.Generator(device=DEV).manual_seed(0)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=DEV, generator=g) * scale).to(bf)


def run(T, layers, mode, side, evs):
    W2, W1, Wo, Wq = r(H, I, scale=.02), r(I, H, scale=.02), r(H, H, scale=.02), r(3 * H, H, scale=.02)
    sets = [dict(df=r(T, H), du=r(T, I), da=r(T, H), dqkv=r(T, 3 * H), gu=r(T, I), h=r(T, H), x=r(T, H), ctx=r(T, H))
            for _ in range(8)]
    out = dict(a=torch.empty(T, I, dtype=bf, device=DEV), b=torch.empty(T, H, dtype=bf, device=DEV),
               c=torch.empty(T, H, dtype=bf, device=DEV), d=torch.empty(T, H, dtype=bf, device=DEV))
    gW = [dict(W2=torch.zeros(H, I, device=DEV), W1=torch.zeros(I, H, device=DEV), Wq=torch.zeros(3 * H, H, device=DEV),
               Wo=torch.zeros(H, H, device=DEV)) for _ in range(4)]
    main = torch.cuda.current_stream()

    def chain(s):
        pg.ops.gemm(s["df"], W2, out["a"], b_kmajor=False)
        pg.ops.gemm(s["du"], W1, out["b"], b_kmajor=False, residual=s["df"])
        pg.ops.gemm(s["da"], Wo, out["c"], b_kmajor=False)
        pg.ops.gemm(s["dqkv"], Wq, out["d"], b_kmajor=False, residual=s["da"])

    def group(ss):
        probs = []
        for s, gw in zip(ss, gW):
            probs += [(s["df"], s["gu"], gw["W2"]), (s["du"], s["h"], gw["W1"]), (s["dqkv"], s["x"], gw["Wq"]),
                      (s["da"], s["ctx"], gw["Wo"])]
        pg.ops.gemm_group(probs, a_kmajor=False, b_kmajor=False, beta=1.0)

    def once():
        k = 0
        for l0 in range(0, layers, 4):
            ss = [sets[(l0 + i) % 8] for i in range(4)]
            for s in ss:
                chain(s)
            if mode == "serial":
                group(ss)
            else:
                evs[k % len(evs)].record(main)
                side.wait_event(evs[k % len(evs)])
                with torch.cuda.stream(side):
                    group(ss)
                k += 1
        if mode != "serial":
            main.wait_stream(side)

    for _ in range(2):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3 / layers * 1e3


Ts = [int(a) for a in sys.argv[1:]] or [8448, 2816]
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range", lo, hi)
for T in Ts:
    for mode, side in (("serial", None), ("side", torch.cuda.Stream()), ("side_low", torch.cuda.Stream(priority=0)),
                       ("serial", None), ("side", torch.cuda.Stream())):
        evs = [torch.cuda.Event() for _ in range(4)]
        us = run(T, 16, mode, side, evs)
        print(f"T {T:6d} {mode:9s} {us:8.1f} us per layer", flush=True)
