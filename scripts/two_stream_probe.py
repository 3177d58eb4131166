"""Do two independent encoder forwards (T = 4096 and T = 8192 rows, with stash) finish sooner on two streams
than one after the other?  Each stream gets its own workspace."""
import sys, torch
sys.path.insert(0, ".")
import pgload
pg = pgload.load()
from protgen_gan_amd import bert
DEV = torch.device("cuda", 0)
cfg = pg.ProtBertConfig()
m = pg.ProtBertForMaskedLM(cfg, device=DEV, compute_dtype=torch.bfloat16).init_random(seed=1)
g = torch.Generator().manual_seed(0)
ids8 = torch.randint(5, 25, (8, 512), generator=g).to(DEV)
ids16 = torch.randint(5, 25, (16, 512), generator=g).to(DEV)
m8 = torch.ones(8, 512, device=DEV)
m16 = torch.ones(16, 512, device=DEV)
m.store.shadow_ptr()
side = torch.cuda.Stream()
ws_main = {}
ws_side = {}


def run(seq):
    if seq:
        m._encode(ids8, None, m8, want_stash=True)
        m._encode(ids16, None, m16, want_stash=True)
    else:
        side.wait_stream(torch.cuda.current_stream())
        bert._WORKSPACES.clear(); bert._WORKSPACES.update(ws_side)
        with torch.cuda.stream(side):
            m._encode(ids16, None, m16, want_stash=True)
        ws_side.update(bert._WORKSPACES)
        bert._WORKSPACES.clear(); bert._WORKSPACES.update(ws_main)
        m._encode(ids8, None, m8, want_stash=True)
        ws_main.update(bert._WORKSPACES)
        torch.cuda.current_stream().wait_stream(side)


for seq in (True, False, True, False):
    for _ in range(2):
        run(seq)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(seq)
    e1.record()
    torch.cuda.synchronize()
    print("sequential" if seq else "two streams", f"{e0.elapsed_time(e1) / 5:.2f} ms", flush=True)
