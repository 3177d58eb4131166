"""Per-tensor bf16-vs-fp32 gradient distances of the S 1024 critic update (tests/test_gpu_bench_shapes.py's case)."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import pgload
pg = pgload.load()
import test_gpu_bench_shapes as tb
from protgen_gan_amd import train as T
DEV, BF = "cuda", torch.bfloat16
cfg = pg.ProtBertConfig(num_hidden_layers=4, max_position_embeddings=1024)
real, fake, alpha = tb._bench_batches(pg, 4, 1024, 7)
print("lengths", real["lengths"])
out = {}
for dt in (torch.float32, BF):
    torch.manual_seed(3)
    critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=DEV, compute_dtype=dt).init_random(seed=2))
    with torch.no_grad():
        for k in (0, 2, 4):
            critic.classifier[k].bias.fill_(10.0)
    seen = []
    T.HEAD_STASH_HOOK = lambda hs: seen.append(hs.clone())
    loss = T.critic_update(critic, tb._NoStep(critic), real["input_ids"], real["attention_mask"], fake,
                           T.GanHyper(lambda_gp=5.0, max_len=1024), alpha=alpha, real_lengths=real["lengths"])
    torch.cuda.synchronize()
    T.HEAD_STASH_HOOK = None
    hs = seen[0]
    acts = hs[:, 1024:1024 + 512 + 256 + 64]
    print(dt, "cls |x| max", float(hs[:, :1024].abs().max()), "closed gates per row", (acts <= 0).sum(1).tolist(),
          "min act", float(acts.min()), "scores-ish a3 mean", float(hs[:, 1024 + 768:1024 + 832].mean()), flush=True)
    out_cls = hs[:, :1024].clone()
    out.setdefault("cls", {})[dt] = out_cls
    out[dt] = (float(loss), tb._grads(critic))
print("cls rows bf16 vs fp32 rel err per row", [round(float((out["cls"][BF][i] - out["cls"][torch.float32][i]).norm() / out["cls"][torch.float32][i].norm()), 4) for i in range(12)])
(l0, g0), (l1, g1) = out[torch.float32], out[BF]
print("loss", l0, l1)
d = {n: (tb.rel_l2(g1[n].numpy(), g0[n].numpy()), float(g0[n].norm()), float(g1[n].norm())) for n in g0}
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{v[0]:.4f}  |fp32| {v[1]:.4e}  |bf16| {v[2]:.4e}  {n}")
