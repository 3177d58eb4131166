#!/usr/bin/env python3
"""ProtBERT GAN benchmark on MI355X.

A step is one iteration of the reference's blind-mode training loop
(fully_masked_train.py:284-352) at BASELINE.json's configs[1]: n_critic = 8 critic updates
(each: 10-iteration fill of a fresh fake batch of 8 with the generator, gradient penalty with
its double backward, real and fake scores, c_loss.backward(), AdamW) followed by one generator
update (fill of 4 fakes, critic score, g_loss.backward()), len 512, bf16 compute with fp32
master weights and accumulation, both ProtBERTs unfrozen, lambda_gp = 5.  Weights are the
full-size ProtBERT architecture with seeded random values and the data are synthetic
DNMT-length sequences (neither checkpoints nor the dataset ship with the reference).

  python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torchrun)
  python bench.py --mode generate                         (sequences/s of the fill loop)

Prints one JSON line (rank 0).  `value` is summed over ranks (weak scaling: every rank runs
the per-GPU step on its own data and the critic gradients are averaged over RCCL)."""
import argparse
import ctypes
import glob
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_BF16 = 2.5e15  # dense bf16 MFMA peak of MI355X (spec)
VARIANT_NAMES = {8 + 2 + 1: "gemm_bf16_nt (X W^T, Q K^T)", 8 + 2 + 0: "gemm_bf16_nn (dY W, P V)",
                 8 + 4 + 0: "gemm_bf16_tn_f32out (dW = dY^T X)", 8 + 0: "gemm_bf16_tn (dS^T Q, P^T dO)",
                 8 + 4 + 2 + 1: "gemm_bf16_nt_f32out (scores)", 8 + 1: "gemm_bf16_tn_kB", 8 + 4 + 2: "gemm_bf16_nn_f32out",
                 8 + 4 + 1: "gemm_bf16_tn_kB_f32out"}


def f_enc(S, H=1024, I=4096, L=30):
    return L * (2 * (4 * H * H + 2 * H * I) + 4 * S * H)


def f_gen_tok(S, H=1024, V=30):
    return f_enc(S) + 2 * H * H + 2 * H * V


def f_c(B, S):
    return B * S * f_enc(S) + B * 1343616


def f_row(n):
    """One sequence of n token rows through the encoder + critic head, forward."""
    return n * f_enc(n) + 1343616


def gemm_source_digest():
    """Digest of the GEMM kernel sources: a counter capture under profiles/ carries the digest of the sources it was
    taken on, and only a capture of the current sources is quoted."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "protgen-gan_amd", "csrc", "gemm*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["train", "generate"], default="train")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--gen-batch", type=int, default=4)
    ap.add_argument("--n-critic", type=int, default=8)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--micro-batch", type=int, default=0,
                    help="critic updates made this many rows at a time (same update, the activation memory of the smaller batch)")
    ap.add_argument("--lambda-gp", type=float, default=5.0)
    ap.add_argument("--layers", type=int, default=30, help="(debug only) fewer layers invalidates the number")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--gan-mode", choices=["blind", "seeded"], default="blind",
                    help="seeded = 10p_train.py's loop: fakes start from real sequences with 90 %% of the residues masked")
    ap.add_argument("--regime", choices=["unfrozen", "frozen"], default="unfrozen",
                    help="frozen = epoch 0 of the trainers: both ProtBERTs frozen, only the critic head learns")
    ap.add_argument("--soft-generator-update", action="store_true",
                    help="generator update through soft embeddings (gradients reach the generator and are all-reduced); "
                         "not the reference loop's update, so not the default line")
    ap.add_argument("--no-fill-ahead", action="store_true",
                    help="fill an iteration's fake batches at its start instead of beside the previous iteration's updates "
                         "(train.FillAhead; blind mode with the reference's generator update only)")
    ap.add_argument("--no-lanes", action="store_true",
                    help="critic updates on one stream (train.TWO_STREAM_BACKWARD_DEFAULT off): with --no-fill-ahead, every "
                         "launch has the chip to itself, which is what a kernel trace for per-kernel rates wants")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-generate-leg", action="store_true", help="skip the generated-sequences/s leg after the train leg")
    ap.add_argument("--generate-batches", type=int, default=3, help="batches of 64 sequences per rank in that leg")
    ap.add_argument("--profile-sample", type=int, default=16)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="(kernel A/B runs) forwarded to pg_tune before the first step")
    args = ap.parse_args()

    import pgload
    pg = pgload.load()
    from protgen_gan_amd import distributed as pgd
    from protgen_gan_amd import train as pgt
    from protgen_gan_amd._lib import lib

    if args.no_lanes:
        pgt.TWO_STREAM_BACKWARD_DEFAULT = False
    rank, local, world = pgd.init_from_env()
    if torch.distributed.is_initialized():
        # the ranks the process group actually has, not what the environment announced
        world = torch.distributed.get_world_size()
        rank = torch.distributed.get_rank()
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s); reporting n_gpus = {world}",
              file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    for kv in args.tune:
        k, v = kv.split("=")
        lib.pg_tune(int(k), int(v))

    cfg = pg.ProtBertConfig(num_hidden_layers=args.layers)
    generator = pg.Generator(pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=dtype).init_random(seed=0))
    torch.manual_seed(2)
    critic = pg.Critic(pg.ProtBertForMaskedLM(cfg, device=dev, compute_dtype=dtype).init_random(seed=1))
    if args.regime == "frozen":
        pgt.freeze_protbert(generator)
        pgt.freeze_protbert(critic)
    hp = pgt.GanHyper(n_critic=args.n_critic, lambda_gp=args.lambda_gp, max_len=args.seq, gen_batch_blind=args.gen_batch,
                      micro_batch=args.micro_batch or None)
    gen_opt = pg.AdamW(generator.parameters(), lr=5e-6, betas=(0.9, 0.999), weight_decay=0.01)
    critic_opt = pg.AdamW(critic.parameters(), lr=5e-5, betas=(0.9, 0.999), weight_decay=0.01)
    sync = pgd.GradSync([critic.protbert.store, critic.classifier.store], world, overlap=critic.protbert,
                        position_rows=args.seq)

    # lengths (real rows and fake templates) come from one stream every rank seeds alike: a rank draws the global
    # batch's lengths and keeps its balanced share (data.balanced_deal), so the ranks' token counts stay within a few
    # per cent of each other and none is the straggler of every gradient reduction; contents are per rank
    rng = random.Random(1234)
    torch.manual_seed(1234 + rank)
    tgen = torch.Generator().manual_seed(1234 + rank)
    scale = args.seq / 512.0
    pool = [int(L * scale) for L in range(200, 501)]  # DNMT-length stand-in: uniform 200..500

    if args.mode == "generate":
        return bench_generate(args, pg, generator, pool, rng, rank, world, dev)

    total = args.warmup + args.steps
    real = [[pgt.synthetic_real_batch(args.batch, pool, args.seq, dev, rng, tgen, rank, world) for _ in range(args.n_critic)]
            for _ in range(total)]

    use_sync = world > 1
    if world == 1 and os.environ.get("PG_REHEARSE_SYNC") == "1":
        # one-rank RCCL group: the host and stream cost of the layer-wise gradient reduction, nothing to reduce with
        import tempfile
        torch.distributed.init_process_group("nccl", init_method=f"file://{tempfile.mkdtemp()}/rdv", rank=0, world_size=1,
                                             device_id=dev)
        sync = pgd.GradSync([critic.protbert.store, critic.classifier.store], 2, overlap=critic.protbert,
                            position_rows=args.seq)
        use_sync = True

    seeds = None
    if args.gan_mode == "seeded":
        # (one set more than iterations: with the fill run ahead, the last timed iteration fills for a next one too, so
        # that the timed region holds as many fills as iterations)
        seeds = [[pgt.synthetic_real_batch(args.batch, pool, args.seq, dev, rng, tgen, rank, world) for _ in range(args.n_critic + 1)]
                 for _ in range(total + 1)]

    soft = args.soft_generator_update
    gen_sync = None
    if soft and world > 1:
        gen_sync = pgd.GradSync([generator.protbert.store], world, overlap=generator.protbert, position_rows=args.seq)

    # the next iteration's fakes are filled on a side stream beside this iteration's updates (every timed step still
    # runs one whole fill and one whole set of updates; what step i consumes was filled during step i - 1)
    # (at the device's lowest stream priority when no gradient reduction runs beside it: train.FILL_STREAM_LOW_PRIORITY)
    ahead = pgt.FillAhead(dev, low_priority=not use_sync) if (not soft and not args.no_fill_ahead) else None

    def one_step(i, ahead=ahead):
        if seeds is not None:
            nxt = seeds[i + 1] if (ahead is not None and i + 1 < len(seeds)) else None
            return pgt.gan_step_seeded(generator, critic, gen_opt, critic_opt, real[i], seeds[i], hp,
                                       grad_sync=sync if use_sync else None, soft_generator_update=soft, gen_grad_sync=gen_sync,
                                       fill_ahead=ahead, next_gen_batches=nxt)
        return pgt.gan_step_blind(generator, critic, gen_opt, critic_opt, real[i], pool, hp, args.batch, dev, rng=rng,
                                  grad_sync=sync if use_sync else None, soft_generator_update=soft, gen_grad_sync=gen_sync,
                                  rank=rank, world=world, fill_ahead=ahead)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    verbose = os.environ.get("PG_BENCH_VERBOSE") == "1"
    if verbose:
        import faulthandler
        faulthandler.dump_traceback_later(60, repeat=False, file=sys.stderr)
    for i in range(args.warmup):
        one_step(i)
        if verbose:
            torch.cuda.synchronize()
            print(f"[rank {rank}] warm-up step {i} done", file=sys.stderr, flush=True)
    fence()
    generator.fill_shapes.clear()
    generator.fill_widths.clear()
    del pgt.EXEC_LOG[:]
    lib.pg_profile_begin(args.profile_sample)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fills_seen, fill_rows = 0, []
    t0 = time.perf_counter()
    marks[0].record()
    last = None
    for i in range(args.warmup, total):
        last = one_step(i)
        marks[i - args.warmup + 1].record()
        fill_rows.append(sum(B * S for B, S in generator.fill_shapes[fills_seen:]))
        fills_seen = len(generator.fill_shapes)
    fence()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    prof = (ctypes.c_double * 48)()
    nl = ctypes.c_int64(0)
    lib.pg_profile_end(prof, ctypes.byref(nl))
    if world > 1:
        t = torch.tensor([dt], device=dev if torch.distributed.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    c_loss, g_loss = float(last[0]), float(last[1])
    fill_log = (list(generator.fill_shapes), list(generator.fill_widths))
    exec_log = list(pgt.EXEC_LOG)
    # after the timed region: a few iterations one after the other (fill, then updates), for the kernels' rates
    # when each launch has the chip to itself and for what the overlap is worth
    serial = None
    if ahead is not None:
        ahead.close()
        n_serial = min(args.steps, 4)
        lanes_were = pgt.TWO_STREAM_BACKWARD_DEFAULT
        pgt.TWO_STREAM_BACKWARD_DEFAULT = False  # one stream only: a launch's duration is then its own
        one_step(args.warmup, ahead=None)
        fence()
        lib.pg_profile_begin(args.profile_sample)
        t1 = time.perf_counter()
        for i in range(n_serial):
            one_step(args.warmup + i, ahead=None)
        fence()
        dt1 = time.perf_counter() - t1
        prof1 = (ctypes.c_double * 48)()
        nl1 = ctypes.c_int64(0)
        lib.pg_profile_end(prof1, ctypes.byref(nl1))
        v1 = variant_table(prof1)
        pgt.TWO_STREAM_BACKWARD_DEFAULT = lanes_were
        serial = {"steps": n_serial, "ms_per_step": round(dt1 / n_serial * 1e3, 2),
                  "what": "the same iterations on one stream (fill at their start, critic updates without their second lane), "
                          "after the timed region",
                  "variants": v1, "gemm_launches_total": int(nl1.value)}
    generator.fill_shapes[:], generator.fill_widths[:] = fill_log
    pgt.EXEC_LOG[:] = exec_log

    # algorithmic FLOPs of the timed region on this rank
    fill_flops = sum(B * S * f_gen_tok(S) for B, S in generator.fill_shapes)
    # forward-equivalents per critic update: unfrozen 12 (penalty 6 + real/fake forward 2 and backward 4);
    # frozen 5 (penalty forward, input gradient and tangent pass + real/fake forward; no encoder backward)
    fe_critic, fe_gen = (12, 3) if args.regime == "unfrozen" else (5, 1)
    if soft:  # generator forward + backward with weight gradients (3), critic forward + input-gradient backward (2)
        fe_gen = 5 if args.regime == "unfrozen" else 2
    critic_flops = args.steps * args.n_critic * fe_critic * f_c(args.batch, args.seq)
    gen_flops = args.steps * fe_gen * f_c(args.gen_batch if seeds is None else args.batch, args.seq)
    flops_step = (fill_flops + critic_flops + gen_flops) / args.steps
    # the same accounting on the token rows the updates actually launched: a ragged update runs every sequence on its
    # own leading columns (no pad rows), so the padded-width figure above counts work that is never executed.  Per
    # sequence of n rows: gradient-penalty interpolate 6 (frozen 3) forward-equivalents, scored row 3 (frozen 1).
    fe_gp, fe_sc = (6, 3) if args.regime == "unfrozen" else (3, 1)
    exec_critic = sum(fe_gp * sum(f_row(n) for n in gl) + fe_sc * sum(f_row(n) for n in sl)
                      for kind, gl, sl in pgt.EXEC_LOG if kind == "critic")
    exec_gen = sum(fe_sc * sum(f_row(n) for n in sl) for kind, gl, sl in pgt.EXEC_LOG if kind == "generator")
    if soft:
        exec_gen = gen_flops  # (the soft update runs rectangular)
    exec_rows = sum(sum(gl) + sum(sl) for _, gl, sl in pgt.EXEC_LOG)
    flops_step_exec = (fill_flops + exec_critic + exec_gen) / args.steps
    ms_step = dt / args.steps * 1e3

    variants = variant_table(prof)
    dom = variants[0] if variants else None
    # With two streams sharing the chip a launch's duration is not its own (its workgroups wait for CUs the other
    # stream holds), so FLOPs / duration of a launch in the timed region understates the kernel.  The kernel's roofline
    # is then taken from the iterations run one launch at a time right after the timed region (same HIP-event sampling,
    # same process); the timed region's figure is kept beside it.
    overlapped = None
    if serial is not None and serial["variants"]:
        if dom:
            overlapped = {"kernel": dom["kernel"], "achieved": round(dom["tflops"], 2), "unit": "TFLOP/s",
                          "frac": round(dom["tflops"] * 1e12 / MFMA_PEAK_BF16, 4), "avg_launch_us": round(dom["avg_us"], 2),
                          "sampled_launches": dom["sampled_launches"],
                          "note": "launch durations inside the timed region, where the fill's and the updates' launches share the CUs"}
        variants_timed, variants = variants, serial.pop("variants")
        dom = variants[0]
        nl = ctypes.c_int64(serial.pop("gemm_launches_total"))
    else:
        variants_timed = None
    if serial is not None:
        serial.pop("variants", None)
        serial.pop("gemm_launches_total", None)
    # fabric-side bytes per launch of the dominant kernel's largest shape, from the newest counter capture committed
    # under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over scripts/pmc_gemm.py, summarised by
    # scripts/pmc_summary.py): counters cannot be read from inside this process
    traffic = traffic_note = None
    caps = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if caps:
        try:
            with open(caps[-1]) as fh:
                cap = json.load(fh)
            shapes = {e["shape"]: e for e in cap["shapes"]}
            e = shapes.get("fill_w1")
            if e and cap.get("gemm_source_digest") != gemm_source_digest():
                # the GEMM kernels changed after the capture: its bytes describe another binary
                traffic_note = (f"{os.path.basename(caps[-1])} was captured on GEMM sources {cap.get('gemm_source_digest')}, "
                                f"this run's are {gemm_source_digest()}: not quoted")
            elif e:
                traffic = int(e["fetch_bytes_median"] + e["write_bytes_median"])
                traffic_note = (f"{os.path.basename(caps[-1])} (GEMM sources {cap['gemm_source_digest']}): fill-pass W1 product "
                                f"M {e['M']} N {e['N']} K {e['K']} (the dominant "
                                f"kernel's largest launch): fetched {e['fetch_bytes_median'] / 1e6:.0f} MB = {e['fetch_over_operands']} x its "
                                f"operands, wrote {e['write_bytes_median'] / 1e6:.0f} MB = {e['write_over_output']} x its output")
        except (OSError, KeyError, ValueError):
            pass
    roofline = None
    if dom:
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(dom["tflops"], 2),
                    "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": round(dom["tflops"] * 1e12 / MFMA_PEAK_BF16, 4),
                    "traffic": traffic, "traffic_note": traffic_note, "avg_launch_us": round(dom["avg_us"], 2), "sampled_launches": dom["sampled_launches"],
                    "gemm_launches_total": int(nl.value),
                    "measured_over": ("%d iterations run one launch at a time right after the timed region (HIP events on the "
                                      "launch stream, every %dth launch)" % (serial["steps"], args.profile_sample) if overlapped
                                      else "the timed region (HIP events on the launch stream, every %dth launch)" % args.profile_sample),
                    "overlapped": overlapped}

    out = {
        "metric": "ProtBERT GAN train steps/s (gen+critic) & generated seqs/s, 1/2/4/8 GPUs",
        "value": round(world * args.steps / dt, 5), "unit": f"GAN steps/s (n_critic={args.n_critic} critic updates + 1 generator update; summed over ranks)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("configs[2] on one GPU: one seeded-mode GAN step (10p_train.py), 90 % of the residues masked"
                                if seeds is not None else
                                ("configs[1]" if (args.batch, args.gen_batch, args.seq, args.n_critic, args.layers) == (8, 4, 512, 8, 30)
                                 else "configs[1] at other sizes") +
                                ": one blind-mode GAN step (fully_masked_train.py), ProtBERT generator + ProtBERT critic head"),
                   "critic_batch": args.batch, "generator_batch": args.gen_batch if seeds is None else args.batch, "seq_len": args.seq, "n_critic": args.n_critic,
                   "lambda_gp": args.lambda_gp, "layers": args.layers, "regime": args.regime, "micro_batch": args.micro_batch or None,
                   "generator_update": "soft embeddings through the critic's 3-D branch" if soft else "token ids (reference loop)",
                   "fill_iterations_per_fake_batch": 10, "parallelism": f"dp{world}",
                   "weights": "ProtBERT architecture, seeded random init", "lengths": f"uniform {pool[0]}..{pool[-1]} (DNMT-length stand-in)"},
        # two prices of the same step.  "algorithmic": every update at the padded width the reference runs (len 512);
        # "executed": every update at the token rows actually launched (ragged rows, no pads).  The first says how
        # fast the reference's step is being done, the second how busy the matrix pipes are.
        "algorithmic_tflop_per_step": round(flops_step / 1e12, 2),
        "step_tflops_per_gpu": round(flops_step / (dt / args.steps) / 1e12, 2),
        "step_mfma_frac": round(flops_step / (dt / args.steps) / MFMA_PEAK_BF16, 4),
        "executed_tflop_per_step": round(flops_step_exec / 1e12, 2),
        "step_mfma_frac_executed": round(flops_step_exec / (dt / args.steps) / MFMA_PEAK_BF16, 4),
        "executed_update_rows_per_step": round(exec_rows / args.steps, 1),
        "step_ms": {"min": round(min(step_ms), 2), "median": round(sorted(step_ms)[len(step_ms) // 2], 2),
                    "max": round(max(step_ms), 2), "each": [round(v, 1) for v in step_ms],
                    "fill_token_rows_each": fill_rows},
        "c_loss": c_loss, "g_loss": g_loss,
        "schedule": (("next iteration's fill beside this iteration's updates (train.FillAhead"
                      + (", on a stream of the lowest priority)" if not use_sync and pgt.FILL_STREAM_LOW_PRIORITY else ")")
                      if ahead is not None
                      else "fill, then updates")
                     + ("; critic updates in two lanes" if pgt.TWO_STREAM_BACKWARD_DEFAULT and args.regime == "unfrozen" else "")),
        "one_after_the_other": serial,
        "roofline": roofline, "gemm_variants": variants, "gemm_variants_overlapped": variants_timed,
    }
    # the other half of the metric, after and outside the train leg's timed region: generated sequences/s
    if not args.no_generate_leg:
        out["generate"] = time_generate(pg, generator, pool, rank, world, dev, batches=args.generate_batches,
                                        profile_sample=args.profile_sample)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, pg, dev)
            if "generate" in out:
                out["generate"]["cpu_baseline"] = out["cpu_baseline"].pop("generate")
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def variant_table(prof):
    """pg_profile_end's table (per GEMM operand layout: sampled launches, their FLOPs, their milliseconds from HIP
    events on the launch stream) as a list, largest time share first."""
    variants = []
    for v in range(16):
        n, fl, ms = prof[v * 3], prof[v * 3 + 1], prof[v * 3 + 2]
        if n > 0:
            variants.append({"kernel": VARIANT_NAMES.get(v, f"gemm_variant_{v}"), "sampled_launches": int(n),
                             "avg_us": ms / n * 1e3, "tflops": fl / (ms * 1e-3) / 1e12, "time_share_of_gemm": ms})
    tot_ms = sum(v["time_share_of_gemm"] for v in variants) or 1.0
    for v in variants:
        v["time_share_of_gemm"] = round(v["time_share_of_gemm"] / tot_ms, 4)
    variants.sort(key=lambda v: -v["time_share_of_gemm"])
    return variants


def time_generate(pg, generator, pool, rank, world, dev, batches, warm_batches=1, bs=64, profile_sample=16):
    """generate.py's full mode (batch 64 per GPU, lengths from the pool, 10 fill iterations): sequences/s over
    `batches` batches per rank, summed over ranks (every rank fills its own sequences: no collective)."""
    from protgen_gan_amd import generate as gen_py
    tok = pg.ProtTokenizer()
    n = bs * max(1, batches)
    gen_py.generate_full_mode_sequences(generator, tok, bs * max(1, warm_batches), pool, batch_size=bs, device=str(dev),
                                        rng_seed=7 + rank)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    generator.fill_shapes.clear()
    generator.fill_widths.clear()
    from protgen_gan_amd._lib import lib
    lib.pg_profile_begin(profile_sample)
    t0 = time.perf_counter()
    seqs = gen_py.generate_full_mode_sequences(generator, tok, n, pool, batch_size=bs, device=str(dev), rng_seed=42 + rank)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 48)()
    nl = ctypes.c_int64(0)
    lib.pg_profile_end(prof, ctypes.byref(nl))
    variants = variant_table(prof)
    if world > 1:
        t = torch.tensor([dt], device=dev if torch.distributed.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # the same two prices as the train leg: "algorithmic" = every batch at the width generate.py pads it to (its longest
    # row + 2, generate.py:93-95); "executed" = every row at its own length, which is what the packed passes launch
    flops_pad = sum(B * S * f_gen_tok(S) for B, S in generator.fill_shapes)
    flops_exec = sum(B * S * f_gen_tok(S) if w is None else sum(x * f_gen_tok(x) for x in w)
                     for (B, S), w in zip(generator.fill_shapes, generator.fill_widths))
    roofline = None
    if variants:
        d = variants[0]
        roofline = {"bound": "mfma", "kernel": d["kernel"], "achieved": round(d["tflops"], 2), "peak": MFMA_PEAK_BF16 / 1e12,
                    "unit": "TFLOP/s", "frac": round(d["tflops"] * 1e12 / MFMA_PEAK_BF16, 4), "traffic": None,
                    "avg_launch_us": round(d["avg_us"], 2), "sampled_launches": d["sampled_launches"],
                    "gemm_launches_total": int(nl.value)}
    return {"value": round(world * n / dt, 3), "unit": "sequences/s", "sequences_per_rank": n, "batch": bs, "seconds": round(dt, 4),
            "mean_len": round(sum(len(s) for s in seqs) / len(seqs), 1),
            "step_mfma_frac": round(flops_pad / dt / MFMA_PEAK_BF16, 4),
            "step_mfma_frac_executed": round(flops_exec / dt / MFMA_PEAK_BF16, 4),
            "roofline": roofline,
            "workload": "configs[3]-style mass generation (generate.py full mode), lengths uniform %d..%d, 10 fill iterations"
                        % (pool[0], pool[-1])}


def bench_generate(args, pg, generator, pool, rng, rank, world, dev):
    """Generated sequences/s of generate.py's full mode (batch 64 per GPU, lengths from the pool)."""
    g = time_generate(pg, generator, pool, rank, world, dev, args.steps, args.warmup, profile_sample=args.profile_sample)
    out = {"metric": "generated sequences/s (generate.py full mode)", "value": g["value"], "unit": "sequences/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(g["seconds"] / max(1, args.steps) * 1e3, 2),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": g["workload"] + ", batch 64 per GPU", "sequences_per_rank": g["sequences_per_rank"],
                      "parallelism": f"shard{world}"},
           "mean_len": g["mean_len"], "step_mfma_frac": g["step_mfma_frac"],
           "step_mfma_frac_executed": g["step_mfma_frac_executed"], "roofline": g["roofline"]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, None, dev, generate_only=True)["generate"]
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def cpu_baseline(args, pg=None, dev=None, generate_only=False):
    """The CPU oracle (a plain-PyTorch fp32 restatement of the reference path) timed on this
    box's host cores on a bounded sample, one warm-up and the median of three each:
      * one fill iteration of generate.py's full mode at batch 4, L 510 (S 512) -> generated sequences/s
        (ten iterations per sequence, generate.py:144-158);
      * one critic update (GP double backward + real + fake + backward + AdamW) at batch 2, len 512;
        the GAN step time is composed linearly in batch and iteration counts.
    Full-size ProtBERT; about 40 s of CPU work on 16 threads."""
    from oracle import protbert_cpu as O
    torch.set_num_threads(min(torch.get_num_threads(), os.cpu_count() or 1))
    cores = torch.get_num_threads()
    c = O.BertCfg(num_hidden_layers=args.layers, max_position_embeddings=args.seq)
    P = O.init_params(c, seed=0)
    C = O.init_classifier(c.hidden_size, seed=2)
    g = torch.Generator().manual_seed(0)
    S = args.seq
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    Bf = 4  # configs[0]: batch 4, L 510
    ids, attn = O.build_fully_masked_batch([S - 2] * Bf)
    noise = O.draw_noise(Bf * S, c.vocab_size, g)
    t_iter = []
    for rep in range(4):  # the first is the warm-up
        t0 = time.perf_counter()
        with torch.no_grad():
            O.generate_step(P, c, ids.clone(), attn, 1.0, 0.1, 1.0, noise)
        t_iter.append(time.perf_counter() - t0)
    t_iter = t_iter[1:]
    t_fill = med(t_iter) / Bf  # per sequence and iteration
    gen_cpu = {"value": round(1.0 / (10 * t_fill), 5), "unit": "sequences/s", "cores": cores, "kind": "port",
               "sample": (f"oracle/protbert_cpu.py fp32 on {cores} host threads: one fill iteration of generate.py's full mode at "
                          f"batch {Bf}, L {S - 2} (S {S}), warm-up + median of 3 ({', '.join(f'{v:.2f}' for v in t_iter)} s); a "
                          f"sequence takes ten iterations")}
    if generate_only:
        return {"generate": gen_cpu}
    Bs = 2  # sample batch of the critic update; times below are per sequence
    real = torch.randint(5, 25, (Bs, S), generator=g)
    fake = torch.randint(5, 25, (Bs, S), generator=g)
    ones = torch.ones(Bs, S, dtype=torch.long)
    alpha = torch.rand(Bs, 1, 1, generator=g)
    t_updates = []
    for rep in range(3):
        t0 = time.perf_counter()
        loss, gp, rs, fs, grads = O.critic_step_grads(P, C, c, real, fake, ones, ones, alpha, args.lambda_gp)
        t_updates.append(time.perf_counter() - t0)
    k = "protbert.bert.encoder.layer.0.output.dense.weight"
    p, gr = P[k[len("protbert."):]].clone(), grads[k]
    t_adams = []
    for rep in range(3):
        t1 = time.perf_counter()
        O.adamw_step(p, gr, torch.zeros_like(p), torch.zeros_like(p), 1, 5e-5)
        t_adams.append((time.perf_counter() - t1) * (420.6e6 / p.numel()))
    t_adam = med(t_adams)
    t_bwd = med(t_updates)               # Bs sequences through GP + real + fake + backward
    t_critic = t_bwd / Bs + t_adam / args.batch  # per sequence of a batch-`batch` update (one AdamW step per update)
    t_step = args.n_critic * (args.batch * t_critic + 10 * args.batch * t_fill) + (10 * args.gen_batch * t_fill + args.gen_batch * t_critic / 4)
    probe = None
    if pg is not None:
        # the same critic update (same weights, inputs, alpha) on the engine in the benchmark's precision: the
        # oracle as the checker of what was timed above (loss, penalty and scores; gradients are the tests' job)
        from protgen_gan_amd import train as pgt
        dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        cfg = pg.ProtBertConfig(num_hidden_layers=args.layers, max_position_embeddings=args.seq)
        critic = pg.Critic(pg.ProtBertForMaskedLM.from_params(cfg, P, device=dev, compute_dtype=dt))
        critic.classifier.load_state_dict(C)
        d = lambda t: t.to(dev)  # noqa: E731
        gp_h = pg.compute_gradient_penalty(critic, d(real), d(fake), d(ones), d(ones), dev, alpha=alpha)
        with torch.no_grad():
            rs_h, fs_h = critic(d(real), attention_mask=d(ones)), critic(d(fake), attention_mask=d(ones))
        l_h = pg.critic_loss(rs_h, fs_h, gp_h, args.lambda_gp)
        sc = torch.cat([rs, fs]).detach()
        sc_h = torch.cat([rs_h, fs_h]).detach().cpu()
        probe = {"what": f"one critic update B={Bs} S={S}, {args.layers} layers, engine ({args.dtype}) against the fp32 CPU oracle, same weights and draws",
                 "c_loss_oracle": float(loss.detach()), "c_loss_engine": float(l_h.detach()), "gp_oracle": float(gp.detach()),
                 "gp_engine": float(gp_h.detach()),
                 "scores_max_abs_err": float((sc_h - sc).abs().max()), "scores_max_abs": float(sc.abs().max())}
        del critic
    return {"value": round(1.0 / t_step, 6), "unit": "GAN steps/s", "cores": cores, "kind": "port", "parity_probe": probe,
            "generate": gen_cpu,
            "sample": (f"oracle/protbert_cpu.py fp32 on {cores} host threads, medians of 3: 1 fill iteration B={Bf} S={S} "
                       f"({t_fill:.2f} s/seq) + 1 critic "
                       f"update B={Bs} S={S} incl. GP double backward ({t_bwd:.2f} s; runs {', '.join(f'{v:.1f}' for v in t_updates)}) "
                       f"and AdamW ({t_adam:.2f} s/step); per sequence "
                       f"{t_critic:.2f} s; GAN step composed as "
                       f"n_critic*(B*critic + 10*B*fill) + (10*Bg*fill + Bg*critic/4) = {t_step:.0f} s")}


if __name__ == "__main__":
    main()
