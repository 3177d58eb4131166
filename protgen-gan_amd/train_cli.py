#!/usr/bin/env python3
"""The two training scripts of the reference as one trainer on the HIP engine:

    python train_gan.py --mode blind   ...     # fully_masked_train.py
    python train_gan.py --mode seeded  ...     # 10p_train.py
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train_gan.py --mode ...   # data-parallel over RCCL
(train_gan.py at the repository root registers the package and calls main() below.)

Flags --n_critic --lambda_gp --lr_gen --lr_critic --wd_gen --wd_critic --n_epochs --run_name --batch_size
--num_eval_sequences --eval_batch_size are the reference's (fully_masked_train.py:19-43, 10p_train.py:19-43),
same defaults.  What the loop reproduces:

  * epoch 0 trains the critic head only: both ProtBERTs frozen, critic learning rate 1e-5; from epoch 1 on
    everything trains at --lr_critic (fully_masked_train.py:119-136,274-282);
  * blind mode: groups of n_critic critic updates, each on the next real batch and a fresh batch of fully masked
    fakes, then one generator update on 4 fakes; the epoch ends when the critic loader runs dry, the critic
    updates already made in an incomplete group stand (fully_masked_train.py:284-365);
  * seeded mode: every epoch re-splits the data into N / (n_critic + 1) generator rows and the rest for the critic
    (dataset.py:72-94); a critic update takes the next critic batch (the iterator starts over when it runs dry) and
    fakes grown from the next generator batch; the epoch ends when the generator loader runs dry, before a critic
    update or before the generator update (10p_train.py:296-380);
  * evaluation points: first iteration of epoch 0 ("baseline"), the middle iteration ("mid"), the end of every
    epoch ("end") (fully_masked_train.py:358-373).  The reference folds generated sequences with ESMFold and
    scores them with ProteinMPNN / Progres / TM-align there: other models, out of this engine's scope.  Here the
    hook generates --num_eval_sequences sequences with the evaluation fill schedule (val_metrics.py:96-140)
    and logs them with the losses;
  * a checkpoint per epoch in the reference's layout (fully_masked_train.py:378-389):
    <checkpoint_dir>/<run_name>/epoch_<n>/{critic_bert/, critic_classifier.pth, generator_bert/}, plus the critic
    optimizer's state and the epoch number (`--resume <dir>` continues from them; the reference cannot resume).

The loop itself (`run_epoch_blind`, `run_epoch_seeded`, `fit`) works on a small `Steps` interface, so its schedule
is tested on the CPU with recording stand-ins; `EngineSteps` binds it to train.py's updates."""
from __future__ import annotations

import argparse
import json
import os
import random
import time
from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import torch

WARMUP_HEAD_LR = 1e-5   # fully_masked_train.py:124
GEN_BATCH_BLIND = 4     # fully_masked_train.py:339


# ----------------------------------------------------------------------------- the interface the loop drives
class Steps:
    """What one trainer iteration needs.  Batches are dicts with "input_ids", "attention_mask" and optionally "lengths"."""

    n_critic: int = 8

    def set_frozen(self, frozen: bool) -> None: ...
    def set_critic_lr(self, lr: float) -> None: ...
    # -> (c_loss mean or None, g_loss or None); more_follows: another full group comes after this one in the epoch
    def blind_iteration(self, real_batches: Sequence[dict], with_generator_update: bool, more_follows: bool = False): ...
    def seeded_iteration(self, real_batches: Sequence[dict], gen_batches: Sequence[dict], gen_update_batch: Optional[dict]): ...
    def evaluate(self, tag: str, epoch: int, batch: int, c_loss: float, g_loss: float) -> None: ...
    def save(self, epoch: int) -> None: ...
    def new_epoch_loaders(self, epoch: int): ...   # blind: critic loader; seeded: (generator loader, critic loader)


def _take(it: Iterator, n: int) -> List:
    out = []
    for _ in range(n):
        try:
            out.append(next(it))
        except StopIteration:
            break
    return out


def run_epoch_blind(steps: Steps, epoch: int, critic_loader) -> Dict[str, float]:
    """fully_masked_train.py:262-373 for one epoch."""
    it = iter(critic_loader)
    mid = len(critic_loader) // 2          # (the reference compares the ITERATION count with half the number of critic batches)
    n_iter, c_sum, g_sum = 0, 0.0, 0.0
    nxt = _take(it, steps.n_critic)
    while True:
        group, nxt = nxt, []
        full = len(group) == steps.n_critic
        if full:
            nxt = _take(it, steps.n_critic)  # (looked at now: the iteration may start on the next one's fakes if it is whole)
        if group:
            c, g = steps.blind_iteration(group, with_generator_update=full, more_follows=len(nxt) == steps.n_critic)
        if not full:
            break  # the loader ran dry inside the group: its critic updates stand, no generator update, epoch over
        c_sum, g_sum = c_sum + c, g_sum + g
        if epoch == 0 and n_iter == 0:
            steps.evaluate("baseline", epoch, 0, c, g)
        if n_iter == mid:
            steps.evaluate("mid", epoch, n_iter, c, g)
        n_iter += 1
    return {"iterations": n_iter, "c_loss": c_sum / max(1, n_iter), "g_loss": g_sum / max(1, n_iter)}


def run_epoch_seeded(steps: Steps, epoch: int, gen_loader, critic_loader) -> Dict[str, float]:
    """10p_train.py:270-386 for one epoch."""
    gen_it, critic_holder = iter(gen_loader), {"it": iter(critic_loader)}
    mid = len(gen_loader) // 2

    def next_critic():
        try:
            return next(critic_holder["it"])
        except StopIteration:  # the critic iterator starts over (10p_train.py:302-306)
            critic_holder["it"] = iter(critic_loader)
            return next(critic_holder["it"])

    n_iter, c_sum, g_sum = 0, 0.0, 0.0
    while True:
        gens = _take(gen_it, steps.n_critic)
        reals = [next_critic() for _ in gens]   # (a critic batch is drawn before the generator batch that may be missing:
        if len(gens) < steps.n_critic:          #  one more draw from the critic iterator, which nothing observes)
            if gens:
                steps.seeded_iteration(reals, gens, None)
            break
        last = _take(gen_it, 1)
        c, g = steps.seeded_iteration(reals, gens, last[0] if last else None)
        c_sum += c
        if not last:
            break  # no batch left for the generator update: the epoch ends with this group's critic loss counted
        g_sum += g
        if epoch == 0 and n_iter == 0:
            steps.evaluate("baseline", epoch, 0, c, g)
        if n_iter == mid:
            steps.evaluate("mid", epoch, n_iter, c, g)
        n_iter += 1
    return {"iterations": n_iter, "c_loss": c_sum / max(1, n_iter), "g_loss": g_sum / max(1, n_iter)}


def fit(steps: Steps, mode: str, n_epochs: int, lr_critic: float, first_epoch: int = 0) -> List[Dict[str, float]]:
    """The epoch loop with the freeze schedule.  first_epoch > 0: a resumed run (already past the frozen epoch)."""
    history = []
    steps.set_frozen(first_epoch == 0)
    steps.set_critic_lr(WARMUP_HEAD_LR if first_epoch == 0 else lr_critic)
    for epoch in range(first_epoch, n_epochs):
        if epoch == 1:  # un-freeze both backbones, critic back to its long-term rate
            steps.set_frozen(False)
            steps.set_critic_lr(lr_critic)
        loaders = steps.new_epoch_loaders(epoch)
        stats = run_epoch_blind(steps, epoch, loaders) if mode == "blind" else run_epoch_seeded(steps, epoch, *loaders)
        steps.evaluate("end", epoch, stats["iterations"], stats["c_loss"], stats["g_loss"])
        steps.save(epoch)
        history.append(stats)
    return history


# ----------------------------------------------------------------------------- the engine behind the interface
class EngineSteps(Steps):
    def __init__(self, pg, args, generator, critic, gen_opt, critic_opt, data, pool, tokenizer, device, rank, world,
                 grad_sync, log=print, gen_grad_sync=None, first_epoch=0):
        from . import train as T
        self.pg, self.T, self.args = pg, T, args
        self.generator, self.critic, self.gen_opt, self.critic_opt = generator, critic, gen_opt, critic_opt
        self.data, self.pool, self.tokenizer, self.device = data, pool, tokenizer, device
        self.rank, self.world, self.grad_sync, self.log = rank, world, grad_sync, log
        self.gen_grad_sync = gen_grad_sync
        self.soft = bool(getattr(args, "soft_generator_update", False))
        self.stacked = not bool(getattr(args, "reference_draw_order", False))
        # the next iteration's fakes beside this iteration's updates (blind mode, stacked fill, token-id generator update)
        self.ahead = None
        if (args.mode == "blind" and self.stacked and not bool(getattr(args, "soft_generator_update", False))
                and not bool(getattr(args, "no_fill_ahead", False))):
            self.ahead = T.FillAhead(device, low_priority=world <= 1)  # (see train.FILL_STREAM_LOW_PRIORITY)
        if self.soft and not self.stacked:
            raise ValueError("--soft_generator_update needs the stacked fill: drop --reference_draw_order")
        self.n_critic = args.n_critic
        self.hp = T.GanHyper(n_critic=args.n_critic, lambda_gp=args.lambda_gp, max_len=args.max_len,
                             gen_batch_blind=GEN_BATCH_BLIND)
        # a resumed run continues with streams of its own (the epoch it starts at is folded into every seed): seeded
        # from args.seed alone it would replay the first epoch's permutations and length draws
        salt = 7919 * first_epoch
        self.py_rng = random.Random(args.seed + salt + rank)
        # the same on every rank: ranks share one permutation of the data and one draw of the global fake lengths
        self.data_rng = torch.Generator().manual_seed(args.seed + salt)
        self.len_rng = random.Random(args.seed + salt) if world > 1 else self.py_rng
        self.updates, self.t0 = 0, time.perf_counter()

    def set_frozen(self, frozen):
        self.T.freeze_protbert(self.generator, frozen)
        self.T.freeze_protbert(self.critic, frozen)

    def set_critic_lr(self, lr):
        for g in self.critic_opt.param_groups:
            g["lr"] = lr

    def new_epoch_loaders(self, epoch):
        from .data import BatchLoader, dynamic_split
        kw = dict(device=self.device, generator=self.data_rng, rank=self.rank, world=self.world)
        if self.args.mode == "blind":
            return BatchLoader(self.data, self.args.batch_size, **kw)
        gen_idx, critic_idx = dynamic_split(len(self.data), self.args.n_critic, self.data_rng)
        return (BatchLoader(self.data, self.args.batch_size, subset=gen_idx, **kw),
                BatchLoader(self.data, self.args.batch_size, subset=critic_idx, **kw))

    def _sync(self):
        return self.grad_sync if self.world > 1 else None

    def blind_iteration(self, real_batches, with_generator_update, more_follows=False):
        T, hp = self.T, self.hp
        if with_generator_update:
            c, g = T.gan_step_blind(self.generator, self.critic, self.gen_opt, self.critic_opt, list(real_batches), self.pool,
                                    hp, self.args.batch_size, self.device, rng=self.len_rng, grad_sync=self._sync(),
                                    soft_generator_update=self.soft,
                                    gen_grad_sync=self.gen_grad_sync if self.world > 1 else None,
                                    rank=self.rank, world=self.world, stacked_generation=self.stacked,
                                    fill_ahead=self.ahead, more_follows=more_follows)
            self.updates += len(real_batches)
            return float(c), float(g)
        # the incomplete last group of an epoch: its critic updates only
        lengths = [T.sample_fake_lengths(self.pool, self.args.batch_size, self.len_rng, self.rank, self.world)
                   for _ in real_batches]
        temps = [T.fill_temperatures(hp) for _ in real_batches]
        fakes = T.generate_fake_batches(self.generator, lengths, temps, hp, self.device)
        for batch, fake, fl in zip(real_batches, fakes, lengths):
            T.critic_update(self.critic, self.critic_opt, batch["input_ids"], batch["attention_mask"], fake, hp,
                            grad_sync=self._sync(), real_lengths=batch.get("lengths"), fake_lengths=fl)
        self.updates += len(real_batches)
        return None, None

    def seeded_iteration(self, real_batches, gen_batches, gen_update_batch):
        T, hp = self.T, self.hp
        n = len(real_batches)
        if n == hp.n_critic and gen_update_batch is not None:
            c, g = T.gan_step_seeded(self.generator, self.critic, self.gen_opt, self.critic_opt, list(real_batches),
                                     list(gen_batches) + [gen_update_batch], hp, grad_sync=self._sync(),
                                     soft_generator_update=self.soft, stacked_generation=self.stacked,
                                     gen_grad_sync=self.gen_grad_sync if self.world > 1 else None)
            self.updates += n
            return float(c), float(g)
        fakes = T.generate_fakes_for_batches(self.generator, [gb["input_ids"] for gb in gen_batches], hp)
        c_sum = None
        for batch, fake, gb in zip(real_batches, fakes, gen_batches):
            c = T.critic_update(self.critic, self.critic_opt, batch["input_ids"], batch["attention_mask"], fake, hp,
                                grad_sync=self._sync(), real_lengths=batch.get("lengths"), fake_lengths=gb.get("lengths"))
            c_sum = c if c_sum is None else c_sum + c
        self.updates += n
        return (float(c_sum) / hp.n_critic if n == hp.n_critic else None), None

    def evaluate(self, tag, epoch, batch, c_loss, g_loss):
        seqs = []
        if self.rank == 0 and self.args.num_eval_sequences > 0:
            seqs = self.T.generate_eval_sequences(self.generator, self.tokenizer, self.args.num_eval_sequences, self.pool,
                                                  self.device, rng=self.py_rng)
        if self.rank == 0:
            dt = time.perf_counter() - self.t0
            self.log(json.dumps({"tag": tag, "epoch": epoch + 1, "batch": batch, "critic_loss": c_loss, "generator_loss": g_loss,
                                 "critic_updates": self.updates, "critic_updates_per_s": round(self.updates / max(dt, 1e-9), 4),
                                 "mean_eval_len": (sum(map(len, seqs)) / len(seqs)) if seqs else None,
                                 "eval_sequences": seqs[:2]}))

    def save(self, epoch):
        if self.rank != 0:
            return
        d = os.path.join(self.args.checkpoint_dir, self.args.run_name, f"epoch_{epoch + 1}")
        os.makedirs(d, exist_ok=True)
        self.critic.protbert.save_pretrained(os.path.join(d, "critic_bert"))
        torch.save(self.critic.classifier.state_dict(), os.path.join(d, "critic_classifier.pth"))
        self.generator.protbert.save_pretrained(os.path.join(d, "generator_bert"))
        if not self.args.no_optimizer_state:
            torch.save(self.critic_opt.state_dict(), os.path.join(d, "critic_optimizer.pt"))
            if self.soft:  # (otherwise the generator's optimizer never holds a moment)
                torch.save(self.gen_opt.state_dict(), os.path.join(d, "generator_optimizer.pt"))
        with open(os.path.join(d, "trainer_state.json"), "w") as fh:
            json.dump({"epoch": epoch + 1, "mode": self.args.mode, "critic_updates": self.updates}, fh)
        self.log(f"epoch {epoch + 1}: checkpoint in {d}")


# ----------------------------------------------------------------------------- command line
def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="ProtGen GAN training on the MI355X engine (blind = fully masked, seeded = 10 % kept)")
    ap.add_argument("--mode", choices=["blind", "seeded"], default="blind")
    for flag, kind, default, text in (
            ("--n_critic", int, 8, "critic updates per generator update"), ("--lambda_gp", float, 5.0, "gradient penalty weight"),
            ("--lr_gen", float, 5e-5, "generator learning rate"), ("--lr_critic", float, 5e-5, "critic learning rate"),
            ("--wd_gen", float, 0.01, "generator weight decay"), ("--wd_critic", float, 0.01, "critic weight decay"),
            ("--n_epochs", int, 20, "training epochs"), ("--run_name", str, None, "run name (checkpoint sub-directory)"),
            ("--batch_size", int, 8, "per-GPU batch size"), ("--num_eval_sequences", int, 10, "sequences generated per evaluation"),
            ("--eval_batch_size", int, 4, "(kept for flag compatibility: the structural evaluation it sized is out of scope)")):
        ap.add_argument(flag, type=kind, default=default, help=text)
    src = os.environ.get("SOURCE_DIR", "")
    ap.add_argument("--model_dir", default=os.path.join(src, "models", "dnmt_protbert") if src else None,
                    help="HuggingFace-format ProtBERT directory both models start from (default $SOURCE_DIR/models/dnmt_protbert)")
    ap.add_argument("--init_random", action="store_true", help="seeded random ProtBERT-shaped weights instead of --model_dir")
    ap.add_argument("--layers", type=int, default=30, help="with --init_random: encoder layers")
    ap.add_argument("--data_file", default="data/dnmt_full.txt", help="training sequences, one per line")
    ap.add_argument("--lengths_file", default="data/dnmt_unformatted.txt", help="blind mode: file whose line lengths are sampled")
    ap.add_argument("--checkpoint_dir", default=os.path.join(src, "checkpoints") if src else "./checkpoints")
    ap.add_argument("--max_len", type=int, default=512)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--resume", default=None, help="an epoch_<n> checkpoint directory of a previous run")
    ap.add_argument("--no_fill_ahead", action="store_true",
                    help="blind mode: fill an iteration's fake batches at its start.  By default they are filled on a second "
                         "stream while the previous iteration's updates run (the generator does not change in between); same "
                         "fakes and losses, except that an evaluation point between two iterations then draws its random "
                         "numbers after the next iteration's lengths and temperatures instead of before them")
    ap.add_argument("--reference_draw_order", action="store_true",
                    help="fill each fake batch right before the update that uses it, as the reference loop does, so the CPU "
                         "generator hands out temperatures and gradient-penalty alphas in the reference's order.  The default "
                         "fills all fake batches of an iteration as one stacked batch (faster, same tokens per batch on the "
                         "same noise) and therefore draws every temperature of the iteration before its first alpha")
    ap.add_argument("--no_optimizer_state", action="store_true", help="checkpoints as the reference writes them (weights only)")
    ap.add_argument("--soft_generator_update", action="store_true",
                    help="generator update through soft embeddings and the critic's 3-D branch (gradients reach the generator) "
                         "instead of the reference loop's update on token ids (which reaches the critic only)")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.run_name is None:
        args.run_name = "mn5_default_full_run" if args.mode == "blind" else "mn5_default_10p_run"
    import importlib
    pg = importlib.import_module(__package__)  # (the package registers itself through pgload.py: see train_gan.py)
    from . import data as D
    from . import distributed as pgd
    from . import train as T  # noqa: F401

    rank, local, world = pgd.init_from_env()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    first_epoch = 0
    if args.resume:
        gen_dir, critic_dir = os.path.join(args.resume, "generator_bert"), os.path.join(args.resume, "critic_bert")
        with open(os.path.join(args.resume, "trainer_state.json")) as fh:
            first_epoch = int(json.load(fh)["epoch"])
    else:
        gen_dir = critic_dir = args.model_dir
    torch.manual_seed(args.seed + 7919 * first_epoch + rank)
    if args.init_random and not args.resume:
        cfg = pg.ProtBertConfig(num_hidden_layers=args.layers)
        gen_bert = pg.ProtBertForMaskedLM(cfg, device=device, compute_dtype=dtype).init_random(seed=0)
        critic_bert = pg.ProtBertForMaskedLM(cfg, device=device, compute_dtype=dtype).init_random(seed=0)
    else:
        if not gen_dir:
            raise EnvironmentError("no model directory: pass --model_dir (or set SOURCE_DIR), or --init_random")
        gen_bert = pg.ProtBertForMaskedLM.from_pretrained(gen_dir, device=device, compute_dtype=dtype)
        critic_bert = pg.ProtBertForMaskedLM.from_pretrained(critic_dir, device=device, compute_dtype=dtype)
    tok_dir = args.model_dir if args.model_dir and os.path.isdir(args.model_dir) else ""
    tokenizer = pg.ProtTokenizer.from_pretrained(tok_dir, do_lower_case=False) if tok_dir else pg.ProtTokenizer()
    generator = pg.Generator(protbert_model=gen_bert, mask_token_id=tokenizer.mask_token_id)
    critic = pg.Critic(protbert_model=critic_bert)
    if args.resume:
        critic.classifier.load_state_dict(torch.load(os.path.join(args.resume, "critic_classifier.pth"), weights_only=True))
    gen_opt = pg.AdamW(generator.parameters(), lr=args.lr_gen, betas=(0.9, 0.999), weight_decay=args.wd_gen)
    critic_opt = pg.AdamW(critic.parameters(), lr=args.lr_critic, betas=(0.9, 0.999), weight_decay=args.wd_critic)
    opt_state = os.path.join(args.resume, "critic_optimizer.pt") if args.resume else None
    if opt_state and os.path.exists(opt_state):
        critic_opt.load_state_dict(torch.load(opt_state, weights_only=True))
    gen_state = os.path.join(args.resume, "generator_optimizer.pt") if args.resume else None
    if gen_state and os.path.exists(gen_state):
        gen_bert._reach_head()  # the saved moments cover the MLM head
        gen_opt.load_state_dict(torch.load(gen_state, weights_only=True))

    data = D.TokenizedSet.from_file(args.data_file, tokenizer, args.max_len)
    pool = D.length_pool(args.lengths_file) if args.mode == "blind" else [n - 2 for n in data.lengths]
    sync = pgd.GradSync([critic.protbert.store, critic.classifier.store], world, overlap=critic.protbert,
                        position_rows=args.max_len)
    gen_sync = None
    if args.soft_generator_update:
        gen_sync = pgd.GradSync([generator.protbert.store], world, overlap=generator.protbert, position_rows=args.max_len)
    steps = EngineSteps(pg, args, generator, critic, gen_opt, critic_opt, data, pool, tokenizer, device, rank, world, sync,
                        log=lambda s: print(s, flush=True), gen_grad_sync=gen_sync, first_epoch=first_epoch)
    if rank == 0:
        print(f"{args.mode} mode, {len(data)} sequences, {world} GPU(s), batch {args.batch_size} per GPU, epochs "
              f"{first_epoch + 1}..{args.n_epochs}", flush=True)
    history = fit(steps, args.mode, args.n_epochs, args.lr_critic, first_epoch)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
