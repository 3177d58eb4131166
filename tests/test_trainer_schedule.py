"""The trainer's schedule (protgen-gan_amd/train_cli.py) on the CPU, with recording stand-ins for the engine:
which batches feed which update, when an epoch ends, when the backbones are frozen and at which learning rate,
where evaluations and checkpoints fall (fully_masked_train.py:262-389, 10p_train.py:270-406), plus the native data
path (dataset.py:19-42,72-94)."""
import random

import pytest
import torch


@pytest.fixture(scope="module")
def pg():
    import pgload
    return pgload.load()


class Rec:
    def __init__(self, n_critic, loaders):
        self.n_critic, self.loaders, self.log = n_critic, loaders, []

    def set_frozen(self, frozen):
        self.log.append(("frozen", frozen))

    def set_critic_lr(self, lr):
        self.log.append(("lr", lr))

    def blind_iteration(self, reals, with_generator_update, more_follows=False):
        self.log.append(("blind", tuple(reals), with_generator_update))
        self.log.append(("more_follows", more_follows))
        return (1.0, 2.0) if with_generator_update else (None, None)

    def seeded_iteration(self, reals, gens, last):
        self.log.append(("seeded", tuple(reals), tuple(gens), last))
        return (1.0 if len(reals) == self.n_critic else None, 2.0 if last is not None else None)

    def evaluate(self, tag, epoch, batch, c, g):
        self.log.append(("eval", tag, epoch, batch))

    def save(self, epoch):
        self.log.append(("save", epoch))

    def new_epoch_loaders(self, epoch):
        self.log.append(("loaders", epoch))
        return self.loaders


def test_blind_epoch_groups_and_end(pg):
    from protgen_gan_amd import train_cli as TC
    s = Rec(3, None)
    stats = TC.run_epoch_blind(s, 0, list(range(7)))
    # two full groups with a generator update each, then one real batch left: its critic update stands, the epoch ends
    assert [e for e in s.log if e[0] == "blind"] == [("blind", (0, 1, 2), True), ("blind", (3, 4, 5), True), ("blind", (6,), False)]
    assert [e for e in s.log if e[0] == "eval"] == [("eval", "baseline", 0, 0)]   # "mid" needs iteration 7 // 2 = 3: never reached
    assert stats == {"iterations": 2, "c_loss": 1.0, "g_loss": 2.0}
    # only an iteration that another WHOLE group follows may start on that group's fakes
    assert [e[1] for e in s.log if e[0] == "more_follows"] == [True, False, False]
    s = Rec(3, None)
    TC.run_epoch_blind(s, 1, list(range(6)))   # a multiple of n_critic: the epoch ends on an empty group, no partial update
    assert [e[1] for e in s.log if e[0] == "blind"] == [(0, 1, 2), (3, 4, 5)] and not [e for e in s.log if e[0] == "eval"]
    s = Rec(1, None)
    TC.run_epoch_blind(s, 1, list(range(4)))   # n_critic = 1: iteration 2 is the middle one
    assert [e for e in s.log if e[0] == "eval"] == [("eval", "mid", 1, 2)]


def test_seeded_epoch_recycles_critic_batches_and_ends_on_generator_batches(pg):
    from protgen_gan_amd import train_cli as TC
    s = Rec(3, None)
    gen, critic = [f"g{i}" for i in range(7)], [f"c{i}" for i in range(4)]
    stats = TC.run_epoch_seeded(s, 0, gen, critic)
    seeded = [e for e in s.log if e[0] == "seeded"]
    # iteration 0: three critic updates (g0-g2) and a generator update on g3; then g4-g6 with the critic iterator
    # starting over after c3, and no batch left for the generator update: critic loss counted, epoch over
    assert seeded == [("seeded", ("c0", "c1", "c2"), ("g0", "g1", "g2"), "g3"),
                      ("seeded", ("c3", "c0", "c1"), ("g4", "g5", "g6"), None)]
    assert stats["iterations"] == 1 and stats["c_loss"] == 2.0 and stats["g_loss"] == 2.0
    s = Rec(3, None)
    TC.run_epoch_seeded(s, 1, gen[:5], critic)   # the generator loader runs dry inside the second group
    assert [e for e in s.log if e[0] == "seeded"][1] == ("seeded", ("c3",), ("g4",), None)


def test_fit_freeze_schedule_evaluations_and_checkpoints(pg):
    from protgen_gan_amd import train_cli as TC
    s = Rec(2, list(range(4)))
    TC.fit(s, "blind", 3, lr_critic=5e-5)
    marks = [e for e in s.log if e[0] in ("frozen", "lr", "loaders", "save")]
    assert marks == [("frozen", True), ("lr", 1e-5), ("loaders", 0), ("save", 0), ("frozen", False), ("lr", 5e-5), ("loaders", 1),
                     ("save", 1), ("loaders", 2), ("save", 2)]
    ends = [e for e in s.log if e[0] == "eval" and e[1] == "end"]
    assert ends == [("eval", "end", 0, 2), ("eval", "end", 1, 2), ("eval", "end", 2, 2)]
    s = Rec(2, list(range(4)))
    TC.fit(s, "blind", 3, lr_critic=5e-5, first_epoch=2)   # resumed after the frozen epoch
    assert [e for e in s.log if e[0] in ("frozen", "lr")] == [("frozen", False), ("lr", 5e-5)]


def test_cli_flags_are_the_references(pg):
    from protgen_gan_amd import train_cli as TC
    a = TC.build_parser().parse_args([])
    assert (a.n_critic, a.lambda_gp, a.lr_gen, a.lr_critic, a.wd_gen, a.wd_critic, a.n_epochs, a.batch_size,
            a.num_eval_sequences, a.eval_batch_size) == (8, 5.0, 5e-5, 5e-5, 0.01, 0.01, 20, 8, 10, 4)


def test_tokenized_set_loader_and_split(pg, tmp_path):
    from protgen_gan_amd import data as D
    tok = pg.ProtTokenizer()
    rng = random.Random(0)
    seqs = ["".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(rng.randint(5, 40))) for _ in range(23)]
    p = tmp_path / "d.txt"
    p.write_text("\n".join(" ".join(s) if i % 2 else s for i, s in enumerate(seqs)) + "\n\n")   # spaced and unspaced lines
    ds = D.TokenizedSet.from_file(str(p), tok, max_length=32)
    assert len(ds) == 23 and ds.ids.shape == (23, 32) and ds.mask.dtype == torch.float32
    for i, s in enumerate(seqs):
        n = min(len(s), 30) + 2
        assert ds.lengths[i] == n and int(ds.mask[i].sum()) == n
        assert ds.ids[i, 0] == tok.cls_token_id and ds.ids[i, n - 1] == tok.sep_token_id and bool((ds.ids[i, n:] == tok.pad_token_id).all())
        assert tok.decode(ds.ids[i], skip_special_tokens=True).replace(" ", "") == s[:30]
    g = torch.Generator().manual_seed(1)
    ld = D.BatchLoader(ds, 4, "cpu", generator=g)
    seen = [tuple(b["input_ids"][r].tolist()) for b in ld for r in range(4)]
    assert len(ld) == 5 and len(seen) == 20 and len(set(seen)) == 20   # incomplete last batch dropped, no row twice
    b = next(iter(ld))
    assert b["lengths"] == [int(v) for v in b["attention_mask"].sum(dim=1).tolist()]
    # two ranks, same seed: disjoint shares of the same global batches
    parts = []
    for rank in range(2):
        ld = D.BatchLoader(ds, 2, "cpu", generator=torch.Generator().manual_seed(5), rank=rank, world=2)
        parts.append([tuple(b["input_ids"][r].tolist()) for b in ld for r in range(2)])
    assert len(parts[0]) == len(parts[1]) == 10 and not set(parts[0]) & set(parts[1])
    gen_idx, critic_idx = D.dynamic_split(23, 4, torch.Generator().manual_seed(2))
    assert len(gen_idx) == 23 // 5 and sorted(gen_idx + critic_idx) == list(range(23))
    assert D.length_pool(str(p)) == [len(s) for s in seqs]
