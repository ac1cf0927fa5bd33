"""Data-parallel gradient exchange on CPU: world_size 2 and 4, gloo backend.  Covers ddp.GradSync (bucket ranges in backward
order, asynchronous launch + wait, mean semantics), parameter broadcast and the trainer's sync-point routing, without
any HIP call (ParamSlab and GradSync are device-agnostic)."""
import importlib
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

PKG = "gesture-generation-from-trimodal-context_amd"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_generator(pkg):
    from tests.harness import make_args
    a = make_args()
    torch.manual_seed(0)
    return pkg.PoseGenerator(a, 27, 64, 300, None, pkg.Vocab.speakers(9))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        ddp = importlib.import_module(PKG + ".ddp")
        params = importlib.import_module(PKG + ".params")
        G = _make_generator(pkg)
        slab = params.ParamSlab(G)
        # replicas start different, broadcast makes them rank 0's
        if rank != 0:
            slab.flat.add_(float(rank))
        ddp.broadcast_parameters([slab])
        ref = _make_generator(pkg)
        ok = all(torch.equal(p.detach(), q.detach()) for p, q in zip(G.parameters(), ref.parameters()))
        # rank-dependent gradients; buckets in the backward order the trainer uses
        g = torch.Generator().manual_seed(100 + rank)
        slab.grad.copy_(torch.randn(slab.numel, generator=g))
        mine = slab.grad.clone()
        sync = ddp.GradSync(chunk_floats=1 << 18)
        buckets = (("out", "gru"), ("speaker_embedding", "speaker_mu", "speaker_logvar", "text_encoder"), ("audio_encoder",))
        covered = torch.zeros(slab.numel, dtype=torch.bool)
        for b in buckets:
            lo, hi = sync.bucket_range(slab, b)
            assert not covered[lo:hi].any()
            covered[lo:hi] = True
            sync.run(("bucket", slab, b))
        sync.run(("wait",))
        # every parameter element is in exactly one bucket (only alignment padding is left out)
        for p, off in zip(slab.params, slab.offsets):
            assert covered[off:off + p.numel()].all()
        expect = sum(torch.randn(slab.numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
        err = float((slab.grad - expect)[covered].abs().max())
        # whole-slab form (discriminator step)
        slab.grad.copy_(mine)
        sync.run(("all", slab))
        err2 = float((slab.grad - expect).abs().max())
        # the nn.Parameter .grad views see the reduced values (they alias the slab)
        p0 = slab.params[0]
        view_ok = torch.equal(p0.grad.reshape(-1), slab.grad[slab.offsets[0]:slab.offsets[0] + p0.numel()])
        out.put((rank, ok, err, err2, view_ok))
    finally:
        dist.destroy_process_group()


def _run_world(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, ok, err, err2, view_ok in res:
        assert ok and view_ok, (rank, ok, view_ok)
        assert err < 1e-6 and err2 < 1e-6, (rank, err, err2)


def test_grad_sync_world2_gloo():
    _run_world(2)


def test_grad_sync_world4_gloo():
    """Rehearsal of a wider job on CPU ranks: bucketed mean over four replicas, broadcast from rank 0."""
    _run_world(4)


def test_trainer_routes_sync_points(pkg, monkeypatch):
    """GanTrainer._sync: eager -> GradSync.run(action); capturing -> the cut callback gets the action instead."""
    tg = importlib.import_module(PKG + ".train_gan")

    class FakeSync:
        def __init__(self): self.seen = []
        def run(self, action): self.seen.append(action[0])

    tr = tg.GanTrainer.__new__(tg.GanTrainer)
    tr.grad_sync, tr._cut = FakeSync(), None
    tr._sync("all", "slab"); tr._sync("bucket", "slab", ("gru",)); tr._sync("wait")
    assert tr.grad_sync.seen == ["all", "bucket", "wait"]
    cuts = []
    tr._cut = cuts.append
    tr._sync("bucket", "slab", ("out", "gru"))
    assert cuts == [("bucket", "slab", ("out", "gru"))] and tr.grad_sync.seen == ["all", "bucket", "wait"]
    tr.grad_sync = None
    tr._sync("wait")            # single GPU: no-op


def test_frozen_word_embedding_leaves_optimiser_and_buckets(pkg):
    """args.freeze_wordembed=True (multimodal_context_net.py:40-41): the embedding has requires_grad=False, so the reference's
    optim.Adam(generator.parameters()) never steps it.  Here it sits behind the trainable prefix of the slab: outside the Adam range,
    outside every gradient bucket, no .grad."""
    import numpy as np
    from tests.harness import make_args
    ddp = importlib.import_module(pkg.__name__ + ".ddp")
    emb = np.random.RandomState(0).randn(40, 300).astype(np.float32)
    for freeze in (False, True):
        G = pkg.PoseGenerator(make_args(freeze_wordembed=freeze), 27, 40, 300, emb, pkg.Vocab.speakers(5))
        slab = G.engine.slab
        name = "text_encoder.embedding.weight"
        if not freeze:
            assert not slab.frozen and slab.n_train == slab.numel
            continue
        assert slab.frozen == {name} and slab.names[-1] == name and slab.n_train == slab.offsets[-1] == slab.numel - 40 * 300
        assert G.text_encoder.embedding.weight.grad is None and G.gru.weight_hh_l0.grad is not None
        lo, hi = ddp.GradSync.bucket_range(slab, ("speaker_embedding", "speaker_mu", "speaker_logvar", "text_encoder"))
        assert hi <= slab.n_train                                  # the 48 KB word table is not exchanged
        assert sorted(G.state_dict()) == sorted(pkg.PoseGenerator(make_args(), 27, 40, 300, emb, pkg.Vocab.speakers(5)).state_dict())


# ---------------------------------------------------------------------------------------------------------------------------------
# GanTrainer.train_iter end to end on two gloo ranks, the HIP engines replaced by CPU stand-ins: what is under test is the trainer's
# own exchange schedule (train_gan.py: 'all' on the discriminator slab between its backward and its Adam, the generator's buckets in
# backward order as on_ready names them, 'wait' before the generator's Adam) running over a REAL process group.
def _fake_trainer_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        ddp = importlib.import_module(PKG + ".ddp")
        tg = importlib.import_module(PKG + ".train_gan")
        from tests.harness import make_args
        args = make_args()
        G = _make_generator(pkg)
        D = pkg.ConvDiscriminator(27)
        order = []
        B = 4
        grad_of = lambda slab, seed: torch.randn(slab.numel, generator=torch.Generator().manual_seed(seed))
        BUCKETS = (("out", "gru"), ("speaker_embedding", "speaker_mu", "speaker_logvar", "text_encoder"), ("audio_encoder",))

        class FakeG:                                   # the parts of engine.GeneratorEngine the trainer touches
            z_mode, use_side_stream, in_size = "speaker", False, 108
            def __init__(self, mod): self.slab = importlib.import_module(PKG + ".params").ParamSlab(mod); self.rng = type("R", (), {"state": None, "site": lambda s, n: 1})()
            def forward(self, pre, text, audio, vid, **kw):
                n = pre.shape[0]
                return {"out": torch.zeros(n, 34, 27), "z": torch.zeros(n, 16), "mu": torch.zeros(n, 16), "logvar": torch.zeros(n, 16), "tape": {}}
            def backward(self, tape, d_out, d_mu, d_lv, *, b0, nb, on_ready):
                s = self.slab
                full = grad_of(s, 1000 + rank)          # this rank's gradient; a bucket's entries become final right before on_ready names it
                for bk in BUCKETS:
                    lo, hi = ddp.GradSync.bucket_range(s, bk)
                    s.grad[lo:hi] = full[lo:hi]
                    order.append(("ready", bk))
                    on_ready(bk, last=bk is BUCKETS[-1])     # as engine.GeneratorEngine.backward: the last bucket is named by the engine

        class FakeD:
            def __init__(self, mod): self.slab = importlib.import_module(PKG + ".params").ParamSlab(mod); self.rng = type("R", (), {"state": None})()
            def forward(self, poses, **kw): return {"logit": torch.zeros(poses.shape[0], 1), "tape": {}, "terms": torch.zeros(poses.shape[0])}
            def backward(self, tape, d_logit, **kw):
                if kw.get("param_grads", True):
                    self.slab.grad.copy_(grad_of(self.slab, 2000 + rank))
                    order.append(("d_backward",))

        class FakeOps:                                  # train_gan's direct kernel calls
            @staticmethod
            def iter_head(*a, **kw):
                ng = a[6]
                return torch.zeros(ng * B, 34, 28), torch.zeros(ng * B, 34, dtype=torch.long), torch.zeros(ng * B, dtype=torch.long)
            @staticmethod
            def gan_d_loss(*a): pass
            @staticmethod
            def gan_g_loss(*a): pass
            @staticmethod
            def zeros(*shape, device, dtype=torch.float32): return torch.zeros(*shape, dtype=dtype)
            @staticmethod
            def check_async_errors(): pass
            @staticmethod
            def zero_(t): return t.zero_()

        class FakeAdam:
            def __init__(self, eng, **kw): self.engine, self.seen = eng, None
            @property
            def slab(self): return self.engine.slab
            def step(self, counter_advanced=False):
                self.seen = self.engine.slab.grad.clone()            # what the optimiser would consume
                order.append(("adam", "G" if isinstance(self.engine, FakeG) else "D"))

        class FakePrep:
            def add_slab(self, *a): pass
            def refresh(self, *a): pass
            def active(self):
                import contextlib
                return contextlib.nullcontext()

        object.__setattr__(G, "_engine", FakeG(G))          # the modules build their engines lazily behind a property
        object.__setattr__(D, "_engine", FakeD(D))
        tg.ops, tg.FusedAdam = FakeOps, FakeAdam
        importlib.import_module(PKG + ".params").ops = FakeOps          # ParamSlab.zero_grad
        tg.L = type("FakeL", (), {"WeightPrep": FakePrep})
        sync = ddp.GradSync(chunk_floats=1 << 18)
        real_run = sync.run
        sync.run = lambda action: (order.append(("sync", action[0], tuple(action[2]) if action[0].startswith("bucket") else None)), real_run(action))[1]
        tr = tg.GanTrainer(G, D, args, grad_sync=sync)
        tr.train_iter(11, torch.zeros(B, 34, dtype=torch.long), torch.zeros(B, 100), torch.zeros(B, 34, 27), torch.zeros(B, dtype=torch.long))
        exp_g = sum(grad_of(G.engine.slab, 1000 + r) for r in range(world)) / world
        exp_d = sum(grad_of(D.engine.slab, 2000 + r) for r in range(world)) / world
        covered = torch.zeros(G.engine.slab.numel, dtype=torch.bool)
        for bk in BUCKETS:
            lo, hi = ddp.GradSync.bucket_range(G.engine.slab, bk)
            covered[lo:hi] = True
        err_g = float((tr.g_opt.seen - exp_g)[covered].abs().max())
        err_d = float((tr.d_opt.seen - exp_d).abs().max())
        out.put((rank, order, err_g, err_d, len(sync.pending)))
    finally:
        dist.destroy_process_group()


def test_train_iter_exchange_schedule_over_gloo_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fake_trainer_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    text_bucket = ("speaker_embedding", "speaker_mu", "speaker_logvar", "text_encoder")
    want = [("d_backward",), ("sync", "all", None), ("adam", "D"),
            ("ready", ("out", "gru")), ("sync", "bucket", ("out", "gru")),
            ("ready", text_bucket), ("sync", "bucket", text_bucket),
            ("ready", ("audio_encoder",)), ("sync", "bucket_wait", ("audio_encoder",)), ("adam", "G")]
    for rank, order, err_g, err_d, pending in res:
        assert order == want, (rank, order)
        assert err_g < 1e-6 and err_d < 1e-6 and pending == 0, (rank, err_g, err_d, pending)      # both optimisers consumed the MEAN gradient


def _bench_size_worker(rank, world, port, out):
    """Two data-parallel iterations of the generator's slab at the BENCHMARK's size (V = 20 000 words, 1 371 speaker rows: 13.2 M floats, the
    word embedding inside the {text, speaker} bucket) through the bucket ranges and the exchange order the trainer uses -- the D-slab form first,
    then {out, gru}, then {text, speaker} and {audio} -- with rank-dependent gradients and a host-side Adam on the reduced gradients."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        ddp = importlib.import_module(PKG + ".ddp")
        params = importlib.import_module(PKG + ".params")
        from tests.harness import make_args
        torch.manual_seed(rank)                                        # replicas start DIFFERENT: the broadcast must make them rank 0's
        G = pkg.PoseGenerator(make_args(), 27, 20000, 300, None, pkg.Vocab.speakers(1371))
        slab = params.ParamSlab(G)
        assert slab.numel >= 13_204_939
        ddp.broadcast_parameters([slab])
        sync = ddp.GradSync(chunk_floats=8 * 1024 * 1024)
        buckets = (("out", "gru"), ("speaker_embedding", "speaker_mu", "speaker_logvar", "text_encoder"), ("audio_encoder",))
        sizes = [sync.bucket_range(slab, b) for b in buckets]
        m, v = torch.zeros_like(slab.flat), torch.zeros_like(slab.flat)
        start = slab.flat.clone()
        covered = torch.zeros(slab.numel, dtype=torch.bool)
        for p_, off in zip(slab.params, slab.offsets):
            covered[off:off + p_.numel()] = True
        for it in range(2):
            g = torch.Generator().manual_seed(1000 * it + rank)
            # rank-dependent gradients on every parameter element; alignment padding between tensors stays zero, as zero_grad leaves it
            slab.grad.copy_(torch.randn(slab.numel, generator=g) * (1.0 + rank) * covered)
            sync.run(("bucket", slab, buckets[0]))                     # leaves behind the GRU stack's backward
            sync.run(("bucket", slab, buckets[1]))
            sync.run(("bucket_wait", slab, buckets[2]))                # the backward's last bucket: launched and awaited in one action
            assert not sync.pending
            # Adam(lr 5e-4, betas (0.5, 0.999)) on the reduced gradient, the same float ops on every rank
            m.mul_(0.5).add_(slab.grad, alpha=0.5)
            v.mul_(0.999).addcmul_(slab.grad, slab.grad, value=0.001)
            mh, vh = m / (1 - 0.5 ** (it + 1)), v / (1 - 0.999 ** (it + 1))
            slab.flat.sub_(5e-4 * mh / (vh.sqrt() + 1e-8))
        bits = slab.flat.view(torch.int32).clone()
        lo, hi = bits.clone(), bits.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        moved = float((slab.flat - start)[covered].abs().min())            # every parameter element took two Adam steps
        out.put((rank, bool(torch.equal(lo, hi)), [hi_ - lo_ for lo_, hi_ in sizes], moved > 0))
    finally:
        dist.destroy_process_group()


def test_bench_size_slab_stays_bit_equal_across_ranks_world2():
    """after two iterations of the real bucket ranges of a V = 20 000 slab, every rank holds BIT-IDENTICAL parameters (the
    replicas of scripts/train.py:93-96's DataParallel are one set of weights; here they stay one because every rank applies the same
    reduced gradient)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_size_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, equal, sizes, moved in res:
        assert equal and moved, (rank, equal, moved)
        assert sizes[1] > 6_000_000 and sum(sizes) >= 13_000_000, sizes          # the word embedding sits inside the {text, speaker} bucket
