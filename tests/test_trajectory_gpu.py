"""Consecutive training iterations against the fp64 oracle (Missing 4): Adam moments and bias correction, BatchNorm
running statistics, the device-side step / RNG counters and the warm-up -> GAN switch at epoch 10 -> 11
(train_eval/train_gan.py:27,88; scripts/train.py:104-109) carried from one iteration into the next on BOTH sides -- no fresh state
between iterations."""
import copy

import numpy as np
import pytest
import torch

from oracle import ref_model as O
from tests.harness import (MAX_GATE_FLIPS, NEAR_TIE_FRESH, ZERO_GRAD_KEYS, assert_gate_flips_are_near_ties, build_models, grad_errors, make_args, rel, sample_idx,
                           tcn_gate_sides, to_device_inject, wav_gate_flips, wav_gate_sides)

pytestmark = pytest.mark.gpu

EPOCHS = (9, 10, 11, 11, 12)          # loss_warmup = 10: two warm-up-phase iterations, then the full GAN iteration


def _views(slab_obj):
    """name -> (exp_avg, exp_avg_sq) views of a slab."""
    out = {}
    for n, p, off in zip(slab_obj.names, slab_obj.params, slab_obj.offsets):
        out[n] = (slab_obj.m[off:off + p.numel()].view(p.shape), slab_obj.v[off:off + p.numel()].view(p.shape))
    return out


def test_five_consecutive_iterations_match_fp64_oracle(pkg, dev):
    V, S, B = 512, 17, 4
    gst0, dst0 = O.make_generator_state(3, V, S), O.make_discriminator_state(4)
    og, od = O.clone_state(gst0, torch.float64), O.clone_state(dst0, torch.float64)
    ga, da = {}, {}
    args, G, D = build_models(pkg, dev, gst0, dst0, V, S, make_args())
    tr = pkg.GanTrainer(G, D, args)
    tr.keep_tape = True
    flips_iter, tcn_flips_iter = [], []
    worst_loss, per_iter, grad_iter = 0.0, [], []
    real_g, real_d = {}, {}
    for it, epoch in enumerate(EPOCHS):
        text, audio, vid, poses = O.make_batch(500 + it, B, V, S)          # a new batch every iteration
        rand = O.Rand(seed=2017 + it)
        pre = O.wav_preacts(og, audio.double())                     # LeakyReLU pre-activations of the audio encoder on this iteration's weights
        before = copy.deepcopy((og, od, ga, da))                    # the oracle's complete training state at the start of the iteration
        O.relu_gate_log = {}
        try:
            oret, extra = O.train_iter_gan(og, od, ga, da, epoch, text, audio.double(), poses.double(), vid, rand, dict(O.HP), want_grads=True)
        finally:
            relu_log, O.relu_gate_log = O.relu_gate_log, None
        ret = tr.train_iter(epoch, text.to(dev), audio.to(dev), poses.to(dev), vid.to(dev), inject=to_device_inject(rand.rec, dev)).to_dict()
        assert sorted(ret) == sorted(oret), (it, ret, oret)
        assert ("gen" in ret) == (epoch > 10) and ("dis" in ret) == (epoch > 10)           # the switch happens between epochs 10 and 11
        fl = wav_gate_flips(tr.last_tape, pre)
        flips_iter.append(fl)
        # the text encoder's ReLUs of the differentiated call, as in the full-size test (round 6: a ReLU near-tie of block 2 flipped in iteration 4
        # once another kernel took the B = 4 products, and the block's gradients were 1.6e-2 off with every audio gate in place)
        tcn_sides, tcn_rep = tcn_gate_sides(tr.last_tape, relu_log, 1 if epoch > 10 else 0, B)
        del relu_log
        tcn_flips_iter.append(tcn_rep)
        if sum(f[2] for f in fl) or tcn_rep:
            # A LeakyReLU gate of the audio encoder on the other side of an fp64 NEAR-TIE (|pre-activation| < 2e-6, asserted right here, in every
            # iteration -- round 6: the former 1e-5 window "after a first flip" is gone).  Both sides are correct evaluations of the reference
            # there, but left alone the two weight trajectories drift apart through Adam and later iterations flip gates that are no ties at all.
            # As in the full-size test below, the oracle REPEATS the iteration from the same state and draws with exactly those gates on the
            # side the HIP path took; that run is the reference and the state carried on.
            assert all(f[2] == 0 or f[3] < NEAR_TIE_FRESH for f in fl), (it, fl)
            assert all(r[2] < r[3] for r in tcn_rep) and sum(r[1] for r in tcn_rep) <= MAX_GATE_FLIPS, (it, tcn_rep)
            for dst_, src_ in zip((og, od, ga, da), before):
                dst_.clear(); dst_.update(src_)
            O.wav_gate_override, O.relu_gate_override = wav_gate_sides(tr.last_tape, pre), tcn_sides
            try:
                oret, extra = O.train_iter_gan(og, od, ga, da, epoch, text, audio.double(), poses.double(), vid, O.Rand(seed=2017 + it), dict(O.HP),
                                               want_grads=True)
            finally:
                O.wav_gate_override = O.relu_gate_override = None
        del before
        e = max(abs(ret[k] - oret[k]) / max(abs(oret[k]), 1e-6) for k in oret)
        per_iter.append(e)
        # this iteration's gradients (the slabs hold them until the next backward zeroes them)
        _, Gg, _ = tr.G.views()
        ge, _, gk = grad_errors(Gg, extra["g_grads"])
        de, dk = 0.0, None
        if epoch > 10:
            _, Dg, _ = tr.D.views()
            de, _, dk = grad_errors(Dg, extra["d_grads"])
        grad_iter.append((ge, gk, de, dk))
        worst_loss = max(worst_loss, e)
        # entries whose gradient was real (not rounding noise) in EVERY iteration so far: Adam turns noise-level gradients into
        # arbitrary fractions of lr on both sides, and those entries then random-walk apart (harness.run_train_parity, e_step)
        for store, grads in ((real_g, extra["g_grads"]), (real_d, extra.get("d_grads", {}))):
            for k, g in grads.items():
                if g is None or k in ZERO_GRAD_KEYS:
                    continue
                r = g.abs() > 1e-4 * g.abs().max()
                store[k] = r if k not in store else (store[k] & r)
    print("per-iteration loss errors:", " ".join(f"{e:.1e}" for e in per_iter))
    for it, (ge, gk, de, dk) in enumerate(grad_iter):
        fl = flips_iter[it]
        print(f"  iteration {it} (epoch {EPOCHS[it]}): worst G gradient error {ge:.1e} ({gk}), D {de:.1e} ({dk}); audio-encoder LeakyReLU gates "
              f"that differ from the fp64 oracle's (layer 1, 2, 3): {[f[2] for f in fl]} of {[f[0] for f in fl]}, near-ties {[f[1] for f in fl]}; "
              f"text-encoder ReLU near-ties followed: {[(r[0], r[1], f'{r[2]:.1e}') for r in tcn_flips_iter[it]]}")
    # the wide tolerances below are unlocked by flipped gates ONLY when those are a handful of fp64 near-ties (|pre| < 2e-6 in EVERY iteration: the
    # oracle follows the HIP path's side of such a gate, so no iteration inherits a difference): anything else is a bug
    assert all(f[2] == 0 or f[3] < NEAR_TIE_FRESH for fl_ in flips_iter for f in fl_), flips_iter
    total_flips = assert_gate_flips_are_near_ties(flips_iter, "five-iteration trajectory")
    assert worst_loss <= 1e-4, per_iter

    # ---- optimiser state after the last iteration: step counters, first and second moments
    gs, ds = tr.G.slab.ensure(), tr.D.slab.ensure()
    assert int(gs.step.item()) == ga["step"] == len(EPOCHS)
    assert int(ds.step.item()) == da["step"] == sum(e > 10 for e in EPOCHS)             # the discriminator steps only after the warm-up
    # A flipped LeakyReLU gate in the audio encoder (a pre-activation within rounding of zero lands on the other side in fp32) moves the
    # gradients below it discontinuously -- by ~1 / sqrt(positions summed) of a tensor's max -- and every later
    # iteration inherits the difference through Adam.  So: every tensor OUTSIDE the audio encoder must stay within 1e-4 of the fp64
    # trajectory (1e-4 on exp_avg, 2e-4 on exp_avg_sq) if no gate flipped; if one did (the count is printed above) the audio encoder's
    # tensors within 5e-2 and the others, which see the changed audio features, within 5e-4.
    wk, bad = {}, []
    for net, slab_obj, ostate, real in (("G.", gs, ga, real_g), ("D.", ds, da, real_d)):
        mv = _views(slab_obj)
        for k, r in real.items():
            if "m." + k not in ostate or not bool(r.any()):
                continue
            m, v = mv[k]
            om, ov = ostate["m." + k], ostate["v." + k]
            em = float((m.double().cpu() - om)[r].abs().max() / om.abs().max().clamp_min(1e-30))
            ev = float((v.double().cpu() - ov)[r].abs().max() / ov.abs().max().clamp_min(1e-30))
            wk[net + k] = (em, ev)
            # (a flipped gate also moves the audio features the GRU reads, hence every tensor downstream, by a few 1e-5: measured <= 1.1e-4)
            tol = (5e-2 if k.startswith("audio_encoder") else 5e-4) if total_flips > 0 else 1e-4
            if em > tol or ev > 2 * tol:                       # the second moment is a square: twice the gradient's relative error
                bad.append((net + k, em, ev, tol))
    print(f"optimiser state after the last iteration (normalised max error per tensor, exp_avg / exp_avg_sq); {total_flips} flipped gates in all:")
    for k, e in sorted(wk.items(), key=lambda kv: -kv[1][0])[:6]:
        print(f"  {k}: {e[0]:.1e} / {e[1]:.1e}")
    worst_rest = max(max(e) for k, e in wk.items() if not k.startswith("G.audio_encoder"))
    print(f"  worst tensor outside the audio encoder: {worst_rest:.1e}")
    assert not bad, bad

    # ---- BatchNorm buffers after the last iteration (G advances 2 per warm-up iteration and 3 after, D 1 and 3)
    gsd, dsd = G.state_dict(), D.state_dict()
    for sd, o in ((gsd, og), (dsd, od)):
        for k in o:
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(o[k]), (k, int(sd[k]), int(o[k]))
            elif "running_var" in k:
                assert rel(sd[k], o[k]) <= 1e-4, (k, rel(sd[k], o[k]))
            elif "running_mean" in k:      # inherits the +-lr noise walk of the zero-gradient biases in front of it (harness.py)
                assert float((sd[k].double().cpu() - o[k]).abs().max()) <= 3e-3, k
    n_warm, n_post = sum(e <= 10 for e in EPOCHS), sum(e > 10 for e in EPOCHS)
    assert int(gsd["audio_encoder.feat_extractor.1.num_batches_tracked"]) == 2 * n_warm + 3 * n_post
    assert int(dsd["pre_conv.1.num_batches_tracked"]) == 1 * n_warm + 3 * n_post


def test_graph_replayed_five_times_equals_five_eager_iterations(pkg, dev, monkeypatch):
    """A GraphedGanStep replayed 5 times lands on the same parameters as 5 eager iterations started from the same state with the same
    device RNG seeds (draws come from the device-side Philox counters, which advance identically on both paths).
    The weight-gradient products run with their deterministic two-pass combine here (ops.TN_TWO_PASS_ROWS = 0): with the float-atomic combine
    the two runs differ by the ORDER of those sums, and at B = 8 whole tensors of the text encoder have rounding-noise gradients that Adam's
    first steps turn into +-lr moves -- one full-suite run in six then had 5-10 % of a conv's entries apart by > 0.01 lr (the run-to-run
    spread of the atomics has its own test, test_atomic_weight_gradient_combine_run_to_run_spread)."""
    monkeypatch.setattr(pkg.ops, "TN_TWO_PASS_ROWS", 0)
    V, S, B = 64, 9, 8
    gst, dst = O.make_generator_state(7, V, S), O.make_discriminator_state(8)
    text, audio, vid, poses = (t.to(dev) for t in O.make_batch(11, B, V, S))
    args, G, D = build_models(pkg, dev, gst, dst, V, S)
    tr = pkg.GanTrainer(G, D, args)
    step = pkg.GraphedGanStep(tr, 11, text, audio, poses, vid, warmup_iters=2)     # construction advances the state by two iterations
    snap = tr.snapshot()
    graph_losses = [step().to_dict() for _ in range(5)]
    Pg = {k: v.detach().clone() for k, v in tr.G.slab.views()[0].items()}
    Pd = {k: v.detach().clone() for k, v in tr.D.slab.views()[0].items()}
    bn_g = {k: v.detach().clone() for k, v in G.state_dict().items() if "running" in k or "tracked" in k}
    steps_g = (int(tr.G.slab.step.item()), int(tr.D.slab.step.item()), tr.G.rng.state.tolist(), tr.D.rng.state.tolist())
    tr.restore(snap)
    eager_losses = [tr.train_iter(11, text, audio, poses, vid).to_dict() for _ in range(5)]
    steps_e = (int(tr.G.slab.step.item()), int(tr.D.slab.step.item()), tr.G.rng.state.tolist(), tr.D.rng.state.tolist())
    assert steps_g == steps_e
    for a, b in zip(graph_losses, eager_losses):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(a[k])), (k, a[k], b[k])
    worst, n_off, n_all = 0.0, 0, 0
    rows = []
    for mine, ref, lr in ((tr.G.slab.views()[0], Pg, 5e-4), (tr.D.slab.views()[0], Pd, 1e-4)):
        for k, r in ref.items():
            if k in ZERO_GRAD_KEYS:
                continue
            d = (mine[k] - r).abs()
            rows.append((rel(mine[k], r), k, float(d.max()) / lr, float((d > 0.01 * lr).float().mean())))
            n_off += int((d > 0.01 * lr).sum()); n_all += d.numel()
            worst = max(worst, rows[-1][0])
    for e, k, dl, frac in sorted(rows, reverse=True)[:8]:
        print(f"  {k}: normalised {e:.1e}, max |diff| {dl:.2f} lr, fraction of entries off by > 0.01 lr: {frac:.1e}")
    print(f"graph x5 vs eager x5: worst normalised parameter difference {worst:.1e}")
    # Both runs are the SAME arithmetic up to the order of float atomic sums (~1e-7 relative on a gradient): almost every entry agrees to a
    # tiny fraction of a learning-rate step.  An entry whose gradient is itself rounding noise can take Adam's +-lr step in opposite
    # directions (seen once in ~3 runs: 4e-2 normalised on a bias tensor), so the bound that always holds is the step budget --
    # 5 steps of at most ~lr each way -- and the tight statement is about the bulk of the entries.
    # Inside the full suite (other memory layout, other atomic order than in a fresh process) one run in three had whole text-encoder convs --
    # at B = 8 most word ids are padding and most of their weight gradients are rounding noise -- with 5-10 % of the entries apart by more than
    # 0.01 lr, so a single tensor is held to 20 % and the network as a whole to 2 %.
    for e, k, dl, frac in rows:
        assert dl <= 12.0, (k, dl)                             # |diff| <= 2 x 5 steps x ~1.2 lr
        assert frac <= 0.2, (k, frac)
    assert n_off <= 2e-2 * n_all, (n_off, n_all)               # > 98 % of all entries within 0.01 lr
    assert sorted(r[0] for r in rows)[len(rows) // 2] <= 1e-5  # the median tensor agrees to 1e-5 normalised
    sd = G.state_dict()
    for k, r in bn_g.items():
        if "tracked" in k:
            assert int(sd[k]) == int(r), k
        elif "running_var" in k:
            assert rel(sd[k], r) <= 1e-5, k


class _LaunchLog:
    """Records which C entry points an iteration calls and, for the grouped GEMM launches, the kernel plan the library picks for them
    (tg_gemm_nt_kernel_plan / tg_gemm_tn_kernel_plan: 2 = the mover-wave kernels)."""

    def __init__(self, pkg):
        self.ops, self.lib = pkg.ops, pkg._lib.load()
        self.names, self.nt, self.tn = [], [], []

    def __enter__(self):
        import ctypes as C
        self._orig = self.ops.call

        def call(name, *args):
            self.names.append(name)
            if name == "tg_gemm_nt_group":
                arr, n = args[0], int(args[1])
                q0 = arr if n == 1 and not hasattr(arr, "__len__") else arr
                tm, tn = C.c_int32(0), C.c_int32(0)
                plan = int(self.lib.tg_gemm_nt_kernel_plan(q0, n, C.byref(tm), C.byref(tn)))
                first = arr._obj if hasattr(arr, "_obj") else arr[0]
                self.nt.append((plan, int(first.M), int(first.N), int(first.A.K), n))
            elif name == "tg_gemm_tn_group":
                arr, n = args[0], int(args[1])
                plan = int(self.lib.tg_gemm_tn_kernel_plan(arr, n))
                first = arr._obj if hasattr(arr, "_obj") else arr[0]
                self.tn.append((plan, int(first.M), int(first.N), n))
            return self._orig(name, *args)
        self.ops.call = call
        return self

    def __exit__(self, *exc):
        self.ops.call = self._orig
        return False


def test_full_size_trajectory_b128_dropout_on_matches_fp64_oracle(pkg, dev):
    """The configuration bench.py times (B = 128 clips per iteration, every dropout ON) stepped THREE consecutive iterations -- epochs 10, 11,
    11: one warm-up-phase iteration, then two full GAN iterations (train_eval/train_gan.py:13-103) -- against the fp64 oracle stepping the
    same state, with every random draw (dropout masks, eps, the speaker permutation) recorded by oracle.Rand and injected.  Compared per
    iteration: losses, every gradient's norm and 64 sampled entries; after the last one: Adam's step counters and moments (sampled),
    BatchNorm buffers and counters.  The launch log proves that this ran on the kernel set of the headline number: the mover-wave NT / TN
    GEMMs at the stacked forward's 13 056 rows and the cluster-synchronised recurrences at B = 384 / 128 (injected masks ride in the same
    epilogues as regenerated ones: test_full_size_iteration_regenerated_dropout_equals_stored_masks ties the two bit for bit)."""
    V, S, B = 2000, 17, 128
    epochs = (10, 11, 11)
    gst0, dst0 = O.make_generator_state(3, V, S), O.make_discriminator_state(4)
    og, od = O.clone_state(gst0, torch.float64), O.clone_state(dst0, torch.float64)
    ga, da = {}, {}
    args, G, D = build_models(pkg, dev, gst0, dst0, V, S, make_args())
    tr = pkg.GanTrainer(G, D, args)
    tr.keep_tape = True
    assert pkg.ops.get_math_mode() == "f32"
    flips_iter, report, tcn_flips = [], [], []
    real_g, real_d = {}, {}
    log = _LaunchLog(pkg)
    import copy
    for it, epoch in enumerate(epochs):
        text, audio, vid, poses = O.make_batch(700 + it, B, V, S)
        rand = O.Rand(seed=3017 + it)
        pre = O.wav_preacts(og, audio.double())
        before = copy.deepcopy((og, od, ga, da))                     # the oracle's complete training state at the start of the iteration
        O.relu_gate_log = {}
        try:
            oret, extra = O.train_iter_gan(og, od, ga, da, epoch, text, audio.double(), poses.double(), vid, rand, dict(O.HP), want_grads=True)
        finally:
            relu_log, O.relu_gate_log = O.relu_gate_log, None
        inj = to_device_inject(rand.rec, dev)
        with log:
            ret = tr.train_iter(epoch, text.to(dev), audio.to(dev), poses.to(dev), vid.to(dev), inject=inj).to_dict()
        del inj
        assert sorted(ret) == sorted(oret), (it, ret, oret)
        fl = wav_gate_flips(tr.last_tape, pre)
        flips_iter.append(fl)
        n_fl = sum(f[2] for f in fl)
        # the same for the text encoder's ReLUs (31 M gates per stacked forward; the differentiated call g2 is what the gradients see)
        tcn_sides, tcn_rep = tcn_gate_sides(tr.last_tape, relu_log, 1 if epoch > 10 else 0, B)
        del relu_log
        n_tcn = sum(r[1] for r in tcn_rep)
        tcn_flips.append(tcn_rep)
        if n_fl or n_tcn:
            # At this size the audio encoder evaluates 23 M LeakyReLU gates per forward, ~30 of them on pre-activations within 2e-6 of zero, where
            # the fp32 path may take the other side -- both sides are correct evaluations of the reference there, but the gradients below differ
            # by up to ~3e-3 and the two weight trajectories would drift apart (more flips every iteration).  The oracle therefore REPEATS the
            # iteration from the same state and the same draws with exactly those near-tie gates on the side the HIP path took
            # (oracle.wav_gate_override / relu_gate_override; asserted near-ties right below), and that run is the reference and the state
            # that is carried on.
            # (how many: 54 M gates, pre-activations of O(1) with a density of O(1) at zero, two paths whose weights agree to ~1e-7 after an
            # iteration or two -> of the order of ten per iteration; what is asserted is that EVERY one of them is a near-tie)
            assert (not n_fl or max(f[3] for f in fl) < NEAR_TIE_FRESH) and n_fl + n_tcn <= 8 * MAX_GATE_FLIPS, (it, fl, tcn_rep)
            assert all(r[2] < r[3] for r in tcn_rep), (it, tcn_rep)
            for dst_, src_ in zip((og, od, ga, da), before):
                dst_.clear(); dst_.update(src_)
            O.wav_gate_override, O.relu_gate_override = wav_gate_sides(tr.last_tape, pre), tcn_sides
            try:
                oret, extra = O.train_iter_gan(og, od, ga, da, epoch, text, audio.double(), poses.double(), vid, O.Rand(seed=3017 + it), dict(O.HP),
                                               want_grads=True)
            finally:
                O.wav_gate_override = O.relu_gate_override = None
        del before
        e_loss = max(abs(ret[k] - oret[k]) / max(abs(oret[k]), 1e-6) for k in oret)
        assert e_loss <= 1e-4, (it, ret, oret)
        rows = []
        for net, eng, grads in (("G.", tr.G, extra["g_grads"]), ("D.", tr.D, extra.get("d_grads") if epoch > 10 else None)):
            if grads is None:
                continue
            _, Gv, _ = eng.views()
            for k, r in grads.items():
                if r is None or k in ZERO_GRAD_KEYS:
                    continue
                mine = Gv[k].detach().double().cpu()
                e_n = abs(float(mine.norm()) - float(r.norm())) / (float(r.norm()) + 1e-30)
                idx = torch.from_numpy(sample_idx(r.numel(), 64))
                e_s = float((mine.reshape(-1)[idx] - r.reshape(-1)[idx]).abs().max() / r.abs().max().clamp_min(1e-30))
                rows.append((net + k, e_n, e_s))
        # (with the near-tie gates aligned, every tensor -- the audio encoder's included -- is held to 1e-4)
        bad = [(k, e_n, e_s) for k, e_n, e_s in rows if e_n > 1e-4 or e_s > 1e-4]
        wk = max(rows, key=lambda r: max(r[1], r[2]))
        report.append(f"iteration {it} (epoch {epoch}): loss error {e_loss:.1e}; worst gradient {wk[0]} norm {wk[1]:.1e} sampled {wk[2]:.1e}; "
                      f"audio-encoder gate flips (layer 1, 2, 3) {[f[2] for f in fl]} of {[f[0] for f in fl]}, near-ties {[f[1] for f in fl]}; "
                      f"text-encoder ReLU flips {[(r[0], r[1], f'{r[2]:.1e}') for r in tcn_rep]}")
        print(report[-1])
        assert not bad, (it, n_fl, n_tcn, bad)
        for store, grads in ((real_g, extra["g_grads"]), (real_d, extra.get("d_grads", {}))):
            for k, g in grads.items():
                if g is None or k in ZERO_GRAD_KEYS:
                    continue
                r = g.abs() > 1e-4 * g.abs().max()
                store[k] = r if k not in store else (store[k] & r)
    total_flips = sum(f[2] for fl_ in flips_iter for f in fl_)
    print(f"flipped near-tie gates in all: audio encoder {total_flips}, text encoder {sum(r[1] for rep_ in tcn_flips for r in rep_)}")

    # ---- the kernel set of the headline number ran: mover-wave NT products on the stacked forward's 3 * 128 * 34 rows, mover-wave weight
    # gradients, the cluster-synchronised recurrences (forward at B = 384 = kernel <MT = 2, NS = 3>, backward at B = 128)
    n_post = sum(e > 10 for e in epochs)                             # the warm-up-phase iteration stacks two generator calls (8 704 rows), the others three
    big_nt = [p for p in log.nt if p[1] == 3 * B * 34 and p[0] == 2]
    assert len(big_nt) == n_post * (4 + 8), log.nt                   # per post-warm-up iteration: 4 GRU projection groups + 8 text-encoder convs
    assert sum(1 for p in log.tn if p[0] == 2) >= 3 * 4, log.tn      # per iteration: the four GRU layers' weight-gradient groups (+ text encoder)
    assert log.names.count("tg_gru_forward_cluster_rows") == 3 * 4 and sum(log.names.count(n_) for n_ in ("tg_gru_backward_cluster", "tg_gru_backward_cluster_stats")) == 3 * 4
    assert "tg_gru_forward" not in log.names and "tg_gru_backward" not in log.names
    assert pkg._lib.load().tg_gru_cluster_supported(3 * B, 300) and pkg._lib.load().tg_gru_cluster_bwd_supported(B, 300)
    pkg.ops.check_async_errors()

    # ---- optimiser state after the last iteration
    gs, ds = tr.G.slab.ensure(), tr.D.slab.ensure()
    assert int(gs.step.item()) == ga["step"] == len(epochs)
    assert int(ds.step.item()) == da["step"] == sum(e > 10 for e in epochs)
    wk, bad = {}, []
    for net, slab_obj, ostate, real in (("G.", gs, ga, real_g), ("D.", ds, da, real_d)):
        mv = _views(slab_obj)
        for k, r in real.items():
            if "m." + k not in ostate or not bool(r.any()):
                continue
            m, v = mv[k]
            om, ov = ostate["m." + k], ostate["v." + k]
            em = float((m.double().cpu() - om)[r].abs().max() / om.abs().max().clamp_min(1e-30))
            ev = float((v.double().cpu() - ov)[r].abs().max() / ov.abs().max().clamp_min(1e-30))
            wk[net + k] = (em, ev)
            tol = 2e-4                                   # three iterations of gradients that agree to <= 5e-5 (measured 1.1e-4 on one GRU matrix)
            if em > tol or ev > 2 * tol:
                bad.append((net + k, em, ev, tol))
    print(f"optimiser state after iteration {len(epochs) - 1} (normalised max error, exp_avg / exp_avg_sq); {total_flips} flipped near-tie gates in all:")
    for k, e in sorted(wk.items(), key=lambda kv: -kv[1][0])[:6]:
        print(f"  {k}: {e[0]:.1e} / {e[1]:.1e}")
    assert not bad, bad

    # ---- BatchNorm buffers and counters
    gsd, dsd = G.state_dict(), D.state_dict()
    for sd, o in ((gsd, og), (dsd, od)):
        for k in o:
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(o[k]), (k, int(sd[k]), int(o[k]))
            elif "running_var" in k:
                assert rel(sd[k], o[k]) <= 1e-4, (k, rel(sd[k], o[k]))
            elif "running_mean" in k:
                assert float((sd[k].double().cpu() - o[k]).abs().max()) <= 3e-3, k
    n_warm, n_post = sum(e <= 10 for e in epochs), sum(e > 10 for e in epochs)
    assert int(gsd["audio_encoder.feat_extractor.1.num_batches_tracked"]) == 2 * n_warm + 3 * n_post
    assert int(dsd["pre_conv.1.num_batches_tracked"]) == 1 * n_warm + 3 * n_post


@pytest.mark.parametrize("B", [64, 128])
def test_deterministic_mode_runs_are_bit_identical(pkg, dev, B):
    """tg_set_deterministic(1) (ops.set_deterministic): every cross-workgroup combine in a fixed order -- two-pass weight gradients, bias
    gradients by fixed-order column sums, one-writer embedding scatters, the generic forms of the two fused backward kernels that combine by
    float atomics.  Two runs of five hipGraph replays from the SAME state must leave bit-identical weights, gradients, Adam moments, BatchNorm
    buffers and RNG counters (the reference on CPU is reproducible given a seed); five eager iterations from that state land on the same
    bits as well (same kernels, same order)."""
    # (B = 128: the headline batch -- size-dependent kernel choices such as the two-launch BatchNorm and the cluster recurrences are the ones
    # the benchmark runs)
    V, S = 256, 9
    gst, dst = O.make_generator_state(7, V, S), O.make_discriminator_state(8)
    text, audio, vid, poses = (t.to(dev) for t in O.make_batch(11, B, V, S))
    pkg.ops.set_deterministic(True)
    try:
        assert pkg.ops.deterministic()
        args, G, D = build_models(pkg, dev, gst, dst, V, S)
        tr = pkg.GanTrainer(G, D, args)
        step = pkg.GraphedGanStep(tr, 11, text, audio, poses, vid, warmup_iters=2)
        snap = tr.snapshot()
        runs = []
        for _ in range(2):
            tr.restore(snap)
            losses = [step().to_dict() for _ in range(5)]
            runs.append((losses, [t.detach().clone() for t in tr._state_tensors()]))
        tr.restore(snap)
        eager_losses = [tr.train_iter(11, text, audio, poses, vid).to_dict() for _ in range(5)]
        eager_state = [t.detach().clone() for t in tr._state_tensors()]
    finally:
        pkg.ops.set_deterministic(False)
    assert not pkg.ops.deterministic()
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    n_diff = [int((a != b).sum()) for a, b in zip(runs[0][1], runs[1][1])]
    assert sum(n_diff) == 0, n_diff                                    # bit for bit: weights, gradients, moments, counters, BatchNorm buffers
    n_diff_e = [int((a != b).sum()) for a, b in zip(runs[0][1], eager_state)]
    worst = max(rel(a.float(), b.float()) for a, b in zip(runs[0][1], eager_state) if a.is_floating_point() and float(b.abs().max()) > 0)
    print(f"deterministic mode: graph x5 vs graph x5: 0 differing entries; graph x5 vs eager x5: {sum(n_diff_e)} differing entries, worst normalised {worst:.1e}")
    assert worst <= 1e-6, (worst, n_diff_e)
    assert eager_losses == runs[0][0] or all(abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(b[k])) for a, b in zip(eager_losses, runs[0][0]) for k in a)


def _device_draws(pkg, tr, B, T=34):
    """Every random draw of the LAST iteration of `tr` (device RNG, nothing injected), regenerated from the Philox states it left behind -- a
    draw is a pure function of (state, site, element index): ops.Drop / tg_normal / tg_randperm give the same values again until the next
    iteration advances the states -- and renamed to the oracle's per-call names (oracle.ref_model.Rand)."""
    ops = pkg.ops
    inj = {}
    for eng, stacked in ((tr.G, {"g": ["g1", "g2", "g3"]}), (tr.D, {"d": ["d_real", "d_fake"], "d_out": ["d_out"]})):
        st, sites = eng.rng.state, eng.rng._sites
        dev = st.device
        for name, site in sites.items():
            if name == "perm":
                inj["perm"] = ops.randperm(torch.empty(B, dtype=torch.int64, device=dev), st, site).cpu()
                continue
            tag, what = name.split(".", 1)
            calls = stacked[tag]
            Bs = B * len(calls)
            if what == "eps":
                full = ops.normal(torch.empty(Bs, 16, device=dev), st, site)
                parts = {"eps": full}
            elif what == "emb_drop":
                parts = {"emb_drop": ops.Drop(st, site, 0.1, (Bs, T, 300)).materialize()}
            elif what == "tcn.drop":
                full = ops.Drop(st, site, 0.3, (8, Bs, T, 300)).materialize()
                parts = {f"tcn{j // 2}.drop{j % 2 + 1}": full[j] for j in range(8)}
            elif what == "gru.drop":                      # the fused form: one draw for the three inter-layer masks
                Td, C2 = (28, 128) if eng is tr.D else (T, 600)
                full = ops.Drop(st, site, 0.3, (3, Bs, Td, C2)).materialize()
                parts = {f"gru.drop{l}": full[l] for l in range(3)}
            elif what.startswith("gru.drop"):
                Td, C2 = (28, 128) if eng is tr.D else (T, 600)
                parts = {what: ops.Drop(st, site, 0.3, (Bs, Td, C2)).materialize()}
            else:
                raise AssertionError(f"unknown draw site {name}")
            for k, v in parts.items():
                for ci, call in enumerate(calls):
                    piece = v[ci * B:(ci + 1) * B]
                    inj[f"{call}.{k}"] = (piece.transpose(1, 2) if k.startswith("tcn") else piece).contiguous().double().cpu()
    return inj


def test_captured_default_step_at_bench_size_matches_fp64_oracle(pkg, dev):
    """What bench.py times, pinned directly: the hipGraph-captured DEFAULT iteration (GraphedGanStep: three streams, weight-gradient side rows,
    device RNG, dropout masks regenerated by their consumers, fp16 x 2 products) at B = 128, V = 20 000 words, S = 1 371 speaker rows, epoch 11,
    ONE replay from a known state -- against the fp64 oracle (train_eval/train_gan.py:13-103 restated) fed with the draws that replay used.
    The draws are read back from the device: an eager iteration from the same state uses the same Philox states, and every draw is regenerated
    from them by site (_device_draws); if the replay had drawn anything else its losses would disagree with the oracle's.  Gates: losses 1e-4,
    every gradient's norm and 64 sampled entries 1e-4, as everywhere."""
    from importlib import import_module
    GraphedGanStep = import_module(pkg.__name__ + ".train_gan").GraphedGanStep
    V, S, B = 20000, 1371, 128
    gst0, dst0 = O.make_generator_state(20, V, S), O.make_discriminator_state(21)
    text, audio, vid, poses = O.make_batch(1201, B, V, S)
    args, G, D = build_models(pkg, dev, gst0, dst0, V, S, make_args())
    tr = pkg.GanTrainer(G, D, args)
    assert pkg.ops.get_math_mode() == "f32"
    td, ad, pd, vd = text.to(dev), audio.to(dev), poses.to(dev), vid.to(dev)
    snap = tr.snapshot()
    step = GraphedGanStep(tr, 11, td, ad, pd, vd)                     # warm-up iterations + capture: advances weights, Adam and RNG counters
    tr.restore(snap)
    ret = step().to_dict()                                            # ONE replay from the snapshot's state
    torch.cuda.synchronize()
    pkg.ops.check_async_errors()
    g_graph = {k: v.detach().double().cpu() for k, v in tr.G.views()[1].items()}
    d_graph = {k: v.detach().double().cpu() for k, v in tr.D.views()[1].items()}
    # the same iteration eagerly, from the same state: same Philox states -> the replay's draws; its tape shows which gates the HIP path took
    tr.restore(snap)
    tr.keep_tape = True
    ret_eager = tr.train_iter(11, td, ad, pd, vd).to_dict()
    for k in ret:
        assert abs(ret[k] - ret_eager[k]) <= 1e-5 * max(1.0, abs(ret_eager[k])), (k, ret[k], ret_eager[k])      # (float-atomic order only)
    inj = _device_draws(pkg, tr, B)
    assert torch.equal(tr.last_tape["vid"][2 * B:].cpu(), vid[inj["perm"]]), "tg_randperm(site 'perm') is not the permutation the iteration used"
    keep = [float((inj[f"g{c}.tcn0.drop1"] > 0).double().mean()) for c in (1, 2, 3)]
    assert all(0.69 < k_ < 0.71 for k_ in keep), keep                 # real masks (p = 0.3), different per call
    assert not torch.equal(inj["g1.emb_drop"], inj["g2.emb_drop"])
    og, od = O.clone_state(gst0, torch.float64), O.clone_state(dst0, torch.float64)
    pre = O.wav_preacts(og, audio.double())
    before = copy.deepcopy((og, od))
    O.relu_gate_log = {}
    try:
        oret, extra = O.train_iter_gan(og, od, {}, {}, 11, text, audio.double(), poses.double(), vid, O.Rand(inject=inj), dict(O.HP), fast_gru=True,
                                       want_grads=True)
    finally:
        relu_log, O.relu_gate_log = O.relu_gate_log, None
    fl = wav_gate_flips(tr.last_tape, pre)
    tcn_sides, tcn_rep = tcn_gate_sides(tr.last_tape, relu_log, 1, B)
    del relu_log
    n_fl, n_tcn = sum(f[2] for f in fl), sum(r[1] for r in tcn_rep)
    if n_fl or n_tcn:
        # near-tie gates (asserted): the oracle repeats the iteration on the HIP path's side of them, as in the full-size trajectory test above
        assert (not n_fl or max(f[3] for f in fl) < NEAR_TIE_FRESH) and n_fl + n_tcn <= 8 * MAX_GATE_FLIPS, (fl, tcn_rep)
        assert all(r[2] < r[3] for r in tcn_rep), tcn_rep
        og, od = before
        O.wav_gate_override, O.relu_gate_override = wav_gate_sides(tr.last_tape, pre), tcn_sides
        try:
            oret, extra = O.train_iter_gan(og, od, {}, {}, 11, text, audio.double(), poses.double(), vid, O.Rand(inject=inj), dict(O.HP), fast_gru=True,
                                           want_grads=True)
        finally:
            O.wav_gate_override = O.relu_gate_override = None
    assert sorted(ret) == sorted(oret), (ret, oret)
    e_loss = max(abs(ret[k] - oret[k]) / max(abs(oret[k]), 1e-6) for k in oret)
    rows = []
    for net, mine_all, grads in (("G.", g_graph, extra["g_grads"]), ("D.", d_graph, extra["d_grads"])):
        for k, r in grads.items():
            if r is None or k in ZERO_GRAD_KEYS:
                continue
            mine = mine_all[k]
            e_n = abs(float(mine.norm()) - float(r.norm())) / (float(r.norm()) + 1e-30)
            idx = torch.from_numpy(sample_idx(r.numel(), 64))
            e_s = float((mine.reshape(-1)[idx] - r.reshape(-1)[idx]).abs().max() / r.abs().max().clamp_min(1e-30))
            rows.append((net + k, e_n, e_s))
    wk = max(rows, key=lambda r_: max(r_[1], r_[2]))
    print(f"captured default replay at B = 128, V = 20 000 vs fp64 oracle on its own draws: loss error {e_loss:.1e}; worst gradient {wk[0]} norm "
          f"{wk[1]:.1e} sampled {wk[2]:.1e}; near-tie gate flips: audio encoder {n_fl}, text encoder {n_tcn}")
    assert e_loss <= 1e-4, (ret, oret)
    bad = [r_ for r_ in rows if r_[1] > 1e-4 or r_[2] > 1e-4]
    assert not bad, bad
    eg = g_graph["text_encoder.embedding.weight"].norm(dim=1)
    oe = extra["g_grads"]["text_encoder.embedding.weight"].double().norm(dim=1)
    assert int((eg > 0).sum()) == int((oe > 0).sum()) and abs(float(eg[0]) - float(oe[0])) <= 1e-4 * float(oe[0])     # the PAD row's hot gradient, the touched rows
