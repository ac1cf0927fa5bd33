"""The reference's own iteration ORDER driven through the drop-in nn.Module API (Missing 2; INTEGRATION.md section 1's "runs
unchanged on these modules"): three `G(...)` calls, two `.detach()`s, `D(...)` three times, `loss.backward()` twice and two
`torch.optim.Adam.step()`s -- the call sequence of scripts/train_eval/train_gan.py:13-103 restated here over `hip.PoseGenerator` /
`hip.ConvDiscriminator`, with the random draws the REAL reference recorded for the g2 fixtures (dropout masks, eps, the speaker
permutation) replayed through the modules' test-only `_replay_draws` queues.  Held to the g2 tolerances: losses, every generator and
discriminator gradient, BatchNorm counters and the parameters after both optimiser steps.

Nothing of GanTrainer runs here: the modules' autograd bridge (modules._Bridge) and stock torch.optim.Adam on the slab views do."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from harness import O, ZERO_GRAD_KEYS, build_models, rel, sample_idx
from test_oracle_golden import unpack_masks

pytestmark = pytest.mark.gpu


def gan_iteration_in_reference_order(args, epoch, in_text, in_audio, target, vid, G, D, g_opt, d_opt, perm, on_d_step=None):
    """One iteration in the order of train_eval/train_gan.py:13-103 (a restatement for the test: losses in torch on the device, the
    speaker permutation supplied by the caller instead of torch.randperm so that the fixture's draw can be replayed)."""
    n_pre = args.n_pre_poses
    B, T, Dp = target.shape
    pre_seq = target.new_zeros(B, T, Dp + 1)                      # :20-22 seed poses + constraint bit
    pre_seq[:, :n_pre, :Dp] = target[:, :n_pre]
    pre_seq[:, :n_pre, Dp] = 1
    gan_phase = epoch > args.loss_warmup and args.loss_gan_weight > 0.0
    dis_error = None
    if gan_phase:                                                # :27-43 discriminator step
        d_opt.zero_grad()
        fake = G(pre_seq, in_text, in_audio, vid)[0]
        p_real = D(target, in_text)
        p_fake = D(fake.detach(), in_text)
        dis_error = -(torch.log(p_real + 1e-8) + torch.log(1 - p_fake + 1e-8)).mean()
        dis_error.backward()
        if on_d_step is not None:
            on_d_step()
        d_opt.step()
    g_opt.zero_grad()                                            # :47-92 generator step
    out, z, mu, logvar = G(pre_seq, in_text, in_audio, vid)
    huber = F.smooth_l1_loss(out / 0.1, target / 0.1) * 0.1
    p_out = D(out, in_text)
    gen_error = -torch.log(p_out + 1e-8).mean()
    out_rand, z_rand, _, _ = G(pre_seq, in_text, in_audio, vid[perm])
    pose_l1 = (F.smooth_l1_loss(out / 0.05, out_rand.detach() / 0.05, reduction="none") * 0.05).sum(dim=(1, 2))
    z_l1 = (z.detach() - z_rand.detach()).abs().mean(dim=1)
    div_reg = torch.clamp(-(pose_l1 / (z_l1 + 1.0e-5)), min=-1000).mean()
    kld = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
    loss = args.loss_regression_weight * huber + args.loss_kld_weight * kld + args.loss_reg_weight * div_reg
    if epoch > args.loss_warmup:
        loss = loss + args.loss_gan_weight * gen_error
    loss.backward()
    g_opt.step()
    ret = {"loss": args.loss_regression_weight * huber.item()}   # :94-102 (tensor truthiness: a term that is exactly 0 drops its key)
    if kld:
        ret["KLD"] = args.loss_kld_weight * kld.item()
    if div_reg:
        ret["DIV_REG"] = args.loss_reg_weight * div_reg.item()
    if gan_phase:
        ret["gen"] = args.loss_gan_weight * gen_error.item()
        ret["dis"] = dis_error.item()
    return ret


@pytest.mark.parametrize("label", ["warmup", "gan"])
def test_reference_iteration_order_through_module_api_matches_g2(pkg, dev, label):
    g = np.load(os.path.join(GOLDEN, f"g2_train_{label}.npz"), allow_pickle=False)
    epoch, V, S, B = int(g["epoch"]), int(g["n_words"]), int(g["n_speakers"]), 4
    gst, dst = O.make_generator_state(int(g["g_seed"]), V, S), O.make_discriminator_state(int(g["d_seed"]))
    text, audio, vid, poses = (t.to(dev) for t in O.make_batch(int(g["batch_seed"]), B, V, S))
    args, G, D = build_models(pkg, dev, gst, dst, V, S)
    G.train(); D.train()
    # the draws the reference made, call by call: the golden run's G calls are tagged g1 (D step), g2, g3; nn.GRU's internal dropout was
    # switched off there (its draws happen inside ATen and cannot be recorded) = all-ones masks here
    tags = ["g1", "g2", "g3"] if epoch > 10 else ["g2", "g3"]
    rec = unpack_masks(g, tags)
    for tg, e in zip(tags, g["eps"]):
        rec[f"{tg}.eps"] = torch.from_numpy(e)
    ones = lambda *s: torch.ones(*s, device=dev)
    for tg in tags:
        call = {}
        for k, v in rec.items():
            if k.startswith(tg + "."):
                v = v.float()
                call["g." + k[len(tg) + 1:]] = (v.transpose(1, 2) if ".tcn" in k else v).contiguous().to(dev)
        for l in range(3):
            call[f"g.gru.drop{l}"] = ones(B, 34, 600)
        G._replay_draws.append(call)
    for _ in range(3 if epoch > 10 else 1):                       # D(real), D(fake), D(out) -- the warm-up phase still runs D(out) (:55)
        D._replay_draws.append({f"d.gru.drop{l}": ones(B, 28, 128) for l in range(3)})
    perm = torch.from_numpy(g["perm"]).to(dev)
    g_opt = torch.optim.Adam(G.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(D.parameters(), lr=args.learning_rate * args.discriminator_lr_weight, betas=(0.5, 0.999))
    d_grads = {}

    def grab_d():
        d_grads.update({k: p.grad.detach().clone() for k, p in D.named_parameters() if p.grad is not None})
    ret = gan_iteration_in_reference_order(args, epoch, text, audio, poses, vid, G, D, g_opt, d_opt, perm, on_d_step=grab_d)
    pkg.ops.check_async_errors()
    assert not G._replay_draws and not D._replay_draws            # 3 (2) generator calls and 3 (1) discriminator calls were made

    assert sorted(ret) == list(g["loss_keys"]), (sorted(ret), list(g["loss_keys"]))
    for k, v in zip(g["loss_keys"], g["loss_vals"]):
        assert abs(ret[k] - v) <= 2e-5 * max(1.0, abs(v)), (label, k, ret[k], v)
    worst = 0.0
    g_real = {}
    for k, p in G.named_parameters():
        if O.is_tcn_alias(k) or "gg/" + k not in g.files:
            continue
        idx = sample_idx(p.numel())
        mine = p.grad.reshape(-1).cpu().numpy()[idx]
        ref = g["gg/" + k]
        if k in ZERO_GRAD_KEYS:
            assert float(np.abs(mine).max()) < 1e-4, k
            continue
        e = rel(mine, ref)
        worst = max(worst, e)
        assert e < 1e-4, (label, k, e)
        g_real[k] = np.abs(ref) > 1e-5 * np.abs(ref).max()
    d_real = {}
    if epoch > 10:
        assert d_grads
        for k, gr in d_grads.items():
            if k in ZERO_GRAD_KEYS or "dg/" + k not in g.files:
                continue
            ref = g["dg/" + k]
            e = rel(gr.reshape(-1).cpu().numpy()[sample_idx(gr.numel())], ref)
            worst = max(worst, e)
            assert e < 1e-4, (label, k, e)
            d_real[k] = np.abs(ref) > 1e-5 * np.abs(ref).max()
    # state after the iteration: BatchNorm counters (Q2: the generator's advance 2 / 3 times, the discriminator's 1 / 3 times), running
    # statistics, and the parameters after torch.optim.Adam's step where the gradient is real (Adam's first step turns rounding-noise
    # gradients into arbitrary fractions of lr on both sides: harness.run_train_parity)
    step_worst = 0.0
    for sd, pre, real, lr in ((G.state_dict(), "gp/", g_real, 5e-4), (D.state_dict(), "dp/", d_real, 1e-4)):
        for k, v in sd.items():
            if pre + k not in g.files:
                continue
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(g[pre + k]), (k, int(v), int(g[pre + k]))
            elif "running_var" in k:
                assert rel(v.reshape(-1).cpu().numpy()[sample_idx(v.numel())], g[pre + k]) < 1e-4, k
            elif k in real and bool(real[k].any()):
                mine = v.reshape(-1).cpu().numpy()[sample_idx(v.numel())]
                d = float(np.abs(mine.astype(np.float64) - g[pre + k])[real[k]].max()) / lr
                step_worst = max(step_worst, d)
                assert d < 2e-2, (label, k, d)
    n_g = int(G.state_dict()["audio_encoder.feat_extractor.1.num_batches_tracked"])
    n_d = int(D.state_dict()["pre_conv.1.num_batches_tracked"])
    assert (n_g, n_d) == ((3, 3) if epoch > 10 else (2, 1)), (n_g, n_d)
    print(f"module-API iteration ({label}): worst gradient error {worst:.1e}, worst post-step parameter error {step_worst:.1e} lr")


def test_reference_iteration_order_through_module_api_full_size_g3(pkg, dev):
    """The same call sequence at the benchmark's size against the reference's OWN B = 128 iteration (g3 fixture: epoch 11, every dropout off,
    the reference's eps / permutation draws): the module API then runs the kernels of the headline number -- mover-wave GEMMs at 4 352 and
    13 056... no stacking here: three separate 128-clip generator calls, the cluster recurrences at B = 128 -- under torch.optim.Adam.  Losses,
    every gradient's norm and 64 sampled entries; audio-encoder tensors with the near-tie allowance of the engine-level g3 test."""
    g = np.load(os.path.join(GOLDEN, "g3_train_b128.npz"), allow_pickle=False)
    V, S, B = int(g["n_words"]), int(g["n_speakers"]), int(g["batch"])
    gst, dst = O.make_generator_state(int(g["g_seed"]), V, S), O.make_discriminator_state(int(g["d_seed"]))
    text, audio, vid, poses = (t.to(dev) for t in O.make_batch(int(g["batch_seed"]), B, V, S))
    from harness import make_args
    args, G, D = build_models(pkg, dev, gst, dst, V, S, make_args(dropout_prob=0.0))
    G.train(); D.train()
    ones = lambda *s: torch.ones(*s, device=dev)
    for e in g["eps"]:
        call = {"g.eps": torch.from_numpy(e).to(dev), "g.emb_drop": ones(B, 34, 300)}
        for l in range(3):
            call[f"g.gru.drop{l}"] = ones(B, 34, 600)
        G._replay_draws.append(call)
    for _ in range(3):
        D._replay_draws.append({f"d.gru.drop{l}": ones(B, 28, 128) for l in range(3)})
    g_opt = torch.optim.Adam(G.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(D.parameters(), lr=args.learning_rate * args.discriminator_lr_weight, betas=(0.5, 0.999))
    ret = gan_iteration_in_reference_order(args, 11, text, audio, poses, vid, G, D, g_opt, d_opt, torch.from_numpy(g["perm"]).to(dev))
    pkg.ops.check_async_errors()
    assert not G._replay_draws and not D._replay_draws
    for k, v in zip(g["loss_keys"], g["loss_vals"]):
        assert abs(ret[k] - v) <= 2e-5 * max(1.0, abs(v)), (k, ret[k], v)
    bad, worst = [], 0.0
    for k, p in G.named_parameters():
        if k in ZERO_GRAD_KEYS or "ggn/" + k not in g.files:
            continue
        gr = p.grad
        e_n = abs(float(gr.double().norm()) - float(g["ggn/" + k])) / (float(g["ggn/" + k]) + 1e-30)
        mine = gr.reshape(-1).cpu().numpy()[sample_idx(gr.numel(), 64)]
        e_s = float(np.abs(mine - g["gg/" + k]).max()) / float(gr.abs().max())
        tol_s, tol_n = (5e-3, 1e-3) if k.startswith("audio_encoder") else (1e-4, 1e-4)      # (LeakyReLU near-ties: test_engine_gpu's g3 test)
        worst = max(worst, e_n if not k.startswith("audio_encoder") else 0.0)
        if e_n > tol_n or e_s > tol_s:
            bad.append((k, e_n, e_s))
    assert not bad, bad
    for sd, pre in ((G.state_dict(), "gp/"), (D.state_dict(), "dp/")):
        for k, v in sd.items():
            if k.endswith("num_batches_tracked") and pre + k in g.files:
                assert int(v) == int(g[pre + k]), k
    print(f"module-API iteration at B = 128 (g3): worst gradient-norm error outside the audio encoder {worst:.1e}")
