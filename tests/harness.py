"""Shared parity harness (tests, __graft_entry__.smoke, bench cpu_baseline): builds the HIP-backed modules from the
oracle's deterministic weights, replays the oracle's recorded random draws on the GPU path and compares."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_model as O  # noqa: E402

# parameters whose true gradient is exactly zero (they feed a train-mode BatchNorm through linear maps only):
# both sides hold rounding noise there, and Adam turns noise into +-lr steps
ZERO_GRAD_KEYS = {"audio_encoder.feat_extractor.0.bias", "audio_encoder.feat_extractor.3.bias", "audio_encoder.feat_extractor.6.bias",
                  "pre_conv.0.bias", "pre_conv.3.bias", "pre_conv.1.bias", "pose_encoder.net.0.0.bias", "pose_encoder.net.1.0.bias",
                  "pose_encoder.net.2.0.bias", "pose_encoder.out_net.0.bias", "pose_encoder.out_net.3.bias", "decoder.pre_net.0.bias",
                  "decoder.net.0.bias", "decoder.net.3.bias", "pose_encoder.net.3.bias", "pose_encoder.out_net.1.bias",
                  "pose_encoder.out_net.4.bias", "pose_encoder.out_net.6.bias", "pose_encoder.fc_mu.bias"}


def make_args(**over):
    """The Namespace the reference's parse_args builds from config/multimodal_context.yml (package config.py reads the YAML)."""
    import importlib
    cfg = importlib.import_module("gesture-generation-from-trimodal-context_amd.config")
    return cfg.load_config("multimodal_context", **over)


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def z_obj_for(pkg, z_type, n_speakers):
    """train.py:82-87: the speaker Vocab, 1 (random noise) or None."""
    return pkg.Vocab.speakers(n_speakers) if z_type == "speaker" else (1 if z_type == "random" else None)


def build_models(pkg, dev, gst, dst, n_words, n_speakers, args=None):
    args = args or make_args()
    G = pkg.PoseGenerator(args, 27, n_words, 300, None, z_obj_for(pkg, args.z_type, n_speakers))
    D = pkg.ConvDiscriminator(27)
    G.load_state_dict(O.clone_state(gst, torch.float32), strict=True)
    D.load_state_dict(O.clone_state(dst, torch.float32), strict=True)
    return args, G.to(dev), D.to(dev)


def to_device_inject(rec, dev):
    """Oracle-recorded draws -> GPU-path layout: TCN masks are (B, C, T) in the oracle, (B, T, C) here."""
    inj = {}
    for k, v in rec.items():
        if k == "perm":
            inj[k] = v.to(dev)
        elif ".tcn" in k:
            inj[k] = v.float().transpose(1, 2).contiguous().to(dev)
        else:
            inj[k] = v.float().contiguous().to(dev)
    return inj


def grad_errors(mine, ref):
    """(worst normalised error over real gradients, worst |grad| over zero-by-construction ones, worst key)"""
    worst, wkey, zmax = 0.0, None, 0.0
    for k, r in ref.items():
        if r is None:
            continue
        if k in ZERO_GRAD_KEYS:
            zmax = max(zmax, float(mine[k].abs().max()))
            continue
        e = rel(mine[k], r)
        if e > worst:
            worst, wkey = e, k
    return worst, zmax, wkey


def run_train_parity(pkg, dev, batch=4, epochs=(0, 11), n_words=512, n_speakers=17, seed=77, verbose=False, dropout=True,
                     input_context="both", z_type="speaker", rand_seed=1017, check_step=True, n_layers=4):
    """Oracle (fp64, CPU) and HIP path on identical weights, inputs and random draws, one iteration per epoch value,
    fresh models each.  Returns the worst normalised error over losses, gradients, BN buffers and updated parameters."""
    worst = 0.0
    z_mode = z_type if z_type in ("speaker", "random") else None
    gst0 = O.make_generator_state(3, n_words, n_speakers, input_context=input_context, z_mode=z_mode, layers=n_layers)
    dst0 = O.make_discriminator_state(4)
    text, audio, vid, poses = O.make_batch(seed, batch, n_words, n_speakers)
    for epoch in epochs:
        og, od = O.clone_state(gst0, torch.float64), O.clone_state(dst0, torch.float64)
        hp = dict(O.HP)
        hp["n_layers"] = n_layers
        if not dropout:
            hp["dropout_prob"] = 0.0
        # seed note: the fp64 oracle and the fp32 path can disagree on the sign of a ReLU pre-activation that is ~0
        # (one gate flip moves a TCN weight gradient by ~1e-2).  Seeds 1017+epoch have no such tie at B=4; 4 of 5
        # seeds tried were tie-free and agreed to ~1e-6 (parity notes).
        rand = O.Rand(seed=rand_seed + epoch) if dropout else _NoDrop(seed=rand_seed + epoch)
        oret, extra = O.train_iter_gan(og, od, {}, {}, epoch, text, audio.double(), poses.double(), vid, rand, hp, want_grads=True,
                                       input_context=input_context, z_type=z_type)
        args, G, D = build_models(pkg, dev, gst0, dst0, n_words, n_speakers, make_args(input_context=input_context, z_type=z_type, n_layers=n_layers))
        tr = pkg.GanTrainer(G, D, args)
        losses = tr.train_iter(epoch, text.to(dev), audio.to(dev), poses.to(dev), vid.to(dev), inject=to_device_inject(rand.rec, dev))
        ret = losses.to_dict()
        assert sorted(ret) == sorted(oret), (ret, oret)
        e_loss = max(abs(ret[k] - oret[k]) / max(abs(oret[k]), 1e-6) for k in oret)
        _, Gg, _ = tr.G.views()
        e_g, z_g, k_g = grad_errors(Gg, extra["g_grads"])
        e_d, z_d, k_d = (0.0, 0.0, None)
        if epoch > 10:
            _, Dg, _ = tr.D.views()
            e_d, z_d, k_d = grad_errors(Dg, extra["d_grads"])
        gsd, dsd = G.state_dict(), D.state_dict()
        e_bn = max(rel(sd[k], o[k]) for sd, o in ((gsd, og), (dsd, od)) for k in o if "running_var" in k)
        # running_mean inherits the +-lr noise step of the zero-gradient biases in front of it: compare loosely
        e_bnm = max(float((sd[k].double().cpu() - o[k]).abs().max()) for sd, o in ((gsd, og), (dsd, od)) for k in o if "running_mean" in k)
        for sd, o in ((gsd, og), (dsd, od)):
            for k in o:
                if k.endswith("num_batches_tracked"):
                    assert int(sd[k]) == int(o[k]), (k, int(sd[k]), int(o[k]))
        # Adam's first step is lr * g / (|g| + 1e-8): entries whose gradient is rounding noise (|g| <~ 1e-7) move by an
        # arbitrary fraction of lr on both sides, so the updated parameters are compared where the gradient is real
        e_step = 0.0
        for sd, o, lr, gr in ((gsd, og, 5e-4, extra["g_grads"]), (dsd, od, 1e-4, extra.get("d_grads", {}))):
            for k in o:
                if k in gr and gr[k] is not None and k not in ZERO_GRAD_KEYS:
                    real = gr[k].abs() > 1e-5 * gr[k].abs().max()
                    if bool(real.any()):
                        e_step = max(e_step, float((sd[k].double().cpu() - o[k])[real].abs().max()) / lr)
        if verbose:
            print(f"epoch {epoch}: loss {e_loss:.2e} g_grad {e_g:.2e} ({k_g}) zero-grad |g| {z_g:.1e} d_grad {e_d:.2e} ({k_d}) "
                  f"bn_var {e_bn:.2e} bn_mean_abs {e_bnm:.1e} step/lr {e_step:.2e}")
        assert not check_step or (e_bnm < 5e-4 and e_step < 2e-2), (e_bnm, e_step)
        worst = max(worst, e_loss, e_g, e_d, e_bn)
    return worst


def wav_gate_flips(tape, preacts, near=2e-6):
    """LeakyReLU gates of the audio encoder: HIP path against the fp64 oracle.  tape: GanTrainer.last_tape of the iteration; preacts:
    oracle.wav_preacts on the same weights and audio.  Returns per layer (elements, near-ties |pre| < near (normalised pre-activations are O(1)),
    elements where the HIP path's post-activation has the other sign, the largest fp64 |pre| among those)."""
    out = []
    for li, pre in enumerate(preacts, start=1):
        x = tape["wav"][li][0]                                   # (Ba, L, C) channel-last post-activation = the input of conv li + 1
        mine = x[:pre.shape[0]].detach().double().cpu().transpose(1, 2)
        assert mine.shape == pre.shape, (mine.shape, pre.shape)
        flipped = (mine > 0) != (pre > 0)
        out.append((pre.numel(), int((pre.abs() < near).sum()), int(flipped.sum()), float(pre.abs()[flipped].max()) if bool(flipped.any()) else 0.0))
    return out


def wav_gate_sides(tape, preacts):
    """{layer: (flat indices into the oracle's (B, C, L) pre-activation, the side the HIP path took)} for the gates that differ from the fp64
    oracle's -- the form oracle.ref_model.wav_gate_override takes."""
    out = {}
    for li, pre in enumerate(preacts, start=1):
        x = tape["wav"][li][0]
        mine = x[:pre.shape[0]].detach().double().cpu().transpose(1, 2)
        flipped = ((mine > 0) != (pre > 0)).reshape(-1)
        idx = flipped.nonzero().view(-1)
        out[li] = (idx, (mine.reshape(-1)[idx] > 0))
    return out


def tcn_gate_sides(tape, gate_log, call, B, tag="g2", rel_window=5e-6):
    """ReLU gates of the text encoder, HIP path against the fp64 oracle, for the stacked generator call number `call` (rows call * B ..) whose
    oracle run logged its pre-activations under `tag` (oracle.relu_gate_log).  Returns ({site: (flat indices, HIP sides)} for the gates that
    differ -- the form oracle.relu_gate_override takes -- and [(site, flips, largest |pre-activation| among them relative to the site's
    max, window)]).  A gate behind a dropped element (dropout scale 0) cannot be observed and does not matter: skipped."""
    sides, report = {}, []
    rows = slice(call * B, (call + 1) * B)
    for i, blk in enumerate(tape["tcn"]):
        for k, (act, mask) in enumerate(((blk["o0"], blk["m0"]), (blk["o1"], blk["m1"]), (blk["y"], None)), start=1):
            site = f"{tag}.tcn{i}.relu{k}"
            pre = gate_log[site]                                                   # (B, C, T) fp64
            mine = act[rows].detach().double().cpu().transpose(1, 2)
            if mask is not None and not isinstance(mask, torch.Tensor):              # ops.Drop: the mask is regenerated, not stored
                mask = mask[rows].materialize()
                valid = mask.detach().cpu().transpose(1, 2) > 0
            else:
                valid = torch.ones_like(pre, dtype=torch.bool) if mask is None else (mask[rows].detach().cpu().transpose(1, 2) > 0)
            flipped = (valid & ((mine > 0) != (pre > 0))).reshape(-1)
            idx = flipped.nonzero().view(-1)
            if idx.numel():
                sides[site] = (idx, mine.reshape(-1)[idx] > 0)
                report.append((site, int(idx.numel()), float(pre.reshape(-1)[idx].abs().max() / pre.abs().max()), rel_window))
    return sides, report


MAX_GATE_FLIPS = 8
NEAR_TIE_FRESH, NEAR_TIE_AFTER_FLIP = 2e-6, 1e-5


def assert_gate_flips_are_near_ties(flips, what=""):
    """The allowance for flipped LeakyReLU gates is for a handful of fp64 NEAR-TIES, nothing else.  `flips`:
    wav_gate_flips results, one list per iteration in order (or a single iteration's list).  While both sides still step identical
    weights a gate may differ only where the fp64 pre-activation is within 2e-6 of zero (fp32 rounding of an O(1) value).  Once a gate
    has flipped, the audio encoder's gradients of that iteration differ by up to ~7e-3 of their max, Adam carries that into the weights
    (lr * 7e-3 per entry) and the NEXT iterations' pre-activations can differ by more than fp32 rounding: from then on the window is 1e-5
    (the largest flipped pre-activation ever measured is 2.3e-6; round 4 allowed 5e-4 here).  At most 8 flips in all.  Returns the total number of flipped gates."""
    iters = flips if isinstance(flips[0], list) else [flips]
    total, window = 0, NEAR_TIE_FRESH
    for it, fl in enumerate(iters):
        for li, f in enumerate(fl, start=1):
            assert f[2] == 0 or f[3] < window, (f"{what}: iteration {it}, audio-encoder layer {li}: {f[2]} gates differ from the fp64 oracle's, the "
                                                f"largest at |pre-activation| = {f[3]:.2e} -- not a near-tie (window {window:.0e}): {iters}")
        n = sum(f[2] for f in fl)
        total += n
        if n:
            window = NEAR_TIE_AFTER_FLIP
    assert total <= MAX_GATE_FLIPS, f"{what}: {total} flipped near-tie gates (cap {MAX_GATE_FLIPS}): {iters}"
    return total


class _NoDrop(O.Rand):
    def keep_mask(self, name, shape, p, dtype=torch.float32):
        m = torch.ones(shape, dtype=dtype)
        self.rec[name] = m
        return m


def sample_idx(numel, n=2048, seed=7):
    if numel <= n:
        return np.arange(numel)
    return np.sort(np.random.RandomState(seed + numel % 9973).choice(numel, n, replace=False))


# --------------------------------------------------------------------------- fixtures generated by tests/golden/make_golden_eval.py
MEAN_DIR_VEC_KEY = "mean_dir_vec"
FIXTURE_WORDS = ("so what i want to talk about today is how we move our hands when we speak and why it matters for the people who listen "
                 "because gesture carries meaning that words alone do not").split()


def fixture_lang(vocab_cls, n_words):
    """The word Vocab make_golden_eval.py built with the reference's class: the fixture's words, then fillers up to n_words."""
    lang = vocab_cls("words")
    for w in FIXTURE_WORDS:
        lang.index_word(w)
    i = 0
    while lang.n_words < n_words:
        lang.index_word(f"filler{i}")
        i += 1
    lang.word_embedding_weights = None
    return lang


class OracleLang:
    """get_word_index of that Vocab without any product / reference class (oracle side of the tests)."""

    def __init__(self, n_words):
        self.index = {}
        n = 4
        for w in FIXTURE_WORDS:
            if w not in self.index:
                self.index[w] = n
                n += 1
        self.n_words = n_words

    def get_word_index(self, w):
        return self.index.get(w, 3)


def synth_case(g, name):
    """One generate_gestures case of g9_generate_gestures.npz -> dict of python / numpy inputs and expected outputs."""
    words = [[str(w), float(t[0]), float(t[1])] for w, t in zip(g[f"{name}/words"], g[f"{name}/word_times"])]
    seed = g[f"{name}/seed_seq"]
    zt = str(g[f"{name}/z_type"])
    draws = g[f"{name}/eps"] if zt == "speaker" else g[f"{name}/z"]
    return dict(z_type=zt, audio=g[f"{name}/audio"].astype(np.float32), words=words, fade_out=bool(g[f"{name}/fade_out"]),
                vid_arg=None if int(g[f"{name}/vid_arg"]) < 0 else int(g[f"{name}/vid_arg"]),
                vid_used=None if int(g[f"{name}/vid_used"]) < 0 else int(g[f"{name}/vid_used"]),
                seed_seq=seed if seed.shape[0] else None, draws=draws, out=g[f"{name}/out"], win_pre_seq=g[f"{name}/win_pre_seq"],
                win_text=g[f"{name}/win_text"], win_audio_head=g[f"{name}/win_audio_head"], win_audio_tail=g[f"{name}/win_audio_tail"],
                win_audio_abs_sum=g[f"{name}/win_audio_abs_sum"])


def check_window_audio(case, i, audio_window):
    a = np.asarray(audio_window, dtype=np.float32)
    assert np.array_equal(a[:64], case["win_audio_head"][i]) and np.array_equal(a[-3000:], case["win_audio_tail"][i]), i
    assert abs(float(np.abs(a.astype(np.float64)).sum()) - float(case["win_audio_abs_sum"][i])) < 1e-9, i


def dataset_samples(g):
    """The raw samples of g10_dataset.npz in the preprocessor's stored format."""
    out = []
    for i in range(int(g["n_samples"])):
        words = [[str(w), float(t[0]), float(t[1])] for w, t in zip(g[f"raw{i}/words"], g[f"raw{i}/word_times"])]
        aux = g[f"raw{i}/aux"]
        out.append([words, g[f"raw{i}/pose"], g[f"raw{i}/vec"], g[f"raw{i}/audio"].astype(np.float32), g[f"raw{i}/spec"],
                    {"vid": str(g[f"raw{i}/vid"]), "start_frame_no": int(aux[0]), "end_frame_no": int(aux[1]), "start_time": float(aux[2]),
                     "end_time": float(aux[3])}])
    return out
