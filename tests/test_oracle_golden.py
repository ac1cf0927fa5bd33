"""Pins oracle/ref_model.py to golden vectors produced by the REAL reference (tests/golden/make_golden.py).
CPU only.  The weights are regenerated from seeds (same torch build on both machines) and cross-checked by checksum."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import ref_model as O


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sample_idx(numel, n=2048, seed=7):
    if numel <= n:
        return np.arange(numel)
    return np.sort(np.random.RandomState(seed + numel % 9973).choice(numel, n, replace=False))


def unpack_masks(g, tags, p_tcn=0.3):
    """Recorded keep-masks (packed bits, reference layout (B,T,300) for the embedding and (B,300,T) for the TCN)."""
    shape_emb = tuple(g["mask_shape"])
    names = ["emb_drop"] + [f"tcn{i}.drop{j}" for i in range(4) for j in (1, 2)]
    n = int(np.prod(shape_emb))
    bits = np.unpackbits(g["masks"], axis=1)[:, :n]
    inj, k = {}, 0
    B = shape_emb[0]
    for tag in tags:
        for name in names:
            p = 0.1 if name == "emb_drop" else p_tcn
            keep = torch.from_numpy(bits[k].astype(np.float32))
            shp = shape_emb if name == "emb_drop" else (B, shape_emb[2], shape_emb[1])
            inj[f"{tag}.{name}"] = keep.view(shp) / (1.0 - p)
            k += 1
    return inj


def test_eval_report_is_tight():
    """golden_report_eval.json: oracle-vs-reference errors recorded while the g8-g11 fixtures were generated."""
    with open(os.path.join(GOLDEN, "golden_report_eval.json")) as f:
        rep = json.load(f)
    for zt in ("speaker", "random", "none"):
        r = rep["evaluate_testset_" + zt]
        assert max(r["loss"], r["joint_mae"], r["accel"], r["out"]) < 2e-6 and max(r["frechet"], r["feat_dist"]) < 2e-5, r
    assert rep["dir_vec_to_pose"] < 1e-7 and rep["pose_seq_to_dir_vec"] < 1e-12
    cases = [k for k in rep if k.startswith("generate_gestures_")]
    assert len(cases) == 9
    for k in cases:
        assert rep[k]["out"] < 2e-6 and rep[k]["pre_seq"] < 1e-6 and rep[k]["text_equal"] and rep[k]["audio_equal"], (k, rep[k])
    assert all(all(v.values()) if isinstance(v, dict) else v for v in rep["dataset"].values())
    assert max(rep["ae_eval"].values()) < 2e-6


def test_report_is_tight():
    rep = json.load(open(os.path.join(GOLDEN, "golden_report.json")))
    assert rep["torch"] == torch.__version__
    assert max(rep["G1"].values()) < 2e-6
    for k in ("G2_warmup", "G2_gan"):
        assert rep[k]["g_grad_max"] < 2e-5 and max(rep[k]["loss"].values()) < 1e-6
    assert rep["G3"]["g_grad_max"] < 2e-5 and rep["G5"]["fgd_rel"] < 1e-9


def test_g1_eval_forward_matches_reference():
    g = load("g1_eval_forward.npz")
    gst, dst = O.make_generator_state(int(g["g_seed"]), int(g["n_words"]), int(g["n_speakers"])), O.make_discriminator_state(int(g["d_seed"]))
    chk = np.array([float(v.double().abs().sum()) for v in gst.values() if v.is_floating_point()])
    assert np.allclose(chk, g["w_checksum"], rtol=1e-12), "seeded weights differ from the ones the golden run used"
    text, audio, vid, poses = O.make_batch(int(g["batch_seed"]), 4, int(g["n_words"]), int(g["n_speakers"]))
    assert np.array_equal(text.numpy(), g["text"]) and np.array_equal(audio.numpy(), g["audio"])
    pre = O.make_pre_seq(poses, 4)
    out, z, mu, lv, parts = O.generator_forward(gst, pre, text, audio, vid, training=False,
                                                rand=O.Rand(inject={"g.eps": torch.from_numpy(g["eps"])}), return_parts=True)
    assert rel(out, g["out"]) < 2e-6 and rel(z, g["z"]) < 1e-6 and rel(mu, g["mu"]) < 1e-6 and rel(lv, g["logvar"]) < 1e-6
    assert rel(parts["audio_feat"], g["wav_feat"]) < 2e-6 and rel(parts["text_feat"], g["text_feat"]) < 2e-6
    d = O.discriminator_forward(dst, poses, training=False, rand=O.Rand())
    assert rel(d, g["d_out"]) < 1e-6


def _run_g2(label):
    g = load(f"g2_train_{label}.npz")
    epoch = int(g["epoch"])
    V, S = int(g["n_words"]), int(g["n_speakers"])
    gst, dst = O.make_generator_state(int(g["g_seed"]), V, S), O.make_discriminator_state(int(g["d_seed"]))
    text, audio, vid, poses = O.make_batch(int(g["batch_seed"]), 4, V, S)
    tags = ["g1", "g2", "g3"] if epoch > 10 else ["g2", "g3"]
    inj = unpack_masks(g, tags)
    for t, e in zip(tags, g["eps"]):
        inj[f"{t}.eps"] = torch.from_numpy(e)
    inj["perm"] = torch.from_numpy(g["perm"])
    for t in tags:
        for l in range(3):
            inj[f"{t}.gru.drop{l}"] = torch.ones(4, 34, 600)
    for t in ("d_real", "d_fake", "d_out"):
        for l in range(3):
            inj[f"{t}.gru.drop{l}"] = torch.ones(4, 28, 128)
    ret, extra = O.train_iter_gan(gst, dst, {}, {}, epoch, text, audio, poses, vid, O.Rand(inject=inj), want_grads=True)
    return g, ret, extra, gst, dst


def test_g2_train_steps_match_reference():
    for label in ("warmup", "gan"):
        g, ret, extra, gst, dst = _run_g2(label)
        assert sorted(ret) == list(g["loss_keys"])
        for k, v in zip(g["loss_keys"], g["loss_vals"]):
            assert abs(ret[k] - v) <= 1e-6 * max(1.0, abs(v)), (label, k, ret[k], v)
        worst = 0.0
        for k, gr in extra["g_grads"].items():
            ref = g["gg/" + k]
            mine = gr.reshape(-1).numpy()[sample_idx(gr.numel())]
            if float(np.abs(ref).max()) < 1e-5:      # zero-by-construction gradients (bias in front of BatchNorm)
                assert float(np.abs(mine).max()) < 1e-4
                continue
            worst = max(worst, rel(mine, ref))
        assert worst < 2e-5, (label, worst)
        for k in gst:
            if k.endswith("num_batches_tracked"):
                assert int(gst[k]) == int(g["gp/" + k]), k


def test_g6_generator_variants_match_reference():
    """input_context 'audio' | 'text' | 'none' and z_type 'random' | 'none' (multimodal_context_net.py:71-97, train_gan.py:59-84):
    one reference train_iter_gan per variant, draws replayed into the oracle."""
    g = load("g6_variants.npz")
    V, S, epoch = int(g["n_words"]), int(g["n_speakers"]), int(g["epoch"])
    text, audio, vid, poses = O.make_batch(int(g["batch_seed"]), 4, V, S)
    sidx = lambda n: sample_idx(n, 256)
    for var in g["variants"]:
        ctx, zt = str(var).split("/")
        key = f"{ctx}_{zt}"
        z_mode = zt if zt in ("speaker", "random") else None
        gst = O.make_generator_state(int(g["g_seed"]), V, S, input_context=ctx, z_mode=z_mode)
        dst = O.make_discriminator_state(int(g["d_seed"]))
        tags = ["g1", "g2"] + (["g3"] if z_mode else [])
        inj = {}
        if g[f"{key}/masks"].size:
            inj.update(unpack_masks({"mask_shape": g[f"{key}/mask_shape"], "masks": g[f"{key}/masks"]}, tags))
        for t, e in zip(tags, g[f"{key}/eps"] if zt == "speaker" else []):
            inj[f"{t}.eps"] = torch.from_numpy(e)
        for t, z in zip(tags, g[f"{key}/z"] if zt == "random" else []):
            inj[f"{t}.z"] = torch.from_numpy(z)
        if g[f"{key}/perm"].size:
            inj["perm"] = torch.from_numpy(g[f"{key}/perm"])
        for t in tags:
            for l in range(3):
                inj[f"{t}.gru.drop{l}"] = torch.ones(4, 34, 600)
        for t in ("d_real", "d_fake", "d_out"):
            for l in range(3):
                inj[f"{t}.gru.drop{l}"] = torch.ones(4, 28, 128)
        ret, extra = O.train_iter_gan(gst, dst, {}, {}, epoch, text, audio, poses, vid, O.Rand(inject=inj), want_grads=True,
                                      input_context=ctx, z_type=zt)
        assert sorted(ret) == list(g[f"{key}/loss_keys"]), (key, ret)
        for k, v in zip(g[f"{key}/loss_keys"], g[f"{key}/loss_vals"]):
            assert abs(ret[k] - v) <= 1e-6 * max(1.0, abs(v)), (key, k, ret[k], v)
        grad_keys = list(g[f"{key}/grad_keys"])
        # parameters the reference leaves without a gradient (unused encoder / absent z) have none here either
        assert sorted(k for k, gr in extra["g_grads"].items() if gr is not None) == grad_keys, key
        worst = 0.0
        for k in grad_keys:
            gr, ref = extra["g_grads"][k], g[f"{key}/gg/{k}"]
            mine = gr.reshape(-1).numpy()[sidx(gr.numel())]
            if float(np.abs(ref).max()) < 1e-5:
                assert float(np.abs(mine).max()) < 1e-4
                continue
            worst = max(worst, rel(mine, ref))
        assert worst < 2e-5, (key, worst)
        for k in gst:                       # incl. the UNUSED audio encoder's BatchNorm buffers for input_context='text'
            if "running" in k:
                assert rel(gst[k], g[f"{key}/gp/{k}"]) < 1e-6, (key, k)
            elif k.endswith("num_batches_tracked"):
                assert int(gst[k]) == int(g[f"{key}/gp/{k}"]), (key, k)


def test_g5_fgd_matches_reference():
    g = load("g5_fgd.npz")
    ast = O.make_autoencoder_state(int(g["ae_seed"]))
    gp = torch.Generator().manual_seed(int(g["pose_seed"]))
    real = 0.1 * torch.randn(256, 34, 27, generator=gp)
    fake = real + 0.05 * torch.randn(256, 34, 27, generator=gp)
    fr, _, _, _ = O.ae_forward(O.clone_state(ast), real, False)
    fk, _, _, _ = O.ae_forward(O.clone_state(ast), fake, False)
    assert rel(fr, g["feat_real"]) < 2e-6 and rel(fk, g["feat_fake"]) < 2e-6
    fd, feat_dist = O.fgd_scores(g["feat_fake"], g["feat_real"])
    assert abs(fd - float(g["fgd"])) <= 1e-9 * abs(float(g["fgd"]))
    fd8, _ = O.fgd_scores(g["feat_fake"][:8], g["feat_real"][:8])
    assert abs(fd8 - float(g["fgd8"])) <= 1e-6 * max(1.0, abs(float(g["fgd8"])))
    ret, _ = O.ae_train_iter(O.clone_state(ast), {}, real[:32])
    assert abs(ret["loss"] - float(g["train_loss"])) < 1e-5 * float(g["train_loss"])


def test_g8_evaluate_testset_matches_reference():
    """oracle evaluate_testset == scripts/train.py:evaluate_testset (:234-329) incl. FGD, for every z_type; bone integration."""
    g = load("g8_evaluate_testset.npz")
    V, S = int(g["n_words"]), int(g["n_speakers"])
    ast = O.make_autoencoder_state(int(g["ae_seed"]))
    sizes = [int(x) for x in g["sizes"]]
    for zt in ("speaker", "random", "none"):
        z_mode = zt if zt != "none" else None
        gst = O.make_generator_state(int(g["g_seed"]), V, S, z_mode=z_mode)
        batches = []
        for i, b in enumerate(sizes):
            text, audio, _, poses = O.make_batch(int(g["batch_seed0"]) + i, b, V, S)
            batches.append((text, poses, audio))
        offs = np.cumsum([0] + sizes)
        inj, vids = {}, None
        for i in range(len(sizes)):
            if zt == "speaker":
                inj[f"e{i}.eps"] = torch.from_numpy(g[f"{zt}/eps"][offs[i]:offs[i + 1]])
            elif zt == "random":
                inj[f"e{i}.z"] = torch.from_numpy(g[f"{zt}/z"][offs[i]:offs[i + 1]])
        if zt == "speaker":
            vids = [torch.from_numpy(g[f"{zt}/vids"][offs[i]:offs[i + 1]]) for i in range(len(sizes))]
        ret, outs = O.evaluate_testset(O.clone_state(gst), batches, g["mean_dir_vec"], O.Rand(inject=inj), ast=O.clone_state(ast),
                                       vids=vids, z_mode=z_mode)
        want = dict(zip([str(k) for k in g[f"{zt}/ret_keys"]], g[f"{zt}/ret_vals"]))
        assert sorted(want) == ["feat_dist", "frechet", "joint_mae", "loss"]          # the reference's return keys (:316-325)
        for k, v in want.items():
            assert abs(ret[k] - v) <= 2e-5 * abs(v), (zt, k, ret[k], v)
        assert abs(ret["accel"] - float(g[f"{zt}/accel"])) <= 1e-6 * float(g[f"{zt}/accel"])
        assert rel(torch.cat(outs), g[f"{zt}/out"]) < 2e-6
    v = g["dir_vec/in"]
    assert rel(O.dir_vec_to_pose(v), g["dir_vec/pose_b_t"]) < 1e-7 and rel(O.dir_vec_to_pose(v[0]), g["dir_vec/pose_t"]) < 1e-7
    assert rel(O.dir_vec_to_pose(v[0, 0]), g["dir_vec/pose_single"]) < 1e-7
    assert rel(O.pose_seq_to_dir_vec(g["dir_vec/pose_in"]), g["dir_vec/vec_from_pose"]) < 1e-12
    assert rel(O.pose_seq_to_dir_vec(g["dir_vec/pose_in"][1]), g["dir_vec/vec_from_pose_t"]) < 1e-12


def test_g9_generate_gestures_matches_reference():
    """oracle generate_gestures == scripts/synthesize.py:generate_gestures (:36-209): window slicing, seed hand-over, cross-fade,
    fade-out, speaker-id handling, for 1 / 2 / 3 / 4 windows and every z_type."""
    from tests.harness import OracleLang, check_window_audio, synth_case
    g = load("g9_generate_gestures.npz")
    V, S = int(g["n_words"]), int(g["n_speakers"])
    lang = OracleLang(V)
    seen_fade = set()
    for name in [str(c) for c in g["cases"]]:
        c = synth_case(g, name)
        z_mode = c["z_type"] if c["z_type"] != "none" else None
        gst = O.make_generator_state(int(g["g_seed"]), V, S, z_mode=z_mode)
        n = c["win_text"].shape[0]
        key = "eps" if c["z_type"] == "speaker" else "z"
        inj = {f"w{i}.{key}": torch.from_numpy(c["draws"][i:i + 1]) for i in range(c["draws"].shape[0])}
        wins = []
        out = O.generate_gestures(O.clone_state(gst), c["audio"], c["words"], lang.get_word_index, O.Rand(inject=inj), vid=c["vid_used"],
                                  seed_seq=c["seed_seq"], fade_out=c["fade_out"], z_mode=z_mode, windows=wins)
        assert out.shape == c["out"].shape and rel(out, c["out"]) < 5e-6, (name, out.shape, c["out"].shape)
        assert len(wins) == n == O.num_windows(len(c["audio"]) / 16000)
        for i, (pre, text, audio) in enumerate(wins):
            assert np.array_equal(text.numpy(), c["win_text"][i:i + 1])
            assert float(np.abs(pre.numpy() - c["win_pre_seq"][i:i + 1]).max()) < 1e-6
            check_window_audio(c, i, audio.numpy()[0])
        seen_fade.add(c["fade_out"])
    assert seen_fade == {True, False}


def test_g10_dataset_matches_reference():
    """oracle data_getitem == SpeechMotionDataset.__getitem__ (lmdb_data_loader.py:107-171), bit for bit, both word-timing modes."""
    from tests.harness import OracleLang, dataset_samples
    g = load("g10_dataset.npz")
    lang = OracleLang(int(g["vocab_size"]))
    samples = dataset_samples(g)
    for i, s in enumerate(samples):
        for tag, rwt in (("timed", False), ("rwt", True)):
            words, ext, pose, vec, audio, spec, aux = O.data_getitem(s, lang.get_word_index, remove_word_timing=rwt, full=True)
            assert np.array_equal(words, g[f"{tag}{i}/words"]) and np.array_equal(ext, g[f"{tag}{i}/ext"]), (tag, i)
        assert np.array_equal(pose, g[f"item{i}/pose"]) and np.array_equal(vec, g[f"item{i}/vec"])
        assert audio.shape[0] == int(g[f"item{i}/audio_len"]) == 36267
        assert np.array_equal(audio[:64], g[f"item{i}/audio_head"]) and np.array_equal(audio[-1500:], g[f"item{i}/audio_tail"])
        assert abs(float(np.abs(audio.astype(np.float64)).sum()) - float(g[f"item{i}/audio_abs_sum"])) < 1e-9
        assert list(spec.shape) == list(g[f"item{i}/spec_shape"])
    a = g["fixlen/in"]
    assert np.array_equal(O.data_make_audio_fixed_length(a, 1500), g["fixlen/longer"])
    assert np.array_equal(O.data_make_audio_fixed_length(a, 700), g["fixlen/shorter"])
    assert np.array_equal(O.data_make_audio_fixed_length(a, 1000), g["fixlen/same"])


def test_g11_autoencoder_eval_matches_reference():
    """oracle eval_embed / ae_evaluate_testset == train_joint_embed.py:54-62 / train_feature_extractor.py:26-51."""
    g = load("g11_ae_eval.npz")
    ast = O.make_autoencoder_state(int(g["ae_seed"]))
    gen = torch.Generator().manual_seed(int(g["pose_seed"]))
    batches = [0.1 * torch.randn(int(b), 34, 27, generator=gen) for b in g["sizes"]]
    loss, recon = O.eval_embed(O.clone_state(ast), batches[0])
    assert abs(loss - float(g["eval_embed_loss"])) < 1e-6 * float(g["eval_embed_loss"]) and rel(recon, g["recon"]) < 2e-6
    ret = O.ae_evaluate_testset(O.clone_state(ast), batches)
    assert abs(ret["loss"] - float(g["evaluate_testset_loss"])) < 1e-6 * float(g["evaluate_testset_loss"])


def test_window_blend_and_count_hand_computed():
    """synthesize.py:57-63,142-160 restated; values worked out by hand."""
    assert O.num_windows(1.0) == 1 and O.num_windows(34 / 15) == 1 and O.num_windows(34 / 15 + 0.01) == 2 and O.num_windows(10.0) == 5
    a, b = np.zeros((34, 2), np.float32), np.ones((34, 2), np.float32)
    out = O.blend_windows([a, b])
    assert out.shape == (64, 2)
    assert np.allclose(out[30:34, 0], [1 / 5, 2 / 5, 3 / 5, 4 / 5]) and np.all(out[:30] == 0) and np.all(out[34:] == 1)


def test_dir_vec_to_pose_hand_computed():
    v = np.zeros((1, 1, 27)); v[0, 0, 0:3] = [0, 1, 0]; v[0, 0, 3:6] = [1, 0, 0]
    p = O.dir_vec_to_pose(v)
    assert np.allclose(p[0, 0, 1], [0, 0.26, 0]) and np.allclose(p[0, 0, 2], [0.18, 0.26, 0]) and np.allclose(p[0, 0, 4], [0, 0.26, 0])
