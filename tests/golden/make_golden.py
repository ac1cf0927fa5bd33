#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (build container only).

Imports /root/reference/scripts (read-only, no bytecode written), loads this build's deterministic
weights into the reference's own nn.Modules, runs the reference's forward passes and its
``train_iter_gan`` with recording hooks on every random draw, and stores inputs / outputs as small
.npz fixtures next to this file.  While doing so it checks oracle/ref_model.py against the reference
on the FULL tensors and writes the max errors to golden_report.json.

Nothing here travels as reference code: fixtures hold tensors and scalars only.

    python tests/golden/make_golden.py
"""
import argparse
import json
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

REF = "/root/reference/scripts"


def import_reference():
    sys.path.insert(0, REF)
    for name in ("fasttext", "umap"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import model.embedding_net as embedding_net            # must come first (circular import)
    import model.multimodal_context_net as mcn
    import model.vocab as vocab
    import train_eval.train_gan as train_gan
    from model.embedding_space_evaluator import EmbeddingSpaceEvaluator
    return embedding_net, mcn, vocab, train_gan, EmbeddingSpaceEvaluator


def ref_args(hidden=300, layers=4):
    return argparse.Namespace(n_pre_poses=4, n_poses=34, input_context="both", hidden_size=hidden, n_layers=layers,
                              dropout_prob=0.3, freeze_wordembed=False, z_type="speaker", loss_warmup=10,
                              loss_gan_weight=5.0, loss_regression_weight=500, loss_kld_weight=0.1,
                              loss_reg_weight=0.05, learning_rate=0.0005, discriminator_lr_weight=0.2)


def sample_idx(numel, n=2048, seed=7):
    if numel <= n:
        return np.arange(numel)
    return np.sort(np.random.RandomState(seed + numel % 9973).choice(numel, n, replace=False))


def sampled(t, n=2048):
    a = t.detach().cpu().numpy().reshape(-1)
    return a[sample_idx(a.size, n)]


def maxerr(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def pre_bn_bias(k):
    """Conv/linear biases that feed straight into a train-mode BatchNorm: their true gradient is exactly
    zero (BN subtracts the batch mean), so both sides hold rounding noise only."""
    return k in ("audio_encoder.feat_extractor.0.bias", "audio_encoder.feat_extractor.3.bias",
                 "audio_encoder.feat_extractor.6.bias", "pre_conv.0.bias", "pre_conv.3.bias",
                 "pose_encoder.net.0.0.bias", "pose_encoder.net.1.0.bias", "pose_encoder.net.2.0.bias",
                 "pose_encoder.out_net.0.bias", "pose_encoder.out_net.3.bias", "decoder.pre_net.0.bias",
                 "decoder.net.0.bias", "decoder.net.3.bias",
                 # ... and everything that reaches the next BatchNorm through linear maps only
                 # (LeakyReLU(True) is the identity): BN betas and biases in front of it
                 "pre_conv.1.bias", "pose_encoder.net.3.bias", "pose_encoder.out_net.1.bias",
                 "pose_encoder.out_net.4.bias", "pose_encoder.out_net.6.bias", "pose_encoder.fc_mu.bias")


def grad_err(mine, ref):
    """(max error over real gradients, max |grad| over the zero-by-construction ones)."""
    e = max(maxerr(mine[k], ref[k]) for k in mine if not pre_bn_bias(k) and ref[k] is not None)
    z = max([float(ref[k].abs().max()) for k in mine if pre_bn_bias(k) and ref[k] is not None] + [0.0])
    return e, z


def step_err(mine, ref, before, lr):
    """max |p_mine - p_ref| / lr over parameters with a real gradient (Adam's first step is lr*sign(g):
    elements whose gradient is rounding noise flip sign freely, so the error is quoted in units of lr)."""
    return max(float((mine[k].double() - ref[k].double()).abs().max()) / lr for k in mine
               if mine[k].is_floating_point() and "running" not in k and not pre_bn_bias(k))


class Recorder:
    """Hooks for every random draw on the reference path."""

    def __init__(self, embedding_net, drop_p_override=None):
        self.masks, self.eps, self.perms = [], [], []
        self.embedding_net = embedding_net
        self.p_override = drop_p_override
        self.gen = torch.Generator().manual_seed(4242)

    def install(self):
        import torch.nn.functional as F
        self._dropout, self._reparam, self._randperm = F.dropout, self.embedding_net.reparameterize, torch.randperm

        def dropout(x, p=0.5, training=True, inplace=False):
            if self.p_override is not None:
                p = self.p_override
            if not training:
                return x
            keep = torch.bernoulli(torch.full(x.shape, 1.0 - p), generator=self.gen) if p > 0 else torch.ones(x.shape)
            m = keep / (1.0 - p)
            self.masks.append(keep.bool().numpy())
            return x * m

        def reparameterize(mu, logvar):
            std = torch.exp(0.5 * logvar)
            eps = torch.randn(std.shape, generator=self.gen)
            self.eps.append(eps.numpy().copy())
            return mu + eps * std

        def randperm(n, *a, **k):
            p = self._randperm(n, generator=self.gen)
            self.perms.append(p.numpy().copy())
            return p

        F.dropout = dropout
        self.embedding_net.reparameterize = reparameterize
        torch.randperm = randperm

    def remove(self):
        import torch.nn.functional as F
        F.dropout, self.embedding_net.reparameterize, torch.randperm = self._dropout, self._reparam, self._randperm


MASK_NAMES = ["emb_drop"] + [f"tcn{i}.drop{j}" for i in range(4) for j in (1, 2)]


def masks_to_inject(rec, tags, p_list):
    """Map the recorder's call-ordered keep masks onto oracle Rand names (scale masks)."""
    inj, it = {}, iter(rec.masks)
    for tag in tags:
        for name in MASK_NAMES:
            p = 0.1 if name == "emb_drop" else p_list
            keep = next(it)
            inj[f"{tag}.{name}"] = torch.from_numpy(keep.astype(np.float32)) / (1.0 - p)
    return inj


def build_ref_models(mcn, vocab, gst, dst, n_words, n_speakers):
    args = ref_args()
    spk = vocab.Vocab("vid", insert_default_tokens=False)
    for i in range(n_speakers - 1):
        spk.index_word(f"spk{i}")
    assert spk.n_words == n_speakers
    G = mcn.PoseGenerator(args, pose_dim=27, n_words=n_words, word_embed_size=300,
                          word_embeddings=np.zeros((n_words, 300), dtype=np.float32), z_obj=spk)
    D = mcn.ConvDiscriminator(27)
    G.load_state_dict(gst, strict=True)
    D.load_state_dict(dst, strict=True)
    return args, G, D


def main():
    from oracle import ref_model as O
    embedding_net, mcn, vocab, train_gan, Evaluator = import_reference()
    report = OrderedDict(torch=torch.__version__)
    torch.set_num_threads(8)

    # ------------------------------------------------------------------ G1: eval-mode forwards, B=4
    V, S, B = 512, 17, 4
    gst0, dst0 = O.make_generator_state(0, V, S), O.make_discriminator_state(1)
    text, audio, vid, poses = O.make_batch(100, B, V, S)
    args, G, D = build_ref_models(mcn, vocab, O.clone_state(gst0), O.clone_state(dst0), V, S)
    assert len(G.state_dict()) == len(gst0) == 117, (len(G.state_dict()), len(gst0))
    assert set(G.state_dict().keys()) == set(gst0.keys())
    assert set(D.state_dict().keys()) == set(dst0.keys())
    G.eval(); D.eval()
    rec = Recorder(embedding_net); rec.install()
    pre_seq = O.make_pre_seq(poses, 4)
    with torch.no_grad():
        out, z, mu, logvar = G(pre_seq, text, audio, vid)
        wav = G.audio_encoder(audio)
        txt, _ = G.text_encoder(text)
        d_out = D(poses)
    rec.remove()
    eps = torch.from_numpy(rec.eps[0])
    r = O.Rand(inject={"g.eps": eps})
    o_out, o_z, o_mu, o_lv, parts = O.generator_forward(O.clone_state(gst0), pre_seq, text, audio, vid, training=False,
                                                        rand=r, return_parts=True)
    o_d = O.discriminator_forward(O.clone_state(dst0), poses, training=False, rand=O.Rand())
    report["G1"] = dict(out=maxerr(o_out, out), z=maxerr(o_z, z), mu=maxerr(o_mu, mu), logvar=maxerr(o_lv, logvar),
                        wav=maxerr(parts["audio_feat"], wav), text=maxerr(parts["text_feat"], txt), d=maxerr(o_d, d_out))
    np.savez_compressed(os.path.join(HERE, "g1_eval_forward.npz"), n_words=V, n_speakers=S, g_seed=0, d_seed=1,
                        batch_seed=100, text=text.numpy(), audio=audio.numpy(), vid=vid.numpy(), poses=poses.numpy(),
                        eps=eps.numpy(), out=out.numpy(), z=z.numpy(), mu=mu.numpy(), logvar=logvar.numpy(),
                        wav_feat=wav.numpy(), text_feat=txt.numpy(), d_out=d_out.numpy(),
                        w_checksum=np.array([float(v.double().abs().sum()) for v in gst0.values() if v.is_floating_point()]))

    # ------------------------------------------------------------------ G2: reference train_iter_gan, B=4
    for label, epoch in (("warmup", 0), ("gan", 11)):
        args, G, D = build_ref_models(mcn, vocab, O.clone_state(gst0), O.clone_state(dst0), V, S)
        G.train(); D.train()
        G.gru.dropout = 0.0          # nn.GRU's inter-layer dropout draws inside ATen: not recordable
        D.gru.dropout = 0.0
        g_opt = torch.optim.Adam(G.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
        d_opt = torch.optim.Adam(D.parameters(), lr=args.learning_rate * args.discriminator_lr_weight, betas=(0.5, 0.999))
        d_grads = {}
        d_step = d_opt.step

        def rec_step(*a, **k):
            for n_, p_ in D.named_parameters():
                d_grads[n_] = None if p_.grad is None else p_.grad.detach().clone()
            return d_step(*a, **k)
        d_opt.step = rec_step
        rec = Recorder(embedding_net); rec.install()
        ret = train_gan.train_iter_gan(args, epoch, text, audio, poses, vid, G, D, g_opt, d_opt)
        rec.remove()
        tags = ["g1", "g2", "g3"] if epoch > 10 else ["g2", "g3"]
        inj = masks_to_inject(rec, tags, 0.3)
        for tg, e in zip(tags, rec.eps):
            inj[f"{tg}.eps"] = torch.from_numpy(e)
        inj["perm"] = torch.from_numpy(rec.perms[0])
        hp = dict(O.HP)
        og, od = O.clone_state(gst0), O.clone_state(dst0)
        ga, da = {}, {}
        # oracle with GRU inter-layer dropout off (p=0 -> all-ones masks), everything else replayed
        hp["dropout_prob"] = 0.3
        r = O.Rand(inject={**inj, **{f"{t}.gru.drop{l}": torch.ones(1) .expand(B, 34, 600) for t in tags for l in range(3)},
                           **{f"{t}.gru.drop{l}": torch.ones(1).expand(B, 28, 128) for t in ("d_real", "d_fake", "d_out")
                              for l in range(3)}})
        oret, extra = O.train_iter_gan(og, od, ga, da, epoch, text, audio, poses, vid, r, hp, want_grads=True)
        errs = {}
        ref_g_grads = {n_: p_.grad for n_, p_ in G.named_parameters()}
        errs["g_grad_max"], errs["g_zero_grad_abs"] = grad_err(extra["g_grads"], ref_g_grads)
        if epoch > 10:
            errs["d_grad_max"], errs["d_zero_grad_abs"] = grad_err(extra["d_grads"], d_grads)
        gsd, dsd = G.state_dict(), D.state_dict()
        errs["g_step_err_over_lr"] = step_err(og, gsd, gst0, 5e-4)
        errs["d_step_err_over_lr"] = step_err(od, dsd, dst0, 1e-4)
        errs["bn_buffers_max"] = max(maxerr(o_[k], r_[k]) for o_, r_ in ((og, gsd), (od, dsd)) for k in o_ if "running" in k)
        assert all(int(o_[k]) == int(r_[k]) for o_, r_ in ((og, gsd), (od, dsd)) for k in o_ if "num_batches" in k)
        errs["loss"] = {k: abs(oret[k] - ret[k]) / max(abs(ret[k]), 1e-12) for k in ret}
        assert set(oret) == set(ret), (oret, ret)
        report["G2_" + label] = errs
        store = dict(epoch=epoch, n_words=V, n_speakers=S, g_seed=0, d_seed=1, batch_seed=100,
                     perm=rec.perms[0], mask_shape=np.array(rec.masks[0].shape),
                     masks=np.packbits(np.stack([m.reshape(-1) for m in rec.masks]), axis=1),
                     eps=np.stack(rec.eps), loss_keys=np.array(sorted(ret)), loss_vals=np.array([ret[k] for k in sorted(ret)]))
        for k, g_ in ref_g_grads.items():
            store["gg/" + k] = sampled(g_)
            store["ggn/" + k] = np.array(float(g_.double().norm()))
        for k, g_ in d_grads.items():
            if g_ is not None:
                store["dg/" + k] = sampled(g_)
        for k, v_ in gsd.items():
            if not O.is_tcn_alias(k):
                store["gp/" + k] = sampled(v_) if v_.is_floating_point() else v_.numpy()
        for k, v_ in dsd.items():
            store["dp/" + k] = sampled(v_) if v_.is_floating_point() else v_.numpy()
        np.savez_compressed(os.path.join(HERE, f"g2_train_{label}.npz"), **store)

    # ------------------------------------------------------------------ G3: B=128 full-size step, scalars only
    V3, S3, B3 = 2000, 1371, 128
    gst3, dst3 = O.make_generator_state(10, V3, S3), O.make_discriminator_state(11)
    text3, audio3, vid3, poses3 = O.make_batch(300, B3, V3, S3)
    args, G, D = build_ref_models(mcn, vocab, O.clone_state(gst3), O.clone_state(dst3), V3, S3)
    G.train(); D.train(); G.gru.dropout = 0.0; D.gru.dropout = 0.0
    g_opt = torch.optim.Adam(G.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(D.parameters(), lr=args.learning_rate * args.discriminator_lr_weight, betas=(0.5, 0.999))
    rec = Recorder(embedding_net, drop_p_override=0.0); rec.install()
    ret = train_gan.train_iter_gan(args, 11, text3, audio3, poses3, vid3, G, D, g_opt, d_opt)
    rec.remove()
    store = dict(epoch=11, n_words=V3, n_speakers=S3, g_seed=10, d_seed=11, batch_seed=300, batch=B3,
                 perm=rec.perms[0], eps=np.stack(rec.eps), loss_keys=np.array(sorted(ret)),
                 loss_vals=np.array([ret[k] for k in sorted(ret)]))
    for n_, p_ in G.named_parameters():
        store["ggn/" + n_] = np.array(float(p_.grad.double().norm()))
        store["gg/" + n_] = sampled(p_.grad, 64)
    for k, v_ in G.state_dict().items():
        if "num_batches_tracked" in k or "running" in k:
            store["gp/" + k] = v_.numpy()
    for k, v_ in D.state_dict().items():
        if "num_batches_tracked" in k or "running" in k:
            store["dp/" + k] = v_.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_train_b128.npz"), **store)
    inj = {f"{t}.eps": torch.from_numpy(e) for t, e in zip(("g1", "g2", "g3"), rec.eps)}
    inj["perm"] = torch.from_numpy(rec.perms[0])
    hp = dict(O.HP); hp["dropout_prob"] = 0.0
    # at B=128 the BN/conv reductions run over ~1e6 elements: the oracle is evaluated in fp64 (the truth both
    # fp32 implementations are measured against; the reference's fp32 result agrees with it to ~1e-6)
    og, od = O.clone_state(gst3, torch.float64), O.clone_state(dst3, torch.float64)

    class NoDrop(O.Rand):
        def keep_mask(self, name, shape, p, dtype=torch.float32):
            return torch.ones(shape, dtype=dtype)
    r = NoDrop(inject={k: (v.double() if v.is_floating_point() else v) for k, v in inj.items()})
    oret, extra = O.train_iter_gan(og, od, {}, {}, 11, text3, audio3.double(), poses3.double(), vid3, r, hp,
                                   fast_gru=True, want_grads=True)
    report["G3"] = dict(loss={k: abs(oret[k] - ret[k]) / max(abs(ret[k]), 1e-12) for k in ret},
                        g_grad_max=grad_err(extra["g_grads"], {n_: p_.grad for n_, p_ in G.named_parameters()})[0])

    # ------------------------------------------------------------------ G5: FGD autoencoder + Frechet distance
    ast = O.make_autoencoder_state(2)
    AE = embedding_net.EmbeddingNet(ref_args(), 27, 34, None, None, None, mode="pose")
    assert set(AE.state_dict().keys()) == set(ast.keys()), set(AE.state_dict().keys()) ^ set(ast.keys())
    AE.load_state_dict(O.clone_state(ast)); AE.eval()
    gp = torch.Generator().manual_seed(55)
    real = 0.1 * torch.randn(256, 34, 27, generator=gp)
    fake = real + 0.05 * torch.randn(256, 34, 27, generator=gp)
    with torch.no_grad():
        _, _, _, f_real, _, _, rec_real = AE(None, None, None, real, "pose", variational_encoding=False)
        _, _, _, f_fake, _, _, rec_fake = AE(None, None, None, fake, "pose", variational_encoding=False)
    o_real, _, _, o_rec = O.ae_forward(O.clone_state(ast), real, False)
    fr, fk = f_real.numpy(), f_fake.numpy()
    fd = Evaluator.calculate_frechet_distance(fk.mean(0), np.cov(fk, rowvar=False), fr.mean(0), np.cov(fr, rowvar=False))
    # near-singular: rank-deficient features (only 8 samples in 32-d) exercises the eps-offset / complex branch
    fr8, fk8 = fr[:8], fk[:8]
    try:
        fd8 = Evaluator.calculate_frechet_distance(fk8.mean(0), np.cov(fk8, rowvar=False), fr8.mean(0), np.cov(fr8, rowvar=False))
    except ValueError:
        fd8 = 1e10
    ofd, ofeat = O.fgd_scores(fk, fr)
    ofd8, _ = O.fgd_scores(fk8, fr8)
    report["G5"] = dict(feat=maxerr(o_real, f_real), recon=maxerr(o_rec, rec_real), fgd_rel=abs(ofd - fd) / abs(fd),
                        fgd8_rel=abs(ofd8 - fd8) / max(abs(fd8), 1e-12))
    # AE train step (train_feature_extractor.py:54-97 restated around the imported module)
    AE2 = embedding_net.EmbeddingNet(ref_args(), 27, 34, None, None, None, mode="pose")
    AE2.load_state_dict(O.clone_state(ast)); AE2.train()
    opt = torch.optim.Adam(AE2.parameters(), lr=0.0005, betas=(0.5, 0.999))
    tgt = real[:32]
    opt.zero_grad()
    _, _, _, _, _, _, recon = AE2(None, None, None, tgt, None, variational_encoding=False)
    import torch.nn.functional as F
    rl = torch.mean(F.l1_loss(recon, tgt, reduction="none"), dim=(1, 2))
    rl = rl + torch.mean(F.l1_loss(recon[:, 1:] - recon[:, :-1], tgt[:, 1:] - tgt[:, :-1], reduction="none"), dim=(1, 2))
    rl = torch.sum(rl)
    rl.backward(); opt.step()
    oast = O.clone_state(ast)
    oret, ogr = O.ae_train_iter(oast, {}, tgt)
    asd = AE2.state_dict()
    report["G5_train"] = dict(loss=abs(oret["loss"] - float(rl)) / float(rl),
                              step_err_over_lr=step_err(oast, asd, ast, 5e-4),
                              grad_max=grad_err({k: v for k, v in ogr.items() if v is not None},
                                                {n_: p_.grad for n_, p_ in AE2.named_parameters()})[0])
    store = dict(ae_seed=2, pose_seed=55, feat_real=fr, feat_fake=fk, fgd=fd, fgd8=fd8, feat_dist=ofeat,
                 recon_real=sampled(rec_real), train_loss=float(rl))
    for n_, p_ in AE2.named_parameters():
        if p_.grad is not None:
            store["ag/" + n_] = sampled(p_.grad)
    for k, v_ in asd.items():
        store["ap/" + k] = sampled(v_) if v_.is_floating_point() else v_.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_fgd.npz"), **store)

    with open(os.path.join(HERE, "golden_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
