#!/usr/bin/env python3
"""Golden vectors for the other PoseGenerator variants (input_context 'audio' | 'text' | 'none', z_type 'random' | 'none'),
from the REAL reference (build container only; see make_golden.py for the method).

One post-warm-up ``train_iter_gan`` per variant at B=4 with every random draw recorded; stores losses, gradient norms,
sampled gradients, BatchNorm buffers -> g6_variants.npz, and the oracle-vs-reference errors -> golden_report_variants.json.

    python tests/golden/make_golden_variants.py
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

VARIANTS = (("audio", "random"), ("text", "none"), ("none", "speaker"), ("text", "speaker"), ("both", "random"))


class Recorder(MG.Recorder):
    """+ torch.randn (the 'random' z vector, multimodal_context_net.py:134)."""

    def install(self):
        super().install()
        self.zs = []
        self._randn = torch.randn

        def randn(*size, **kw):
            kw.pop("device", None)
            if "generator" in kw:
                return self._randn(*size, **kw)
            z = self._randn(*size, generator=self.gen, **kw)
            self.zs.append(z.numpy().copy())
            return z
        torch.randn = randn

    def remove(self):
        super().remove()
        torch.randn = self._randn


def make_reference_checkpoint(mcn, vocab):
    """A (tiny) checkpoint written by the reference's own classes in the reference's format (train.py:153-157 through
    utils/train_utils.py:save_checkpoint == torch.save): pickled argparse.Namespace + model.vocab.Vocab objects + state dicts.
    tests/test_checkpoint_cpu.py loads it through the package's checkpoint module."""
    args = MG.ref_args(hidden=8, layers=1)
    args.model, args.wordembed_dim, args.name = "multimodal_context", 8, "tiny"
    lang = vocab.Vocab("words")
    for w in "the quick brown fox jumps over a lazy dog".split():
        lang.index_word(w)
    lang.word_embedding_weights = np.linspace(-1, 1, lang.n_words * 8, dtype=np.float32).reshape(lang.n_words, 8)
    spk = vocab.Vocab("vid", insert_default_tokens=False)
    for i in range(4):
        spk.index_word(f"spk{i}")
    torch.manual_seed(99)
    G = mcn.PoseGenerator(args, pose_dim=27, n_words=lang.n_words, word_embed_size=8, word_embeddings=lang.word_embedding_weights,
                          z_obj=spk)
    D = mcn.ConvDiscriminator(27)
    path = os.path.join(HERE, "g7_reference_checkpoint.bin")
    torch.save({"args": args, "epoch": 7, "lang_model": lang, "speaker_model": spk, "pose_dim": 27,
                "gen_dict": G.state_dict(), "dis_dict": D.state_dict()}, path)
    return dict(bytes=os.path.getsize(path), gen_keys=len(G.state_dict()), dis_keys=len(D.state_dict()),
                n_words=lang.n_words, n_speakers=spk.n_words,
                gen_abs_sum=float(sum(v.double().abs().sum() for v in G.state_dict().values() if v.is_floating_point())))


def main():
    from oracle import ref_model as O
    embedding_net, mcn, vocab, train_gan, _ = MG.import_reference()
    torch.set_num_threads(8)
    V, S, B, epoch = 512, 17, 4, 11
    text, audio, vid, poses = O.make_batch(100, B, V, S)
    dst0 = O.make_discriminator_state(1)
    report, store = OrderedDict(), dict(n_words=V, n_speakers=S, batch_seed=100, d_seed=1, g_seed=20, epoch=epoch,
                                        variants=np.array(["/".join(v) for v in VARIANTS]))
    for ctx, zt in VARIANTS:
        z_mode = zt if zt in ("speaker", "random") else None
        gst0 = O.make_generator_state(20, V, S, input_context=ctx, z_mode=z_mode)
        args = MG.ref_args()
        args.input_context, args.z_type = ctx, zt
        spk = vocab.Vocab("vid", insert_default_tokens=False)
        for i in range(S - 1):
            spk.index_word(f"spk{i}")
        z_obj = spk if zt == "speaker" else (1 if zt == "random" else None)          # train.py:82-87
        G = mcn.PoseGenerator(args, pose_dim=27, n_words=V, word_embed_size=300,
                              word_embeddings=np.zeros((V, 300), dtype=np.float32), z_obj=z_obj)
        D = mcn.ConvDiscriminator(27)
        assert set(G.state_dict().keys()) == set(gst0.keys()), set(G.state_dict().keys()) ^ set(gst0.keys())
        G.load_state_dict(O.clone_state(gst0), strict=True)
        D.load_state_dict(O.clone_state(dst0), strict=True)
        G.train(); D.train(); G.gru.dropout = 0.0; D.gru.dropout = 0.0
        g_opt = torch.optim.Adam(G.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
        d_opt = torch.optim.Adam(D.parameters(), lr=args.learning_rate * args.discriminator_lr_weight, betas=(0.5, 0.999))
        rec = Recorder(embedding_net); rec.install()
        ret = train_gan.train_iter_gan(args, epoch, text, audio, poses, vid, G, D, g_opt, d_opt)
        rec.remove()

        tags = ["g1", "g2"] + (["g3"] if z_mode else [])
        inj = {}
        if ctx != "none":                        # the text encoder draws its 9 dropout masks whenever any context is used
            inj.update(MG.masks_to_inject(rec, tags, 0.3))
        for tg, e in zip(tags, rec.eps):
            inj[f"{tg}.eps"] = torch.from_numpy(e)
        for tg, z in zip(tags, rec.zs):
            inj[f"{tg}.z"] = torch.from_numpy(z)
        if rec.perms:
            inj["perm"] = torch.from_numpy(rec.perms[0])
        ones = {f"{t}.gru.drop{l}": torch.ones(1).expand(B, 34, 600) for t in tags for l in range(3)}
        ones.update({f"{t}.gru.drop{l}": torch.ones(1).expand(B, 28, 128) for t in ("d_real", "d_fake", "d_out") for l in range(3)})
        og, od = O.clone_state(gst0), O.clone_state(dst0)
        oret, extra = O.train_iter_gan(og, od, {}, {}, epoch, text, audio, poses, vid, O.Rand(inject={**inj, **ones}), dict(O.HP),
                                       want_grads=True, input_context=ctx, z_type=zt)
        assert set(oret) == set(ret), (oret, ret)
        ref_g = {n_: p_.grad for n_, p_ in G.named_parameters() if p_.grad is not None}
        mine = {k: v for k, v in extra["g_grads"].items() if k in ref_g and v is not None}
        assert set(mine) == set(ref_g), set(mine) ^ set(ref_g)
        gsd = G.state_dict()
        key = f"{ctx}_{zt}"
        report[key] = dict(loss={k: abs(oret[k] - ret[k]) / max(abs(ret[k]), 1e-12) for k in ret},
                           g_grad_max=MG.grad_err(mine, ref_g)[0],
                           g_step_err_over_lr=MG.step_err({k: og[k] for k in mine}, gsd, gst0, 5e-4),
                           bn_buffers_max=max(MG.maxerr(og[k], gsd[k]) for k in og if "running" in k),
                           n_grads=len(ref_g), n_masks=len(rec.masks), n_eps=len(rec.eps), n_z=len(rec.zs))
        store[f"{key}/loss_keys"] = np.array(sorted(ret))
        store[f"{key}/loss_vals"] = np.array([ret[k] for k in sorted(ret)])
        store[f"{key}/grad_keys"] = np.array(sorted(ref_g))
        store[f"{key}/masks"] = (np.packbits(np.stack([m.reshape(-1) for m in rec.masks]), axis=1) if rec.masks else np.zeros((0, 0), np.uint8))
        store[f"{key}/mask_shape"] = np.array(rec.masks[0].shape if rec.masks else (0,))
        store[f"{key}/eps"] = np.stack(rec.eps) if rec.eps else np.zeros((0,))
        store[f"{key}/z"] = np.stack(rec.zs) if rec.zs else np.zeros((0,))
        store[f"{key}/perm"] = rec.perms[0] if rec.perms else np.zeros((0,), np.int64)
        for k, g_ in ref_g.items():
            store[f"{key}/gg/{k}"] = MG.sampled(g_, 256)
            store[f"{key}/ggn/{k}"] = np.array(float(g_.double().norm()))
        for k, v_ in gsd.items():
            if "running" in k or "num_batches" in k:
                store[f"{key}/gp/{k}"] = v_.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_variants.npz"), **store)
    report["checkpoint"] = make_reference_checkpoint(mcn, vocab)
    with open(os.path.join(HERE, "golden_report_variants.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
