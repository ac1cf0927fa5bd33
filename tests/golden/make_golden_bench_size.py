#!/usr/bin/env python3
"""g12: the REAL reference's train_iter_gan (scripts/train_eval/train_gan.py:13-103) at the size bench.py times -- B = 128 clips, V = 20 000 words,
S = 1 371 speaker rows, epoch 11 (full GAN iteration) -- stored as scalars only (build container only; imports /root/reference like make_golden.py):

  * the loss dict, every gradient's norm and 64 sampled entries (as g3);
  * the word-embedding gradient (multimodal_context_net.py:40-41, SURVEY Q8: trainable, no padding_idx): norm of row 0 (PAD, the dense hot row),
    number of rows with a non-zero gradient, the norm over all other rows;
  * 64 sampled entries of every parameter AFTER both Adam steps (scripts/train.py:104-109), BatchNorm buffers and counters.

As in g3 every F.dropout runs with p = 0 and nn.GRU's inter-layer dropout is off (it draws inside ATen and cannot be recorded); eps and the speaker
permutation are the reference's own draws, recorded.  Also checks oracle/ref_model.py (fp64) against the reference on the full tensors and writes
the errors to golden_report_bench_size.json.

    python tests/golden/make_golden_bench_size.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import Recorder, build_ref_models, grad_err, import_reference, sampled          # noqa: E402  (the same hooks and model builder)


def main():
    from oracle import ref_model as O
    embedding_net, mcn, vocab, train_gan, _ = import_reference()
    torch.set_num_threads(8)
    V, S, B = 20000, 1371, 128
    gst, dst = O.make_generator_state(20, V, S), O.make_discriminator_state(21)
    text, audio, vid, poses = O.make_batch(1200, B, V, S)
    args, G, D = build_ref_models(mcn, vocab, O.clone_state(gst), O.clone_state(dst), V, S)
    G.train(); D.train(); G.gru.dropout = 0.0; D.gru.dropout = 0.0
    g_opt = torch.optim.Adam(G.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(D.parameters(), lr=args.learning_rate * args.discriminator_lr_weight, betas=(0.5, 0.999))
    d_grads = {}
    d_step = d_opt.step

    def rec_step(*a, **k):                      # the discriminator's gradients exist only until the generator step's D(out) backward overwrites them
        for n_, p_ in D.named_parameters():
            d_grads[n_] = None if p_.grad is None else p_.grad.detach().clone()
        return d_step(*a, **k)
    d_opt.step = rec_step
    rec = Recorder(embedding_net, drop_p_override=0.0); rec.install()
    ret = train_gan.train_iter_gan(args, 11, text, audio, poses, vid, G, D, g_opt, d_opt)
    rec.remove()
    store = dict(epoch=11, n_words=V, n_speakers=S, g_seed=20, d_seed=21, batch_seed=1200, batch=B, perm=rec.perms[0], eps=np.stack(rec.eps),
                 loss_keys=np.array(sorted(ret)), loss_vals=np.array([ret[k] for k in sorted(ret)]))
    for n_, p_ in G.named_parameters():
        store["ggn/" + n_] = np.array(float(p_.grad.double().norm()))
        store["gg/" + n_] = sampled(p_.grad, 64)
    for n_, g_ in d_grads.items():
        if g_ is not None:
            store["dgn/" + n_] = np.array(float(g_.double().norm()))
            store["dg/" + n_] = sampled(g_, 64)
    eg = G.text_encoder.embedding.weight.grad.double()
    row_norm = eg.norm(dim=1)
    store["emb_row0_norm"] = np.array(float(row_norm[0]))
    store["emb_touched_rows"] = np.array(int((row_norm > 0).sum()))
    store["emb_other_rows_norm"] = np.array(float(row_norm[1:].norm()))
    store["text_pad_fraction"] = np.array(float((text == 0).float().mean()))
    for k, v_ in G.state_dict().items():
        if O.is_tcn_alias(k):
            continue
        store["gp/" + k] = sampled(v_, 64) if v_.is_floating_point() else v_.numpy()
    for k, v_ in D.state_dict().items():
        store["dp/" + k] = sampled(v_, 64) if v_.is_floating_point() else v_.numpy()
    np.savez_compressed(os.path.join(HERE, "g12_train_bench_size.npz"), **store)

    # ---- the oracle (fp64) on the same inputs and draws, against the reference on the FULL tensors
    inj = {f"{t}.eps": torch.from_numpy(e).double() for t, e in zip(("g1", "g2", "g3"), rec.eps)}
    inj["perm"] = torch.from_numpy(rec.perms[0])
    hp = dict(O.HP); hp["dropout_prob"] = 0.0
    og, od = O.clone_state(gst, torch.float64), O.clone_state(dst, torch.float64)

    class NoDrop(O.Rand):
        def keep_mask(self, name, shape, p, dtype=torch.float32):
            return torch.ones(shape, dtype=dtype)
    oret, extra = O.train_iter_gan(og, od, {}, {}, 11, text, audio.double(), poses.double(), vid, NoDrop(inject=inj), hp, fast_gru=True, want_grads=True)
    report = dict(torch=torch.__version__, size=dict(B=B, V=V, S=S),
                  loss={k: abs(oret[k] - ret[k]) / max(abs(ret[k]), 1e-12) for k in ret},
                  g_grad_max=grad_err(extra["g_grads"], {n_: p_.grad for n_, p_ in G.named_parameters()})[0],
                  d_grad_max=grad_err(extra["d_grads"], d_grads)[0],
                  emb_row0_norm=float(store["emb_row0_norm"]), emb_touched_rows=int(store["emb_touched_rows"]),
                  emb_other_rows_norm=float(store["emb_other_rows_norm"]), text_pad_fraction=float(store["text_pad_fraction"]))
    with open(os.path.join(HERE, "golden_report_bench_size.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
