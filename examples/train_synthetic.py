#!/usr/bin/env python3
"""End-to-end example on synthetic data: the reference's training loop (scripts/train.py:65-232) on the MI355X path.

    python examples/train_synthetic.py --epochs 14 --iters-per-epoch 20 --batch 128 --out /tmp/run

Dataset -> DataLoader -> collate -> DeviceBatchFeeder -> captured GAN step (warm-up epochs, then the full GAN iteration) ->
evaluate_testset (L1 / joint MAE / accel / FGD) -> reference-format checkpoint.  Everything model-related goes through the
package's drop-in classes; only the data is synthetic (no TED LMDB in this environment).
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hip = importlib.import_module("gesture-generation-from-trimodal-context_amd")
data = importlib.import_module(hip.__name__ + ".data")
ckpt = importlib.import_module(hip.__name__ + ".checkpoint")
fgd = importlib.import_module(hip.__name__ + ".fgd")
metrics = importlib.import_module(hip.__name__ + ".eval_metrics")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=14)
    ap.add_argument("--iters-per-epoch", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--n-words", type=int, default=2000)
    ap.add_argument("--n-speakers", type=int, default=200)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    args = importlib.import_module(hip.__name__ + ".config").load_config("multimodal_context", name="synthetic", pose_dim=27)
    lang = hip.Vocab("words")
    for i in range(a.n_words - 4):
        lang.index_word(f"w{i}")
    lang.word_embedding_weights = (np.random.RandomState(0).randn(lang.n_words, 300) / np.sqrt(300)).astype(np.float32)
    spk = hip.Vocab.speakers(a.n_speakers)
    train_set = data.SyntheticSpeechMotionDataset(a.iters_per_epoch * a.batch, lang, spk, seed=1)
    val_set = data.SyntheticSpeechMotionDataset(2 * a.batch, lang, spk, seed=2)
    loader = torch.utils.data.DataLoader(train_set, batch_size=a.batch, shuffle=True, drop_last=True, num_workers=4,
                                         collate_fn=lambda items: data.collate(items, spk))

    generator, discriminator, _ = ckpt.init_model(args, lang, spk, args.pose_dim, dev)              # train.py:36-62
    trainer = hip.GanTrainer(generator, discriminator, args)
    ae = hip.EmbeddingNet(args, args.pose_dim, args.n_poses, None, None, None, mode="pose").to(dev)  # stands in for the trained FGD net
    evaluator = fgd.EmbeddingSpaceEvaluator.from_net(ae, args.n_pre_poses)
    args.mean_dir_vec = np.zeros(27, dtype=np.float32)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=a.batch, collate_fn=data.collate_reference)

    steps = {}                                    # one captured step per phase (warm-up / GAN): static shapes
    for epoch in range(a.epochs):
        generator.train(); discriminator.train()
        phase = epoch > args.loss_warmup
        t0 = time.time()
        last = None
        for text, vec, audio, vid in loader:
            if phase not in steps:
                step = hip.GraphedGanStep(trainer, epoch, text.to(dev), audio.to(dev), vec.to(dev), vid.to(dev), warmup_iters=1)
                steps[phase] = (step, data.DeviceBatchFeeder(*step.static))
            step, feeder = steps[phase]
            feeder.put(text, vec, audio, vid)
            feeder.ready()
            last = step()
        losses = last.to_dict()                   # the only host read of the epoch
        torch.cuda.synchronize()
        rate = a.iters_per_epoch * a.batch / (time.time() - t0)
        val = metrics.evaluate_testset(val_loader, generator, None, evaluator, args)                # train.py:234-329 (same signature)
        print(f"epoch {epoch:3d}  {rate:8.0f} clips/s  " + "  ".join(f"{k} {v:.4f}" for k, v in losses.items()) +
              "  | val " + "  ".join(f"{k} {v:.4f}" for k, v in val.items()) + f"  accel {val.accel:.4f}", flush=True)
    if a.out:
        os.makedirs(a.out, exist_ok=True)
        path = os.path.join(a.out, "synthetic_checkpoint_best.bin")
        ckpt.save_checkpoint({"args": args, "epoch": a.epochs, "lang_model": lang, "speaker_model": spk, "pose_dim": args.pose_dim,
                              "gen_dict": generator.state_dict(), "dis_dict": discriminator.state_dict()}, path)
        args2, gen2, *_ = ckpt.load_checkpoint_and_model(path, dev)
        print("saved and re-loaded", path, "->", type(gen2).__name__, "in eval mode:", not gen2.training)


if __name__ == "__main__":
    main()
