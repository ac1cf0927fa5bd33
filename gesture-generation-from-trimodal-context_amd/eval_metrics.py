"""Test-set evaluation of the generator: scripts/train.py:evaluate_testset (:234-329), multimodal_context and gesture_autoencoder
branches, with the reference's signature so that train.py:117 can call it unchanged.

Per batch: eval-mode forward with randomly drawn speaker ids (:255-260; None when the generator has no speaker Vocab), L1 loss
(:282), FGD feature push (:290), joint-position MAE through convert_dir_vec_to_pose (utils/data_utils.py:77-98) over the non-seed
frames (:293-305) and the acceleration difference (:308-310).  The pose arithmetic runs in one device kernel (tg_pose_metrics);
only three sums per batch reach the host.
"""
import logging
import random
import time

import numpy as np
import torch

from . import ops
from .checkpoint import get_speaker_model
from .fgd import eval_embed

# utils/data_utils.py:14-15 (adjacency and bone length): data, also baked into tg_pose_metrics
dir_vec_pairs = [(0, 1, 0.26), (1, 2, 0.18), (2, 3, 0.14), (1, 4, 0.22), (4, 5, 0.36),
                 (5, 6, 0.33), (1, 7, 0.22), (7, 8, 0.36), (8, 9, 0.33)]


class AverageMeter:
    """utils/average_meter.py: running average weighted by the update's n."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class EvalResult(dict):
    """The reference's return dict ({'loss', 'joint_mae'[, 'frechet', 'feat_dist']}); the acceleration difference, which the
    reference computes (:308-310) but only logs, and the wall time ride along as attributes."""
    accel = 0.0
    elapsed_s = 0.0


def convert_dir_vec_to_pose(vec):
    """utils/data_utils.py:77-98 on the host (numpy): (…, 27) or (…, 9, 3) direction vectors -> (…, 10, 3) joint positions.
    For inspection / plotting; the evaluation loop integrates bones on the device (tg_pose_metrics)."""
    vec = np.array(vec)
    if vec.shape[-1] != 3:
        vec = vec.reshape(vec.shape[:-1] + (-1, 3))
    assert 2 <= vec.ndim <= 4
    joint_pos = np.zeros(vec.shape[:-2] + (10, 3))
    for j, (a, b, length) in enumerate(dir_vec_pairs):
        joint_pos[..., b, :] = joint_pos[..., a, :] + length * vec[..., j, :]
    return joint_pos


def batch_metrics(out_dir_vec, target_dir_vec, mean_dir_vec, n_pre):
    """(l1, joint_mae, accel) of one batch, as train.py:282,304,310 compute them."""
    out, tgt = out_dir_vec.float().contiguous(), target_dir_vec.float().contiguous()
    B, T, D = out.shape
    mean = torch.as_tensor(np.asarray(mean_dir_vec, dtype=np.float32).reshape(-1), device=out.device).contiguous()
    sums = ops.pose_metrics(out, tgt, mean, n_pre, torch.empty(3, dtype=torch.float64, device=out.device)).tolist()
    return sums[2] / (B * T * D), sums[0] / (B * (T - n_pre) * 30), sums[1] / (B * (T - 2) * 30)


def evaluate_testset(test_data_loader, generator, loss_fn, embed_space_evaluator, args):
    """train.py:234-329.  test_data_loader yields (in_text, text_lengths, in_text_padded, _, target_vec, in_audio, in_spec,
    aux_info) like the reference's DataLoader (default_collate_fn).  `loss_fn` is unused by the two models on the hot path
    (kept for the signature).  Leaves the generator in train mode, like the reference (:313)."""
    if args.model not in ("multimodal_context", "gesture_autoencoder"):
        raise NotImplementedError(f"model {args.model!r} is a baseline outside the hot path")
    device = next(generator.parameters()).device
    generator.train(False)
    if embed_space_evaluator:
        embed_space_evaluator.reset()
    losses, joint_mae, accel = AverageMeter("loss"), AverageMeter("mae_on_joint"), AverageMeter("accel")
    start = time.time()
    with torch.no_grad():
        for iter_idx, data in enumerate(test_data_loader, 0):
            in_text, text_lengths, in_text_padded, _, target_vec, in_audio, in_spec, aux_info = data
            batch_size = target_vec.size(0)
            in_text_padded, in_audio = in_text_padded.to(device), in_audio.to(device)
            target = target_vec.to(device).float()
            # speaker input (:255-260)
            speaker_model = get_speaker_model(generator)
            if speaker_model:
                vid_indices = [random.choice(list(speaker_model.word2index.values())) for _ in range(batch_size)]
                vid_indices = torch.LongTensor(vid_indices).to(device)
            else:
                vid_indices = None
            if args.model == "gesture_autoencoder":            # :270-271; no joint metrics for the autoencoder (:287)
                loss, _ = eval_embed(in_text_padded, in_audio, target[:, 0:args.n_pre_poses], target, generator)
                losses.update(float(loss), batch_size)
                continue
            pre_seq = ops.make_pre_seq(target.contiguous(), torch.empty(batch_size, target.shape[1], target.shape[2] + 1, device=device),
                                       args.n_pre_poses)
            out_dir_vec, *_ = generator(pre_seq, in_text_padded, in_audio, vid_indices)
            l1, mae, acc = batch_metrics(out_dir_vec, target, args.mean_dir_vec, args.n_pre_poses)
            losses.update(l1, batch_size)
            if embed_space_evaluator:
                embed_space_evaluator.push_samples(in_text_padded, in_audio, out_dir_vec, target)
            joint_mae.update(mae, batch_size)
            accel.update(acc, batch_size)
    generator.train(True)                                       # back to training mode (:313)
    ret_dict = EvalResult({"loss": losses.avg, "joint_mae": joint_mae.avg})
    ret_dict.accel, ret_dict.elapsed_s = accel.avg, time.time() - start
    if embed_space_evaluator and embed_space_evaluator.get_no_of_samples() > 0:
        frechet_dist, feat_dist = embed_space_evaluator.get_scores()
        logging.info("[VAL] loss: {:.3f}, joint mae: {:.5f}, accel diff: {:.5f}, FGD: {:.3f}, feat_D: {:.3f} / {:.1f}s".format(
            losses.avg, joint_mae.avg, accel.avg, frechet_dist, feat_dist, ret_dict.elapsed_s))
        ret_dict["frechet"] = frechet_dist
        ret_dict["feat_dist"] = feat_dist
    else:
        logging.info("[VAL] loss: {:.3f}, joint mae: {:.3f} / {:.1f}s".format(losses.avg, joint_mae.avg, ret_dict.elapsed_s))
    return ret_dict
