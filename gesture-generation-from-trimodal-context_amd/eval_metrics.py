"""Test-set evaluation of the generator: the multimodal branch of scripts/train.py:evaluate_testset (:234-329).

Per batch: eval-mode forward with randomly drawn speaker ids (:257-260), L1 loss (:282), FGD feature push (:290), joint-position
MAE through convert_dir_vec_to_pose (utils/data_utils.py:77-98) over the non-seed frames (:293-305) and the acceleration
difference (:308-310).  The pose arithmetic runs in one device kernel (tg_pose_metrics); only three sums per batch reach the host.
"""
import random
import time

import torch

from . import ops


class _Meter:
    """utils/average_meter.py:AverageMeter (value weighted by batch size)."""
    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


def batch_metrics(out_dir_vec, target_dir_vec, mean_dir_vec, n_pre):
    """(l1, joint_mae, accel) of one batch, as train.py:282,304,310 compute them."""
    out, tgt = out_dir_vec.float().contiguous(), target_dir_vec.float().contiguous()
    B, T, D = out.shape
    mean = torch.as_tensor(mean_dir_vec, dtype=torch.float32, device=out.device).reshape(-1).contiguous()
    sums = ops.pose_metrics(out, tgt, mean, n_pre, torch.empty(3, dtype=torch.float64, device=out.device)).tolist()
    return sums[2] / (B * T * D), sums[0] / (B * (T - n_pre) * 30), sums[1] / (B * (T - 2) * 30)


def evaluate_testset(test_data_loader, generator, embed_space_evaluator, args, device=None):
    """test_data_loader yields (in_text, text_lengths, in_text_padded, _, target_vec, in_audio, in_spec, aux_info) like the
    reference's DataLoader.  Returns {'loss', 'joint_mae', 'accel' [, 'frechet', 'feat_dist']}."""
    device = device or next(generator.parameters()).device
    was_training = generator.training
    generator.train(False)
    if embed_space_evaluator:
        embed_space_evaluator.reset()
    losses, joint_mae, accel = _Meter(), _Meter(), _Meter()
    start = time.time()
    speaker_model = getattr(generator, "z_obj", None)
    with torch.no_grad():
        for data in test_data_loader:
            _, _, in_text_padded, _, target_vec, in_audio, _, _ = data
            B = target_vec.size(0)
            in_text_padded, in_audio, target = in_text_padded.to(device), in_audio.to(device), target_vec.to(device).float()
            ids = list(speaker_model.word2index.values())
            vid = torch.LongTensor([random.choice(ids) for _ in range(B)]).to(device)
            pre_seq = ops.make_pre_seq(target.contiguous(), torch.empty(B, target.shape[1], target.shape[2] + 1, device=device),
                                       args.n_pre_poses)
            out_dir_vec, *_ = generator(pre_seq, in_text_padded, in_audio, vid)
            l1, mae, acc = batch_metrics(out_dir_vec, target, args.mean_dir_vec, args.n_pre_poses)
            losses.update(l1, B); joint_mae.update(mae, B); accel.update(acc, B)
            if embed_space_evaluator:
                embed_space_evaluator.push_samples(in_text_padded, in_audio, out_dir_vec, target)
    generator.train(was_training)
    ret = {"loss": losses.avg, "joint_mae": joint_mae.avg, "accel": accel.avg, "elapsed_s": time.time() - start}
    if embed_space_evaluator and embed_space_evaluator.get_no_of_samples() > 0:
        ret["frechet"], ret["feat_dist"] = embed_space_evaluator.get_scores()
    return ret
