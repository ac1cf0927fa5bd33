"""Frechet Gesture Distance on the HIP autoencoder (model/embedding_space_evaluator.py:15-156) and the autoencoder's
training step (train_feature_extractor.py:54-97).  Feature extraction runs on the GPU; the 32x32 statistics are fp64
host maths exactly as in the reference (numpy mean / cov(rowvar=False) / matrix square root)."""
import numpy as np
import torch

from . import ops
from .optim import FusedAdam


def _sqrtm(a):
    """Principal matrix square root of a (possibly non-symmetric) real matrix, complex if needed."""
    try:
        from scipy import linalg
        r = linalg.sqrtm(a)
        return r[0] if isinstance(r, tuple) else r
    except ImportError:            # eigen-decomposition fallback (host maths only, not a kernel fallback)
        w, v = np.linalg.eig(a)
        return (v * np.sqrt(w.astype(complex))) @ np.linalg.inv(v)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """||mu1-mu2||^2 + Tr(S1) + Tr(S2) - 2 Tr sqrt(S1 S2)  (embedding_space_evaluator.py:103-156)."""
    mu1, mu2 = np.atleast_1d(mu1).astype(np.float64), np.atleast_1d(mu2).astype(np.float64)
    sigma1, sigma2 = np.atleast_2d(sigma1).astype(np.float64), np.atleast_2d(sigma2).astype(np.float64)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean = _sqrtm(sigma1.dot(sigma2))
    if not np.isfinite(covmean).all():                                  # singular product: offset the diagonals (:139-144)
        off = np.eye(sigma1.shape[0]) * eps
        covmean = _sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(covmean):                                        # (:147-151)
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def fgd_scores(generated_feats, real_feats):
    """(FGD, mean L1 distance of paired latents): EmbeddingSpaceEvaluator.get_scores (:74-101)."""
    g, r = np.asarray(generated_feats), np.asarray(real_feats)
    try:
        fd = frechet_distance(np.mean(g, axis=0), np.cov(g, rowvar=False), np.mean(r, axis=0), np.cov(r, rowvar=False))
    except ValueError:
        fd = 1e10
    return fd, float(np.mean(np.sum(np.abs(r - g), axis=1)))


class EmbeddingSpaceEvaluator:
    """model/embedding_space_evaluator.py:15-101 with the reference's constructor `(args, embed_net_path, lang_model, device)`
    (train.py:99-101 calls it unchanged): the checkpoint at embed_net_path is the autoencoder trainer's
    {'args', 'epoch', 'pose_dim', 'gen_dict'} (train_feature_extractor.py:155-157).  from_net() wraps an already built network."""

    def __init__(self, args, embed_net_path, lang_model, device):
        from .checkpoint import load_checkpoint
        from .modules import EmbeddingNet
        self.n_pre_poses = args.n_pre_poses
        ckpt = load_checkpoint(embed_net_path, device)
        self.pose_dim = ckpt["pose_dim"]
        word_embeddings = getattr(lang_model, "word_embedding_weights", None)
        self.net = EmbeddingNet(args, self.pose_dim, args.n_poses, getattr(lang_model, "n_words", None),
                                getattr(args, "wordembed_dim", None), word_embeddings, "pose").to(device)
        self.net.load_state_dict(ckpt["gen_dict"])
        self.net.train(False)
        self.reset()

    @classmethod
    def from_net(cls, net, n_pre_poses=4):
        self = cls.__new__(cls)
        self.net, self.n_pre_poses, self.pose_dim = net, n_pre_poses, net.pose_dim
        self.net.train(False)
        self.reset()
        return self

    @classmethod
    def from_checkpoint(cls, args, ckpt, device):
        """From an already loaded autoencoder checkpoint dict."""
        from .modules import EmbeddingNet
        net = EmbeddingNet(args, ckpt["pose_dim"], args.n_poses, None, None, None, mode="pose").to(device)
        net.load_state_dict(ckpt["gen_dict"])
        return cls.from_net(net, args.n_pre_poses)

    def reset(self):
        self.context_feat_list, self.real_feat_list, self.generated_feat_list, self.recon_err_diff = [], [], [], []

    def get_no_of_samples(self):
        return len(self.real_feat_list)

    def push_samples(self, context_text, context_spec, generated_poses, real_poses):
        eng = self.net.engine
        with torch.no_grad():
            r = eng.forward(real_poses.float(), training=False)
            g = eng.forward(generated_poses.float(), training=False)
            err = torch.empty(2, device=real_poses.device)
            ops.l1_mean(real_poses.float().contiguous(), r["recon"], err[0:1])
            ops.l1_mean(generated_poses.float().contiguous(), g["recon"], err[1:2])
        self.real_feat_list.append(r["feat"].cpu().numpy())
        self.generated_feat_list.append(g["feat"].cpu().numpy())
        e = err.tolist()
        self.recon_err_diff.append(e[1] - e[0])

    def get_scores(self):
        return fgd_scores(np.vstack(self.generated_feat_list), np.vstack(self.real_feat_list))


def eval_embed(in_text, in_audio, pre_poses, target_poses, net, mode=None):
    """train_eval/train_joint_embed.py:54-62 for the pose-mode network: (mean over clips of the per-clip mean L1 between
    reconstruction and target, reconstruction).  The loss stays a device scalar (the caller's .item() is the one host read)."""
    _, _, _, _, _, _, recon_poses = net(in_text, in_audio, pre_poses, target_poses, mode, variational_encoding=False)
    target = target_poses.float().contiguous()
    loss = ops.l1_mean(recon_poses.detach().contiguous(), target, torch.empty(1, device=target.device))   # clips have equal size: mean of means
    return loss.view(()), recon_poses


def evaluate_testset(test_data_loader, generator):
    """scripts/train_feature_extractor.py:26-51: average reconstruction L1 of the autoencoder over a loader that yields
    (target_poses, target_vec) like data_loader/h36m_loader.py; leaves the network in train mode like the reference."""
    device = next(generator.parameters()).device
    generator.train(False)
    s, n = 0.0, 0
    with torch.no_grad():
        for target_poses, target_vec in test_data_loader:
            loss, _ = eval_embed(None, None, None, target_vec.to(device), generator)
            s += float(loss) * target_vec.size(0)
            n += target_vec.size(0)
    generator.train(True)
    return {"loss": s / n}


def _ae_plan(net, target):
    """The fused step's plan (ops.AeStep: argument block + workspace) for this network and batch size, cached on its engine; None where the
    fused step does not apply (other shapes, CPU tensors)."""
    E = net.engine
    slab = E.slab.ensure()
    B = target.shape[0]
    if not target.is_cuda or not ops.AeStep.supported(slab, B, target.shape):
        return None
    p = getattr(E, "_ae_plan", None)
    buffers = dict(net.named_buffers())
    if p is None or p.cache_key != ops.AeStep.key(slab, buffers, B):
        assert not torch.cuda.is_current_stream_capturing(), "run one eager step before capturing (the plan allocates its workspace)"
        slab.zero_grad()                 # fc_logvar gets no gradient (:58): its slots stay zero, every other one is written by each step
        p = E._ae_plan = ops.AeStep(slab, buffers, B)
    return p


def _adam_args(opt):
    s = opt.slab
    return s.m, s.v, opt.lr, opt.betas[0], opt.betas[1], opt.eps


def train_iter(args, epoch, target_data, net, optim):
    """scripts/train_feature_extractor.py:54-97 with variational_encoding=False; `optim` is a FusedAdam over `net`."""
    E = net.engine
    target = target_data.float().contiguous()
    loss = torch.empty(1, device=target.device)
    plan = _ae_plan(net, target)
    if plan is not None:                 # 18 launches, the optimiser step inside the last one (csrc/ae_step.hip)
        plan.run(target, loss, adam=_adam_args(optim))
        return {"loss": float(loss)}
    E.slab.ensure().zero_grad()
    res = E.forward(target, training=True, save=True)
    d_recon = torch.empty_like(target)
    ops.ae_loss(res["recon"], target, loss, d_recon)
    E.backward(res["tape"], d_recon)
    optim.step()
    return {"loss": float(loss)}


class AutoencoderTrainer:
    """train_feature_extractor.py:train_iter with variational_encoding=False: reconstruction L1 + L1 of frame differences,
    summed over the batch; Adam(lr 5e-4, betas (0.5, 0.999)).
    fused (default: where supported): the whole step, optimiser included, as the 18 launches of csrc/ae_step.hip instead of the ~105 launches of
    the layer-by-layer engine (kept for other shapes and as the test reference)."""

    def __init__(self, net, lr=5e-4, fused=None):
        self.net, self.E = net, net.engine
        self.opt = FusedAdam(self.E, lr=lr, betas=(0.5, 0.999))
        self.fused = fused
        self._plan = None
        self.last = {}

    def _fused_plan(self, target):
        plan = None if self.fused is False else _ae_plan(self.net, target)
        if plan is None and self.fused:
            raise RuntimeError(f"fused autoencoder step: unsupported batch / shapes {tuple(target.shape)}")
        self._plan = plan
        return plan

    def train_iter(self, target, keep_outputs=False):
        E = self.E
        target = target.float().contiguous()
        plan = self._fused_plan(target)
        loss = torch.empty(1, device=target.device)
        if plan is not None:
            recon = torch.empty_like(target) if keep_outputs else None
            feat = torch.empty(target.shape[0], 32, device=target.device) if keep_outputs else None
            plan.run(target, loss, recon=recon, feat=feat, adam=_adam_args(self.opt))          # gradients AND the Adam step
            if keep_outputs:
                self.last = {"recon": recon, "feat": feat}
            return loss
        E.slab.ensure().zero_grad()
        res = E.forward(target, training=True, save=True)
        d_recon = torch.empty_like(target)
        ops.ae_loss(res["recon"], target, loss, d_recon)
        E.backward(res["tape"], d_recon)
        self.opt.step()
        if keep_outputs:
            self.last = {"recon": res["recon"], "feat": res["feat"]}
        return loss
