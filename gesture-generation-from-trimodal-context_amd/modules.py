"""Drop-in nn.Module surface of the reference's hot path, backed by the HIP engines.

Same constructor signatures, forward signatures / returns, attribute names and state_dict key set as
model/multimodal_context_net.py (PoseGenerator :64-160, ConvDiscriminator :207-252) and the pose-mode
model/embedding_net.py:EmbeddingNet (:262-308), so reference checkpoints load with strict=True and the reference's
own train_iter_gan / evaluate_testset / generate_gestures can drive these modules unchanged.

The torch.nn layers instantiated here are PARAMETER CONTAINERS only (they give the reference's names, shapes and
default initialisation); their forward() is never called.  All arithmetic runs in libtrimodal_hip.so through
engine.py; a missing library raises at first use -- there is no PyTorch fallback.
"""
import warnings

import torch
import torch.nn as nn

from . import vocab
from . import ops
from .engine import AutoencoderEngine, DiscriminatorEngine, GeneratorEngine


def _wn_conv(cin, cout, k, dilation):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.nn.utils.weight_norm(nn.Conv1d(cin, cout, k, stride=1, padding=(k - 1) * dilation, dilation=dilation))


class _TemporalBlockParams(nn.Module):
    """Parameters of one TCN block with the reference's double registration (conv1 == net.0, conv2 == net.4)."""

    def __init__(self, cin, cout, k, dilation, p):
        super().__init__()
        assert cin == cout, "the path uses embed_size == hidden_size (no 1x1 downsample conv)"
        self.conv1 = _wn_conv(cin, cout, k, dilation)
        self.conv2 = _wn_conv(cout, cout, k, dilation)
        idle = [nn.Identity() for _ in range(6)]
        self.net = nn.Sequential(self.conv1, idle[0], idle[1], idle[2], self.conv2, idle[3], idle[4], idle[5])


class _TCNParams(nn.Module):
    def __init__(self, cin, channels, k, p):
        super().__init__()
        self.network = nn.Sequential(*[_TemporalBlockParams(cin if i == 0 else channels[i - 1], c, k, 2 ** i, p)
                                       for i, c in enumerate(channels)])


class _WavEncoderParams(nn.Module):
    def __init__(self):
        super().__init__()
        self.feat_extractor = nn.Sequential(
            nn.Conv1d(1, 16, 15, stride=5, padding=1600), nn.BatchNorm1d(16), nn.Identity(),
            nn.Conv1d(16, 32, 15, stride=6), nn.BatchNorm1d(32), nn.Identity(),
            nn.Conv1d(32, 64, 15, stride=6), nn.BatchNorm1d(64), nn.Identity(),
            nn.Conv1d(64, 32, 15, stride=6))


class _TextEncoderParams(nn.Module):
    def __init__(self, args, n_words, embed_size, pre_trained_embedding, dropout):
        super().__init__()
        if pre_trained_embedding is not None:
            assert pre_trained_embedding.shape[0] == n_words and pre_trained_embedding.shape[1] == embed_size
            self.embedding = nn.Embedding.from_pretrained(torch.as_tensor(pre_trained_embedding, dtype=torch.float32),
                                                          freeze=args.freeze_wordembed)
        else:
            self.embedding = nn.Embedding(n_words, embed_size)
        self.tcn = _TCNParams(embed_size, [args.hidden_size] * args.n_layers, 2, dropout)
        self.decoder = nn.Linear(args.hidden_size, 32)
        self.decoder.bias.data.fill_(0)
        self.decoder.weight.data.normal_(0, 0.01)


class _Bridge(torch.autograd.Function):
    """Connects an engine's hand-written backward to torch.autograd so `loss.backward()` works on these modules.
    Parameter gradients are accumulated into param.grad (views of the gradient slab) as a side effect; only
    gradients w.r.t. tensor INPUTS are returned to autograd."""

    @staticmethod
    def forward(ctx, anchor, runner, *inputs):
        outs, back = runner(*inputs)
        ctx.set_materialize_grads(False)
        ctx.back = back
        ctx.n_in = len(inputs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        din = ctx.back(*grads)
        return (None, None) + tuple(din)


class PoseGenerator(nn.Module):
    def __init__(self, args, pose_dim, n_words, word_embed_size, word_embeddings, z_obj=None):
        super().__init__()
        self.pre_length = args.n_pre_poses
        self.gen_length = args.n_poses - args.n_pre_poses
        self.z_obj = z_obj
        self.input_context = args.input_context
        if self.input_context == "both":
            self.in_size = 32 + 32 + pose_dim + 1          # audio_feat + text_feat + last pose + constraint bit
        elif self.input_context == "none":
            self.in_size = pose_dim + 1
        elif self.input_context in ("audio", "text"):
            self.in_size = 32 + pose_dim + 1
        else:
            raise ValueError(f"input_context {self.input_context!r}")
        self.audio_encoder = _WavEncoderParams()
        self.text_encoder = _TextEncoderParams(args, n_words, word_embed_size, word_embeddings, args.dropout_prob)
        self.speaker_embedding = None
        if self.z_obj:
            self.z_size = 16
            self.in_size += self.z_size
            if isinstance(self.z_obj, vocab.Vocab):
                self.speaker_embedding = nn.Sequential(nn.Embedding(z_obj.n_words, self.z_size),
                                                       nn.Linear(self.z_size, self.z_size))
                self.speaker_mu = nn.Linear(self.z_size, self.z_size)
                self.speaker_logvar = nn.Linear(self.z_size, self.z_size)
            # else: z is a plain random vector (multimodal_context_net.py:95-96)
        self.hidden_size = args.hidden_size
        self.n_layers = args.n_layers
        self.dropout_prob = args.dropout_prob
        self.pose_dim = pose_dim
        self.gru = nn.GRU(self.in_size, hidden_size=self.hidden_size, num_layers=args.n_layers, batch_first=True,
                          bidirectional=True, dropout=args.dropout_prob)
        self.out = nn.Sequential(nn.Linear(self.hidden_size, self.hidden_size // 2), nn.Identity(),
                                 nn.Linear(self.hidden_size // 2, pose_dim))
        self.do_flatten_parameters = False
        self._engine = None
        self._replay_draws = []      # parity tests: queue of {'g.eps' | 'g.z' | mask name: tensor}, one entry consumed per forward call

    @property
    def engine(self) -> GeneratorEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", GeneratorEngine(self))
        return self._engine

    def forward(self, pre_seq, in_text, in_audio, vid_indices=None):
        assert vid_indices is not None or self.speaker_embedding is None
        eng = self.engine
        eng.rng.advance()            # every call draws fresh dropout masks / eps (the trainer advances once per iteration itself)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

        inject = self._replay_draws.pop(0) if self._replay_draws else None

        def runner():
            res = eng.forward(pre_seq.float(), in_text, in_audio.float(), vid_indices, training=self.training, save=need_grad,
                              inject=inject)

            def back(d_out, d_z, d_mu, d_lv):
                # z = mu + eps*std: a gradient on z folds into mu/logvar inside engine.backward via the GRU input path;
                # an explicit d_z from the caller (rare) is added the same way
                tp = res["tape"]
                dm = d_mu if d_mu is not None else None
                dl = d_lv if d_lv is not None else None
                if d_z is not None and tp.get("eps") is not None:
                    from . import ops
                    dm = ops.zeros_like(res["mu"]) if dm is None else dm.contiguous().clone()
                    dl = ops.zeros_like(res["mu"]) if dl is None else dl.contiguous().clone()
                    ops.reparam_bwd(d_z.contiguous(), tp["logvar"], tp["eps"], dm, dl)
                eng.backward(tp, d_out if d_out is not None else ops.zeros_like(res["out"]), dm, dl)
                return ()
            return (res["out"], res["z"], res["mu"], res["logvar"]), back

        if not need_grad:
            outs, _ = runner()
            return outs
        anchor = next(p for p in self.parameters() if p.requires_grad)
        return _Bridge.apply(anchor, runner)


class ConvDiscriminator(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = 64
        self.pre_conv = nn.Sequential(nn.Conv1d(input_size, 16, 3), nn.BatchNorm1d(16), nn.Identity(),
                                      nn.Conv1d(16, 8, 3), nn.BatchNorm1d(8), nn.Identity(), nn.Conv1d(8, 8, 3))
        self.gru = nn.GRU(8, hidden_size=self.hidden_size, num_layers=4, bidirectional=True, dropout=0.3, batch_first=True)
        self.out = nn.Linear(self.hidden_size, 1)
        self.out2 = nn.Linear(28, 1)
        self.do_flatten_parameters = False
        self._engine = None
        self._replay_draws = []      # parity tests: queue of {'d.gru.drop<l>': mask}, one entry consumed per forward call

    @property
    def engine(self) -> DiscriminatorEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", DiscriminatorEngine(self))
        return self._engine

    def forward(self, poses, in_text=None):
        eng = self.engine
        eng.rng.advance()
        params_need = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        need_grad = params_need or (torch.is_grad_enabled() and poses.requires_grad)
        inject = self._replay_draws.pop(0) if self._replay_draws else None

        def runner(poses_in):
            res = eng.forward(poses_in.float(), training=self.training, save=need_grad, inject=inject)

            def back(d_prob):
                from . import ops
                d_logit = ops.sigmoid_bwd(d_prob.contiguous(), res["prob"], torch.empty_like(res["prob"]))
                dp = eng.backward(res["tape"], d_logit, param_grads=params_need, need_dposes=poses_in.requires_grad)
                return (dp,)
            return (res["prob"],), back

        if not need_grad:
            outs, _ = runner(poses)
            return outs[0]
        anchor = next(p for p in self.parameters() if p.requires_grad) if params_need else poses
        return _Bridge.apply(anchor, runner, poses)[0]


class _ConvNormReluParams(nn.Sequential):
    def __init__(self, cin, cout, downsample=False):
        k, s = (4, 2) if downsample else (3, 1)
        super().__init__(nn.Conv1d(cin, cout, k, stride=s), nn.BatchNorm1d(cout), nn.Identity())


class _PoseEncoderParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.Sequential(_ConvNormReluParams(dim, 32), _ConvNormReluParams(32, 64), _ConvNormReluParams(64, 64, True),
                                 nn.Conv1d(64, 32, 3))
        self.out_net = nn.Sequential(nn.Linear(384, 256), nn.BatchNorm1d(256), nn.Identity(), nn.Linear(256, 128),
                                     nn.BatchNorm1d(128), nn.Identity(), nn.Linear(128, 32))
        self.fc_mu = nn.Linear(32, 32)
        self.fc_logvar = nn.Linear(32, 32)


class _PoseDecoderParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.pre_net = nn.Sequential(nn.Linear(32, 64), nn.BatchNorm1d(64), nn.Identity(), nn.Linear(64, 136))
        self.net = nn.Sequential(nn.ConvTranspose1d(4, 32, 3), nn.BatchNorm1d(32), nn.Identity(),
                                 nn.ConvTranspose1d(32, 32, 3), nn.BatchNorm1d(32), nn.Identity(),
                                 nn.Conv1d(32, 32, 3), nn.Conv1d(32, dim, 3))


class EmbeddingNet(nn.Module):
    """FGD feature extractor / pose autoencoder: the mode='pose' variant of model/embedding_net.py:262-314."""

    def __init__(self, args, pose_dim, n_frames, n_words=None, word_embed_size=None, word_embeddings=None, mode="pose"):
        super().__init__()
        if mode != "pose" or n_frames != 34:
            raise NotImplementedError("HIP path implements mode='pose' with 34-frame clips (config/gesture_autoencoder.yml)")
        self.context_encoder = None
        self.pose_dim = pose_dim
        self.pose_encoder = _PoseEncoderParams(pose_dim)
        self.decoder = _PoseDecoderParams(pose_dim)
        self.mode = mode
        self._engine = None

    @property
    def engine(self) -> AutoencoderEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", AutoencoderEngine(self))
        return self._engine

    def forward(self, in_text, in_audio, pre_poses, poses, input_mode=None, variational_encoding=False):
        if variational_encoding:
            raise NotImplementedError("variational_encoding=True is not on the reference's configured path")
        eng = self.engine
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

        def runner():
            res = eng.forward(poses.float(), training=self.training, save=need_grad)

            def back(d_feat, d_mu, d_lv, d_recon):
                assert d_feat is None and d_mu is None and d_lv is None, "only the reconstruction carries gradient"
                eng.backward(res["tape"], d_recon)
                return ()
            return (res["feat"], res["mu"], res["logvar"], res["recon"]), back

        if need_grad:
            anchor = next(p for p in self.parameters() if p.requires_grad)
            feat, mu, logvar, recon = _Bridge.apply(anchor, runner)
        else:
            (feat, mu, logvar, recon), _ = runner()
        return None, None, None, feat, mu, logvar, recon
