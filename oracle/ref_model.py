"""Functional CPU restatement of the reference's trimodal GAN hot path (oracle).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never by the product package.

Every function operates on a flat ``state`` dict that uses exactly the
reference's ``state_dict`` key names, so the same tensors can be loaded into
the real reference modules (tests/golden/make_golden.py does that to pin this
file) and into the HIP-backed modules of the product package.

All randomness (dropout masks, the reparameterisation noise, the speaker
shuffle) goes through :class:`Rand`, which either draws from a torch generator
and records what it drew, or replays injected tensors.  That is what lets the
GPU path be compared with this oracle draw for draw.

Reference line cites are relative to /root/reference/scripts/.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# FAST=True routes BatchNorm and the GRU through ATen's native CPU kernels (F.batch_norm, _VF.gru) --
# the same kernels the reference itself runs on CPU.  Used for the timed cpu_baseline; the default
# (explicit formulas, fp64 statistics) is the restatement the parity tests check against.
FAST = False


# --------------------------------------------------------------------------- randomness
class Rand:
    """Named random draws: replay ``inject[name]`` if present, else draw and record."""

    def __init__(self, seed: int | None = None, inject: dict | None = None):
        self.gen = None
        if seed is not None:
            self.gen = torch.Generator(device="cpu")
            self.gen.manual_seed(int(seed))
        self.inject = dict(inject or {})
        self.rec: dict[str, torch.Tensor] = OrderedDict()

    def keep_mask(self, name, shape, p, dtype=torch.float32):
        """Inverted-dropout scale mask: 0 with prob p, else 1/(1-p) (torch.nn.Dropout semantics)."""
        if name in self.inject:
            m = torch.as_tensor(self.inject[name]).to(dtype).reshape(shape)
        elif p <= 0.0:
            m = torch.ones(shape, dtype=dtype)
        else:
            keep = torch.bernoulli(torch.full(shape, 1.0 - p, dtype=torch.float32), generator=self.gen)
            m = (keep / (1.0 - p)).to(dtype)
        self.rec[name] = m
        return m

    def normal(self, name, shape, dtype=torch.float32):
        if name in self.inject:
            e = torch.as_tensor(self.inject[name]).to(dtype).reshape(shape)
        else:
            e = torch.randn(shape, generator=self.gen, dtype=torch.float32).to(dtype)
        self.rec[name] = e
        return e

    def perm(self, name, n):
        if name in self.inject:
            p = torch.as_tensor(self.inject[name]).long().reshape(n)
        else:
            p = torch.randperm(n, generator=self.gen)
        self.rec[name] = p
        return p


# --------------------------------------------------------------------------- primitives
def leaky(x, slope):
    # nn.LeakyReLU(True) in the reference == negative_slope 1.0 == identity (README.md:122)
    if slope == 1.0:
        return x
    return torch.where(x >= 0, x, x * slope)


def batch_norm(x, st, prefix, training, update_stats=True):
    """nn.BatchNorm1d on (B,C,L) or (B,C).  Train: biased batch variance normalises, the running
    estimate takes the unbiased one with momentum 0.1; num_batches_tracked += 1."""
    w, b = st[prefix + ".weight"], st[prefix + ".bias"]
    if FAST:
        if training and update_stats:
            st[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, st[prefix + ".running_mean"], st[prefix + ".running_var"], w, b,
                            training, BN_MOMENTUM if update_stats else 0.0, BN_EPS)
    dims = (0, 2) if x.dim() == 3 else (0,)
    shape = (1, -1, 1) if x.dim() == 3 else (1, -1)
    in_dtype = x.dtype
    if training:
        # statistics (and their autograd) in fp64: a naive fp32 mean/var over B*L ~ 1e6 elements is
        # only good to ~1e-3 in the gradients, ATen's own kernel is good to ~1e-6.
        x = x.double()
        w, b = w.double(), b.double()
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=dims)
        var = ((x - mean.view(shape)) ** 2).mean(dim=dims)
        if update_stats:
            with torch.no_grad():
                rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
                rm.mul_(1 - BN_MOMENTUM).add_(mean.detach().to(rm.dtype), alpha=BN_MOMENTUM)
                rv.mul_(1 - BN_MOMENTUM).add_((var.detach() * (n / max(n - 1, 1))).to(rv.dtype), alpha=BN_MOMENTUM)
                st[prefix + ".num_batches_tracked"] += 1
    else:
        mean, var = st[prefix + ".running_mean"].to(x.dtype), st[prefix + ".running_var"].to(x.dtype)
    xh = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS)
    return (xh * w.view(shape) + b.view(shape)).to(in_dtype)


def weight_norm_weight(g, v):
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||, norm over all dims but 0 (model/tcn.py:19,25)."""
    nrm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / nrm)


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.GRU layer, batch_first, h0 = 0.  Gate rows are [r; z; n]:
    r=s(Wir x+bir+Whr h+bhr) z=s(Wiz x+biz+Whz h+bhz) n=tanh(Win x+bin + r*(Whn h+bhn)) h'=(1-z)n+zh."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t() + b_ih
    h = x.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi[:, t].split(H, dim=1)
        h_r, h_z, h_n = gh.split(H, dim=1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, dim=1)


def gru_stack(x, st, prefix, n_layers, p_drop, training, rand: Rand, tag, fast=False):
    """Multi-layer bidirectional nn.GRU (multimodal_context_net.py:98-99,155): the concat of both
    directions feeds the next layer; inter-layer dropout on every layer's output but the last."""
    for l in range(n_layers):
        if fast or FAST:
            names = [f"{prefix}.{k}_l{l}{sfx}" for sfx in ("", "_reverse")
                     for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            y, _ = torch._VF.gru(x, x.new_zeros(2, x.shape[0], st[names[1]].shape[1]),
                                 [st[n] for n in names], True, 1, 0.0, False, True, True)
        else:
            f = gru_direction(x, st[f"{prefix}.weight_ih_l{l}"], st[f"{prefix}.weight_hh_l{l}"],
                              st[f"{prefix}.bias_ih_l{l}"], st[f"{prefix}.bias_hh_l{l}"], False)
            r = gru_direction(x, st[f"{prefix}.weight_ih_l{l}_reverse"], st[f"{prefix}.weight_hh_l{l}_reverse"],
                              st[f"{prefix}.bias_ih_l{l}_reverse"], st[f"{prefix}.bias_hh_l{l}_reverse"], True)
            y = torch.cat((f, r), dim=2)
        if training and l < n_layers - 1:
            y = y * rand.keep_mask(f"{tag}.gru.drop{l}", tuple(y.shape), p_drop, y.dtype)
        x = y
    return x


# --------------------------------------------------------------------------- generator
# Test aid (tests/test_trajectory_gpu.py): {layer 1 | 2 | 3: (flat indices into the (B, C, L) pre-activation, bool sides)} -- LeakyReLU gates
# of the audio encoder that take the GIVEN side instead of sign(pre-activation).  The tests pass the side the HIP path took for the handful
# of elements whose fp64 pre-activation is within rounding of zero (both sides are correct evaluations of the reference there); None = off.
wav_gate_override = None


def _leaky_gated(x, slope, layer):
    ov = None if wav_gate_override is None else wav_gate_override.get(layer)
    if ov is None or ov[0].numel() == 0:
        return leaky(x, slope)
    gate = (x >= 0).reshape(-1).clone()
    gate[ov[0]] = ov[1]
    return torch.where(gate.view(x.shape), x, x * slope)


def wav_encoder(st, audio, training, prefix="audio_encoder.feat_extractor"):
    """WavEncoder (multimodal_context_net.py:9-28): (B,A) -> (B,34,32)."""
    x = audio.unsqueeze(1)
    x = F.conv1d(x, st[f"{prefix}.0.weight"], st[f"{prefix}.0.bias"], stride=5, padding=1600)
    x = _leaky_gated(batch_norm(x, st, f"{prefix}.1", training), 0.3, 1)
    x = F.conv1d(x, st[f"{prefix}.3.weight"], st[f"{prefix}.3.bias"], stride=6)
    x = _leaky_gated(batch_norm(x, st, f"{prefix}.4", training), 0.3, 2)
    x = F.conv1d(x, st[f"{prefix}.6.weight"], st[f"{prefix}.6.bias"], stride=6)
    x = _leaky_gated(batch_norm(x, st, f"{prefix}.7", training), 0.3, 3)
    x = F.conv1d(x, st[f"{prefix}.9.weight"], st[f"{prefix}.9.bias"], stride=6)
    return x.transpose(1, 2)


def wav_preacts(st, audio, prefix="audio_encoder.feat_extractor"):
    """The three train-mode BatchNorm outputs of WavEncoder (multimodal_context_net.py:13-21), i.e. the LeakyReLU PRE-activations, channel-first
    (B, C, L), without touching the running statistics.  Test aid: where a pre-activation is within rounding of zero, two correct
    implementations may take different sides of the LeakyReLU (a 'gate flip'), which moves the gradients below it discontinuously."""
    outs = []
    x = audio.unsqueeze(1)
    for conv, bn, stride, pad in ((0, 1, 5, 1600), (3, 4, 6, 0), (6, 7, 6, 0)):
        x = F.conv1d(x, st[f"{prefix}.{conv}.weight"], st[f"{prefix}.{conv}.bias"], stride=stride, padding=pad)
        pre = batch_norm(x, st, f"{prefix}.{bn}", True, update_stats=False)
        outs.append(pre.detach())
        x = leaky(pre, 0.3)
    return outs


# Test aids (tests/test_trajectory_gpu.py), as wav_gate_override above but for the text encoder's ReLUs, site names "<tag>.relu1|2|3" with
# tag = e.g. "g2.tcn0": relu_gate_log (a dict) collects every site's pre-activation; relu_gate_override {site: (flat indices, bool sides)}
# puts the listed gates on the GIVEN side.  None = off.
relu_gate_log = None
relu_gate_override = None


def _relu_gated(x, site):
    if relu_gate_log is not None:
        relu_gate_log[site] = x.detach()
    ov = None if relu_gate_override is None else relu_gate_override.get(site)
    if ov is None or ov[0].numel() == 0:
        return torch.relu(x)
    gate = (x > 0).reshape(-1).clone()
    gate[ov[0]] = ov[1]
    return torch.where(gate.view(x.shape), x, torch.zeros_like(x))


def tcn_block(x, st, prefix, dilation, p_drop, training, rand, tag):
    """TemporalBlock (model/tcn.py:16-46) on (B,C,T): two weight-normed causal dilated k=2 convs,
    each ReLU + dropout, then relu(out + x).  Causal = pad d both sides, chomp the last d."""
    T = x.shape[2]
    out = x
    for ci, name in enumerate(("conv1", "conv2")):
        w = weight_norm_weight(st[f"{prefix}.{name}.weight_g"], st[f"{prefix}.{name}.weight_v"])
        out = F.conv1d(out, w, st[f"{prefix}.{name}.bias"], stride=1, padding=dilation, dilation=dilation)
        out = out[:, :, :T]
        out = _relu_gated(out, f"{tag}.relu{ci + 1}")
        if training:
            out = out * rand.keep_mask(f"{tag}.drop{ci + 1}", tuple(out.shape), p_drop, out.dtype)
    return _relu_gated(out + x, f"{tag}.relu3")


def text_encoder(st, in_text, n_layers, p_drop, training, rand, tag, prefix="text_encoder"):
    """TextEncoderTCN (multimodal_context_net.py:31-61): (B,T) int64 -> (B,T,32)."""
    emb = st[f"{prefix}.embedding.weight"][in_text]
    if training:
        emb = emb * rand.keep_mask(f"{tag}.emb_drop", tuple(emb.shape), 0.1, emb.dtype)
    y = emb.transpose(1, 2)
    for i in range(n_layers):
        y = tcn_block(y, st, f"{prefix}.tcn.network.{i}", 2 ** i, p_drop, training, rand, f"{tag}.tcn{i}")
    y = y.transpose(1, 2)
    return y @ st[f"{prefix}.decoder.weight"].t() + st[f"{prefix}.decoder.bias"]


def linear(st, prefix, x):
    return x @ st[prefix + ".weight"].t() + st[prefix + ".bias"]


def generator_forward(st, pre_seq, in_text, in_audio, vid, *, training, rand: Rand, tag="g",
                      n_layers=4, hidden=300, p_drop=0.3, fast_gru=False, return_parts=False,
                      input_context="both", z_mode="speaker"):
    """PoseGenerator.forward (multimodal_context_net.py:110-160).  input_context 'both' | 'audio' | 'text' | 'none'
    (:71-76, :139-148); z_mode 'speaker' (z_obj = speaker Vocab) | 'random' (plain noise, :132-134) | None.
    Returns (out, z, mu, logvar)."""
    audio_feat = text_feat = None
    if input_context != "none":                 # both encoders are evaluated for 'audio' and 'text' too (:117-123); the unused
        audio_feat = wav_encoder(st, in_audio, training)        # audio encoder still moves its BatchNorm running statistics
        if input_context == "text":
            audio_feat = None
    if input_context in ("both", "text"):       # (an unused text encoder has no state: only its dropout draws are skipped)
        text_feat = text_encoder(st, in_text, n_layers, p_drop, training, rand, tag)
    mu = logvar = z = None
    if z_mode == "speaker":
        zc = linear(st, "speaker_embedding.1", st["speaker_embedding.0.weight"][vid])
        mu = linear(st, "speaker_mu", zc)
        logvar = linear(st, "speaker_logvar", zc)
        std = torch.exp(0.5 * logvar)                       # embedding_net.py:10-13 (no train/eval switch)
        eps = rand.normal(f"{tag}.eps", tuple(std.shape), std.dtype)
        z = mu + eps * std
    elif z_mode == "random":
        z = rand.normal(f"{tag}.z", (in_text.shape[0], 16), pre_seq.dtype)
    in_data = torch.cat([t for t in (pre_seq, audio_feat, text_feat) if t is not None], dim=2)
    if z is not None:
        in_data = torch.cat((in_data, z.unsqueeze(1).repeat(1, in_data.shape[1], 1)), dim=2)
    g = gru_stack(in_data, st, "gru", n_layers, p_drop, training, rand, tag, fast=fast_gru)
    g = g[:, :, :hidden] + g[:, :, hidden:]
    o = linear(st, "out.0", g.reshape(-1, hidden))      # LeakyReLU(True) == identity between the two
    o = linear(st, "out.2", o)
    out = o.reshape(in_data.shape[0], in_data.shape[1], -1)
    if return_parts:
        return out, z, mu, logvar, dict(audio_feat=audio_feat, text_feat=text_feat, in_data=in_data, gru=g)
    return out, z, mu, logvar


# --------------------------------------------------------------------------- discriminator
def discriminator_forward(st, poses, *, training, rand: Rand, tag="d", fast_gru=False):
    """ConvDiscriminator.forward (multimodal_context_net.py:232-252): (B,34,27) -> (B,1)."""
    x = poses.transpose(1, 2)
    x = F.conv1d(x, st["pre_conv.0.weight"], st["pre_conv.0.bias"])
    x = batch_norm(x, st, "pre_conv.1", training)        # LeakyReLU(True) == identity
    x = F.conv1d(x, st["pre_conv.3.weight"], st["pre_conv.3.bias"])
    x = batch_norm(x, st, "pre_conv.4", training)
    x = F.conv1d(x, st["pre_conv.6.weight"], st["pre_conv.6.bias"])
    x = x.transpose(1, 2)
    g = gru_stack(x, st, "gru", 4, 0.3, training, rand, tag, fast=fast_gru)
    g = g[:, :, :64] + g[:, :, 64:]
    o = linear(st, "out", g.reshape(-1, 64)).view(poses.shape[0], -1)
    o = linear(st, "out2", o)
    return torch.sigmoid(o)


# --------------------------------------------------------------------------- GAN step
def make_pre_seq(target, n_pre):
    """train_eval/train_gan.py:20-22."""
    pre = target.new_zeros(target.shape[0], target.shape[1], target.shape[2] + 1)
    pre[:, :n_pre, :-1] = target[:, :n_pre]
    pre[:, :n_pre, -1] = 1
    return pre


def adam_step(params, grads, state, lr, betas=(0.5, 0.999), eps=1e-8):
    """torch.optim.Adam defaults used at train.py:104-109 (no weight decay, no amsgrad)."""
    b1, b2 = betas
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    for k, p in params.items():
        g = grads[k]
        if g is None:
            continue
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.data.addcdiv_(m, denom, value=-lr / bc1)


HP = dict(n_pre_poses=4, loss_warmup=10, loss_gan_weight=5.0, loss_regression_weight=500.0,
          loss_kld_weight=0.1, loss_reg_weight=0.05, learning_rate=5e-4, discriminator_lr_weight=0.2,
          n_layers=4, hidden_size=300, dropout_prob=0.3)   # config/multimodal_context.yml + parse_args.py:39,57


def unique_params(st):
    """Trainable leaves: float tensors minus BN running stats; TCN alias keys (net.0/net.4) dropped."""
    out = OrderedDict()
    for k, v in st.items():
        if not v.is_floating_point() or k.endswith("running_mean") or k.endswith("running_var"):
            continue
        if is_tcn_alias(k):
            continue
        out[k] = v
    return out


def is_tcn_alias(k):
    """TCN convs are registered twice (model/tcn.py:19-32): conv1 == net.0, conv2 == net.4."""
    return "tcn.network." in k and (".net.0." in k or ".net.4." in k)


def sync_aliases(st):
    for k in list(st.keys()):
        if is_tcn_alias(k):
            st[k] = st[k.replace(".net.0.", ".conv1.").replace(".net.4.", ".conv2.")]


def gan_losses_g(out, target, d_out, out_rand, z, z_rand, mu, logvar, epoch, hp=HP):
    """Generator-side losses, train_eval/train_gan.py:53-89.  out_rand None = no diversity term (z_type neither 'speaker'
    nor 'random', or loss_reg_weight == 0); mu None = no KLD (z_type != 'speaker').  Returns (loss, parts)."""
    beta = 0.1
    huber = F.smooth_l1_loss(out / beta, target / beta) * beta
    gen_error = -torch.mean(torch.log(d_out + 1e-8))
    loss = hp["loss_regression_weight"] * huber
    div_reg = kld = torch.zeros(())
    if out_rand is not None:
        beta = 0.05
        pose_l1 = F.smooth_l1_loss(out / beta, out_rand.detach() / beta, reduction="none") * beta
        pose_l1 = pose_l1.sum(dim=1).sum(dim=1)
        z_l1 = (z.detach() - z_rand.detach()).abs().mean(1)
        div_reg = -(pose_l1 / (z_l1 + 1.0e-5))
        div_reg = torch.clamp(div_reg, min=-1000).mean()
        loss = loss + hp["loss_reg_weight"] * div_reg
        if mu is not None:
            kld = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
            loss = loss + hp["loss_kld_weight"] * kld
    if epoch > hp["loss_warmup"]:
        loss = loss + hp["loss_gan_weight"] * gen_error
    return loss, dict(huber=huber, gen=gen_error, div_reg=div_reg, kld=kld)


def train_iter_gan(gst, dst, g_opt, d_opt, epoch, in_text, in_audio, target, vid, rand: Rand, hp=HP,
                   fast_gru=False, want_grads=False, input_context="both", z_type="speaker"):
    """One GAN iteration, same order of operations as train_eval/train_gan.py:13-103.
    z_type 'speaker' | 'random' | anything else ('none'): train.py:82-87 picks the generator's z_obj from it.
    ``gst``/``dst`` are state dicts (modified in place: params, BN buffers); ``g_opt``/``d_opt`` are
    Adam state dicts.  Returns the reference's loss dict (plus grads when asked)."""
    z_mode = z_type if z_type in ("speaker", "random") else None
    kw = dict(n_layers=hp["n_layers"], hidden=hp["hidden_size"], p_drop=hp["dropout_prob"], fast_gru=fast_gru,
              input_context=input_context, z_mode=z_mode)
    gp, dp = unique_params(gst), unique_params(dst)
    for p in list(gp.values()) + list(dp.values()):
        p.requires_grad_(True)
        p.grad = None
    pre_seq = make_pre_seq(target, hp["n_pre_poses"])
    ret, extra = {}, {}
    post_warmup = epoch > hp["loss_warmup"] and hp["loss_gan_weight"] > 0.0

    if post_warmup:                                                      # train_gan.py:27-43
        out1, *_ = generator_forward(gst, pre_seq, in_text, in_audio, vid, training=True, rand=rand, tag="g1", **kw)
        d_real = discriminator_forward(dst, target, training=True, rand=rand, tag="d_real", fast_gru=fast_gru)
        d_fake = discriminator_forward(dst, out1.detach(), training=True, rand=rand, tag="d_fake", fast_gru=fast_gru)
        dis_error = torch.sum(-torch.mean(torch.log(d_real + 1e-8) + torch.log(1 - d_fake + 1e-8)))
        dgr = torch.autograd.grad(dis_error, list(dp.values()), allow_unused=True)
        if want_grads:
            extra["d_grads"] = {k: g.detach().clone() for k, g in zip(dp.keys(), dgr)}
        with torch.no_grad():
            adam_step(dp, dict(zip(dp.keys(), dgr)), d_opt, hp["learning_rate"] * hp["discriminator_lr_weight"])
        ret["dis"] = float(dis_error)

    out, z, mu, logvar = generator_forward(gst, pre_seq, in_text, in_audio, vid, training=True, rand=rand, tag="g2", **kw)
    d_out = discriminator_forward(dst, out, training=True, rand=rand, tag="d_out", fast_gru=fast_gru)   # :55 always
    out_r = z_r = None
    if z_mode is not None and hp["loss_reg_weight"] > 0.0:               # :59-70
        rand_vids = vid[rand.perm("perm", vid.shape[0])] if z_mode == "speaker" else None
        out_r, z_r, _, _ = generator_forward(gst, pre_seq, in_text, in_audio, rand_vids, training=True, rand=rand,
                                             tag="g3", **kw)
    loss, parts = gan_losses_g(out, target, d_out, out_r, z, z_r, mu, logvar, epoch, hp)
    ggr = torch.autograd.grad(loss, list(gp.values()), allow_unused=True)
    if want_grads:
        extra["g_grads"] = {k: (None if g is None else g.detach().clone()) for k, g in zip(gp.keys(), ggr)}
        extra["out"] = out.detach().clone()
    with torch.no_grad():
        adam_step(gp, dict(zip(gp.keys(), ggr)), g_opt, hp["learning_rate"])
    for p in list(gp.values()) + list(dp.values()):
        p.requires_grad_(False)

    parts = {k: v.detach() for k, v in parts.items()}
    ret["loss"] = hp["loss_regression_weight"] * float(parts["huber"])
    if float(parts["kld"]) != 0.0:                                         # tensor truthiness, :95-98
        ret["KLD"] = hp["loss_kld_weight"] * float(parts["kld"])
    if float(parts["div_reg"]) != 0.0:
        ret["DIV_REG"] = hp["loss_reg_weight"] * float(parts["div_reg"])
    if post_warmup:
        ret["gen"] = hp["loss_gan_weight"] * float(parts["gen"])
    if want_grads:
        return ret, extra
    return ret


# --------------------------------------------------------------------------- FGD autoencoder
def _cnr(x, st, prefix, stride, training):
    """ConvNormRelu (embedding_net.py:16-39): conv + BN + LeakyReLU(0.2)."""
    x = F.conv1d(x, st[f"{prefix}.0.weight"], st[f"{prefix}.0.bias"], stride=stride)
    return leaky(batch_norm(x, st, f"{prefix}.1", training), 0.2)


def ae_encode(st, poses, training, prefix="pose_encoder"):
    """PoseEncoderConv (embedding_net.py:42-82), 34-frame branch, variational_encoding=False: z = mu."""
    x = poses.transpose(1, 2)
    x = _cnr(x, st, f"{prefix}.net.0", 1, training)
    x = _cnr(x, st, f"{prefix}.net.1", 1, training)
    x = _cnr(x, st, f"{prefix}.net.2", 2, training)
    x = F.conv1d(x, st[f"{prefix}.net.3.weight"], st[f"{prefix}.net.3.bias"])
    x = x.flatten(1)
    x = batch_norm(linear(st, f"{prefix}.out_net.0", x), st, f"{prefix}.out_net.1", training)
    x = batch_norm(linear(st, f"{prefix}.out_net.3", x), st, f"{prefix}.out_net.4", training)
    x = linear(st, f"{prefix}.out_net.6", x)
    mu = linear(st, f"{prefix}.fc_mu", x)
    logvar = linear(st, f"{prefix}.fc_logvar", x)
    return mu, mu, logvar


def ae_decode(st, feat, training, prefix="decoder"):
    """PoseDecoderConv (embedding_net.py:165-217), length 34."""
    x = batch_norm(linear(st, f"{prefix}.pre_net.0", feat), st, f"{prefix}.pre_net.1", training)
    x = linear(st, f"{prefix}.pre_net.3", x).view(feat.shape[0], 4, -1)
    x = F.conv_transpose1d(x, st[f"{prefix}.net.0.weight"], st[f"{prefix}.net.0.bias"])
    x = leaky(batch_norm(x, st, f"{prefix}.net.1", training), 0.2)
    x = F.conv_transpose1d(x, st[f"{prefix}.net.3.weight"], st[f"{prefix}.net.3.bias"])
    x = leaky(batch_norm(x, st, f"{prefix}.net.4", training), 0.2)
    x = F.conv1d(x, st[f"{prefix}.net.6.weight"], st[f"{prefix}.net.6.bias"])
    x = F.conv1d(x, st[f"{prefix}.net.7.weight"], st[f"{prefix}.net.7.bias"])
    return x.transpose(1, 2)


def ae_forward(st, poses, training):
    """EmbeddingNet(mode='pose').forward(None, None, None, poses) (embedding_net.py:283-308)."""
    z, mu, logvar = ae_encode(st, poses, training)
    return z, mu, logvar, ae_decode(st, z, training)


def ae_loss(recon, target):
    """train_feature_extractor.py:64-72: per-clip mean L1 + mean L1 of frame differences, summed over batch."""
    l = (recon - target).abs().mean(dim=(1, 2))
    l = l + ((recon[:, 1:] - recon[:, :-1]) - (target[:, 1:] - target[:, :-1])).abs().mean(dim=(1, 2))
    return l.sum()


def ae_train_iter(st, opt, target, lr=5e-4):
    """train_feature_extractor.py:54-97 with variational_encoding=False."""
    ps = unique_params(st)
    for p in ps.values():
        p.requires_grad_(True)
    _, _, _, recon = ae_forward(st, target, True)
    loss = ae_loss(recon, target)
    # fc_logvar gets no gradient (z = mu): grads None -> skipped by Adam, like torch.optim does
    gr = torch.autograd.grad(loss, list(ps.values()), allow_unused=True)
    with torch.no_grad():
        adam_step(ps, dict(zip(ps.keys(), gr)), opt, lr)
    for p in ps.values():
        p.requires_grad_(False)
    return {"loss": float(loss)}, dict(zip(ps.keys(), gr))


# --------------------------------------------------------------------------- FGD / metrics (fp64 host maths)
def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """embedding_space_evaluator.py:103-156: ||mu1-mu2||^2 + Tr(S1) + Tr(S2) - 2 Tr sqrtm(S1 S2)."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean = linalg.sqrtm(sigma1.dot(sigma2))
    if isinstance(covmean, tuple):
        covmean = covmean[0]
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def fgd_scores(generated_feats, real_feats):
    """EmbeddingSpaceEvaluator.get_scores (embedding_space_evaluator.py:74-101)."""
    g, r = np.asarray(generated_feats), np.asarray(real_feats)
    try:
        fd = frechet_distance(np.mean(g, axis=0), np.cov(g, rowvar=False), np.mean(r, axis=0), np.cov(r, rowvar=False))
    except ValueError:
        fd = 1e10
    feat_dist = float(np.mean(np.sum(np.abs(r - g), axis=1)))
    return float(fd), feat_dist


DIR_VEC_PAIRS = [(0, 1, 0.26), (1, 2, 0.18), (2, 3, 0.14), (1, 4, 0.22), (4, 5, 0.36),
                 (5, 6, 0.33), (1, 7, 0.22), (7, 8, 0.36), (8, 9, 0.33)]   # utils/data_utils.py:14-15


def dir_vec_to_pose(vec):
    """utils/data_utils.py:77-98 for (B,T,27) or (B,T,9,3) -> (B,T,10,3) float64."""
    vec = np.asarray(vec, dtype=np.float64)
    if vec.shape[-1] != 3:
        vec = vec.reshape(vec.shape[:-1] + (-1, 3))
    pos = np.zeros(vec.shape[:-2] + (10, 3))
    for j, (a, b, length) in enumerate(DIR_VEC_PAIRS):
        pos[..., b, :] = pos[..., a, :] + length * vec[..., j, :]
    return pos


def eval_metrics(out_dir_vec, target_dir_vec, mean_dir_vec, n_pre=4):
    """train.py:282-310, multimodal branch: (l1, joint_mae, accel) for one batch."""
    out = np.asarray(out_dir_vec, dtype=np.float32)
    tgt = np.asarray(target_dir_vec, dtype=np.float32)
    l1 = float(np.mean(np.abs(out - tgt)))
    mean = np.asarray(mean_dir_vec, dtype=np.float64).squeeze()
    oj = dir_vec_to_pose(out + mean)
    tj = dir_vec_to_pose(tgt + mean)
    mae = float(np.mean(np.abs(oj[:, n_pre:] - tj[:, n_pre:])))
    accel = float(np.mean(np.abs(np.diff(tj, n=2, axis=1) - np.diff(oj, n=2, axis=1))))
    return l1, mae, accel


def blend_windows(windows, n_pre=4):
    """synthesize.py:142-160: drop the last n_pre frames of the previous window and cross-fade them
    into the first n_pre frames of the next: prev*(n-j)/(n+1) + next*(j+1)/(n+1)."""
    out_list = []
    for w in windows:
        w = np.array(w, dtype=np.float32, copy=True)
        if out_list:
            last = out_list[-1][-n_pre:]
            out_list[-1] = out_list[-1][:-n_pre]
            n = len(last)
            for j in range(n):
                w[j] = last[j] * (n - j) / (n + 1) + w[j] * (j + 1) / (n + 1)
        out_list.append(w)
    return np.vstack(out_list)


def num_windows(clip_seconds, n_poses=34, n_pre=4, fps=15):
    """synthesize.py:57-63."""
    unit, stride = n_poses / fps, (n_poses - n_pre) / fps
    if clip_seconds < unit:
        return 1
    return math.ceil((clip_seconds - unit) / stride) + 1


# --------------------------------------------------------------------------- deterministic weights
def _u(gen, shape, bound):
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * bound


def _bn(st, prefix, c, gen, stats=True):
    st[prefix + ".weight"] = 1 + 0.1 * torch.randn(c, generator=gen)
    st[prefix + ".bias"] = 0.1 * torch.randn(c, generator=gen)
    st[prefix + ".running_mean"] = 0.05 * torch.randn(c, generator=gen) if stats else torch.zeros(c)
    st[prefix + ".running_var"] = 1 + 0.2 * torch.rand(c, generator=gen) if stats else torch.ones(c)
    st[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def _lin(st, prefix, n_out, n_in, gen, k=1):
    bound = 1.0 / math.sqrt(n_in * k)
    shape = (n_out, n_in, k) if k > 1 else (n_out, n_in)
    st[prefix + ".weight"] = _u(gen, shape, bound)
    st[prefix + ".bias"] = _u(gen, (n_out,), bound)


def _conv(st, prefix, n_out, n_in, k, gen):
    bound = 1.0 / math.sqrt(n_in * k)
    st[prefix + ".weight"] = _u(gen, (n_out, n_in, k), bound)
    st[prefix + ".bias"] = _u(gen, (n_out,), bound)


def _gru(st, prefix, n_in, hidden, layers, gen):
    bound = 1.0 / math.sqrt(hidden)
    for l in range(layers):
        for sfx in ("", "_reverse"):
            k_in = n_in if l == 0 else 2 * hidden
            st[f"{prefix}.weight_ih_l{l}{sfx}"] = _u(gen, (3 * hidden, k_in), bound)
            st[f"{prefix}.weight_hh_l{l}{sfx}"] = _u(gen, (3 * hidden, hidden), bound)
            st[f"{prefix}.bias_ih_l{l}{sfx}"] = _u(gen, (3 * hidden,), bound)
            st[f"{prefix}.bias_hh_l{l}{sfx}"] = _u(gen, (3 * hidden,), bound)


def make_generator_state(seed=0, n_words=512, n_speakers=17, hidden=300, layers=4, pose_dim=27, embed=300,
                         input_context="both", z_mode="speaker"):
    """Deterministic PoseGenerator state_dict with the reference's exact key set (117 keys
    at 4 layers).  Distributions follow torch defaults in scale; values are this build's own.
    Both encoders are always present (multimodal_context_net.py:78-80); the speaker layers only for z_mode 'speaker'."""
    g = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    fe = "audio_encoder.feat_extractor"
    for idx, (co, ci) in zip((0, 3, 6, 9), ((16, 1), (32, 16), (64, 32), (32, 64))):
        _conv(st, f"{fe}.{idx}", co, ci, 15, g)
        if idx != 9:
            _bn(st, f"{fe}.{idx + 1}", co, g)
    st["text_encoder.embedding.weight"] = torch.randn(n_words, embed, generator=g) / math.sqrt(embed)
    for i in range(layers):
        for name, alias in (("conv1", "net.0"), ("conv2", "net.4")):
            p = f"text_encoder.tcn.network.{i}.{name}"
            ci = embed if (i == 0 and name == "conv1") else hidden
            v = _u(g, (hidden, ci, 2), 1.0 / math.sqrt(ci * 2))
            st[p + ".bias"] = _u(g, (hidden,), 1.0 / math.sqrt(ci * 2))
            st[p + ".weight_g"] = v.flatten(1).norm(dim=1).view(-1, 1, 1) * (0.8 + 0.4 * torch.rand(hidden, 1, 1, generator=g))
            st[p + ".weight_v"] = v
        for name, alias in (("conv1", "net.0"), ("conv2", "net.4")):   # alias keys after both, like nn.Sequential order
            p = f"text_encoder.tcn.network.{i}.{name}"
            a = f"text_encoder.tcn.network.{i}.{alias}"
            for s in (".bias", ".weight_g", ".weight_v"):
                st[a + s] = st[p + s]
    st["text_encoder.decoder.weight"] = 0.01 * torch.randn(32, hidden, generator=g)
    st["text_encoder.decoder.bias"] = torch.zeros(32)
    if z_mode == "speaker":
        st["speaker_embedding.0.weight"] = torch.randn(n_speakers, 16, generator=g)
        _lin(st, "speaker_embedding.1", 16, 16, g)
        _lin(st, "speaker_mu", 16, 16, g)
        _lin(st, "speaker_logvar", 16, 16, g)
    n_ctx = {"both": 64, "audio": 32, "text": 32, "none": 0}[input_context]
    _gru(st, "gru", n_ctx + pose_dim + 1 + (16 if z_mode else 0), hidden, layers, g)
    _lin(st, "out.0", hidden // 2, hidden, g)
    _lin(st, "out.2", pose_dim, hidden // 2, g)
    return st


def make_discriminator_state(seed=1, pose_dim=27):
    g = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    _conv(st, "pre_conv.0", 16, pose_dim, 3, g)
    _bn(st, "pre_conv.1", 16, g)
    _conv(st, "pre_conv.3", 8, 16, 3, g)
    _bn(st, "pre_conv.4", 8, g)
    _conv(st, "pre_conv.6", 8, 8, 3, g)
    _gru(st, "gru", 8, 64, 4, g)
    _lin(st, "out", 1, 64, g)
    _lin(st, "out2", 1, 28, g)
    return st


def make_autoencoder_state(seed=2, pose_dim=27):
    """EmbeddingNet(mode='pose') state_dict, 34-frame branch (190 691 parameters)."""
    g = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    e = "pose_encoder"
    for i, (co, ci, k) in enumerate(((32, pose_dim, 3), (64, 32, 3), (64, 64, 4))):
        _conv(st, f"{e}.net.{i}.0", co, ci, k, g)
        _bn(st, f"{e}.net.{i}.1", co, g)
    _conv(st, f"{e}.net.3", 32, 64, 3, g)
    _lin(st, f"{e}.out_net.0", 256, 384, g)
    _bn(st, f"{e}.out_net.1", 256, g)
    _lin(st, f"{e}.out_net.3", 128, 256, g)
    _bn(st, f"{e}.out_net.4", 128, g)
    _lin(st, f"{e}.out_net.6", 32, 128, g)
    _lin(st, f"{e}.fc_mu", 32, 32, g)
    _lin(st, f"{e}.fc_logvar", 32, 32, g)
    d = "decoder"
    _lin(st, f"{d}.pre_net.0", 64, 32, g)
    _bn(st, f"{d}.pre_net.1", 64, g)
    _lin(st, f"{d}.pre_net.3", 136, 64, g)
    for idx, (ci, co) in zip((0, 3), ((4, 32), (32, 32))):            # ConvTranspose1d weight is (Cin, Cout, k)
        bound = 1.0 / math.sqrt(co * 3)
        st[f"{d}.net.{idx}.weight"] = _u(g, (ci, co, 3), bound)
        st[f"{d}.net.{idx}.bias"] = _u(g, (co,), bound)
        _bn(st, f"{d}.net.{idx + 1}", co, g)
    _conv(st, f"{d}.net.6", 32, 32, 3, g)
    _conv(st, f"{d}.net.7", pose_dim, 32, 3, g)
    return st


def clone_state(st, dtype=None):
    out = OrderedDict()
    for k, v in st.items():
        c = v.detach().clone()
        if dtype is not None and c.is_floating_point():
            c = c.to(dtype)
        out[k] = c
    sync_aliases(out)
    return out


def make_batch(seed, batch, n_words=512, n_speakers=17, n_frames=34, pose_dim=27, audio_len=36267):
    """Synthetic batch per SURVEY 8(d): sparse-onset text ids, N(0,0.1^2) audio/poses, vids in [1,S)."""
    g = torch.Generator().manual_seed(seed)
    text = torch.zeros(batch, n_frames, dtype=torch.int64)
    for b in range(batch):
        k = int(torch.randint(4, 13, (1,), generator=g))
        frames = torch.randperm(n_frames, generator=g)[:k]
        text[b, frames] = torch.randint(4, n_words, (k,), generator=g)
    audio = (0.1 * torch.randn(batch, audio_len, generator=g)).clamp_(-1, 1)
    vid = torch.randint(1, n_speakers, (batch,), generator=g)
    poses = 0.1 * torch.randn(batch, n_frames, pose_dim, generator=g)
    return text, audio, vid, poses


# --------------------------------------------------------------------------- input pipeline (after the LMDB read)
def data_make_audio_fixed_length(audio, expected_audio_length):
    """utils/data_utils.py:68-74."""
    n_padding = expected_audio_length - len(audio)
    if n_padding > 0:
        audio = np.pad(audio, (0, n_padding), mode="symmetric")
    else:
        audio = audio[0:expected_audio_length]
    return audio


def data_words_to_tensor(word_seq, word_index, end_time=None, sos=1, eos=2):
    """lmdb_data_loader.py:142-149: [SOS, ids of the words that start no later than end_time, EOS]."""
    idx = [sos]
    for w in word_seq:
        if end_time is not None and w[1] > end_time:
            break
        idx.append(word_index(w[0]))
    idx.append(eos)
    return np.asarray(idx, dtype=np.int64)


def data_getitem(sample, word_index, n_poses=34, fps=15, remove_word_timing=False, full=False):
    """SpeechMotionDataset.__getitem__ (data_loader/lmdb_data_loader.py:107-171) after `pyarrow.deserialize`; `word_index`
    maps a word to its vocabulary id (lang_model.get_word_index).  Returns (extended_word_seq, vec_seq, audio) as numpy, or with
    full=True the reference's whole tuple (word_seq_tensor, extended_word_seq, pose_seq, vec_seq, audio, spectrogram, aux_info).
    Pinned by tests/golden/g10_dataset.npz (SpeechMotionDataset.__getitem__ of the imported reference, make_golden_eval.py)."""
    word_seq, pose_seq, vec_seq, audio, spectrogram, aux_info = sample
    duration = aux_info["end_time"] - aux_info["start_time"]
    sample_end_time = aux_info["start_time"] + duration * n_poses / vec_seq.shape[0]          # :153
    audio = data_make_audio_fixed_length(audio, int(round(n_poses / fps * 16000)))            # :154, :62
    spec_len = int(round((n_poses / fps * 16000 - 1024) / 512 + 1))                            # utils/data_utils.py:44-46
    spectrogram = np.asarray(spectrogram)[:, 0:spec_len]                                       # :155
    vec_seq = vec_seq[0:n_poses]
    pose_seq = pose_seq[0:n_poses]
    frame_duration = (sample_end_time - aux_info["start_time"]) / n_poses                     # :119
    ext = np.zeros(n_poses)
    onset_frames = [max(0, int(np.floor((w[1] - aux_info["start_time"]) / frame_duration))) for w in word_seq]
    if remove_word_timing:                                                                     # :122-131
        n_words = sum(1 for f in onset_frames if f < n_poses)
        space = int(n_poses / (n_words + 1))
        for i in range(n_words):
            ext[(i + 1) * space] = word_index(word_seq[i][0])
    else:                                                                                      # :132-139
        for w, f in zip(word_seq, onset_frames):
            if f < n_poses:
                ext[f] = word_index(w[0])
    ext = ext.astype(np.int64)
    vec = vec_seq.reshape(vec_seq.shape[0], -1).astype(np.float32)
    audio = np.asarray(audio, dtype=np.float32)
    if not full:
        return ext, vec, audio
    return (data_words_to_tensor(word_seq, word_index, sample_end_time), ext, pose_seq.reshape(pose_seq.shape[0], -1).astype(np.float32),
            vec, audio, spectrogram, aux_info)


def pose_seq_to_dir_vec(pose):
    """utils/data_utils.py:101-121: joint positions (T,10,3) | (B,T,10,3) -> unit bone direction vectors (..., 9, 3).
    sklearn's normalize leaves an all-zero row at zero."""
    pose = np.asarray(pose)
    if pose.shape[-1] != 3:
        pose = pose.reshape(pose.shape[:-1] + (-1, 3))
    out = np.zeros(pose.shape[:-2] + (len(DIR_VEC_PAIRS), 3))
    for i, (a, b, _) in enumerate(DIR_VEC_PAIRS):
        d = pose[..., b, :] - pose[..., a, :]
        n = np.sqrt((d * d).sum(-1, keepdims=True))
        out[..., i, :] = d / np.where(n == 0, 1.0, n)
    return out


# --------------------------------------------------------------------------- evaluation loops
def evaluate_testset(gst, batches, mean_dir_vec, rand: Rand, *, n_pre=4, ast=None, vids=None, input_context="both",
                     z_mode="speaker", dtype=torch.float32):
    """scripts/train.py:234-329, multimodal_context branch.  `batches`: list of (in_text_padded, target_vec, in_audio) torch
    tensors; `vids`: per-batch speaker ids (the reference draws them with random.choice :257-260; None when the generator has
    no speaker Vocab, utils/train_utils.py:152-164).  Draw names: 'e{i}.eps' / 'e{i}.z'.  Returns the reference's dict plus
    'accel' (computed at :308-310 but only logged) and the per-batch outputs."""
    tot = {"loss": 0.0, "joint_mae": 0.0, "accel": 0.0}
    count = 0
    outs, real_feats, gen_feats = [], [], []
    mean = np.asarray(mean_dir_vec).squeeze()
    for i, (text, target, audio) in enumerate(batches):
        B = target.shape[0]
        target_t = target.to(dtype)
        pre_seq = make_pre_seq(target_t, n_pre)                                        # :262-265
        vid = None if (vids is None or z_mode != "speaker") else vids[i]
        out, *_ = generator_forward(gst, pre_seq, text, audio.to(dtype), vid, training=False, rand=rand, tag=f"e{i}",
                                    input_context=input_context, z_mode=z_mode)
        loss = float((out - target_t).abs().mean())                                   # F.l1_loss :282
        if ast is not None:                                                            # push_samples, evaluator :46-64
            real_feats.append(ae_forward(ast, target_t.to(torch.float32), False)[0].numpy())
            gen_feats.append(ae_forward(ast, out.to(torch.float32), False)[0].numpy())
        o = out.detach().to(torch.float32).numpy() + mean                              # :293-298 (float32 + float64 -> float64)
        t = target.to(torch.float32).numpy() + mean
        oj, tj = dir_vec_to_pose(o), dir_vec_to_pose(t)
        mae = float(np.mean(np.absolute(oj[:, n_pre:] - tj[:, n_pre:])))               # :300-305
        acc = float(np.mean(np.abs(np.diff(tj, n=2, axis=1) - np.diff(oj, n=2, axis=1))))   # :308-310
        tot["loss"] += loss * B; tot["joint_mae"] += mae * B; tot["accel"] += acc * B
        count += B
        outs.append(out.detach())
    ret = {k: v / count for k, v in tot.items()}
    if ast is not None and real_feats:
        ret["frechet"], ret["feat_dist"] = fgd_scores(np.vstack(gen_feats), np.vstack(real_feats))
    return ret, outs


def eval_embed(ast, target_poses):
    """train_eval/train_joint_embed.py:54-62 for the pose-mode network: (mean over clips of the per-clip mean L1, recon)."""
    _, _, _, recon = ae_forward(ast, target_poses, False)
    return float((recon - target_poses).abs().mean(dim=(1, 2)).mean()), recon


def ae_evaluate_testset(ast, batches):
    """scripts/train_feature_extractor.py:26-51: batch-size weighted average of eval_embed's loss over (B,34,27) batches."""
    s, n = 0.0, 0
    for target in batches:
        loss, _ = eval_embed(ast, target)
        s += loss * target.shape[0]
        n += target.shape[0]
    return {"loss": s / n}


# --------------------------------------------------------------------------- long-utterance synthesis
def words_in_time_range(word_list, start_time, end_time):
    """data_loader/data_preprocessor.py:174-188."""
    out = []
    for w in word_list:
        if w[1] >= end_time:
            break
        if w[2] <= start_time:
            continue
        out.append(w)
    return out


def window_inputs(audio, words, i, word_index, *, n_poses=34, n_pre=4, fps=15, audio_sr=16000):
    """synthesize.py:82-119 for window i: (in_audio (L,) float32 zero-padded, in_text_padded (n_poses,) int64, padding samples)."""
    unit_time, stride_time = n_poses / fps, (n_poses - n_pre) / fps
    clip_length = len(audio) / audio_sr
    audio_sample_length = int(unit_time * audio_sr)
    start_time = i * stride_time
    end_time = start_time + unit_time
    a0 = math.floor(start_time / clip_length * len(audio))
    piece = np.asarray(audio[a0:a0 + audio_sample_length])
    pad = 0
    if len(piece) < audio_sample_length:
        pad = audio_sample_length - len(piece)
        piece = np.pad(piece, (0, pad), "constant")
    ids = np.zeros(n_poses)
    frame_duration = (end_time - start_time) / n_poses
    for w in words_in_time_range(words, start_time, end_time):
        ids[max(0, int(np.floor((w[1] - start_time) / frame_duration)))] = word_index(w[0])
    return piece.astype(np.float32), ids.astype(np.int64), pad


def fade_out_tail(out_dir_vec, end_padding_samples, *, n_pre=4, fps=15, audio_sr=16000):
    """synthesize.py:188-207: pad if needed, zero the frames after the fade, weighted quadratic fit over 2*n_pre frames."""
    out = np.array(out_dir_vec, copy=True)
    n_smooth = n_pre
    start_frame = len(out) - int(end_padding_samples / audio_sr * fps)
    end_frame = start_frame + n_smooth * 2
    if len(out) < end_frame:
        out = np.pad(out, [(0, end_frame - len(out)), (0, 0)], mode="constant")
    out[end_frame - n_smooth:] = 0
    y = out[start_frame:end_frame]
    x = np.arange(y.shape[0])
    w = np.ones(len(y)); w[0] = 5; w[-1] = 5
    coeffs = np.polyfit(x, y, 2, w=w)
    out[start_frame:end_frame] = np.stack([np.poly1d(coeffs[:, k])(x) for k in range(y.shape[1])], axis=1)
    return out


def generate_gestures(gst, audio, words, word_index, rand: Rand, *, vid=None, seed_seq=None, fade_out=False, n_poses=34,
                      n_pre=4, fps=15, audio_sr=16000, pose_dim=27, input_context="both", z_mode="speaker", windows=None):
    """scripts/synthesize.py:36-209, multimodal_context model.  vid: speaker id (already drawn: the reference's
    random.randrange at :69-71 is the caller's business) or None for z_mode != 'speaker'.  Draw names 'w{i}.eps' / 'w{i}.z'.
    `windows` (list) receives each window's (pre_seq, in_text_padded, in_audio) when given."""
    clip_length = len(audio) / audio_sr
    n_sub = num_windows(clip_length, n_poses, n_pre, fps)
    pre_seq = torch.zeros(1, n_poses, pose_dim + 1)
    if seed_seq is not None:
        pre_seq[0, :n_pre, :-1] = torch.as_tensor(np.asarray(seed_seq)[:n_pre], dtype=torch.float32)
        pre_seq[0, :n_pre, -1] = 1
    vid_t = torch.tensor([int(vid)], dtype=torch.int64) if (z_mode == "speaker") else None
    out_list, end_pad, out = [], 0, None
    for i in range(n_sub):
        a, ids, pad = window_inputs(audio, words, i, word_index, n_poses=n_poses, n_pre=n_pre, fps=fps, audio_sr=audio_sr)
        if i == n_sub - 1:
            end_pad = pad
        if i > 0:
            pre_seq = pre_seq.clone()
            pre_seq[0, :n_pre, :-1] = out[0, -n_pre:]
            pre_seq[0, :n_pre, -1] = 1
        in_audio, in_text = torch.from_numpy(a).unsqueeze(0), torch.from_numpy(ids).unsqueeze(0)
        if windows is not None:
            windows.append((pre_seq.clone(), in_text.clone(), in_audio.clone()))
        out, *_ = generator_forward(gst, pre_seq, in_text, in_audio, vid_t, training=False, rand=rand, tag=f"w{i}",
                                    input_context=input_context, z_mode=z_mode)
        out_list.append(out[0].detach().numpy().copy())
    res = blend_windows(out_list, n_pre)
    if fade_out:
        res = fade_out_tail(res, end_pad, n_pre=n_pre, fps=fps, audio_sr=audio_sr)
    return res
