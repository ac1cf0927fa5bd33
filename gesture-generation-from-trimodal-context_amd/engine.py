"""Network engines: hand-sequenced forward and backward passes over the HIP ops.

Each engine owns the flat parameter slab of one drop-in nn.Module (modules.py) and runs the whole network as a fixed
sequence of kernel launches on the current stream -- no autograd graph, no host synchronisation, so a complete
training iteration can be captured into one hipGraph (train_gan.GraphedGanStep).

Reference maths: model/multimodal_context_net.py (WavEncoder :9-28, TextEncoderTCN :31-61, PoseGenerator :64-160,
ConvDiscriminator :207-252), model/tcn.py, model/embedding_net.py (pose-mode EmbeddingNet).
"""
import os

import torch

from . import layers as L
from . import ops, _lib
from .ops import Win
from .params import ParamSlab

WAV_CONVS = ((0, 16, 1, 5, 1600), (3, 32, 16, 6, 0), (6, 64, 32, 6, 0), (9, 32, 64, 6, 0))   # idx, Co, Ci, stride, pad
WAV_KW = 15


class DeviceRNG:
    """Philox counter RNG state on the device: {seed, step}.  Every draw site gets a stable id."""

    def __init__(self, seed, device):
        self.state = ops.new_rng_state(seed, device)
        self._sites = {}

    def site(self, name):
        return self._sites.setdefault(name, len(self._sites) + 1)

    def advance(self):
        ops.rng_advance(self.state)


class _Engine:
    # overlap independent work on a second HIP stream (layers.Fork); off by default: measured 3 % slower under hipGraph replay on MI355X (12.39 vs 12.00 ms); TG_SIDE_STREAM=1 enables
    use_side_stream = os.environ.get("TG_SIDE_STREAM", "0") != "0"

    def __init__(self, module, seed=0):
        self.mod = module
        self.slab = ParamSlab(module)
        self._seed = seed
        self._rng = None

    @property
    def rng(self):
        dev = self.slab.flat.device
        if self._rng is None or self._rng.state.device != dev:
            self._rng = DeviceRNG(self._seed, dev)
        return self._rng

    def views(self):
        self.slab.ensure()
        P, G = self.slab.views()
        Bf = dict(self.mod.named_buffers())
        return P, G, Bf

    def _drop(self, name, x, p, inject):
        """F.dropout(x, p) in train mode -> (y, scale mask or None): injected mask (tests) or one fused draw-and-apply pass."""
        if inject is not None and name in inject:
            m = inject[name]
            assert m.shape == x.shape, (name, m.shape, x.shape)
            m = m.contiguous()
            return ops.mul(x, m, torch.empty_like(x)), m
        if p <= 0.0:
            return x, None
        return ops.dropout_apply(x.contiguous(), p, self.rng.state, self.rng.site(name))

    def _mask(self, name, like, p, inject):
        """Inverted-dropout scale mask for site `name`: injected (tests) or drawn on device."""
        if inject is not None and name in inject:
            m = inject[name]
            assert m.shape == like.shape, (name, m.shape, like.shape)
            return m.contiguous()
        if p <= 0.0:
            return None
        return ops.dropout_mask(torch.empty_like(like), p, self.rng.state, self.rng.site(name))


# ======================================================================================================= generator
class GeneratorEngine(_Engine):
    def __init__(self, module, seed=0):
        super().__init__(module, seed)
        self.H = module.hidden_size
        self.n_layers = module.n_layers
        self.p_drop = module.dropout_prob
        self.pose_dim = module.pose_dim
        self.T = module.pre_length + module.gen_length
        # input_context / z variants (multimodal_context_net.py:71-97): columns of the GRU input are
        # [pre_seq (D+1) | audio 32 | text 32 | z 16] with the absent blocks removed
        ctx = module.input_context
        assert ctx in ("both", "audio", "text", "none"), ctx
        self.use_audio, self.use_text = ctx in ("both", "audio"), ctx in ("both", "text")
        self.z_mode = None if not module.z_obj else ("speaker" if module.speaker_embedding is not None else "random")
        D = self.pose_dim
        self.c_audio = D + 1
        self.c_text = self.c_audio + 32 * self.use_audio
        self.c_z = self.c_text + 32 * self.use_text
        self.in_size = self.c_z + (16 if self.z_mode else 0)

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, pre_seq, in_text, in_audio, vid, *, training, groups=1, save=False, inject=None, tag="g", save_rows=None):
        """Stacked forward: the batch may hold `groups` reference forward calls back to back (BatchNorm statistics are
        per group).  Returns a dict with out/z/mu/logvar (+ the tape when save=True)."""
        P, G, Bf = self.views()
        Bs, T = in_text.shape
        H, D = self.H, self.pose_dim
        assert T == self.T and pre_seq.shape == (Bs, T, D + 1) and Bs % groups == 0
        tp = {"Bs": Bs, "groups": groups, "training": training}
        in_size = self.in_size
        in_data = L.empty(Bs, T, in_size, like=pre_seq)
        # TG_SIDE_STREAM_FWD=1: only the audio encoder beside the text encoder (no cluster kernel runs until the join)
        fork = L.Fork(pre_seq.device, enabled=(self.use_side_stream or os.environ.get("TG_SIDE_STREAM_FWD", "0") != "0") and self.use_audio)
        # The reference evaluates BOTH encoders for 'audio' and 'text' (:117-123) and discards one.  The discarded text encoder
        # has no state, so it is skipped; the discarded audio encoder moves its BatchNorm running statistics in train mode, so
        # for 'text' it still runs (forward only, into a scratch buffer) to keep the state_dict identical after training.
        if self.use_audio:
            self._audio_fwd(P, Bf, in_audio, in_data, self.c_audio, tp, fork, training, groups)
        elif training and self.use_text:
            self._audio_fwd(P, Bf, in_audio, L.empty(Bs, T, 32, like=pre_seq), 0, {}, fork, training, groups)
        if self.use_text:
            self._text_fwd(P, in_text, in_data, tp, training, inject, tag)
        mu = logvar = z = None
        z_in_place = False
        if self.z_mode == "speaker":
            # ---- speaker embedding -> mu/logvar -> reparameterised z (:125-131; embedding_net.py:10-13)
            assert vid is not None
            if inject is not None and f"{tag}.eps" in inject:
                eps = inject[f"{tag}.eps"].contiguous()
            else:
                eps = ops.normal(L.empty(Bs, 16, like=pre_seq), self.rng.state, self.rng.site(f"{tag}.eps"))
            if ops.SPEAKER_FUSED:
                # gather, three 16 x 16 linears, reparameterisation and the per-frame copy into the GRU input: one launch
                se, zc, mu, logvar, z = ops.speaker_fwd(P["speaker_embedding.0.weight"], vid.contiguous(), P["speaker_embedding.1.weight"],
                                                        P["speaker_embedding.1.bias"], P["speaker_mu.weight"], P["speaker_mu.bias"],
                                                        P["speaker_logvar.weight"], P["speaker_logvar.bias"], eps,
                                                        rep=in_data.view(Bs * T, in_size)[:, self.c_z:], T=T)
                z_in_place = True
            else:
                se = ops.embed_gather(P["speaker_embedding.0.weight"], vid.contiguous(), L.empty(Bs, 16, like=pre_seq))
                zc = L.linear_fwd(se, P["speaker_embedding.1.weight"], P["speaker_embedding.1.bias"])
                mu = L.linear_fwd(zc, P["speaker_mu.weight"], P["speaker_mu.bias"])
                logvar = L.linear_fwd(zc, P["speaker_logvar.weight"], P["speaker_logvar.bias"])
                z = ops.reparam_fwd(mu, logvar, eps, torch.empty_like(mu))
            tp.update(se=se, zc=zc, mu=mu, logvar=logvar, eps=eps, vid=vid)
        elif self.z_mode == "random":
            # ---- plain noise vector (:132-134)
            if inject is not None and f"{tag}.z" in inject:
                z = inject[f"{tag}.z"].contiguous()
            else:
                z = ops.normal(L.empty(Bs, 16, like=pre_seq), self.rng.state, self.rng.site(f"{tag}.z"))

        # ---- concat [pre_seq | audio | text | z repeated over time] (:139-153)
        fork.join()
        flat_in = in_data.view(Bs * T, in_size)
        ops.copy2d(pre_seq.contiguous().view(Bs * T, D + 1), flat_in[:, :D + 1])
        if z is not None and not z_in_place:
            ops.repeat_rows(z, flat_in[:, self.c_z:], Bs, T)

        # ---- 4-layer bidirectional GRU, sum of directions, output MLP (:155-158)
        y, gtape = L.gru_stack_fwd(in_data, P, "gru", self.n_layers, H, p_drop=self.p_drop, training=training, rng=self.rng,
                                   save=save, inject=inject, tag=tag, save_rows=save_rows)      # save_rows: the batch rows backward() will be asked for
        if ops.OUT_MLP_COMPOSED:
            # LeakyReLU(True) == identity (README.md:122): Linear(300, 150) -> Linear(150, 27) is one linear map on the composed weight; written
            # twice side by side it acts on [fwd | rev] directly, so the direction sum (:155-156) is never formed
            w21, w21t, b21 = ops.out_mlp_compose(P["out.0.weight"], P["out.0.bias"], P["out.2.weight"], P["out.2.bias"], dup=2)
            out = L.linear_fwd(y.view(Bs * T, 2 * H), w21, b21).view(Bs, T, D)
            tp.update(in_text=in_text, gru=gtape, o=None, y_last=y, h1=None, w21t=w21t, in_size=in_size)
        else:
            o = ops.add_halves(y, L.empty(Bs * T, H, like=y))
            h1 = L.linear_fwd(o, P["out.0.weight"], P["out.0.bias"])
            out = L.linear_fwd(h1, P["out.2.weight"], P["out.2.bias"]).view(Bs, T, D)
            tp.update(in_text=in_text, gru=gtape, o=o, h1=h1, in_size=in_size)
        return {"out": out, "z": z, "mu": mu, "logvar": logvar, "in_data": in_data, "tape": tp if save else None}

    def _audio_fwd(self, P, Bf, in_audio, in_data, ca, tp, fork, training, groups):
        """WavEncoder: 4 strided convs as window GEMMs, BN + LeakyReLU(0.3) between (:9-28) -> in_data[:, :, audio columns].
        The stacked forward calls of one GAN iteration see the SAME audio and the encoder has no dropout, so when
        in_audio holds one group's rows the encoder runs once: identical batch statistics, running stats updated
        `groups` times (tg_bn_train_stats repeats), output replicated per group."""
        Bs, T, in_size = in_data.shape
        fe = "audio_encoder.feat_extractor"
        Ba = in_audio.shape[0]
        shared = Ba != Bs
        assert Ba == Bs or Ba * groups == Bs, (Ba, Bs, groups)
        x = in_audio.contiguous().view(Ba, -1, 1)
        wav = []
        fork.keep(x, in_data)
        fork.__enter__()                 # audio encoder on the side stream, text encoder + speaker path on the main one
        for idx, Co, Ci, stride, pad in WAV_CONVS:
            if idx == 0 and L.wav_front_supported(P[f"{fe}.0.weight"]):
                # conv1 + BatchNorm + LeakyReLU straight from the raw audio: the 16-channel pre-BatchNorm tensor (65 MB at B = 128) never exists
                y, st = L.wav_front_fwd(x.view(Ba, -1), P[f"{fe}.0.weight"], P[f"{fe}.0.bias"], P[f"{fe}.1.weight"], P[f"{fe}.1.bias"],
                                        Bf[f"{fe}.1.running_mean"], Bf[f"{fe}.1.running_var"], Bf[f"{fe}.1.num_batches_tracked"],
                                        stride=stride, pad=pad, training=training, groups=1 if shared else groups, act_slope=0.3,
                                        repeats=groups if shared else 1)
                wav.append((x, st, y.shape[1]))
                x = y
                continue
            wp = L.pack_conv_weight(P[f"{fe}.{idx}.weight"])
            last = idx == 9
            if last and shared:
                # the encoder ran once for all stacked calls: its last conv writes every group's slice of the GRU input itself -- one grouped
                # launch of `groups` identical (tiny, latency-bound) products instead of one product + a copy launch per group
                Lo = L.conv_out_len(x.shape[1], WAV_KW, stride, pad)
                assert Lo == T, f"audio length gives {Lo} frames, expected {T}"
                a_win = Win.conv(x, WAV_KW, stride=stride, pad=pad, rows_out=Lo)
                probs = []
                for g in range(groups):
                    o = in_data[g * Ba:(g + 1) * Ba, :, ca:ca + 32]
                    probs.append(dict(A=a_win, W=wp, bias=P[f"{fe}.{idx}.bias"], out=o, c_batch_stride=o.stride(0), c_row_stride=o.stride(1), c_rows_out=Lo))
                ops.gemm_nt_group(probs)
                wav.append((x, None, None))
                break
            out = in_data[:, :, ca:ca + 32] if last else None
            c = L.conv_fwd(x, wp, P[f"{fe}.{idx}.bias"], WAV_KW, stride=stride, pad=pad, out=out)
            if last:
                assert c.shape[1] == T, f"audio length gives {c.shape[1]} frames, expected {T}"
                wav.append((x, None, None))
                break
            y, st = L.bn_fwd(c, P[f"{fe}.{idx + 1}.weight"], P[f"{fe}.{idx + 1}.bias"], Bf[f"{fe}.{idx + 1}.running_mean"],
                             Bf[f"{fe}.{idx + 1}.running_var"], Bf[f"{fe}.{idx + 1}.num_batches_tracked"],
                             training=training, groups=1 if shared else groups, act_slope=0.3,
                             repeats=groups if shared else 1)
            wav.append((x, st, c.shape[1]))
            x = y
        fork.__exit__(None, None, None)
        tp["wav"], tp["wav_shared"] = wav, shared

    def _text_fwd(self, P, in_text, in_data, tp, training, inject, tag):
        """TextEncoderTCN (:31-61, model/tcn.py) -> in_data[:, :, text columns]."""
        Bs, T, in_size = in_data.shape
        te = "text_encoder"
        E = P[f"{te}.embedding.weight"].shape[1]
        emb = ops.embed_gather(P[f"{te}.embedding.weight"], in_text.contiguous().view(-1), L.empty(Bs, T, E, like=in_data))
        cur, emb_mask = self._drop(f"{tag}.emb_drop", emb, 0.1, inject) if training else (emb, None)
        # every weight-normed conv of the TCN in one launch: w = g v / ||v|| as the forward GEMM operand [Co][2 Ci] and, transposed,
        # as the input-gradient operand [Ci][2 Co] (model/tcn.py:19,25)
        names = [f"{te}.tcn.network.{i}.{c}" for i in range(self.n_layers) for c in ("conv1", "conv2")]
        wps, wts = ops.weight_norm_fwd_batch([P[n + ".weight_v"] for n in names], [P[n + ".weight_g"] for n in names], want_t=training)
        # many-row forward (the stacked calls of a training iteration): the eight packed weights pre-split in ONE pass, so that the convs run on
        # the mover-wave kernel (csrc/gemm_mw.hip: weights global -> LDS by DMA); conv j's rows start at j * Co of the plane buffer
        Co_w = wps.shape[1]
        w_pl = ops.split3_planes(wps.view(len(names) * Co_w, wps.shape[2])) if (ops.GEMM_PLANES and Bs * T >= 8192) else None
        # the eight dropout masks of the block convs: injected (tests) or ONE draw launch; each rides in its conv's GEMM epilogue
        sites = [f"{tag}.tcn{i}.drop{ci + 1}" for i in range(self.n_layers) for ci in range(2)]
        masks = [None] * len(sites)
        if training:
            if inject is not None and all(sn in inject for sn in sites):
                masks = [inject[sn].contiguous() for sn in sites]
            elif self.p_drop > 0.0:
                drawn = ops.dropout_mask(L.empty(len(sites), Bs, T, wps.shape[1], like=in_data), self.p_drop, self.rng.state,
                                         self.rng.site(f"{tag}.tcn.drop"))
                masks = [inject[sn].contiguous() if (inject is not None and sn in inject) else drawn[j] for j, sn in enumerate(sites)]
        tcn = []
        # the block's closing relu(out + x) as a second output of conv2's launch when that launch runs on a kernel with the epilogue
        # extensions (ops.nt_ext_supported: the big-product path; small batches keep the separate add_relu pass)
        fuse_res = wps.shape[1] == cur.shape[2] and cur.is_contiguous() and ops.nt_ext_supported(
            Win.conv(cur, 2, pad=1, dil=1, rows_out=T), wps[0], cur, c_batch_stride=cur.stride(0), c_row_stride=cur.stride(1), c_rows_out=T)
        for i in range(self.n_layers):
            d = 2 ** i
            blk = {"x": cur, "d": d}
            h = cur
            y = None
            for ci in range(2):
                j = 2 * i + ci
                m = masks[j]
                if m is not None:
                    assert m.shape == (Bs, T, wps.shape[1]), (sites[j], m.shape)
                # causal conv (chomp) + ReLU + dropout scale in one GEMM
                if ci == 1 and fuse_res:
                    y = torch.empty_like(cur)
                    o = L.conv_fwd(h, wps[j], P[names[j] + ".bias"], 2, pad=d, dil=d, rows_out=T, act_slope=0.0, out_scale=m, res=cur, out2=y,
                                   res_slope=0.0, w_planes=w_pl, w_row0=j * Co_w)
                else:
                    o = L.conv_fwd(h, wps[j], P[names[j] + ".bias"], 2, pad=d, dil=d, rows_out=T, act_slope=0.0, out_scale=m, w_planes=w_pl,
                                   w_row0=j * Co_w)
                blk[f"in{ci}"], blk[f"wt{ci}"], blk[f"o{ci}"], blk[f"m{ci}"] = h, (wts[j] if wts is not None else None), o, m
                h = o
            if y is None:
                y = ops.add_relu(h, cur, torch.empty_like(cur))
            blk["y"] = y
            tcn.append(blk)
            cur = y
        tp["tcn"], tp["emb_mask"], tp["text_x"] = tcn, emb_mask, cur
        L.linear_fwd(cur.view(Bs * T, -1), P[f"{te}.decoder.weight"], P[f"{te}.decoder.bias"],
                     out=in_data.view(Bs * T, in_size)[:, self.c_text:self.c_text + 32])

    # ---------------------------------------------------------------------------------------------- backward
    def backward(self, tp, d_out, d_mu=None, d_logvar=None, *, b0=0, nb=None, on_ready=None):
        """Gradients of rows [b0, b0+nb) (one BatchNorm group) of a taped forward.  d_out: (nb, T, D);
        d_mu/d_logvar: (nb, 16) direct gradients (KLD term) or None.  Accumulates into the gradient slab.
        on_ready(prefixes): called as soon as the gradients of the named top-level parameter groups are final, in
        backward order -- the data-parallel trainer starts that bucket's all-reduce while the rest still runs."""
        ready = on_ready if on_ready is not None else (lambda names: None)
        P, G, Bf = self.views()
        Bs, T, H, D = tp["Bs"], self.T, self.H, self.pose_dim
        nb = Bs - b0 if nb is None else nb
        rows = slice(b0, b0 + nb)
        grp = b0 // (Bs // tp["groups"])
        assert nb == Bs // tp["groups"] and b0 % nb == 0, "backward runs on exactly one BatchNorm group"
        M = nb * T
        in_size = tp["in_size"]
        d_out2 = d_out.contiguous().view(M, D)

        # out MLP
        if tp["h1"] is None:
            # composed output MLP: the batch-sized operands enter only P = d_out^T o (+ its column sums) and d o = d_out (W2 W1)
            y_rows = tp["y_last"].view(-1, 2 * H)[b0 * T:b0 * T + M]
            buf = ops.zeros(D * H + D, device=d_out2.device)                    # one fill for both accumulators
            Pm, sv = buf[:D * H].view(D, H), buf[D * H:]
            # P = d_out^T (y_fwd + y_rev): two products of one grouped launch accumulating into the same [D x H] block
            ops.gemm_tn_group([dict(dY=d_out2, A=Win.plain(y_rows[:, :H]), dW=Pm, dbias=sv), dict(dY=d_out2, A=Win.plain(y_rows[:, H:]), dW=Pm)])
            ops.out_mlp_param_grads(Pm, sv, P["out.0.weight"], P["out.0.bias"], P["out.2.weight"], G["out.0.weight"], G["out.0.bias"],
                                    G["out.2.weight"], G["out.2.bias"])
            dy = ops.gemm_nt(Win.plain(d_out2), tp["w21t"], None, L.empty(M, 2 * H, like=d_out2)).view(nb, T, 2 * H)   # both halves at once
        else:
            dh1 = L.linear_bwd(d_out2, tp["h1"][b0 * T:b0 * T + M], P["out.2.weight"], G["out.2.weight"], G["out.2.bias"])
            do = L.linear_bwd(dh1, tp["o"][b0 * T:b0 * T + M], P["out.0.weight"], G["out.0.weight"], G["out.0.bias"])
            dy = ops.dup_halves(do, L.empty(nb, T, 2 * H, like=do))
        fork = L.Fork(dy.device, enabled=self.use_side_stream)
        d_in = L.gru_stack_bwd(dy, tp["gru"], P, G, "gru", self.n_layers, b0=b0, nb=nb, fork=fork)      # (nb, T, in_size)
        d_in2 = d_in.view(M, in_size)

        if self.z_mode == "speaker":
            # speaker path
            dz = ops.sum_rows(d_in2[:, self.c_z:], L.empty(nb, 16, like=d_in), nb, T)
            if ops.speaker_bwd_supported(nb):
                # reparameterisation, the three linears' weight / bias / input gradients and the embedding scatter: one launch
                ops.speaker_bwd(dz, d_mu.contiguous() if d_mu is not None else None, d_logvar.contiguous() if d_logvar is not None else None,
                                tp["logvar"][rows], tp["eps"][rows], tp["zc"][rows], tp["se"][rows], tp["vid"][rows].contiguous(),
                                P["speaker_embedding.1.weight"], P["speaker_mu.weight"], P["speaker_logvar.weight"],
                                G["speaker_embedding.1.weight"], G["speaker_embedding.1.bias"], G["speaker_mu.weight"], G["speaker_mu.bias"],
                                G["speaker_logvar.weight"], G["speaker_logvar.bias"], G["speaker_embedding.0.weight"])
                dz = None
        if self.z_mode == "speaker" and dz is not None:
            dmu = d_mu.clone() if d_mu is not None else ops.zeros_like(dz)
            dlv = d_logvar.clone() if d_logvar is not None else ops.zeros_like(dz)
            ops.reparam_bwd(dz, tp["logvar"][rows], tp["eps"][rows], dmu, dlv)
            zc = tp["zc"][rows]
            dzc = L.linear_bwd(dmu, zc, P["speaker_mu.weight"], G["speaker_mu.weight"], G["speaker_mu.bias"])
            L.linear_bwd(dlv, zc, P["speaker_logvar.weight"], G["speaker_logvar.weight"], G["speaker_logvar.bias"],
                         dx_out=dzc, accumulate_dx=True)
            dse = L.linear_bwd(dzc, tp["se"][rows], P["speaker_embedding.1.weight"], G["speaker_embedding.1.weight"],
                               G["speaker_embedding.1.bias"])
            ops.embed_scatter_add(dse, tp["vid"][rows].contiguous(), G["speaker_embedding.0.weight"])

        if self.use_text:
            # text encoder
            te = "text_encoder"
            d_text = d_in2[:, self.c_text:self.c_text + 32]
            dcur = L.linear_bwd(d_text, tp["text_x"][rows].reshape(M, -1), P[f"{te}.decoder.weight"], G[f"{te}.decoder.weight"],
                                G[f"{te}.decoder.bias"])
            Co_t = P[f"{te}.tcn.network.0.conv1.weight_v"].shape[0]
            dwp_all = ops.zeros(2 * self.n_layers, Co_t, 2 * Co_t, device=dcur.device)       # packed weight gradients of the 8 convs: one fill
            wn_pre = []                                                                       # weight-norm backward of all convs: one launch after the loop
            wg_probs = []
            for i in range(self.n_layers - 1, -1, -1):
                blk = tp["tcn"][i]
                d = blk["d"]
                Cc = dcur.shape[1]
                # relu(out + x) and the second conv's relu + dropout gate in one pass
                o1, m1 = blk["o1"][rows].reshape(M, -1), blk["m1"]
                dsum, dc_top = ops.act_mask_bwd2(dcur, blk["y"][rows].reshape(M, Cc), o1, None if m1 is None else m1[rows].reshape(M, -1), 0.0,
                                                 torch.empty_like(dcur), torch.empty_like(o1))
                dh = dsum
                dc_fused = None                  # conv1's output gradient when conv2's input-gradient launch gated it in its epilogue
                for ci, name in ((1, "conv2"), (0, "conv1")):
                    pre = f"{te}.tcn.network.{i}.{name}"
                    o = blk[f"o{ci}"][rows].reshape(M, -1)
                    m = blk[f"m{ci}"]
                    if ci == 1:
                        dc = dc_top
                    elif dc_fused is not None:
                        dc = dc_fused
                    else:
                        dc = ops.act_mask_bwd(dh, o, None if m is None else m[rows].reshape(M, -1), 0.0, torch.empty_like(o))
                    dc3 = dc.view(nb, T, -1)
                    xin = blk[f"in{ci}"][rows]
                    v = P[pre + ".weight_v"]
                    dwp = dwp_all[2 * i + ci]
                    # the weight gradient is off the dependency chain: all convs' products go into ONE grouped launch after the loop
                    # (eight launches of 33 us, each a tail-heavy 700-workgroup grid, against one that fills the chip)
                    wg_probs.append(dict(dY=dc, A=Win.conv(xin, 2, pad=d, dil=d, rows_out=T), dW=dwp, dbias=G[pre + ".bias"]))
                    wn_pre.append((dwp, pre))
                    # dx[t] = dy[t] . W[:, :, 1] + dy[t + d] . W[:, :, 0]  -> taps (t + d, t) with B = w^T per tap (from the forward's batch)
                    wT = blk[f"wt{ci}"]
                    if ci == 1:
                        a_win = Win.taps(dc3, 2, shift=d, dil=-d, rows_out=T)
                        dh = L.empty(M, xin.shape[2], like=dc)
                        o0, m0 = blk["o0"][rows].reshape(M, -1), blk["m0"]
                        if o0.shape == dh.shape and ops.nt_ext_supported(a_win, wT, dh):
                            # relu + dropout backward of conv1's output (its saved post-dropout activation is the gate, the dropout scale
                            # the multiplier) in this launch's epilogue: dh arrives as conv1's output gradient
                            ops.gemm_nt(a_win, wT, None, dh, out_scale=None if m0 is None else m0[rows].reshape(M, -1), gate=o0)
                            dc_fused = dh
                        else:
                            ops.gemm_nt(a_win, wT, None, dh)
                    else:   # first conv of the block: add into the residual branch gradient
                        dh = ops.gemm_nt(Win.taps(dc3, 2, shift=d, dil=-d, rows_out=T), wT, None, dsum, accumulate=True)
                dcur = dh
            for j0 in range(0, len(wg_probs), _lib.MAX_GROUP):
                ops.gemm_tn_group(wg_probs[j0:j0 + _lib.MAX_GROUP])
            for j0 in range(0, len(wn_pre), 8):
                chunk = wn_pre[j0:j0 + 8]
                ops.weight_norm_bwd_batch([dw for dw, _ in chunk], [P[pre + ".weight_v"] for _, pre in chunk],
                                          [P[pre + ".weight_g"] for _, pre in chunk], [G[pre + ".weight_g"] for _, pre in chunk],
                                          [G[pre + ".weight_v"] for _, pre in chunk])
            em = tp["emb_mask"]
            demb = ops.mul(dcur, em[rows].reshape(M, -1), torch.empty_like(dcur)) if em is not None else dcur
            if f"{te}.embedding.weight" not in self.slab.frozen:           # freeze_wordembed (:40-41): no gradient, never stepped
                ops.embed_scatter_add(demb, tp["in_text"][rows].contiguous().view(-1), G[f"{te}.embedding.weight"])
        fork.join()        # the GRU weight gradients ran beside the recurrences and the text-encoder backward
        ready(("out", "gru"))
        ready(("speaker_embedding", "speaker_mu", "speaker_logvar", "text_encoder"))

        if not self.use_audio:
            ready(("audio_encoder",))
            return None
        # wav encoder
        fe = "audio_encoder.feat_extractor"
        dyw = d_in[:, :, self.c_audio:self.c_audio + 32]                      # (nb, 34, 32) strided view
        wrow0 = 0 if tp["wav_shared"] else b0              # shared audio: the encoder ran once on the group's rows
        wgrp = 0 if tp["wav_shared"] else grp
        wrows = slice(wrow0, wrow0 + nb)
        for li in range(3, -1, -1):
            idx, Co, Ci, stride, pad = WAV_CONVS[li]
            x_in, st, _ = tp["wav"][li]                     # input of conv li (post BN+act of the previous block)
            x_rows = x_in[wrows]
            if li == 1 and ops.WAV_FUSED and (Co, Ci, stride, pad) == (32, 16, 6, 0) and x_rows.is_contiguous():
                ops.wav_conv2_wgrad(dyw.contiguous(), x_rows, G[f"{fe}.{idx}.weight"], G[f"{fe}.{idx}.bias"])     # whole result per wave: one pass over the 65 MB activation
            else:
                L.conv_wgrad(dyw, x_rows, G[f"{fe}.{idx}.weight"], G[f"{fe}.{idx}.bias"], WAV_KW, stride=stride, pad=pad)
            if li == 0:
                break
            pidx = WAV_CONVS[li - 1][0] + 1
            x_prev, st_prev, _ = tp["wav"][li - 1]
            if isinstance(st_prev, L.WavFrontState) and L.wav_front_bwd_fused_supported(P[f"{fe}.{idx}.weight"], stride):
                # conv2's input gradient, BatchNorm1's backward and conv1's weight gradient in ONE reduction over d c2
                L.wav_front_bwd_fused(dyw, P[f"{fe}.{idx}.weight"], st_prev, x_prev.view(x_prev.shape[0], -1), P[f"{fe}.0.weight"],
                                      P[f"{fe}.0.bias"], P[f"{fe}.1.weight"], G[f"{fe}.0.weight"], G[f"{fe}.0.bias"], G[f"{fe}.1.weight"],
                                      G[f"{fe}.1.bias"], g0=wgrp, row0=wrow0)
                break
            dxa = L.conv_dgrad(dyw, P[f"{fe}.{idx}.weight"], x_rows.shape[1], stride=stride)   # grad w.r.t. act(BN(c_prev))
            if isinstance(st_prev, L.WavFrontState):        # fused front end: BatchNorm backward and conv1's weight gradient in one reduction
                L.wav_front_bwd(dxa, st_prev, x_prev.view(x_prev.shape[0], -1), P[f"{fe}.0.weight"], P[f"{fe}.0.bias"], P[f"{fe}.1.weight"],
                                G[f"{fe}.0.weight"], G[f"{fe}.0.bias"], G[f"{fe}.1.weight"], G[f"{fe}.1.bias"], g0=wgrp, row0=wrow0)
                break
            dyw = L.bn_bwd(dxa, st_prev, P[f"{fe}.{pidx}.weight"], P[f"{fe}.{pidx}.bias"], G[f"{fe}.{pidx}.weight"],
                           G[f"{fe}.{pidx}.bias"], g0=wgrp, ng=1, row0=wrow0)
        ready(("audio_encoder",))
        return None


# =================================================================================================== discriminator
class DiscriminatorEngine(_Engine):
    H = 64
    CONVS = ((0, 16, 27), (3, 8, 16), (6, 8, 8))
    d_fork = os.environ.get("TG_D_FORK", "0") != "0"
    d_defer_wgrad = os.environ.get("TG_D_DEFER_WGRAD", "1") != "0"

    def forward(self, poses, *, training, groups=1, save=False, inject=None, tag="d"):
        """poses: (Bs, 34, 27).  Returns {'logit': (Bs,1) pre-sigmoid, 'prob': sigmoid, 'tape'}."""
        P, G, Bf = self.views()
        Bs, T0, D = poses.shape
        tp = {"Bs": Bs, "groups": groups, "poses": poses.contiguous()}
        x = tp["poses"]
        convs = []
        for idx, Co, Ci in self.CONVS:
            wp = L.pack_conv_weight(P[f"pre_conv.{idx}.weight"])
            c = L.conv_fwd(x, wp, P[f"pre_conv.{idx}.bias"], 3)
            if idx == 6:
                convs.append((x, None))
                x = c
                break
            y, st = L.bn_fwd(c, P[f"pre_conv.{idx + 1}.weight"], P[f"pre_conv.{idx + 1}.bias"],
                             Bf[f"pre_conv.{idx + 1}.running_mean"], Bf[f"pre_conv.{idx + 1}.running_var"],
                             Bf[f"pre_conv.{idx + 1}.num_batches_tracked"], training=training, groups=groups, act_slope=1.0)
            convs.append((x, st))
            x = y
        T = x.shape[1]
        y, gtape = L.gru_stack_fwd(x, P, "gru", 4, self.H, p_drop=0.3, training=training, rng=self.rng, save=save,
                                   inject=inject, tag=tag)
        # head: direction sum -> Linear(64 -> 1) per frame -> Linear(28 -> 1) -> sigmoid (:243-252), one launch
        l1, logit, prob = ops.d_head_fwd(y, P["out.weight"], P["out.bias"], P["out2.weight"], P["out2.bias"])
        tp.update(convs=convs, gru=gtape, y=y, l1=l1, T=T)
        return {"logit": logit, "prob": prob, "tape": tp if save else None}

    def backward(self, tp, d_logit, *, b0=0, nb=None, param_grads=True, need_dposes=False):
        """d_logit: (nb, 1) gradient w.r.t. the pre-sigmoid output for rows [b0, b0+nb): whole BatchNorm groups."""
        P, G, Bf = self.views()
        Bs, T, H = tp["Bs"], tp["T"], self.H
        nb = Bs - b0 if nb is None else nb
        rows = slice(b0, b0 + nb)
        per = Bs // tp["groups"]
        assert nb % per == 0 and b0 % per == 0
        grp, ng = b0 // per, nb // per
        M = nb * T
        pg = param_grads
        dy = ops.d_head_bwd(d_logit.contiguous().view(nb), tp["y"][rows], tp["l1"][rows], P["out.weight"], P["out2.weight"],
                            L.empty(nb, T, 2 * H, like=d_logit),
                            (G["out.weight"], G["out.bias"], G["out2.weight"], G["out2.bias"]) if pg else None)
        # TG_D_FORK=1: the weight-gradient GEMMs (off the dependency chain) on a second stream beside the next layer's recurrence, which
        # occupies 2 * nb / 16 workgroups; the H = 64 kernels have no cross-workgroup synchronisation, so sharing the device is safe
        # (TG_D_FORK=1, lab: the weight-gradient GEMMs on a second stream beside the next layer's recurrence -- measured 0.19 ms SLOWER per
        # iteration under graph replay: seven fork / join pairs cost more than the 140 us they hide)
        fork = L.Fork(dy.device, enabled=pg and self.d_fork)
        # default: every weight gradient of the backward pass is deferred and launched in three grouped launches at the end (they are off
        # the dependency chain; one launch per layer was mostly launch latency and tail on the 32-workgroup-sized problems)
        deferred = [] if (pg and not self.d_fork and self.d_defer_wgrad) else None
        dx = L.gru_stack_bwd(dy, tp["gru"], P, G, "gru", 4, b0=b0, nb=nb, param_grads=pg, fork=fork, defer=deferred)    # (nb, 28, 8)
        def finish(ret):
            if deferred:
                L.tn_group_deferred(deferred)
            fork.join()
            return ret
        for li in (2, 1, 0):
            idx, Co, Ci = self.CONVS[li]
            x_in, _ = tp["convs"][li]
            x_rows = x_in[rows]
            if pg:
                fork.keep(dx, x_rows)
                with fork:
                    L.conv_wgrad(dx, x_rows, G[f"pre_conv.{idx}.weight"], G[f"pre_conv.{idx}.bias"], 3, defer=deferred)
            if li == 0 and not need_dposes:
                return finish(None)
            dxa = L.conv_dgrad(dx, P[f"pre_conv.{idx}.weight"], x_rows.shape[1])
            if li == 0:
                return finish(dxa)
            pidx = self.CONVS[li - 1][0] + 1
            _, st_prev = tp["convs"][li - 1]
            dx = L.bn_bwd(dxa, st_prev, P[f"pre_conv.{pidx}.weight"], P[f"pre_conv.{pidx}.bias"],
                          G[f"pre_conv.{pidx}.weight"] if pg else None, G[f"pre_conv.{pidx}.bias"] if pg else None,
                          g0=grp, ng=ng, row0=b0)
        return None


# ===================================================================================================== autoencoder
class AutoencoderEngine(_Engine):
    """EmbeddingNet(mode='pose'): PoseEncoderConv + PoseDecoderConv, 34-frame branch (embedding_net.py:42-82,165-217)."""
    ENC = (("net.0", 32, 27, 3, 1), ("net.1", 64, 32, 3, 1), ("net.2", 64, 64, 4, 2))

    def _bn(self, P, Bf, pre, x, training, slope):
        return L.bn_fwd(x, P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"], Bf[pre + ".running_var"],
                        Bf[pre + ".num_batches_tracked"], training=training, act_slope=slope)

    def encode(self, poses, *, training, tape=None):
        P, G, Bf = self.views()
        e = "pose_encoder"
        x = poses.contiguous()
        B = x.shape[0]
        rec = []
        for name, Co, Ci, kw, stride in self.ENC:
            c = L.conv_fwd(x, L.pack_conv_weight(P[f"{e}.{name}.0.weight"]), P[f"{e}.{name}.0.bias"], kw, stride=stride)
            y, st = self._bn(P, Bf, f"{e}.{name}.1", c, training, 0.2)
            rec.append((x, st))
            x = y
        c4 = L.conv_fwd(x, L.pack_conv_weight(P[f"{e}.net.3.weight"]), P[f"{e}.net.3.bias"], 3)     # (B, 12, 32)
        # flatten(1) of the reference's (B, C=32, L=12) layout: channel-major -> permute our (B, L, C)
        flat = ops.permute3(c4, L.empty(B, 32 * c4.shape[1], like=c4), (0, 2, 1))
        f1 = L.linear_fwd(flat, P[f"{e}.out_net.0.weight"], P[f"{e}.out_net.0.bias"])
        y1, st1 = self._bn(P, Bf, f"{e}.out_net.1", f1, training, 1.0)
        f2 = L.linear_fwd(y1, P[f"{e}.out_net.3.weight"], P[f"{e}.out_net.3.bias"])
        y2, st2 = self._bn(P, Bf, f"{e}.out_net.4", f2, training, 1.0)
        f3 = L.linear_fwd(y2, P[f"{e}.out_net.6.weight"], P[f"{e}.out_net.6.bias"])
        mu = L.linear_fwd(f3, P[f"{e}.fc_mu.weight"], P[f"{e}.fc_mu.bias"])
        logvar = L.linear_fwd(f3, P[f"{e}.fc_logvar.weight"], P[f"{e}.fc_logvar.bias"])
        if tape is not None:
            tape.update(enc=rec, x4=x, c4=c4, flat=flat, st1=st1, y1=y1, st2=st2, y2=y2, f3=f3)
        return mu, logvar

    def decode(self, feat, *, training, tape=None):
        P, G, Bf = self.views()
        d = "decoder"
        B = feat.shape[0]
        p0 = L.linear_fwd(feat, P[f"{d}.pre_net.0.weight"], P[f"{d}.pre_net.0.bias"])
        yp, stp = self._bn(P, Bf, f"{d}.pre_net.1", p0, training, 1.0)
        p3 = L.linear_fwd(yp, P[f"{d}.pre_net.3.weight"], P[f"{d}.pre_net.3.bias"])                 # (B, 136) = (B, 4, 34)
        x0 = ops.permute3(p3.view(B, 4, 34), L.empty(B, 34, 4, like=p3), (0, 2, 1))                # channel-last
        t0 = L.conv_transpose_fwd(x0, P[f"{d}.net.0.weight"], P[f"{d}.net.0.bias"])                 # (B, 36, 32)
        y0, s0 = self._bn(P, Bf, f"{d}.net.1", t0, training, 0.2)
        t1 = L.conv_transpose_fwd(y0, P[f"{d}.net.3.weight"], P[f"{d}.net.3.bias"])                 # (B, 38, 32)
        y1, s1 = self._bn(P, Bf, f"{d}.net.4", t1, training, 0.2)
        c6 = L.conv_fwd(y1, L.pack_conv_weight(P[f"{d}.net.6.weight"]), P[f"{d}.net.6.bias"], 3)    # (B, 36, 32)
        out = L.conv_fwd(c6, L.pack_conv_weight(P[f"{d}.net.7.weight"]), P[f"{d}.net.7.bias"], 3)   # (B, 34, 27)
        if tape is not None:
            tape.update(feat=feat, stp=stp, yp=yp, x0=x0, s0=s0, y0=y0, s1=s1, y1d=y1, c6=c6)
        return out

    def forward(self, poses, *, training, save=False):
        tape = {} if save else None
        mu, logvar = self.encode(poses, training=training, tape=tape)
        recon = self.decode(mu, training=training, tape=tape)      # variational_encoding=False: z = mu
        return {"feat": mu, "mu": mu, "logvar": logvar, "recon": recon, "tape": tape}

    def backward(self, tp, d_recon):
        """Gradients of the reconstruction loss (train_feature_extractor.py:54-97, z = mu so fc_logvar gets none)."""
        P, G, Bf = self.views()
        d, e = "decoder", "pose_encoder"
        B = d_recon.shape[0]
        dy = d_recon.contiguous()
        # decoder
        L.conv_wgrad(dy, tp["c6"], G[f"{d}.net.7.weight"], G[f"{d}.net.7.bias"], 3)
        dy = L.conv_dgrad(dy, P[f"{d}.net.7.weight"], 36)
        L.conv_wgrad(dy, tp["y1d"], G[f"{d}.net.6.weight"], G[f"{d}.net.6.bias"], 3)
        dy = L.conv_dgrad(dy, P[f"{d}.net.6.weight"], 38)
        dy = L.bn_bwd(dy, tp["s1"], P[f"{d}.net.4.weight"], P[f"{d}.net.4.bias"], G[f"{d}.net.4.weight"], G[f"{d}.net.4.bias"])
        dy = L.conv_transpose_bwd(dy, tp["y0"], P[f"{d}.net.3.weight"], G[f"{d}.net.3.weight"], G[f"{d}.net.3.bias"])
        dy = L.bn_bwd(dy, tp["s0"], P[f"{d}.net.1.weight"], P[f"{d}.net.1.bias"], G[f"{d}.net.1.weight"], G[f"{d}.net.1.bias"])
        dx0 = L.conv_transpose_bwd(dy, tp["x0"], P[f"{d}.net.0.weight"], G[f"{d}.net.0.weight"], G[f"{d}.net.0.bias"])   # (B,34,4)
        dp3 = ops.permute3(dx0, L.empty(B, 136, like=dx0), (0, 2, 1))
        dyp = L.linear_bwd(dp3, tp["yp"], P[f"{d}.pre_net.3.weight"], G[f"{d}.pre_net.3.weight"], G[f"{d}.pre_net.3.bias"])
        dp0 = L.bn_bwd(dyp, tp["stp"], P[f"{d}.pre_net.1.weight"], P[f"{d}.pre_net.1.bias"], G[f"{d}.pre_net.1.weight"],
                       G[f"{d}.pre_net.1.bias"])
        dmu = L.linear_bwd(dp0, tp["feat"], P[f"{d}.pre_net.0.weight"], G[f"{d}.pre_net.0.weight"], G[f"{d}.pre_net.0.bias"])
        # encoder
        df3 = L.linear_bwd(dmu, tp["f3"], P[f"{e}.fc_mu.weight"], G[f"{e}.fc_mu.weight"], G[f"{e}.fc_mu.bias"])
        dy2 = L.linear_bwd(df3, tp["y2"], P[f"{e}.out_net.6.weight"], G[f"{e}.out_net.6.weight"], G[f"{e}.out_net.6.bias"])
        df2 = L.bn_bwd(dy2, tp["st2"], P[f"{e}.out_net.4.weight"], P[f"{e}.out_net.4.bias"], G[f"{e}.out_net.4.weight"],
                       G[f"{e}.out_net.4.bias"])
        dy1 = L.linear_bwd(df2, tp["y1"], P[f"{e}.out_net.3.weight"], G[f"{e}.out_net.3.weight"], G[f"{e}.out_net.3.bias"])
        df1 = L.bn_bwd(dy1, tp["st1"], P[f"{e}.out_net.1.weight"], P[f"{e}.out_net.1.bias"], G[f"{e}.out_net.1.weight"],
                       G[f"{e}.out_net.1.bias"])
        dflat = L.linear_bwd(df1, tp["flat"], P[f"{e}.out_net.0.weight"], G[f"{e}.out_net.0.weight"], G[f"{e}.out_net.0.bias"])
        L4 = tp["c4"].shape[1]
        dc4 = ops.permute3(dflat.view(B, 32, L4), L.empty(B, L4, 32, like=dflat), (0, 2, 1))
        L.conv_wgrad(dc4, tp["x4"], G[f"{e}.net.3.weight"], G[f"{e}.net.3.bias"], 3)
        dy = L.conv_dgrad(dc4, P[f"{e}.net.3.weight"], tp["x4"].shape[1])
        for li in (2, 1, 0):
            name, Co, Ci, kw, stride = self.ENC[li]
            x_in, st = tp["enc"][li]
            dc = L.bn_bwd(dy, st, P[f"{e}.{name}.1.weight"], P[f"{e}.{name}.1.bias"], G[f"{e}.{name}.1.weight"],
                          G[f"{e}.{name}.1.bias"])
            L.conv_wgrad(dc, x_in, G[f"{e}.{name}.0.weight"], G[f"{e}.{name}.0.bias"], kw, stride=stride)
            if li > 0:
                dy = L.conv_dgrad(dc, P[f"{e}.{name}.0.weight"], x_in.shape[1], stride=stride)
