"""Forward / backward building blocks composed from the HIP ops.

Activations are channel-last (batch, time, channels) everywhere, so every Conv1d / ConvTranspose1d / Linear is a
row-window GEMM (ops.Win) and im2col is never materialised.  Each backward routine accumulates parameter
gradients straight into the caller's gradient views (which live in one flat slab per network, see params.py).
"""
import os

import torch

from . import ops, _lib
from .ops import Win


def empty(*shape, like=None, device=None, dtype=torch.float32):
    return torch.empty(*shape, device=like.device if like is not None else device, dtype=dtype)


# ----------------------------------------------------------------------------------------------- weights that stand still
_FROZEN = None          # the memo of the active FrozenWeights scope


class FrozenWeights:
    """A scope in which the model's parameters and buffers do NOT change -- one synthesis call (scripts/synthesize.py:36-209 runs the generator in
    eval mode, window after window, on fixed weights).  Inside it the operands of the eval forward that depend on weights only -- the
    weight-normed TCN kernels, conv packs, eval-mode BatchNorm scale / shift, the composed output map -- are computed on first use and reused
    by every later window (and left OUT of a window graph captured after the first, eager, window): 12 launches of the 54 a window had."""

    def __init__(self):
        self.memo = {}

    def __enter__(self):
        global _FROZEN
        self._prev, _FROZEN = _FROZEN, self.memo
        return self

    def __exit__(self, *exc):
        global _FROZEN
        _FROZEN = self._prev
        return False


def frozen(key, make):
    """make() once per FrozenWeights scope and key (a tag + the data pointers of the weights it reads); plain make() outside a scope."""
    if _FROZEN is None:
        return make()
    v = _FROZEN.get(key)
    if v is None:
        v = _FROZEN[key] = make()
    return v


class WeightPrep:
    """Weight-derived GEMM operands (transposes, conv packs, bf16 planes) of slab-resident parameters, refreshed ONCE per optimiser step by
    batched launches instead of one small permute launch per use (~60 per training iteration = 0.26 ms of launch overhead).

    A trainer registers its parameter slabs as groups, activates the cache around its iteration (`with prep.active():`) and calls
    refresh(group) where that group's weights may have changed (start of the iteration; after the discriminator's optimiser
    step).  transpose2d / pack_conv_weight then return the cached operand for sources that live in a registered slab.  Entries are
    discovered on first use in eager mode (computed in place, then kept); under graph capture unknown sources fall back to the
    inline permute, so a capture never allocates or uploads a table.

    Every operand belongs to the PART of the iteration that first reads it, and each part has its own table and launch (refresh(gid, part)),
    so only what the head of the dependency chain really reads is refreshed there (GanTrainer; refresh(gid) launches all parts, as before):
      "main0"  read on the main stream before the generator forward's fork is joined           -> refreshed at the start, main stream
      "side0"  read on the forked branch, or on the main stream after the join (`zone`)        -> head of the forked branch
      "late"   first read after the generator's forward (`late`: backward passes, discriminator) -> on the forked branch, behind the encoder
    A later request from an EARLIER part than the operand's own promotes it (eager) or raises (under capture: the table is frozen)."""
    PARTS = ("main0", "side0", "late")

    def __init__(self):
        self.by_key = {}            # (src ptr, dims, perm) -> dst tensor
        self.part_of = {}           # same key -> part
        self.groups = {}            # gid -> {"range": (lo, hi), "jobs": [(src3, dst, perm)], "desc": {part: tensor}, "wgs": {part: int}, "n": {part: int}}
        self.late = False           # set by the trainer once the generator's forward is enqueued
        self.zone = "pre_join"      # set by the generator engine: "post_join" between its fork's join and the end of its forward

    def _part_now(self):
        if self.late:
            return "late"
        return "side0" if (Fork.on_side or self.zone == "post_join") else "main0"

    def add_slab(self, gid, flat):
        lo = flat.data_ptr()
        g = self.groups.get(gid)
        if g is None or g["range"][0] != lo:          # new or re-allocated slab: forget what pointed into the old one
            if g is not None:
                for src3, _, perm in g["jobs"]:
                    key = (src3.data_ptr(), tuple(src3.shape), perm)
                    self.by_key.pop(key, None)
                    self.part_of.pop(key, None)
            self.groups[gid] = {"range": (lo, lo + flat.numel() * flat.element_size()), "jobs": [], "desc": {}, "wgs": {}, "n": {}}

    def _group_of(self, ptr):
        for gid, g in self.groups.items():
            if g["range"][0] <= ptr < g["range"][1]:
                return gid
        return None

    def _rebuild(self, g, device):
        for part in self.PARTS:
            rows, wg0 = [], 0
            for s3, d, pm in g["jobs"]:
                if self.part_of[(s3.data_ptr(), tuple(s3.shape), pm)] != part:
                    continue
                if pm[0] == 8:                                        # table entry: d0 = rows, d1 = cw, d2 = cwp
                    nwg = max(1, min(512, (d.numel() // 3 + 16383) // 16384))
                    rows.append([s3.data_ptr(), d.data_ptr(), s3.shape[1], s3.shape[2], d.shape[2], 8, 0, 0, wg0, nwg])
                    wg0 += nwg
                    continue
                if pm[0] == 11:                                       # fp16 x 2 planes of [src^T | other^T] (pm[1] = the other matrix): 8 output rows per workgroup
                    nwg = (s3.shape[2] + 8) // 8
                    rows.append([s3.data_ptr(), d.data_ptr(), s3.shape[1], s3.shape[2], ops.planes_cwp(2 * s3.shape[1]), 11, pm[1], 0, wg0, nwg])
                    wg0 += nwg
                    continue
                if pm[0] == 10:                                       # fp16 x 2 planes: d = the flat buffer [2 planes | inverse scales], one wave per row
                    nwg = max(1, min(512, (s3.shape[1] + 16) // 16))
                    rows.append([s3.data_ptr(), d.data_ptr(), s3.shape[1], s3.shape[2], ops.planes_cwp(s3.shape[2]), 10, 0, 0, wg0, nwg])
                    wg0 += nwg
                    continue
                # ~2 LDS tiles (32 x 32 per batch index) per workgroup.  Counted in TILES, not elements: a conv pack (Co, Ci, kw) is Co small
                # tiles -- sized by elements it got one workgroup that walked 16-64 tiles one after the other (16-27 us per launch)
                tiles = s3.shape[0] * ((s3.shape[1] + 31) // 32) * ((s3.shape[2] + 31) // 32) if tuple(pm) == (0, 2, 1) else (d.numel() + 1023) // 1024
                nwg = max(1, min(512, (tiles + 1) // 2))
                rows.append([s3.data_ptr(), d.data_ptr(), *s3.shape, *pm, wg0, nwg])
                wg0 += nwg
            if g.setdefault("rows", {}).get(part) == rows:
                continue                                              # this part's table is unchanged
            if g["desc"].get(part) is not None:
                g.setdefault("retired", []).append(g["desc"][part])   # a captured graph may still read the previous table: never freed
            g["desc"][part] = torch.tensor(rows, dtype=torch.int64).to(device) if rows else None
            g["wgs"][part], g["n"][part], g["rows"][part] = wg0, len(rows), rows

    def get(self, src3, perm, out_shape, other=None):
        """The permuted copy of src3 (3-D view of a slab parameter) or None when the caller has to permute inline.
        perm = (9, stride, 0) is the conv input-gradient pack (ops.conv_dgrad_pack) of a (Co, Ci, kw) weight."""
        key = (src3.data_ptr(), tuple(src3.shape), perm)
        hit = self.by_key.get(key)
        if hit is not None:
            now, own = self._part_now(), self.part_of[key]
            if self.PARTS.index(now) < self.PARTS.index(own):
                # read earlier in the iteration than the part that refreshes it: it would be read stale, or while it is being rewritten
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError(f"WeightPrep: operand {tuple(src3.shape)} perm {perm} is refreshed in part '{own}' but read in '{now}' "
                                       "during a capture; run an eager iteration of this configuration first")
                self.part_of[key] = now
                self._rebuild(self.groups[self._group_of(src3.data_ptr())], src3.device)
                # ... and THIS iteration the operand's own (later) part has not refreshed it yet -- it still holds the weights from before the last
                # optimiser step, or is being rewritten right now on the forked stream: recompute it here, on the reader's stream
                self._compute(src3, hit, perm, other)
            return hit
        gid = self._group_of(src3.data_ptr())
        if gid is None or torch.cuda.is_current_stream_capturing():
            return None
        if perm[0] == 8:                                          # bf16 x 3 planes of a weight [rows][cw] (src3 = (1, rows, cw)): [3][rows + 1][cwp] bf16
            dst = ops.split3_planes(src3[0]).t
        elif perm[0] == 10:                                       # fp16 x 2 planes + inverse scales of the same: one flat fp16 buffer (ops.h2_planes_alloc)
            dst = ops.h2_planes_alloc(src3.shape[1], src3.shape[2], src3.device)[0]
            ops.split2h_planes(src3[0], buf=dst)
        elif perm[0] == 11:                                       # ... of the K-concatenated transpose of src3[0] and `other`
            dst = ops.h2_planes_alloc(src3.shape[2], 2 * src3.shape[1], src3.device)[0]
            ops.split2h_planes_tcat(src3[0], other, buf=dst)
        elif perm[0] == 9:
            dst = empty(*out_shape, like=src3)
            ops.conv_dgrad_pack(src3, dst, perm[1])
        else:
            dst = empty(*out_shape, like=src3)
            ops.permute3(src3, dst.view(-1), perm)                # fresh now; later refreshes keep it so
        g = self.groups[gid]
        g["jobs"].append((src3, dst, perm))
        self.by_key[key] = dst
        self.part_of[key] = self._part_now()
        self._rebuild(g, src3.device)
        return dst

    @staticmethod
    def _compute(src3, dst, perm, other=None):
        """One operand, inline on the current stream (what a table entry of the batched refresh does)."""
        if perm[0] == 8:
            ops.split3_planes(src3[0], out=dst)
        elif perm[0] == 10:
            ops.split2h_planes(src3[0], buf=dst)
        elif perm[0] == 11:
            ops.split2h_planes_tcat(src3[0], other, buf=dst)
        elif perm[0] == 9:
            ops.conv_dgrad_pack(src3, dst, perm[1])
        else:
            ops.permute3(src3, dst.view(-1), perm)

    def refresh(self, gid, which="all"):
        g = self.groups.get(gid)
        if g is None:
            return
        for part in (self.PARTS if which == "all" else (which,)):
            if g["n"].get(part, 0):
                ops.permute3_batch(g["desc"][part], g["n"][part], g["wgs"][part])

    def active(self):
        return _PrepScope(self)


class _PrepScope:
    def __init__(self, prep):
        self.prep = prep

    def __enter__(self):
        global _PREP
        self.prev, _PREP = _PREP, self.prep
        return self.prep

    def __exit__(self, *exc):
        global _PREP
        _PREP = self.prev
        return False


_PREP = None


def prep_zone(zone):
    """The generator engine tells the active WeightPrep where its forward stands ('pre_join' / 'post_join' of the forked audio branch)."""
    if _PREP is not None:
        _PREP.zone = zone


def transpose2d(w):
    """[N, K] -> [K, N] on device (re-transposed after every optimiser step: inline here, or batched by an active WeightPrep)."""
    N, K = w.shape
    if _PREP is not None and w.is_contiguous():
        hit = _PREP.get(w.view(1, N, K), (0, 2, 1), (K, N))
        if hit is not None:
            return hit
    out = empty(K, N, like=w)
    ops.permute3(w.view(1, N, K), out, (0, 2, 1))
    return out


def _pack_frozen(w):
    Co, Ci, kw = w.shape

    def make():
        out = empty(Co, kw * Ci, like=w)
        ops.permute3(w, out, (0, 2, 1))
        return out
    return frozen(("pack", w.data_ptr(), Co, Ci, kw), make)


def pack_conv_weight(w):
    """nn.Conv1d weight (Co, Ci, kw) -> tap-major [Co, kw*Ci], the B operand of the conv-as-GEMM."""
    Co, Ci, kw = w.shape
    if Ci == 1 or kw == 1:
        return w.reshape(Co, Ci * kw)          # same memory order
    if _FROZEN is not None:
        return _pack_frozen(w)
    if _PREP is not None and w.is_contiguous():
        hit = _PREP.get(w, (0, 2, 1), (Co, kw * Ci))
        if hit is not None:
            return hit
    out = empty(Co, kw * Ci, like=w)
    ops.permute3(w, out, (0, 2, 1))
    return out


def weight_planes(w2d):
    """Pre-split planes (ops.Planes) of a weight matrix [N][K] -- fp16 x 2 (ops.GEMM_H2, the default) or bf16 x 3 -- refreshed once per optimiser
    step by an active WeightPrep, split inline otherwise."""
    N, K = w2d.shape
    h2 = ops.gemm_h2()
    if _PREP is not None and w2d.is_contiguous():
        hit = _PREP.get(w2d.view(1, N, K), (10 if h2 else 8, 0, 0), None)
        if hit is not None:
            if h2:
                cwp = ops.planes_cwp(K)
                n16 = 2 * (N + 1) * cwp
                return ops.Planes(hit[:n16].view(2, N + 1, cwp), N, K, cwp, "h2", hit[n16:].view(torch.float32))
            return ops.Planes(hit, N, K, hit.shape[2])
    return ops.split_planes(w2d)


def weight_planes_tcat(w0, w1):
    """fp16 x 2 Planes of [w0^T | w1^T] for two [rows][cols] parameters (the K-concatenated input-gradient operand of a bidirectional GRU layer):
    refreshed once per optimiser step by an active WeightPrep, split inline otherwise."""
    rows, cols = w0.shape
    if _PREP is not None and w0.is_contiguous() and w1.is_contiguous():
        hit = _PREP.get(w0.view(1, rows, cols), (11, w1.data_ptr(), 0), None, other=w1)
        if hit is not None:
            cwp = ops.planes_cwp(2 * rows)
            n16 = 2 * (cols + 1) * cwp
            return ops.Planes(hit[:n16].view(2, cols + 1, cwp), cols, 2 * rows, cwp, "h2", hit[n16:].view(torch.float32))
    return ops.split2h_planes_tcat(w0, w1)


def dgrad_pack(w, stride):
    """Conv1d weight (Co, Ci, kw) -> [stride][Ci][J * Co], J = ceil(kw / stride): the B operands of the input-gradient phases (conv_dgrad).
    Cached per optimiser step by an active WeightPrep, packed inline otherwise."""
    Co, Ci, kw = w.shape
    J = (kw + stride - 1) // stride
    if _PREP is not None and w.is_contiguous():
        hit = _PREP.get(w, (9, stride, 0), (stride, Ci, J * Co))
        if hit is not None:
            return hit
    packed = empty(stride, Ci, J * Co, like=w)
    ops.conv_dgrad_pack(w, packed, stride)
    return packed


# ----------------------------------------------------------------------------------------------- linear
def linear_fwd(x2d, W, b, out=None, act_slope=1.0):
    if out is None:
        out = empty(x2d.shape[0], W.shape[0], like=x2d)
    return ops.gemm_nt(Win.plain(x2d), W, b, out, act_slope=act_slope)


def linear_bwd(dy2d, x2d, W, dW, db, need_dx=True, dx_out=None, accumulate_dx=False):
    """dW += dy^T x ; db += colsum(dy) ; dx = dy @ W (optional)."""
    if dW is not None:
        ops.gemm_tn(dy2d, Win.plain(x2d), dW, dbias=db)
    elif db is not None:
        ops.colsum(dy2d, db, accumulate=True)
    if not need_dx:
        return None
    if dx_out is None:
        dx_out = empty(dy2d.shape[0], W.shape[1], like=dy2d)
    return ops.gemm_nt(Win.plain(dy2d), transpose2d(W), None, dx_out, accumulate=accumulate_dx)


# ----------------------------------------------------------------------------------------------- conv1d
def conv_out_len(L, kw, stride=1, pad=0, dil=1):
    return (L + 2 * pad - dil * (kw - 1) - 1) // stride + 1


def conv_fwd(x, w_packed, b, kw, *, stride=1, pad=0, dil=1, out=None, act_slope=1.0, rows_out=None, out_scale=None, res=None, out2=None,
             res_slope=0.0, w_planes=None, w_row0=0, **scales):
    """x: (B, L, Ci) view; w_packed: [Co, kw*Ci]; out: (B, Lout, Co) view (may be a channel slice of a wider buffer).
    res / out2 (ops.nt_ext_supported only): out2 = leaky_relu(out + res, res_slope) written by the same launch.
    scales: the fp16 x 2 row-magnitude operands of ops.gemm_nt (a_rowmax, out_rowmax, out2_rowmax)."""
    B, L, _ = x.shape
    Lo = conv_out_len(L, kw, stride, pad, dil) if rows_out is None else rows_out
    Co = w_packed.shape[0]
    if out is None:
        out = empty(B, Lo, Co, like=x)
    assert tuple(out.shape) == (B, Lo, Co) and out.stride(2) == 1
    A = Win.conv(x, kw, stride=stride, pad=pad, dil=dil, rows_out=Lo)
    ops.gemm_nt(A, w_packed, b, out, act_slope=act_slope, c_batch_stride=out.stride(0), c_row_stride=out.stride(1),
                c_rows_out=Lo, out_scale=out_scale, res=res, out2=out2, res_slope=res_slope, w_planes=w_planes, w_row0=w_row0, **scales)
    return out


def tn_group_deferred(probs):
    """Launch deferred weight-gradient problems in as few grouped launches as possible.  A group runs on the bf16 x 3 kernel only when
    every member qualifies (tg_gemm_tn_group: vectorisable layout, M >= 1024, N and K >= 48 and multiples of 4), so the two kinds are
    grouped separately."""
    def x3(p):
        dY, A = p["dY"], p["A"]
        return dY.shape[0] >= 1024 and dY.shape[1] >= 48 and A.K >= 48 and dY.shape[1] % 4 == 0 and A.K % 4 == 0 and A.s.cw % 4 == 0
    for kind in (True, False):
        sel = [p for p in probs if x3(p) == kind]
        for j0 in range(0, len(sel), _lib.MAX_GROUP):
            ops.gemm_tn_group(sel[j0:j0 + _lib.MAX_GROUP])


def conv_wgrad(dy, x, dW, db, kw, *, stride=1, pad=0, dil=1, defer=None):
    """dy: (B, Lout, Co) contiguous; x: (B, L, Ci) view; dW: (Co, Ci, kw) gradient view (accumulates)."""
    B, Lo, Co = dy.shape
    A = Win.conv(x, kw, stride=stride, pad=pad, dil=dil, rows_out=Lo)
    dy2 = dy.reshape(B * Lo, Co)
    if dW is not None and defer is not None:
        defer.append(dict(dY=dy2, A=A, dW=dW.view(Co, -1), out_kw=kw, dbias=db))
    elif dW is not None:
        ops.gemm_tn(dy2, A, dW.view(Co, -1), out_kw=kw, dbias=db)
    elif db is not None:
        ops.colsum(dy2, db, accumulate=True)


def conv_dgrad(dy, w, L_in, *, stride=1, out=None, accumulate=False):
    """Input gradient of Conv1d(stride, no padding, dilation 1).  dy: (B, Lout, Co) view; w: (Co, Ci, kw).
    One GEMM per phase r = p % stride: dx[b, stride*q + r] = sum_j dy[b, q - j] . Wr[:, j, :]  (zero taps outside).
    out (B, L_in, Ci) contiguous + accumulate: the result is added to it (the pose gradient of the GAN term onto the regression term's)."""
    B, Lo, Co = dy.shape
    _, Ci, kw = w.shape
    J = (kw + stride - 1) // stride
    packed = dgrad_pack(w, stride)
    if out is not None:
        assert tuple(out.shape) == (B, L_in, Ci) and out.is_contiguous()
    dx = out if out is not None else empty(B, L_in, Ci, like=w)
    probs = []
    for r in range(stride):
        nq = (L_in - r + stride - 1) // stride
        if nq <= 0:
            continue
        probs.append(dict(A=Win.taps(dy, J, shift=0, dil=-1, rows_out=nq), W=packed[r], bias=None, out=dx[:, r:, :],
                          c_batch_stride=dx.stride(0), c_row_stride=stride * Ci, c_rows_out=nq, accumulate=bool(accumulate and out is not None)))
    ops.gemm_nt_group(probs)                     # the stride phases write disjoint rows of dx: one launch
    return dx


def conv_transpose_fwd(x, w, b, *, out=None):
    """ConvTranspose1d(stride 1, no padding).  x: (B, L, Ci); w: (Ci, Co, kw) -> (B, L + kw - 1, Co).
    Same arithmetic as conv_dgrad with the roles of the channel axes swapped."""
    B, L, Ci = x.shape
    _, Co, kw = w.shape
    packed = dgrad_pack(w, 1)                    # (Ci, Co, kw) seen as a conv weight (Co'=Ci, Ci'=Co)
    Lo = L + kw - 1
    if out is None:
        out = empty(B, Lo, Co, like=x)
    A = Win.taps(x, kw, shift=0, dil=-1, rows_out=Lo)
    ops.gemm_nt(A, packed[0], b, out, c_batch_stride=out.stride(0), c_row_stride=out.stride(1), c_rows_out=Lo)
    return out


def conv_transpose_bwd(dy, x, w, dW, db, need_dx=True):
    """dy: (B, L+kw-1, Co); x: (B, L, Ci); w/dW: (Ci, Co, kw)."""
    B, L, Ci = x.shape
    _, Co, kw = w.shape
    Lo = L + kw - 1
    # dW[ci, co, k] = sum_{b,l} x[b,l,ci] * dy[b,l+k,co]
    A = Win.conv(dy, kw, rows_out=L)
    ops.gemm_tn(x.reshape(B * L, Ci), A, dW.view(Ci, -1), out_kw=kw)
    ops.colsum(dy.reshape(B * Lo, Co), db, accumulate=True)
    if not need_dx:
        return None
    # dx[b,l,ci] = sum_k sum_co dy[b,l+k,co] * w[ci,co,k]  == Conv1d with weight (Ci, Co, kw)
    return conv_fwd(dy, pack_conv_weight(w), None, kw, rows_out=L)


# ----------------------------------------------------------------------------------------------- batch norm
class BNState:
    """Per-call statistics of one BatchNorm1d (kept for the backward pass)."""
    __slots__ = ("mean", "rstd", "groups", "x", "slope")


def _eval_stats(running_mean, running_var, mean, rstd):
    """Eval-mode BatchNorm's (mean, 1 / sqrt(var + eps)) from the running statistics: buffers, so once per FrozenWeights scope."""
    def make():
        ops.bn_eval_stats(running_mean, running_var, mean, rstd)
        return mean, rstd
    return frozen(("bn_eval", running_mean.data_ptr(), running_var.data_ptr()), make)


def bn_fwd(x, gamma, beta, running_mean, running_var, nbt, *, training, groups=1, act_slope=1.0, out=None, repeats=1):
    """x: (..., C) contiguous channel-last.  Returns (y, BNState)."""
    Cc = x.shape[-1]
    x2 = x.view(-1, Cc)
    st = BNState()
    g = groups if training else 1
    st.mean, st.rstd = empty(g, Cc, like=x), empty(g, Cc, like=x)
    st.groups, st.x, st.slope = g, x, act_slope
    if training and x2.shape[0] % g == 0 and ops.bn2_supported(x2.shape[0] // g, Cc, g) and x2.is_contiguous():
        y = torch.empty_like(x) if out is None else out
        ops.bn2_train(x2, y.view(-1, Cc), g, st.mean, st.rstd, running_mean, running_var, nbt, gamma, beta, act_slope, repeats=repeats)
        return y, st
    if training and ops.bn_fused_supported(x2.shape[0], Cc, g):
        y = torch.empty_like(x) if out is None else out
        ops.bn_train_fused(x2, y.view(-1, Cc), g, st.mean, st.rstd, running_mean, running_var, nbt, gamma, beta, act_slope, repeats=repeats)
        return y, st
    if training:
        ws = torch.empty(2 * g * Cc, device=x.device, dtype=torch.float64)
        ops.bn_train_stats(x2, g, ws, st.mean, st.rstd, running_mean, running_var, nbt, repeats=repeats)
    else:
        st.mean, st.rstd = _eval_stats(running_mean, running_var, st.mean, st.rstd)
    y = torch.empty_like(x) if out is None else out
    ops.bn_apply(x2, y.view(-1, Cc), g, st.mean, st.rstd, gamma, beta, act_slope)
    return y, st


def bn_bwd(dy, st, gamma, beta, dgamma, dbeta, *, g0=0, ng=1, row0=0):
    """Backward of act(BN(x)) for `ng` consecutive statistics groups starting at group g0.  dy: (nb, ..., C) holds
    exactly those groups' batch rows; the matching rows of the taped input start at batch index row0."""
    Cc = dy.shape[-1]
    nb = dy.shape[0]
    assert nb % ng == 0
    per = nb // ng
    dy = dy.contiguous()
    dx = torch.empty_like(dy)
    rpg = dy[0].numel() // Cc * per
    x_all = st.x[row0:row0 + nb]
    if ops.bn2_supported(rpg, Cc, ng) and x_all.is_contiguous():
        # every group of the call in one pair of launches
        ops.bn2_backward(dy.view(-1, Cc), x_all.reshape(-1, Cc), dx.view(-1, Cc), ng, st.mean[g0:g0 + ng], st.rstd[g0:g0 + ng], gamma, beta, st.slope,
                         dgamma, dbeta)
        return dx
    ws = torch.empty(2 * Cc, device=dy.device, dtype=torch.float64)
    for g in range(ng):
        sl = slice(g * per, (g + 1) * per)
        x = st.x[row0 + g * per:row0 + (g + 1) * per]
        ops.bn_backward(dy[sl].reshape(-1, Cc), x.reshape(-1, Cc), dx[sl].view(-1, Cc), st.mean[g0 + g], st.rstd[g0 + g],
                        gamma, beta, st.slope, ws, dgamma, dbeta)
    return dx


# ----------------------------------------------------------------------------------------------- WavEncoder front end
class WavFrontState:
    """What the fused Conv1d(1,16,15) -> BatchNorm1d -> LeakyReLU block keeps for its backward pass: per-group statistics, the forward's
    sums (fstat) and one gate bit per output element -- not the pre-BatchNorm tensor."""
    __slots__ = ("mean", "rstd", "fstat", "gate", "groups", "slope", "stride", "pad", "T1", "words_per_clip")


def wav_front_supported(w):
    return ops.WAV_FUSED and tuple(w.shape) == (16, 1, 15)


def wav_front_fwd(audio, w, b, gamma, beta, running_mean, running_var, nbt, *, stride, pad, training, groups=1, act_slope=0.3, repeats=1):
    """audio: (B, L) view.  Returns (y (B, T1, 16), WavFrontState).  Train mode: batch statistics per group of B / groups clips."""
    B, L = audio.shape
    T1 = conv_out_len(L, 15, stride, pad)
    g = groups if training else 1
    assert B % g == 0
    per = B // g
    st = WavFrontState()
    st.mean, st.rstd = empty(g, 16, like=audio), empty(g, 16, like=audio)
    st.groups, st.slope, st.stride, st.pad, st.T1 = g, act_slope, stride, pad, T1
    st.words_per_clip = 4 * ((T1 + 15) // 16)
    st.fstat = st.gate = None
    y = empty(B, T1, 16, like=audio)
    if training:
        st.fstat = torch.empty(g, 272, device=audio.device, dtype=torch.float64)
        st.gate = torch.empty(B * st.words_per_clip, device=audio.device, dtype=torch.int64)
        for q in range(g):
            ops.wav_front_stats(audio[q * per:(q + 1) * per], w, b, stride, pad, st.mean[q], st.rstd[q], running_mean, running_var, nbt,
                                st.fstat[q], repeats=repeats)
    else:
        st.mean, st.rstd = _eval_stats(running_mean, running_var, st.mean, st.rstd)
    for q in range(g):
        sl = slice(q * per, (q + 1) * per)
        ops.wav_front_apply(audio[sl], w, b, stride, pad, st.mean[q], st.rstd[q], gamma, beta, act_slope, y[sl],
                            st.gate[q * per * st.words_per_clip:(q + 1) * per * st.words_per_clip] if training else None)
    return y, st


def wav_front_bwd(dact, st, audio, w, b, gamma, dW, db, dgamma, dbeta, *, g0=0, row0=0):
    """dact: (nb, T1, 16) gradient w.r.t. the block's output for ONE statistics group g0 whose clips start at batch index row0."""
    nb = dact.shape[0]
    wpc = st.words_per_clip
    ops.wav_front_backward(dact.contiguous(), st.gate[row0 * wpc:(row0 + nb) * wpc], audio[row0:row0 + nb], w, b, st.stride, st.pad,
                           st.mean[g0], st.rstd[g0], gamma, st.fstat[g0], st.slope, dW.view(-1) if dW is not None else None, db, dgamma, dbeta)


def wav_front_bwd_fused_supported(w2, stride2):
    return ops.WAV_FUSED_DGRAD and tuple(w2.shape) == (32, 16, 15) and stride2 == 6


def wav_front_bwd_fused(dc2, w2, st, audio, w, b, gamma, dW, db, dgamma, dbeta, *, g0=0, row0=0):
    """wav_front_bwd fed with the gradient w.r.t. the NEXT conv's output (dc2: (nb, T2, 32), w2: (32, 16, 15), stride 6): that conv's input
    gradient is formed tile by tile inside the reduction."""
    nb = dc2.shape[0]
    wpc = st.words_per_clip
    ops.wav_front_backward_fused(dc2.contiguous(), w2, st.gate[row0 * wpc:(row0 + nb) * wpc], audio[row0:row0 + nb], w, b, st.stride, st.pad,
                                 st.mean[g0], st.rstd[g0], gamma, st.fstat[g0], st.slope, dW.view(-1) if dW is not None else None, db, dgamma, dbeta)


# ----------------------------------------------------------------------------------------------- GRU stack
class GRUTape:
    __slots__ = ("x", "y", "save", "masks", "B", "T", "H", "xbound")


def gru_stack_fwd(x, P, prefix, n_layers, H, *, p_drop, training, rng=None, save=False, inject=None, tag="g", save_rows=None, drawn=None):
    """Multi-layer bidirectional GRU.  x: (B, T, Kin) contiguous.  P: name -> parameter tensor.
    Returns (y_last (B,T,2H), tape).  drawn: the (n_layers - 1, B, T, 2H) fused inter-layer dropout masks drawn by the caller beforehand."""
    B, T, _ = x.shape
    tape = GRUTape()
    tape.x, tape.y, tape.save, tape.masks = [], [], [], []
    tape.B, tape.T, tape.H = B, T, H
    tape.xbound = []                      # per layer: a bound on |layer input| known by construction (None: unknown), for the backward's fp16 x 2 scales
    cur = x
    # the inter-layer dropout rides in the recurrence kernels (csrc/gru_h64.hip, csrc/gru_cluster_x3.hip): the masks of all layers of
    # the pass come from ONE draw launch (or from the parity tests)
    fused_drop = ops.gru_fused_dropout(B, H) and training and n_layers > 1 and (p_drop > 0 or inject is not None)
    if drawn is not None:
        assert fused_drop and tuple(drawn.shape) == (n_layers - 1, B, T, 2 * H)
    if drawn is None and fused_drop and p_drop > 0 and not (inject is not None and all(f"{tag}.gru.drop{l}" in inject for l in range(n_layers - 1))):
        drawn = ops.dropout_mask(empty(n_layers - 1, B, T, 2 * H, like=x), p_drop, rng.state, rng.site(f"{tag}.gru.drop"))
    for l in range(n_layers):
        Kin = cur.shape[2]
        gi = empty(2, B, T, 3 * H, like=x)
        a_win = Win.plain(cur.view(B * T, Kin))
        # many-row projections (the stacked forward): pre-split weights let the mover-wave kernel take them (csrc/gemm_mw.hip)
        wpl = (lambda w: weight_planes(w)) if (ops.GEMM_PLANES and B * T >= 8192 and H > 64) else (lambda w: None)
        # fp16 x 2 operands: a layer above the first reads GRU outputs through an inverted dropout -- |h| < 1 by construction (a convex
        # combination of tanh values and the previous h), so |x| <= 1 / (1 - p) is a bound the row scales can take without a pass over x
        # (injected masks may hold anything: those runs measure the rows)
        bound = None
        if l > 0 and inject is None:
            bound = ops.const_rowmax(B * T, 1.0 / (1.0 - p_drop) if (training and p_drop > 0) else 1.0, x.device)
        tape.xbound.append(None if bound is None else (1.0 / (1.0 - p_drop) if (training and p_drop > 0) else 1.0))
        ops.gemm_nt_group([dict(A=a_win, W=P[f"{prefix}.weight_ih_l{l}{sfx}"], bias=P[f"{prefix}.bias_ih_l{l}{sfx}"],
                                out=gi[d].view(B * T, 3 * H), w_planes=wpl(P[f"{prefix}.weight_ih_l{l}{sfx}"]), a_rowmax=bound)
                           for d, sfx in enumerate(("", "_reverse"))])       # both directions, one launch
        y = empty(B, T, 2 * H, like=x)
        sv = empty(2, B, T, 4 * H, like=x) if save else None
        whh = (P[f"{prefix}.weight_hh_l{l}"], P[f"{prefix}.weight_hh_l{l}_reverse"])
        bhh = (P[f"{prefix}.bias_hh_l{l}"], P[f"{prefix}.bias_hh_l{l}_reverse"])
        mask = None
        name = f"{tag}.gru.drop{l}"
        if fused_drop and l < n_layers - 1:
            mask = inject[name].contiguous() if (inject is not None and name in inject) else (drawn[l] if drawn is not None else None)
        if mask is not None:
            nxt = torch.empty_like(y)
            ops.gru_forward(gi, whh, bhh, y, sv, drop_mask=mask, y_drop=nxt, save_rows=save_rows)
            tape.x.append(cur); tape.y.append(y); tape.save.append(sv); tape.masks.append(mask)
            cur = nxt
            continue
        ops.gru_forward(gi, whh, bhh, y, sv, save_rows=save_rows)
        tape.x.append(cur); tape.y.append(y); tape.save.append(sv)
        if training and l < n_layers - 1 and not fused_drop:
            if inject is not None and name in inject:
                mask = inject[name]
                cur = ops.mul(y, mask, torch.empty_like(y))
            elif p_drop > 0:
                # draw + apply in one pass; big batches: the mask is not stored, the input-gradient GEMM's epilogue regenerates it (ops.Drop)
                cur, mask = ops.dropout_apply(y, p_drop, rng.state, rng.site(name), store_mask=not (DROP_REGEN and B * T >= 8192))
            else:
                cur = y
        else:
            cur = y
        tape.masks.append(mask)
    return cur, tape


KSPLIT_DX = 2      # K pieces per direction of the first GRU layer's input gradient (0: off)
DROP_REGEN = True      # dropout masks regenerated by their consumers (ops.Drop) at big batches


class Fork:
    """Second HIP stream for work that is independent of the main dependency chain (weight-gradient GEMMs while the next
    layer's recurrence runs, the audio encoder beside the text encoder).  Works under hipGraph capture: the side stream
    joins the capture through the wait on the main stream and is joined back before the captured region ends.

        with fork:            # launches inside go to the side stream, ordered after everything queued on the main stream
            ...
        fork.join()           # main stream waits for the side stream

    Buffers read by side-stream work must stay referenced until join(): pass them to keep()."""
    _streams = {}
    on_side = False             # launches currently go to a forked stream

    def __init__(self, device, enabled=True, slot=0):
        """slot: which of the device's side streams (forks that are open at the same time take different slots)."""
        self.enabled = enabled and device.type == "cuda"
        self.side = None
        self._ctx = None
        self._keep = []
        self.dirty = False
        if self.enabled:
            key = (device.type, device.index, slot)
            if key not in Fork._streams:
                Fork._streams[key] = torch.cuda.Stream(device=device)
            self.side = Fork._streams[key]

    def keep(self, *tensors):
        self._keep.extend(tensors)

    def __enter__(self):
        if self.enabled:
            self.side.wait_stream(torch.cuda.current_stream())
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
            self.dirty = True
            Fork.on_side = True
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self._ctx.__exit__(*exc)
            self._ctx = None
            Fork.on_side = False
        return False

    def join(self):
        if self.enabled and self.dirty:
            torch.cuda.current_stream().wait_stream(self.side)
            self.dirty = False
        self._keep = []


TN_SIDE_FRACTION = float(os.environ.get("TG_TN_SIDE_FRAC", "0.48"))     # share of a GRU layer's weight-gradient rows that runs beside the next layer's recurrence (gru_stack_bwd side_split)


def gru_stack_bwd(dy, tape, P, G, prefix, n_layers, *, b0=0, nb=None, need_dx=True, param_grads=True, fork=None, defer=None, side_split=None):
    """dy: (nb, T, 2H) gradient w.r.t. the last layer's output for rows [b0, b0+nb) of the taped forward.
    Accumulates into G[...] and returns dx (nb, T, Kin0).  With `fork`, each layer's weight-gradient GEMMs run on the side
    stream while the main stream goes on with dx and the next layer's recurrence; the caller joins.  With `defer` (a list), the
    weight-gradient problems are appended to it instead of launched: the caller groups them (tn_group_deferred) -- they are off the
    dependency chain, and on the small H = 64 stack one launch per layer is mostly launch and tail.
    side_split (a Fork; the cluster-synchronised backward at H > 64): the recurrence is latency-bound and leaves 256 - 160 CUs idle, and the
    weight gradients are off the dependency chain.  The LAST batch rows of every layer's weight-gradient products (TN_SIDE_FRACTION of them)
    are therefore launched on the fork's stream right before the NEXT layer's recurrence, planned for exactly the CUs that recurrence
    leaves free (ops.tn_workgroup_cap) so that all its cluster members stay co-resident; the other rows run on the main stream as before.
    The caller joins the fork before anybody reads the gradients."""
    B, T, H = tape.B, tape.T, tape.H
    nb = B - b0 if nb is None else nb
    rows = slice(b0, b0 + nb)
    dh = empty(4 * nb * H, like=dy)
    fk = fork if fork is not None else Fork(dy.device, enabled=False)
    scaled = False               # dy already carries the dropout scale of the layer below (applied in the input-gradient GEMM's epilogue)
    side_pending = None          # (problems, CU cap, tensors to keep) of the layer above
    free_cus = ops.gru_cluster_bwd_free_cus(nb, H) if (side_split is not None and side_split.enabled and param_grads and defer is None) else 0
    nb_side = int(nb * TN_SIDE_FRACTION) if free_cus >= 64 else 0
    if nb_side * T < 1024 or (nb - nb_side) * T < 2048:          # (the persistent weight-gradient kernel wants 2 048 rows, 1 024 under a cap)
        nb_side = 0
    def launch_side_rows():
        nonlocal side_pending
        if side_pending is not None:
            probs_s, keep_s = side_pending
            side_split.keep(*keep_s)
            with side_split:                                     # ordered behind the mark: the layer above is complete
                with ops.tn_workgroup_cap(free_cus):
                    ops.gemm_tn_group(probs_s)
            side_pending = None
    # fp16 x 2 operands for the products that read dgi / dgh: the cluster recurrence leaves the magnitudes of dgi's rows and of dgi's / dgh's columns
    # behind its hand-offs (tg_gru_backward_cluster_stats), so neither the input-gradient product nor the weight gradients need a pass over them.
    # One zeroed block for all layers: [layer][gi_clipmax 2 nb | gi_colmax 2 x 3H | gh_colmax 2 x 3H]
    want_stats = (ops.gemm_h2() and ops.GRU_CLUSTER and H > 64 and H % 4 == 0 and nb * T >= 2048
                  and ops.gru_cluster_chunks(nb, H, bwd=True) is not None)
    n_rm = (2 * nb + 3) // 4 * 4
    stats_all = ops.zeros(n_layers, n_rm + 12 * H, device=dy.device) if want_stats else None
    for l in range(n_layers - 1, -1, -1):
        # The side rows are enqueued BEFORE the recurrence they run beside.  (Round 5, profiles/r5_z_tn_side_rows.txt: captured the other way
        # round -- recurrence first, an event for the side rows -- hipGraph ran the whole side branch after everything else, 5.00 ms.)
        launch_side_rows()
        dy_mask = None
        if tape.masks[l] is not None and not scaled:
            if ops.gru_fused_dropout(nb, H, bwd=True):
                dy_mask = tape.masks[l][rows].contiguous()         # multiplied in while the recurrence kernel loads dy
            else:
                dy = ops.mul(dy.contiguous(), tape.masks[l][rows].contiguous(), torch.empty_like(dy))
        scaled = False
        wt = tuple(transpose2d(P[f"{prefix}.weight_hh_l{l}{s}"]) for s in ("", "_reverse"))
        dgi, dgh = empty(2, nb, T, 3 * H, like=dy), empty(2, nb, T, 3 * H, like=dy)
        stats = None
        if stats_all is not None:
            st = stats_all[l]
            stats = (st[:2 * nb].view(2, nb), st[n_rm:n_rm + 6 * H].view(2, 3 * H), st[n_rm + 6 * H:].view(2, 3 * H))
        if not ops.gru_backward(dy.contiguous(), tape.y[l], tape.save[l], wt, dgi, dgh, dh, b0=b0, nb=nb, dy_mask=dy_mask, stats=stats):
            stats = None
        x_l = tape.x[l][rows]
        Kin = x_l.shape[2]
        y_l = tape.y[l][rows]
        x_cmax = h_cmax = None
        if stats is not None and param_grads:
            # column magnitudes of the OTHER operands: h_{t-1} is a GRU output (|h| < 1), a layer input above the first is one through an inverted
            # dropout (the bound the forward recorded); anything else is measured (the first layer's 108 columns)
            h_cmax = ops.const_rowmax(H, 1.0, dy.device)
            xb = tape.xbound[l] if getattr(tape, "xbound", None) else None
            if xb is not None:
                x_cmax = ops.const_rowmax(Kin, xb, dy.device)
            elif Kin % 4 == 0 and x_l.is_contiguous():
                x_cmax = ops.absmax_rows_cols(x_l.reshape(nb * T, Kin))[1].view(-1)
            # (else: x_cmax stays None and the layer's weight gradients keep bf16 x 3)
        if param_grads:
            def layer_probs(r0, r1):
                """the layer's four weight gradients (+ four bias gradients) over batch rows [r0, r1)"""
                n, probs = r1 - r0, []
                x_win = Win.plain(x_l[r0:r1].reshape(n * T, Kin))
                for d, sfx in enumerate(("", "_reverse")):
                    gi2, gh2 = dgi[d][r0:r1].view(n * T, 3 * H), dgh[d][r0:r1].view(n * T, 3 * H)
                    cm_i = dict(y_colmax=stats[1][d], a_colmax=x_cmax) if (stats is not None and x_cmax is not None) else {}
                    cm_h = dict(y_colmax=stats[2][d], a_colmax=h_cmax) if (stats is not None and x_cmax is not None) else {}
                    probs.append(dict(dY=gi2, A=x_win, dW=G[f"{prefix}.weight_ih_l{l}{sfx}"], dbias=G[f"{prefix}.bias_ih_l{l}{sfx}"], **cm_i))
                    # h_{t-1} of direction d is the layer output one step back (forward) / ahead (reverse), zero at the ends
                    hwin = Win.taps(y_l[r0:r1, :, d * H:(d + 1) * H], 1, shift=(1 if d else -1), dil=1, rows_out=T)
                    probs.append(dict(dY=gh2, A=hwin, dW=G[f"{prefix}.weight_hh_l{l}{sfx}"], dbias=G[f"{prefix}.bias_hh_l{l}{sfx}"], **cm_h))
                return probs
            split = 0
            if nb_side and l > 0:
                # only if the capped plan takes the side part on the persistent kernel (anything else floods the chip with short workgroups)
                with ops.tn_workgroup_cap(free_cus):
                    if ops.tn_kernel_plan(layer_probs(nb - nb_side, nb), as_launched=True) == 2:
                        split = nb_side
            fk.keep(dgi, dgh, x_l, y_l, stats_all, x_cmax, h_cmax)     # (the side stream reads them after this function's references are gone)
            with fk:
                probs = layer_probs(0, nb - split)
                if defer is not None:
                    defer.extend(probs)
                else:
                    ops.gemm_tn_group(probs)             # one launch
            if split:
                side_pending = (layer_probs(nb - split, nb), (dgi, dgh, x_l, y_l, stats_all, x_cmax, h_cmax))
        dx = None
        if need_dx or l > 0:
            dx = empty(nb * T, Kin, like=dy)
            wt_ih = [transpose2d(P[f"{prefix}.weight_ih_l{l}{sfx}"]) for sfx in ("", "_reverse")]          # [Kin][3H] each
            seg = (wt_ih[1].data_ptr() - wt_ih[0].data_ptr()) // 4
            ksplit = KSPLIT_DX if (l == 0 and nb * T >= 1024 and 48 <= Kin and Kin % 4 == 0 and (3 * H) % 4 == 0 and H > 64) else 0
            if ksplit:
                # few output columns, long reduction (the first layer: N = in_size = 108, K = 6H = 1 800: 136 tiles of 57 slabs on 256 CUs):
                # the K range of each direction cut in `ksplit` pieces, all pieces as problems of ONE launch into partial buffers, then a
                # fixed-order sum -- 2 * ksplit times the workgroups, each with a 1 / ksplit as long dependent slab chain
                parts = empty(2 * ksplit, nb * T, Kin, like=dy)
                kp = (3 * H // ksplit + 3) // 4 * 4
                probs = []
                for d in range(2):
                    a2, w2 = dgi[d].view(nb * T, 3 * H), wt_ih[d]
                    for q in range(ksplit):
                        k0, k1 = q * kp, min(3 * H, (q + 1) * kp)
                        probs.append(dict(A=Win.plain(a2[:, k0:k1]), W=w2[:, k0:k1], bias=None, out=parts[d * ksplit + q]))
                ops.gemm_nt_group(probs)
                ops.sum_parts(parts, dx)
            elif nb * T >= 1024 and Kin >= 48 and (wt_ih[1].data_ptr() - wt_ih[0].data_ptr()) % 16 == 0:
                # dx = [dgi_fwd | dgi_rev] @ [W_ih_fwd ; W_ih_rev]: ONE product over the concatenated K = 6H -- the two directions are
                # two "taps" of the A window (dgi is [2][nb*T][3H]) and two segments of the weight operand.  The inter-layer dropout's
                # backward (the layer below's mask) rides in the epilogue instead of a separate multiply pass.
                a_cat = Win(dgi, batches=1, batch_stride=0, row_stride=3 * H, rows_in=2 * nb * T, rows_out=nb * T, cw=3 * H, K=6 * H,
                            dil=nb * T)
                below = tape.masks[l - 1] if l > 0 else None
                # fp16 x 2 on the mover-wave kernel (128 x 96 tiles: 238 of them for [4352 x 600]) when the recurrence left dgi's row magnitudes:
                # the weight operand is the K-concatenated transpose of both directions' W_ih as pre-split planes (WeightPrep job 11)
                h2kw = {}
                if stats is not None and Kin >= 150 and Kin % 4 == 0:
                    h2kw = dict(w_planes=weight_planes_tcat(P[f"{prefix}.weight_ih_l{l}"], P[f"{prefix}.weight_ih_l{l}_reverse"]), a_rowmax=stats[0].view(-1), a_rowmax_rows=T)
                if below is not None and not ops.gru_fused_dropout(nb, H, bwd=True):
                    ops.gemm_nt(a_cat, wt_ih[0], None, dx, b_seg=(3 * H, seg), out_scale=below[rows].reshape(nb * T, Kin), **h2kw)
                    scaled = True
                else:
                    ops.gemm_nt(a_cat, wt_ih[0], None, dx, b_seg=(3 * H, seg), **h2kw)
            elif Kin == 8 and (3 * H) % 4 == 0:
                # eight input channels (the discriminator's first layer): one bandwidth-sized pass instead of two 13 us narrow products;
                # W_ih [3H][8] is the [K][8] operand as stored
                ops.narrow8_pair(dgi[0].view(nb * T, 3 * H), dgi[1].view(nb * T, 3 * H), P[f"{prefix}.weight_ih_l{l}"],
                                 P[f"{prefix}.weight_ih_l{l}_reverse"], dx)
            else:
                for d in range(2):
                    ops.gemm_nt(Win.plain(dgi[d].view(nb * T, 3 * H)), wt_ih[d], None, dx, accumulate=(d == 1))
        dy = dx.view(nb, T, Kin) if dx is not None else None
    if fork is None:
        fk.join()
    return dy
