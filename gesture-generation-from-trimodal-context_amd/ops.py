"""Checked Python wrappers over the C ABI: torch tensors in, raw device pointers out.

PyTorch is plumbing here (device memory, streams); every arithmetic op below runs in libtrimodal_hip.so.
Shape / dtype / bounds checks live in this layer so that a wrong call raises in Python instead of faulting on
the GPU (kernels themselves only guard their own tile edges).
"""
import ctypes as C

import contextlib
import os

import torch

from . import _lib
from ._lib import Window, call


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{name}: expected a CUDA float32 tensor, got {type(t).__name__} "
                        f"{getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
    return t


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _room(t):
    """floats addressable from t.data_ptr() to the end of its storage"""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


def _flat(t, name):
    _f32(t, name)
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


class Win:
    """Row-window view (struct tg_window) over a channel-last buffer, with a bounds check against the storage."""

    def __init__(self, t, *, batches, batch_stride, row_stride, rows_in, rows_out, cw, K, row_step=1, shift=0, dil=1):
        _f32(t, "window base")
        assert K > 0 and cw > 0 and K % cw == 0 and rows_in > 0 and rows_out > 0 and batches > 0
        max_off = (batches - 1) * batch_stride + (rows_in - 1) * row_stride + cw - 1
        if max_off >= _room(t) or batch_stride < 0 or row_stride < 0:
            raise ValueError(f"window exceeds its tensor: max offset {max_off} >= {_room(t)}")
        self.t, self.batches, self.rows_out, self.K = t, batches, rows_out, K
        self.M = batches * rows_out
        self.s = Window(t.data_ptr(), batch_stride, row_stride, rows_in, rows_out, row_step, shift, dil, cw, K)

    @staticmethod
    def plain(x):
        """2-D matrix view [M, K] (unit inner stride; rows may be strided, e.g. a column slice)."""
        assert x.dim() == 2 and x.stride(1) == 1, (x.shape, x.stride())
        M, K = x.shape
        return Win(x, batches=1, batch_stride=0, row_stride=x.stride(0), rows_in=M, rows_out=M, cw=K, K=K)

    @staticmethod
    def conv(x, kw, *, stride=1, pad=0, dil=1, rows_out=None):
        """Conv1d window over x: (B, L, C) channel-last (unit channel stride).  rows_out defaults to the conv length."""
        assert x.dim() == 3 and (x.stride(2) == 1 or x.shape[2] == 1), (x.shape, x.stride())
        B, L, Cc = x.shape
        if rows_out is None:
            rows_out = (L + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        return Win(x, batches=B, batch_stride=x.stride(0), row_stride=x.stride(1), rows_in=L, rows_out=rows_out,
                   cw=Cc, K=kw * Cc, row_step=stride, shift=-pad, dil=dil)

    @staticmethod
    def taps(x, n_taps, *, shift, dil, rows_out):
        """General tap window over x: (B, L, C): source row = r + shift + tap*dil (zero outside [0, L))."""
        assert x.dim() == 3 and (x.stride(2) == 1 or x.shape[2] == 1)
        B, L, Cc = x.shape
        return Win(x, batches=B, batch_stride=x.stride(0), row_stride=x.stride(1), rows_in=L, rows_out=rows_out,
                   cw=Cc, K=n_taps * Cc, row_step=1, shift=shift, dil=dil)


class Drop:
    """An inverted-dropout scale mask that is never stored: its consumers regenerate element i as element index0 + i of the draw
    dropout_mask(mask, p, state, site) would write (Philox keyed by the element index; `state` only advances in iter_begin, so the forward
    and backward consumers of one F.dropout see the same mask).  Accepted wherever the big-product GEMMs take `out_scale` and by
    act_mask_bwd / act_mask_bwd2; `shape` is the shape of the tensor it scales (checked like a mask tensor's)."""

    def __init__(self, state, site, p, shape, index0=0):
        assert state.dtype == torch.int64 and state.is_cuda and state.numel() >= 2 and 0.0 <= p < 1.0
        self.state, self.site, self.p, self.shape, self.index0 = state, int(site), float(p), tuple(shape), int(index0)

    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def __getitem__(self, rows):
        """Leading-dimension slice (a contiguous range of batch rows), as mask[rows] of the stored form."""
        assert isinstance(rows, slice) and rows.step in (None, 1)
        b0, b1, _ = rows.indices(self.shape[0])
        inner = self.numel() // self.shape[0]
        return Drop(self.state, self.site, self.p, (b1 - b0,) + self.shape[1:], self.index0 + b0 * inner)

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        n = self.numel()
        if -1 in shape:
            known = 1
            for s in shape:
                known *= s if s != -1 else 1
            shape = tuple(n // known if s == -1 else s for s in shape)
        assert Drop(self.state, self.site, self.p, shape).numel() == n
        return Drop(self.state, self.site, self.p, shape, self.index0)

    def contiguous(self):
        return self

    def materialize(self):
        """The stored form (tests): the draw's elements index0 .. index0 + numel."""
        full = torch.empty(self.index0 + self.numel(), device=self.state.device)
        dropout_mask(full, self.p, self.state, self.site)
        return full[self.index0:].view(self.shape).clone()


def _nt_problem(A: Win, W, bias, out, *, act_slope=1.0, accumulate=False, c_batch_stride=None, c_row_stride=None, c_rows_out=None, M=None,
                b_seg=None, out_scale=None, gate=None, res=None, out2=None, res_slope=0.0, w_planes=None, w_row0=0, a_row_scale=None, a_rowmax=None, out_rowmax=None, out2_rowmax=None, a_rowmax_rows=1):
    """Checked tg_gemm_nt_problem.  b_seg = (seg_k, seg_stride_floats): K-concatenated weights, W is the first [N, seg_k] segment and
    segment s starts seg_stride_floats * s floats after it (the caller keeps every segment alive).
    Epilogue extensions (big-product path only, nt_ext_supported): gate -- keep the result where gate > 0, zero elsewhere; res + out2 --
    second output out2 = leaky_relu(out + res, res_slope).  All addressed exactly like `out`."""
    _f32(W, "W"); _f32(out, "out")
    seg_k = A.K if b_seg is None else int(b_seg[0])
    assert W.dim() == 2 and W.stride(1) == 1 and W.shape[1] == seg_k and A.K % seg_k == 0, (W.shape, A.K, seg_k)
    N = W.shape[0]
    M = A.M if M is None else M
    if bias is not None:
        _f32(bias, "bias"); assert bias.numel() == N and bias.is_contiguous()
    if c_row_stride is None:
        assert out.dim() == 2 and out.stride(1) == 1 and tuple(out.shape) == (M, N), (out.shape, M, N)
        c_batch_stride, c_row_stride, c_rows_out = 0, out.stride(0), M
    nb = (M + c_rows_out - 1) // c_rows_out
    max_off = (nb - 1) * c_batch_stride + (c_rows_out - 1) * c_row_stride + N - 1
    if max_off >= _room(out):
        raise ValueError(f"gemm_nt: output exceeds its tensor ({max_off} >= {_room(out)})")
    if (W.shape[0] - 1) * W.stride(0) + seg_k - 1 >= _room(W):
        raise ValueError("gemm_nt: W exceeds its tensor")
    q = _lib.NtProblem()
    q.A = A.s
    q.Bw, q.ldb = W.data_ptr(), W.stride(0)
    q.b_seg_k, q.b_seg_stride = (0, 0) if b_seg is None else (seg_k, int(b_seg[1]))
    q.bias = bias.data_ptr() if bias is not None else None
    q.C, q.c_batch_stride, q.c_row_stride, q.c_rows_out = out.data_ptr(), c_batch_stride, c_row_stride, c_rows_out
    q.M, q.N, q.act_slope, q.accumulate = M, N, float(act_slope), int(bool(accumulate))
    if w_planes is not None:                   # Planes that hold W's rows from w_row0 on (split3_planes / layers.weight_planes): mover-wave kernel
        assert w_planes.cw == A.K and 0 <= w_row0 and w_row0 + N <= w_planes.rows and w_planes.t.is_cuda, (w_planes.rows, N, w_planes.cw, A.K)
        q.b_planes, q.b_plane_stride, q.b_rows, q.b_row0 = w_planes.t.data_ptr(), w_planes.plane_stride, w_planes.rows, int(w_row0)
        if w_planes.kind == "h2":              # fp16 x 2 planes: the product rows' power-of-two scales ride along (h2_row_scales unless the caller has them)
            assert w_row0 % 4 == 0, w_row0
            q.b_planes_kind, q.b_inv_scale = 1, w_planes.inv.data_ptr()
            if a_rowmax is not None:           # the source rows' magnitudes (their producer's out_rowmax, win_row_absmax, const_rowmax): one or two taps
                _f32(a_rowmax, "a_rowmax"); assert a_rowmax_rows >= 1 and a_rowmax.is_contiguous() and A.K <= 2 * A.s.cw
                assert a_rowmax.numel() * a_rowmax_rows >= A.batches * A.s.rows_in, (a_rowmax.numel(), a_rowmax_rows, A.batches, A.s.rows_in)
                q.a_rowmax, q.a_rowmax_rows = a_rowmax.data_ptr(), int(a_rowmax_rows)
            else:
                if a_row_scale is None:
                    a_row_scale = h2_row_scales(A, M)
                _f32(a_row_scale, "a_row_scale"); assert a_row_scale.numel() >= M and a_row_scale.is_contiguous()
                q.a_row_scale = a_row_scale.data_ptr()
            q._keep = (a_row_scale, a_rowmax)
    if isinstance(out_scale, Drop):            # the dropout scale regenerated in the epilogue: `out` must be the contiguous tensor the mask was drawn for
        assert out_scale.numel() == M * N and out_scale.index0 % 4 == 0 and N % 4 == 0, (out_scale.shape, M, N)
        assert c_row_stride == N and (c_rows_out >= M or c_batch_stride == c_rows_out * c_row_stride), "regenerated dropout needs a contiguous output"
        q.drop_state, q.drop_site, q.drop_index0, q.drop_p = out_scale.state.data_ptr(), out_scale.site, out_scale.index0, out_scale.p
    elif out_scale is not None:                # element-wise multiplier applied after the activation, addressed exactly like `out`
        _f32(out_scale, "out_scale")
        assert out_scale.shape == out.shape and out_scale.stride() == out.stride(), (out_scale.shape, out.shape)
        q.out_scale = out_scale.data_ptr()
    for name, t in (("gate", gate), ("res", res), ("out2", out2)):
        if t is not None:
            _f32(t, name)
            assert t.shape == out.shape and t.stride() == out.stride(), (name, t.shape, out.shape)
    assert (res is None) == (out2 is None), "res and out2 go together"
    if gate is not None:
        q.gate = gate.data_ptr()
    if res is not None:
        q.res, q.C2, q.res_slope = res.data_ptr(), out2.data_ptr(), float(res_slope)
    for name, t in (("c_rowmax", out_rowmax), ("c2_rowmax", out2_rowmax)):     # M floats, zeroed by the caller once per pass (mover-wave kernel only)
        if t is not None:
            _f32(t, name); assert t.numel() >= M and t.is_contiguous() and (name == "c_rowmax" or out2 is not None)
            setattr(q, name, t.data_ptr())
    return q


def nt_ext_supported(A: Win, W, out, **kw):
    """True when gemm_nt(A, W, ..., out) would run on a kernel that implements the gate / res / out2 epilogue extensions."""
    if not out.is_cuda:
        return False
    q = _nt_problem(A, W, None, out, **kw)
    return bool(_lib.load().tg_gemm_nt_ext_supported(C.byref(q)))


def set_nt_mover_waves(on):
    """True / False: force the mover-wave kernel on / off; None: back to the default (on)."""
    call("tg_set_nt_mover_waves", -1 if on is None else int(bool(on)))


def nt_kernel_plan(problems):
    """(plan, tile_m, tile_n) for a group given as gemm_nt_group's list of dicts: plan 0 = f32-MFMA / narrow / small kernels, 1 = bf16 x 3
    staged slabs (gemm_split.hip), 2 = bf16 x 3 mover waves (gemm_mw.hip); tiles are 0 unless plan == 2."""
    qs = [_nt_problem(**p) for p in problems]
    arr = (_lib.NtProblem * len(qs))(*qs)
    tm, tn = C.c_int32(0), C.c_int32(0)
    plan = int(_lib.load().tg_gemm_nt_kernel_plan(arr, len(qs), C.byref(tm), C.byref(tn)))
    return plan, tm.value, tn.value


def gemm_nt(A: Win, W, bias, out, **kw):
    """out(m, :) = act(A(m, :) @ W^T + bias) [+ out].  W: [N, K] (row stride may exceed K).
    out: 2-D view [M, N] (unit inner stride) unless explicit C addressing is given."""
    q = _nt_problem(A, W, bias, out, **kw)
    call("tg_gemm_nt_group", C.byref(q), 1, _stream())
    return out


def _share_row_scales(problems):
    """fp16 x 2 problems of a group that read the SAME window (both GRU directions' projections) share one row-scale pre-pass."""
    done, out = {}, []
    for p in problems:
        pl = p.get("w_planes")
        if pl is not None and pl.kind == "h2" and p.get("a_row_scale") is None and p.get("a_rowmax") is None:
            A = p["A"]
            key = (bytes(A.s), p.get("M", A.M))
            if key not in done:
                done[key] = h2_row_scales(A, p.get("M", A.M))
            p = dict(p, a_row_scale=done[key])
        out.append(p)
    return out


_CONST_ROWMAX = {}


def const_rowmax(n, bound, device):
    """n floats all equal to `bound`: the a_rowmax of an activation whose magnitude is bounded by construction (a GRU layer's output through an
    inverted dropout: |h| < 1, so |x| <= 1 / (1 - p)) -- no pass over the tensor.  Cached: the tensor is never written again (graph-safe)."""
    key = (int(n), float(bound), str(device))
    t = _CONST_ROWMAX.get(key)
    if t is None:
        t = _CONST_ROWMAX[key] = torch.full((int(n),), float(bound), device=device, dtype=torch.float32)
    return t


def gemm_nt_group(problems):
    """Several independent products in ONE launch.  problems: list of dicts with the arguments of gemm_nt (A, W, bias, out, ...).
    They must fall into one kernel family (all 'big', i.e. M >= 1024, N >= 48, K >= 64 -- or all narrow / all small); outputs must
    not overlap."""
    problems = _share_row_scales(problems)
    qs = [_nt_problem(**p) for p in problems]
    fam = [_lib.load().tg_gemm_nt_family(C.byref(q)) for q in qs]
    i = 0
    while i < len(qs):                      # runs of equal kernel family, at most MAX_GROUP each, in the caller's order
        j = i
        while j < len(qs) and fam[j] == fam[i] and j - i < _lib.MAX_GROUP:
            j += 1
        arr = (_lib.NtProblem * (j - i))(*qs[i:j])
        call("tg_gemm_nt_group", arr, j - i, _stream())
        i = j


# ------------------------------------------------------------------------------------------------- pre-split (bf16 x 3 planes) weights
# csrc/planes.hip: weight matrices split ONCE per optimiser step (layers.WeightPrep) into hi / mid / lo bf16 planes; on the many-row products
# of the stacked forward the mover waves of gemm_mw.hip fetch them global -> LDS by DMA while the matrix waves multiply.
GEMM_PLANES = True
# Round 6: the planes hold the fp16 x 2 split (two fp16 planes of every row scaled by its own power of two: three matrix instructions per
# product instead of six, csrc/common.hpp); TG_GEMM_H2=0 keeps bf16 x 3 planes (A/B timing).  Never in the bf16 tier (set_math_mode('bf16')).
GEMM_H2 = os.environ.get("TG_GEMM_H2", "1") != "0"


def gemm_h2():
    return GEMM_H2 and _lib.load().tg_get_math_mode() == 0


def split_planes(x2d):
    """Planes of the active operand format (split2h_planes / split3_planes)."""
    return split2h_planes(x2d) if gemm_h2() else split3_planes(x2d)



class Planes:
    """Pre-split planes of an fp32 matrix [rows][cw], each plane slab-tiled [cwp / 32][rows + 1][32] (cwp = cw rounded up to 32) with an
    all-zero row `rows` in every slab (include/trimodal_hip.h).  kind 'x3': three bf16 planes, x = hi + mid + lo exactly, `t` is
    [3][rows + 1][cwp] bf16.  kind 'h2': two fp16 planes of the rows scaled by their own power of two (tg_split2h_planes), `t` is
    [2][rows + 1][cwp] fp16 and `inv` the rows' inverse scales (rows + 1 floats).  Index `t` through element_view(), never as a row-major
    matrix."""
    __slots__ = ("t", "rows", "cw", "cwp", "kind", "inv")

    def __init__(self, t, rows, cw, cwp, kind="x3", inv=None):
        self.t, self.rows, self.cw, self.cwp, self.kind, self.inv = t, rows, cw, cwp, kind, inv

    @property
    def plane_stride(self):
        return (self.rows + 1) * self.cwp

    def element_view(self):
        """[3][rows + 1][cwp] view in matrix order (a permuted view of the tiled memory, for tests)."""
        n = self.t.shape[0]
        return self.t.view(n, self.cwp // 32, self.rows + 1, 32).permute(0, 2, 1, 3).reshape(n, self.rows + 1, self.cwp)


def planes_cwp(cw):
    return (cw + 31) // 32 * 32


def split3_planes(x2d, out=None):
    """fp32 [rows][cw] view (unit inner stride) -> Planes."""
    _f32(x2d, "x"); assert x2d.dim() == 2 and x2d.stride(1) == 1
    rows, cw = x2d.shape
    if (rows - 1) * x2d.stride(0) + cw - 1 >= _room(x2d):
        raise ValueError("split3_planes: x exceeds its tensor")
    cwp = planes_cwp(cw)
    t = torch.empty(3, rows + 1, cwp, device=x2d.device, dtype=torch.bfloat16) if out is None else out
    assert tuple(t.shape) == (3, rows + 1, cwp) and t.is_contiguous() and t.dtype == torch.bfloat16
    call("tg_split3_planes", _p(x2d), x2d.stride(0), rows, cw, C.c_void_p(t.data_ptr()), cwp, (rows + 1) * cwp, _stream())
    return Planes(t, rows, cw, cwp)


def h2_planes_alloc(rows, cw, device):
    """Storage of an fp16 x 2 plane buffer: ONE fp16 tensor [2 planes | inverse scales as fp32], so that a batched refresh addresses it by one pointer."""
    cwp = planes_cwp(cw)
    n16 = 2 * (rows + 1) * cwp
    buf = torch.empty(n16 + 2 * ((rows + 1 + 3) // 4 * 4), device=device, dtype=torch.float16)
    return buf, buf[:n16].view(2, rows + 1, cwp), buf[n16:].view(torch.float32)


def split2h_planes(x2d, buf=None):
    """fp32 [rows][cw] view (unit inner stride) -> fp16 x 2 Planes (hi / lo of every row scaled by its own power of two + inverse scales)."""
    _f32(x2d, "x"); assert x2d.dim() == 2 and x2d.stride(1) == 1
    rows, cw = x2d.shape
    if (rows - 1) * x2d.stride(0) + cw - 1 >= _room(x2d):
        raise ValueError("split2h_planes: x exceeds its tensor")
    cwp = planes_cwp(cw)
    if buf is None:
        buf = h2_planes_alloc(rows, cw, x2d.device)[0]
    n16 = 2 * (rows + 1) * cwp
    t, inv = buf[:n16].view(2, rows + 1, cwp), buf[n16:].view(torch.float32)
    call("tg_split2h_planes", _p(x2d), x2d.stride(0), rows, cw, C.c_void_p(t.data_ptr()), cwp, (rows + 1) * cwp, C.c_void_p(inv.data_ptr()), _stream())
    pl = Planes(t, rows, cw, cwp, "h2", inv)
    return pl


def split2h_planes_tcat(w0, w1, buf=None):
    """fp16 x 2 Planes of [w0^T | w1^T] ([cols][2 rows]) for two contiguous [rows][cols] matrices (tg_split2h_planes_tcat)."""
    _flat(w0, "w0"); _flat(w1, "w1"); assert w0.dim() == 2 and w0.shape == w1.shape
    rows, cols = w0.shape
    cwp = planes_cwp(2 * rows)
    if buf is None:
        buf = h2_planes_alloc(cols, 2 * rows, w0.device)[0]
    n16 = 2 * (cols + 1) * cwp
    t, inv = buf[:n16].view(2, cols + 1, cwp), buf[n16:].view(torch.float32)
    call("tg_split2h_planes_tcat", _p(w0), _p(w1), rows, cols, C.c_void_p(t.data_ptr()), cwp, (cols + 1) * cwp, C.c_void_p(inv.data_ptr()), _stream())
    return Planes(t, cols, 2 * rows, cwp, "h2", inv)


def win_row_absmax(A: Win, out=None):
    """Largest magnitude of every source row of the window's tensor: [batches * rows_in] floats."""
    n = A.batches * A.s.rows_in
    out = torch.empty(n, device=A.t.device, dtype=torch.float32) if out is None else out
    call("tg_win_row_absmax", C.byref(A.s), A.batches, C.c_void_p(out.data_ptr()), _stream())
    return out


def h2_row_scales(A: Win, M=None, src_rowmax=None, out=None):
    """Power-of-two scale of every product row of the window (tg_gemm_nt_problem.a_row_scale): [M] floats."""
    M = A.M if M is None else M
    out = torch.empty(M, device=A.t.device, dtype=torch.float32) if out is None else out
    call("tg_h2_row_scales", C.byref(A.s), M, C.c_void_p(src_rowmax.data_ptr()) if src_rowmax is not None else None, C.c_void_p(out.data_ptr()), _stream())
    return out


def zero_(t):
    """In-place zero fill by a kernel of the library (never a memset: see csrc/common.hpp zero_async)."""
    if not t.is_contiguous() or (t.numel() * t.element_size()) % 4:
        raise ValueError("zero_: needs a contiguous tensor of a multiple of 4 bytes")
    if t.numel():
        call("tg_zero", C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), _stream())
    return t


def zeros(*shape, device, dtype=torch.float32):
    return zero_(torch.empty(*shape, device=device, dtype=dtype))


def zeros_like(t):
    return zero_(torch.empty(t.shape, device=t.device, dtype=t.dtype))


def set_math_mode(mode):
    """'f32' (fp32-accurate, default) or 'bf16' (bf16 operands, fp32 accumulate) for the big forward / input-gradient GEMMs."""
    m = {"f32": 0, "fp32": 0, 0: 0, "bf16": 1, 1: 1}[mode]
    if _lib.load().tg_set_math_mode(m) != 0:
        raise RuntimeError(_lib.load().tg_last_error().decode())


def get_math_mode():
    return ("f32", "bf16")[_lib.load().tg_get_math_mode()]


TN_TWO_PASS_ROWS = 32768     # reductions at least this long combine their partials in fp64 (deterministic) instead of atomics


def set_deterministic(on):
    """tg_set_deterministic: fixed-order combines on the GAN training iteration's path (no float atomics there) -- two runs from the same state
    are bit-identical.  The autoencoder trainer and the evaluation helpers (ae_loss, l1_mean, the small single-launch BatchNorm kernels) are
    outside its scope (include/trimodal_hip.h).  Here:
    every weight-gradient product takes the two-pass workspace and its bias gradient goes through colsum; the engines take the generic
    forms of the two fused backward kernels that combine by atomics (speaker_bwd_supported, engine.DiscriminatorEngine.backward)."""
    call("tg_set_deterministic", int(bool(on)))


def deterministic():
    return bool(_lib.load().tg_get_deterministic())


import os as _os
TN_MW_WS = _os.environ.get("TG_TN_MW_WS", "1") != "0"              # mover-wave weight gradients combine their row splits through a workspace + fixed-order second pass (not float atomics)


def absmax_rows_cols(x2d, *, groups=1, want_rows=False, want_cols=True, rowmax=None, colmax=None):
    """One pass over x2d [M][C] (unit inner stride): (rowmax [M] or None, colmax [groups][C] or None) -- the largest magnitudes the fp16 x 2
    products scale their operands by (rows for gemm_nt's a_rowmax, columns for gemm_tn's y_colmax / a_colmax)."""
    _f32(x2d, "x"); assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.shape[0] % groups == 0
    M, Cc = x2d.shape
    if (M - 1) * x2d.stride(0) + Cc - 1 >= _room(x2d):
        raise ValueError("absmax_rows_cols: x exceeds its tensor")
    if want_rows and rowmax is None:
        rowmax = torch.empty(M, device=x2d.device, dtype=torch.float32)
    if want_cols and colmax is None:
        colmax = torch.empty(groups, Cc, device=x2d.device, dtype=torch.float32)
    call("tg_absmax_rows_cols", _p(x2d), x2d.stride(0), M, Cc, groups, _p(rowmax), _p(colmax), _stream())
    return rowmax, colmax


def _tn_problem(dY, A: Win, dW, *, out_kw=0, dbias=None, keep=None, force_ws=False, y_colmax=None, a_colmax=None):
    _f32(dY, "dY"); _f32(dW, "dW")
    assert dY.dim() == 2 and dY.stride(1) == 1 and dY.shape[0] == A.M, (dY.shape, A.M)
    M, N = dY.shape
    assert dW.is_contiguous() and dW.numel() == N * A.K and dW.shape[0] == N, (dW.shape, N, A.K)
    if dbias is not None:
        _flat(dbias, "dbias"); assert dbias.numel() == N
    if (M - 1) * dY.stride(0) + N - 1 >= _room(dY):
        raise ValueError("gemm_tn: dY exceeds its tensor")
    ws, nws = None, 0
    if M >= TN_TWO_PASS_ROWS or (out_kw > 0 and M >= 1024 and N * A.K >= 8192) or deterministic() or force_ws:
        # two-pass (partial tiles + fp64 combine): long reductions for accuracy, and every conv-layout output (out_kw > 0) of >= 8 K entries
        # (the discriminator's 16 x 81 ... 8 x 24 conv gradients are a few hundred atomics: the combine launch cost more than it saved):
        # the permuted (Co, Ci, kw) scatter makes the one-pass float atomics uncoalesced (measured 224 -> 39 us on the audio
        # conv3 weight gradient), the combine kernel writes that layout from contiguous partials instead
        nws = _lib.load().tg_gemm_tn_ws_floats(M, N, A.K)
        ws = torch.empty(nws, device=dW.device, dtype=torch.float32)
        if keep is not None:
            keep.append(ws)
    q = _lib.TnProblem()
    q.dY, q.ldy, q.A, q.dW, q.ldw = dY.data_ptr(), dY.stride(0), A.s, dW.data_ptr(), A.K
    q.M, q.N, q.out_kw = M, N, int(out_kw)
    q.dbias = dbias.data_ptr() if dbias is not None else None
    q.ws, q.ws_floats = (ws.data_ptr() if ws is not None else None), nws
    if y_colmax is not None or a_colmax is not None:       # fp16 x 2 on the mover-wave kernel: the columns' magnitudes (both or neither)
        _f32(y_colmax, "y_colmax"); _f32(a_colmax, "a_colmax")
        assert y_colmax.numel() >= N and a_colmax.numel() >= A.s.cw and y_colmax.is_contiguous() and a_colmax.is_contiguous()
        q.y_colmax, q.a_colmax = y_colmax.data_ptr(), a_colmax.data_ptr()
        if keep is not None:
            keep.extend((y_colmax, a_colmax))
    return q, ws


def _det_bias(problems):
    """Deterministic mode: the kernels combine bias gradients by float atomics, so they are taken out of the products and summed by
    colsum (one workgroup per 64 columns, fixed order)."""
    if not deterministic():
        return problems
    out = []
    for p in problems:
        if p.get("dbias") is not None:
            colsum(p["dY"], p["dbias"], accumulate=True)
            p = dict(p, dbias=None)
        out.append(p)
    return out


def gemm_tn(dY, A: Win, dW, *, out_kw=0, dbias=None):
    """dW[n, perm(k)] += sum_m dY[m, n] * A(m, k); dbias[n] += sum_m dY[m, n] when given.
    dY: 2-D view [M, N]; dW: contiguous, N rows of K floats."""
    gemm_tn_group([dict(dY=dY, A=A, dW=dW, out_kw=out_kw, dbias=dbias)])
    return dW


def gemm_tn_group(problems):
    """Several independent weight gradients in ONE launch.  problems: list of dicts with the arguments of gemm_tn."""
    assert 1 <= len(problems) <= _lib.MAX_GROUP
    keep = []
    if deterministic() and TN_MW_WS and any(p.get("dbias") is not None for p in problems):
        # deterministic mode: a group the mover-wave kernel takes WITH workspaces keeps its bias gradients in the product (column K of the
        # tile, combined by the fixed-order second pass like every other column): no separate column-sum launch (50 of them, 23 us each,
        # were the largest single cost of the mode)
        qs = [_tn_problem(keep=keep, **p)[0] for p in problems]
        arr = (_lib.TnProblem * len(qs))(*qs)
        if all(q.ws for q in qs) and int(_lib.load().tg_gemm_tn_kernel_plan(arr, len(qs))) == 2:
            call("tg_gemm_tn_group", arr, len(problems), _stream())
            return
        keep = []
    problems = _det_bias(problems)
    qs = [_tn_problem(keep=keep, **p)[0] for p in problems]
    arr = (_lib.TnProblem * len(qs))(*qs)
    if all(q.M >= 1024 and q.N >= 150 for q in qs) and (TN_MW_WS or gemm_h2()):
        plan = int(_lib.load().tg_gemm_tn_kernel_plan(arr, len(qs)))
        if plan == 2 and gemm_h2() and TN_AUTO_COLMAX:
            # the mover-wave kernel with fp16 x 2 operands: column magnitudes of every operand that does not bring them
            problems = _auto_colmax(problems)
        if plan == 2 and TN_MW_WS and not all(q.ws for q in qs):
            # a group the mover-wave kernel takes (csrc/gemm_tn_mw.hip): give every problem the workspace, so that its row splits are combined by
            # the fixed-order second pass instead of float atomics (28 of 116 us of a GRU layer's launch, and the one order-dependent sum left
            # on the default path's big weight gradients)
            qs = [_tn_problem(keep=keep, force_ws=True, **p)[0] for p in problems]
            arr = (_lib.TnProblem * len(qs))(*qs)
        elif plan == 2:
            qs = [_tn_problem(keep=keep, **p)[0] for p in problems]
            arr = (_lib.TnProblem * len(qs))(*qs)
    call("tg_gemm_tn_group", arr, len(problems), _stream())


# Weight-gradient groups whose problems bring no column magnitudes: False (default) = they stay on bf16 x 3 -- a measuring pass over each operand
# costs more than the fp16 x 2 kernel saves (profiles/r6_g_h2_tn_probe.txt: 17 us per [4352 x 900] operand against 12 us per group); True = measure
# (tests, probes).  The training iteration supplies them from the kernels that write the operands (layers.gru_stack_bwd).
TN_AUTO_COLMAX = False


def _auto_colmax(problems):
    """y_colmax / a_colmax for the problems of a mover-wave weight-gradient group that lack them: one absmax pass per distinct operand tensor
    (operands shared inside the group -- the layer input of both directions' W_ih gradients -- are measured once).  A window whose batches are
    not evenly strided keeps the problem (and so the group) on bf16 x 3."""
    done, out = {}, []

    def colmax_of(t2d):
        key = (t2d.data_ptr(), tuple(t2d.shape), t2d.stride(0))
        if key not in done:
            done[key] = absmax_rows_cols(t2d)[1].view(-1)
        return done[key]

    for p in problems:
        if p.get("y_colmax") is None:
            A, dY = p["A"], p["dY"]
            w = A.s
            if not (w.batch_stride == w.rows_in * w.row_stride or A.batches == 1) or w.cw % 4 or w.row_stride % 4 or dY.shape[1] % 4 or dY.stride(0) % 4 \
                    or w.cw > 2048 or dY.shape[1] > 2048 or A.t.data_ptr() % 16 or dY.data_ptr() % 16:
                return problems
            src = torch.as_strided(A.t, (A.batches * w.rows_in, w.cw), (w.row_stride, 1))
            p = dict(p, y_colmax=colmax_of(dY), a_colmax=colmax_of(src))
        out.append(p)
    return out


@contextlib.contextmanager
def tn_workgroup_cap(n):
    """Weight-gradient launches inside are planned for, and occupy, at most n CUs (tg_set_tn_workgroup_cap): for launches on a second stream
    beside a kernel that needs the other CUs to itself."""
    call("tg_set_tn_workgroup_cap", int(n))
    try:
        yield
    finally:
        call("tg_set_tn_workgroup_cap", 0)


def gru_cluster_bwd_free_cus(nb, H):
    """CUs the cluster-synchronised GRU backward leaves free (0: it is not the kernel that would run, or it runs in several launches):
    2 directions x ceil(nb / 16) batch tiles x ceil(H / 32) members, one workgroup per CU (csrc/gru_cluster.hip cluster_plan_bwd)."""
    if not (GRU_CLUSTER and H > 64) or gru_cluster_chunks(nb, H, bwd=True) != [(0, nb)]:
        return 0
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return max(0, cus - 2 * (-(-nb // 16)) * (-(-H // 32)))


def tn_kernel_plan(problems, as_launched=False):
    """0 = f32-MFMA tiles, 1 = bf16 x 3 staged slabs (gemm_split.hip), 2 = mover waves (gemm_tn_mw.hip) for a group given as gemm_tn_group's
    list of dicts.  as_launched: plan the problems as gemm_tn_group would launch them -- with the workspaces of the fixed-order combine
    (TN_MW_WS) when the group qualifies -- so that a caller who depends on the answer (a capped launch beside a cluster recurrence) probes
    exactly what will run."""
    keep = []
    arr = (_lib.TnProblem * len(problems))(*[_tn_problem(keep=keep, **p)[0] for p in problems])
    plan = int(_lib.load().tg_gemm_tn_kernel_plan(arr, len(problems)))
    if as_launched and plan == 2 and TN_MW_WS and not all(q.ws for q in arr):
        arr = (_lib.TnProblem * len(problems))(*[_tn_problem(keep=keep, force_ws=True, **p)[0] for p in problems])
        plan = int(_lib.load().tg_gemm_tn_kernel_plan(arr, len(problems)))
    return plan


def colsum(X, out, *, accumulate=True):
    _f32(X, "X"); _flat(out, "out")
    assert X.dim() == 2 and X.stride(1) == 1 and out.numel() == X.shape[1]
    if (X.shape[0] - 1) * X.stride(0) + X.shape[1] - 1 >= _room(X):
        raise ValueError("colsum: X exceeds its tensor")
    call("tg_colsum", _p(X), X.stride(0), X.shape[0], X.shape[1], _p(out), int(accumulate), _stream())
    return out


# ------------------------------------------------------------------------------------------------- GRU
# persistent cluster-synchronised recurrence (csrc/gru_cluster.hip + gru_cluster_x3.hip) where it fits (tests set GRU_CLUSTER = False to
# reach the per-step launches, the path of batches with more than 256 workgroups).  One workspace per (device, B, H): flag words + exchange buffer; its first word is the timeout marker.
GRU_CLUSTER = True
_gru_ws = {}


def _gru_cluster_ws(dev, B, H, bwd=False):
    key = (dev, B, H, bwd)
    ws = _gru_ws.get(key)
    if ws is None:
        lib = _lib.load()
        nbytes = lib.tg_gru_cluster_bwd_ws_bytes(B, H) if bwd else lib.tg_gru_cluster_ws_bytes(B, H)
        ws = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=dev)
        _gru_ws[key] = ws
    return ws


_gru_vec_ws_cache = {}
GRU_VEC = os.environ.get("TG_GRU_VEC", "1") != "0"      # the few-row inference recurrence (csrc/gru_vec.hip) for B <= 4 forwards without saved gates


def _gru_vec_fill(ws):
    hdr = _lib.load().tg_gru_vec_ws_header_bytes() // 4
    ws.fill_(-1)                 # every exchange word = the sentinel (all ones)
    ws[:hdr].zero_()             # timeout block and launch counter
    return ws


def _gru_vec_ws(dev, H):
    """One workspace per (device, H), like the cluster kernels': launches that share it must be stream-ordered (two decoders replaying on
    different streams at the same time would need one each)."""
    key = (dev, H)
    ws = _gru_vec_ws_cache.get(key)
    if ws is None:
        ws = _gru_vec_fill(torch.empty((_lib.load().tg_gru_vec_ws_bytes(H) + 3) // 4, dtype=torch.int32, device=dev))
        _gru_vec_ws_cache[key] = ws
    return ws


def gru_vec_takes(B, H, save, drop_mask):
    """Will gru_forward run the few-row inference kernel for this call?"""
    return bool(GRU_VEC and H > 64 and save is None and drop_mask is None and _lib.load().tg_gru_vec_supported(int(B), int(H)))


_cluster_caps = {}


def gru_cluster_chunks(B, H, bwd=False):
    """Row ranges [(row0, rows)] in which the cluster kernels walk a batch of B rows: one launch while it fits the chip (forward: 384 rows at
    H = 300 -- 2 directions x 12 tiles of 32 rows x 10 workgroups; backward: 192), otherwise equal chunks of whole 32-row tiles run back to
    back on ONE workspace (its flag words are numbered by generation, so a launch can follow another without a zero fill).  The recurrence
    of a batch row never needs another row, so chunking changes nothing but the launch count: --batch 256 (768 stacked rows) stays on the
    persistent kernels instead of falling to one launch per time step.  None when no chunk size is supported (H > 320, no device)."""
    lib = _lib.load()
    sup = lib.tg_gru_cluster_bwd_supported if bwd else lib.tg_gru_cluster_supported
    if sup(int(B), int(H)):
        return [(0, B)]
    key = (int(H), bool(bwd), torch.cuda.current_device())
    if key not in _cluster_caps:
        _cluster_caps[key] = max([c for c in range(32, 1025, 32) if sup(c, int(H))], default=0)
    cap = _cluster_caps[key]
    if cap == 0:
        return None
    n = -(-B // cap)
    per = -(-B // (32 * n)) * 32
    return [(r, min(per, B - r)) for r in range(0, B, per)]


_dpre_ws = {}             # (device, Bs) -> zero-initialised workspace of the fused discriminator front end (timeout word first)
D_PRECONV_FUSED = True            # tests compare with the unfused chain


def d_preconv_fwd_supported(Bs, groups):
    return D_PRECONV_FUSED and bool(_lib.load().tg_d_preconv_fwd_supported(int(Bs), int(groups)))


def d_preconv_fwd(poses, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, rm1, rv1, nbt1, rm2, rv2, nbt2, groups, eps=1e-5, momentum=0.1):
    """ConvDiscriminator.pre_conv, train-mode forward, one launch -> dict(c1, y1, c2, y2, c3, mean1, rstd1, mean2, rstd2) (tg_d_preconv_fwd)."""
    _flat(poses, "poses"); Bs = poses.shape[0]
    assert tuple(poses.shape[1:]) == (34, 27) and tuple(w1.shape) == (16, 27, 3) and tuple(w2.shape) == (8, 16, 3) and tuple(w3.shape) == (8, 8, 3)
    for t_ in (w1, b1, g1, be1, w2, b2, g2, be2, w3, b3):
        _flat(t_, "parameter")
    dev = poses.device
    key = (dev.type, dev.index, Bs)
    if key not in _dpre_ws:
        _dpre_ws[key] = torch.zeros(_lib.load().tg_d_preconv_ws_bytes(Bs) // 4, dtype=torch.int32, device=dev)
    ws = _dpre_ws[key]
    e = lambda *shape: torch.empty(*shape, device=dev)
    o = dict(c1=e(Bs, 32, 16), y1=e(Bs, 32, 16), c2=e(Bs, 30, 8), y2=e(Bs, 30, 8), c3=e(Bs, 28, 8), mean1=e(groups, 16), rstd1=e(groups, 16),
             mean2=e(groups, 8), rstd2=e(groups, 8))
    call("tg_d_preconv_fwd", _p(poses), _p(w1), _p(b1), _p(g1), _p(be1), _p(w2), _p(b2), _p(g2), _p(be2), _p(w3), _p(b3), _p(o["c1"]), _p(o["y1"]),
         _p(o["c2"]), _p(o["y2"]), _p(o["c3"]), _p(o["mean1"]), _p(o["rstd1"]), _p(o["mean2"]), _p(o["rstd2"]), _p(rm1), _p(rv1), _p(nbt1), _p(rm2),
         _p(rv2), _p(nbt2), _p(ws), ws.numel() * 4, Bs, int(groups), float(eps), float(momentum), _stream())
    return o


def d_preconv_bwd(dc3, poses, c1, y1, c2, y2, mean1, rstd1, mean2, rstd2, w1, w2, w3, g1, g2, grads, dposes, dposes_accumulate, groups):
    """ConvDiscriminator.pre_conv backward, one launch (tg_d_preconv_bwd).  grads: None or the ten gradient tensors (dw1, db1, dgamma1, dbeta1,
    dw2, db2, dgamma2, dbeta2, dw3, db3), accumulated into; dposes: None or (nb, 34, 27), added to when dposes_accumulate."""
    nb = dc3.shape[0]
    for t_ in (dc3, poses, c1, y1, c2, y2, mean1, rstd1, mean2, rstd2):
        _flat(t_, "operand")
    assert tuple(dc3.shape) == (nb, 28, 8) and tuple(poses.shape) == (nb, 34, 27) and tuple(c1.shape) == (nb, 32, 16) and tuple(c2.shape) == (nb, 30, 8)
    assert mean1.numel() == groups * 16 and mean2.numel() == groups * 8
    dev = dc3.device
    key = (dev.type, dev.index, nb)
    if key not in _dpre_ws:
        _dpre_ws[key] = torch.zeros(_lib.load().tg_d_preconv_ws_bytes(nb) // 4, dtype=torch.int32, device=dev)
    ws = _dpre_ws[key]
    gp = [_p(None)] * 10 if grads is None else [_p(_flat(g_, "grad")) for g_ in grads]
    assert len(gp) == 10
    call("tg_d_preconv_bwd", _p(dc3), _p(poses), _p(c1), _p(y1), _p(c2), _p(y2), _p(mean1), _p(rstd1), _p(mean2), _p(rstd2), _p(w1), _p(w2), _p(w3),
         _p(g1), _p(g2), *gp, _p(dposes), int(bool(dposes_accumulate)), _p(ws), ws.numel() * 4, nb, int(groups), _stream())
    return dposes


def check_async_errors():
    """Raise if a bounded spin of a persistent kernel timed out since the last check (synchronises; tests, bench, loss read-out,
    every CHECK_EVERY replays of a captured step).  The timeout word is sticky on the device: only this function clears it."""
    for key, ws in _dpre_ws.items():
        if int(ws[0].item()) != 0:
            ws.zero_()              # counters of the aborted launch are out of step: back to the state of a fresh allocation
            raise RuntimeError(f"fused discriminator front end timed out at a device-wide barrier (device, Bs) = {key}; results are invalid")
    for key, ws in _gru_vec_ws_cache.items():
        if int(ws[0].item()) != 0:
            info = ws[:3].tolist()
            _gru_vec_fill(ws)       # exchange words of the aborted launch are out of step: back to the state of a fresh allocation
            raise RuntimeError(f"few-row GRU kernel timed out waiting for a member's h words (device, H) = {key}: step {info[1]}, workgroup {info[2]}; "
                               "results are invalid")
    for key, ws in _gru_ws.items():
        if int(ws[0].item()) != 0:
            info = ws[:14].tolist()
            # a timed-out launch leaves its cluster's generation word and the late members' flags out of step (the members that gave up
            # advanced the generation, the ones that arrived late published against the new one): the next launch on this workspace
            # would find its waits already satisfied by those stale flags.  Zero the WHOLE workspace -- timeout word, generations,
            # flags -- so the next launch starts from the state of a fresh allocation.
            ws.zero_()
            raise RuntimeError(f"persistent GRU kernel timed out waiting for a cluster member (device, B, H, bwd) = {key}: step {info[1]}, "
                               f"workgroup {info[2]}, flag words seen {info[4:14]}; results are invalid")


def gru_fused_dropout(B, H, bwd=False):
    """True when the recurrence kernel that will run for (B, H) applies the inter-layer dropout itself (drop_mask / dy_mask)."""
    # H = 64 only.  The cluster kernels accept drop_mask / y_drop / dy_mask too (C-ABI; tests call them directly), but measured on the generator
    # (B = 384, H = 300) the mask read + y_drop write inside the latency-critical persistent kernel cost as much as the separate fused
    # draw-and-apply pass saved (211 -> 233 us per launch against 33 us), so the layer code does not use that.
    return H == 64


def gru_forward(gi, w_hh, b_hh, y, save, drop_mask=None, y_drop=None, save_rows=None):
    """gi: [2, B, T, 3H] contiguous; w_hh/b_hh: (fwd, rev) pairs; y: [B, T, 2H]; save: [2, B, T, 4H] or None.
    drop_mask / y_drop ([B, T, 2H], where gru_fused_dropout(B, H)): fused inter-layer dropout, y_drop = y * drop_mask.
    save_rows = (row0, n): only these batch rows of `save` are needed (the cluster kernels then skip the rest; other kernels save all)."""
    _flat(gi, "gi"); _flat(y, "y")
    _, B, T, H3 = gi.shape
    H = H3 // 3
    assert gi.shape[0] == 2 and tuple(y.shape) == (B, T, 2 * H)
    for w, b in zip(w_hh, b_hh):
        _flat(w, "w_hh"); _flat(b, "b_hh"); assert tuple(w.shape) == (3 * H, H) and b.numel() == 3 * H
    if save is not None:
        _flat(save, "save"); assert tuple(save.shape) == (2, B, T, 4 * H)
    if H == 64:
        if drop_mask is not None:
            _flat(drop_mask, "drop_mask"); _flat(y_drop, "y_drop")
            assert tuple(drop_mask.shape) == tuple(y.shape) == tuple(y_drop.shape)
        call("tg_gru_h64_forward", _p(gi), B * T * H3, _p(w_hh[0]), _p(w_hh[1]), _p(b_hh[0]), _p(b_hh[1]), _p(y), _p(save),
             B * T * 4 * H, _p(drop_mask), _p(y_drop), B, T, _stream())
        return y
    if gru_vec_takes(B, H, save, drop_mask):
        ws = _gru_vec_ws(gi.device, H)
        call("tg_gru_forward_vec", _p(gi), B * T * H3, _p(w_hh[0]), _p(w_hh[1]), _p(b_hh[0]), _p(b_hh[1]), _p(y), C.c_void_p(ws.data_ptr()), ws.numel() * 4,
             B, T, H, _stream())
        return y
    chunks = gru_cluster_chunks(B, H) if (GRU_CLUSTER and H > 64) else None
    if chunks is not None:
        if drop_mask is not None:
            _flat(drop_mask, "drop_mask"); _flat(y_drop, "y_drop")
            assert tuple(drop_mask.shape) == tuple(y.shape) == tuple(y_drop.shape)
        r0, rn = (0, B) if save_rows is None else (int(save_rows[0]), int(save_rows[1]))
        assert 0 <= r0 and rn >= 0 and r0 + rn <= B
        ws = _gru_cluster_ws(gi.device, chunks[0][1], H)         # (the first chunk is the largest)
        at = lambda t_, c0, row_floats: C.c_void_p(0) if t_ is None else C.c_void_p(t_.data_ptr() + 4 * c0 * row_floats)
        for c0, cn in chunks:
            # rows [c0, c0 + cn) of every operand; the direction strides stay those of the whole batch.  Gates are saved for the part of
            # [r0, r0 + rn) that falls into the chunk
            s0, s1 = max(r0, c0), min(r0 + rn, c0 + cn)
            call("tg_gru_forward_cluster_rows", at(gi, c0, T * H3), B * T * H3, _p(w_hh[0]), _p(w_hh[1]), _p(b_hh[0]), _p(b_hh[1]), at(y, c0, T * 2 * H),
                 at(save, c0, T * 4 * H), B * T * 4 * H, at(drop_mask, c0, T * 2 * H), at(y_drop, c0, T * 2 * H), C.c_void_p(ws.data_ptr()), ws.numel() * 4,
                 cn, T, H, max(0, s0 - c0), max(0, s1 - s0), _stream())
        return y
    assert drop_mask is None and y_drop is None, "fused dropout: H = 64 or the cluster kernels only"
    call("tg_gru_forward", _p(gi), B * T * H3, _p(w_hh[0]), _p(w_hh[1]), _p(b_hh[0]), _p(b_hh[1]), _p(y), _p(save),
         B * T * 4 * H, B, T, H, _stream())
    return y


def gru_backward(dy, y, save, w_hh_t, dgi, dgh, dh_scratch, *, b0=0, nb=None, dy_mask=None, stats=None):
    """Backward through time for batch rows [b0, b0+nb) of a (possibly larger) stacked forward.
    dy: [nb, T, 2H]; y: [B, T, 2H]; save: [2, B, T, 4H]; w_hh_t: (fwd, rev) each [H, 3H]; dgi/dgh: [2, nb, T, 3H].
    dy_mask ([nb, T, 2H], H = 64 only): multiplied into dy while it is loaded (fused dropout backward).
    stats (cluster kernels only; returns True when they were filled): (gi_clipmax [2, nb], gi_colmax [2, 3H], gh_colmax [2, 3H]), ZEROED by the
    caller -- the magnitudes of dgi per batch row (over all T steps) and of dgi's / dgh's columns that the fp16 x 2 products reading them scale by."""
    _flat(dy, "dy"); _flat(y, "y"); _flat(save, "save"); _flat(dgi, "dgi"); _flat(dgh, "dgh"); _flat(dh_scratch, "dh")
    B, T, H2 = y.shape
    H = H2 // 2
    nb = B - b0 if nb is None else nb
    assert 0 <= b0 and b0 + nb <= B and tuple(dy.shape) == (nb, T, 2 * H)
    assert tuple(save.shape) == (2, B, T, 4 * H) and tuple(dgi.shape) == (2, nb, T, 3 * H) == tuple(dgh.shape)
    assert dh_scratch.numel() >= 4 * nb * H
    for w in w_hh_t:
        _flat(w, "w_hh_t"); assert tuple(w.shape) == (H, 3 * H)
    ys, ss = y[b0:b0 + nb], save[:, b0:b0 + nb]
    if H == 64:
        if dy_mask is not None:
            _flat(dy_mask, "dy_mask"); assert tuple(dy_mask.shape) == tuple(dy.shape)
        call("tg_gru_h64_backward", _p(dy), _p(dy_mask), _p(ys), C.c_void_p(ss.data_ptr()), B * T * 4 * H, _p(w_hh_t[0]), _p(w_hh_t[1]),
             _p(dgi), _p(dgh), nb * T * 3 * H, nb, T, _stream())
        return
    chunks = gru_cluster_chunks(nb, H, bwd=True) if (GRU_CLUSTER and H > 64) else None
    if chunks is not None:
        if dy_mask is not None:
            _flat(dy_mask, "dy_mask"); assert tuple(dy_mask.shape) == tuple(dy.shape)
        ws = _gru_cluster_ws(dy.device, chunks[0][1], H, bwd=True)
        at = lambda t_, c0, row_floats: C.c_void_p(0) if t_ is None else C.c_void_p(t_.data_ptr() + 4 * c0 * row_floats)
        if stats is not None:
            rm, ci, ch = stats
            _flat(rm, "gi_rowmax"); _flat(ci, "gi_colmax"); _flat(ch, "gh_colmax")
            assert tuple(rm.shape) == (2, nb) and tuple(ci.shape) == (2, 3 * H) == tuple(ch.shape)
        for c0, cn in chunks:                                    # (row chunks of one workspace, as in gru_forward)
            if stats is None:
                call("tg_gru_backward_cluster", at(dy, c0, T * 2 * H), at(dy_mask, c0, T * 2 * H), at(ys, c0, T * 2 * H), at(ss, c0, T * 4 * H), B * T * 4 * H,
                     _p(w_hh_t[0]), _p(w_hh_t[1]), at(dgi, c0, T * 3 * H), at(dgh, c0, T * 3 * H), nb * T * 3 * H, C.c_void_p(ws.data_ptr()), ws.numel() * 4,
                     cn, T, H, _stream())
            else:
                call("tg_gru_backward_cluster_stats", at(dy, c0, T * 2 * H), at(dy_mask, c0, T * 2 * H), at(ys, c0, T * 2 * H), at(ss, c0, T * 4 * H),
                     B * T * 4 * H, _p(w_hh_t[0]), _p(w_hh_t[1]), at(dgi, c0, T * 3 * H), at(dgh, c0, T * 3 * H), nb * T * 3 * H,
                     C.c_void_p(ws.data_ptr()), ws.numel() * 4, cn, T, H, at(stats[0], c0, 1), nb, _p(stats[1]), _p(stats[2]), _stream())
        return stats is not None
    assert dy_mask is None, "fused dropout backward: H = 64 or the cluster kernels only"
    call("tg_gru_backward", _p(dy), _p(ys), C.c_void_p(ss.data_ptr()), B * T * 4 * H, _p(w_hh_t[0]), _p(w_hh_t[1]),
         _p(dgi), _p(dgh), nb * T * 3 * H, _p(dh_scratch), nb, T, H, _stream())


# ------------------------------------------------------------------------------------------------- BatchNorm
def bn_train_stats(x2d, groups, ws, mean, rstd, running_mean, running_var, nbt, eps=1e-5, momentum=0.1, repeats=1):
    _flat(x2d, "x"); rows, Cc = x2d.shape
    assert ws.dtype == torch.float64 and ws.numel() >= 2 * groups * Cc and ws.is_cuda
    assert mean.numel() == groups * Cc == rstd.numel()
    assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda)
    call("tg_bn_train_stats", _p(x2d), rows, Cc, groups, _p(ws), _p(_flat(mean, "mean")), _p(_flat(rstd, "rstd")),
         _p(running_mean), _p(running_var), _p(nbt), float(eps), float(momentum), int(repeats), _stream())


def bn_fused_supported(rows, C, groups=1):
    """True when the single-workgroup fused BatchNorm launch handles this shape (csrc/norm.hip)."""
    return bool(_lib.load().tg_bn_fused_supported(int(rows), int(C), int(groups)))


def bn_train_fused(x2d, y2d, groups, mean, rstd, running_mean, running_var, nbt, gamma, beta, act_slope, eps=1e-5, momentum=0.1, repeats=1):
    """Small tensors: batch statistics, running-stat update and y = act(gamma * xhat + beta) in one launch."""
    _flat(x2d, "x"); _flat(y2d, "y"); rows, Cc = x2d.shape
    assert y2d.shape == x2d.shape and bn_fused_supported(rows, Cc, groups) and mean.numel() == groups * Cc == rstd.numel()
    assert gamma.numel() == Cc == beta.numel() and (nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda))
    call("tg_bn_train_fused", _p(x2d), _p(y2d), rows, Cc, groups, _p(_flat(mean, "mean")), _p(_flat(rstd, "rstd")), _p(running_mean),
         _p(running_var), _p(nbt), _p(_flat(gamma, "gamma")), _p(_flat(beta, "beta")), float(act_slope), float(eps), float(momentum),
         int(repeats), _stream())
    return y2d


BN2_MIN_ELEMS = 16384        # below: the single-workgroup fused kernels (one launch) win


def bn2_supported(rows_per_group, C, groups=1):
    """True when the two-launch BatchNorm kernels (csrc/norm.hip bn2_*) take this shape."""
    return rows_per_group * C * groups >= BN2_MIN_ELEMS and bool(_lib.load().tg_bn2_supported(int(rows_per_group), int(C)))


def bn2_train(x2d, y2d, groups, mean, rstd, running_mean, running_var, nbt, gamma, beta, act_slope, eps=1e-5, momentum=0.1, repeats=1):
    _flat(x2d, "x"); rows, Cc = x2d.shape
    assert rows % groups == 0 and (y2d is None or (_flat(y2d, "y").shape == x2d.shape)) and mean.numel() == groups * Cc == rstd.numel()
    assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda)
    rpg = rows // groups
    ws = torch.empty(_lib.load().tg_bn2_ws_doubles(rpg, Cc, groups), device=x2d.device, dtype=torch.float64)
    call("tg_bn2_train", _p(x2d), _p(y2d), rpg, Cc, groups, _p(ws), ws.numel(), _p(_flat(mean, "mean")), _p(_flat(rstd, "rstd")), _p(running_mean),
         _p(running_var), _p(nbt), _p(gamma), _p(beta), float(act_slope), float(eps), float(momentum), int(repeats), _stream())
    return y2d


def bn2_backward(dy2d, x2d, dx2d, groups, mean, rstd, gamma, beta, act_slope, dgamma, dbeta):
    """mean / rstd: [groups, C] statistics of exactly the groups held by dy / x (contiguous rows, group after group)."""
    _flat(dy2d, "dy"); _flat(x2d, "x"); _flat(dx2d, "dx"); rows, Cc = x2d.shape
    assert dy2d.shape == x2d.shape == dx2d.shape and rows % groups == 0 and mean.numel() == groups * Cc == rstd.numel()
    assert mean.is_contiguous() and rstd.is_contiguous()
    rpg = rows // groups
    ws = torch.empty(_lib.load().tg_bn2_ws_doubles(rpg, Cc, groups), device=x2d.device, dtype=torch.float64)
    call("tg_bn2_backward", _p(dy2d), _p(x2d), _p(dx2d), rpg, Cc, groups, _p(mean), _p(rstd), _p(_flat(gamma, "gamma")), _p(_flat(beta, "beta")),
         float(act_slope), _p(ws), ws.numel(), _p(dgamma), _p(dbeta), _stream())
    return dx2d


def bn_eval_stats(running_mean, running_var, mean, rstd, eps=1e-5):
    Cc = running_mean.numel()
    call("tg_bn_eval_stats", _p(_flat(running_mean, "rm")), _p(_flat(running_var, "rv")), Cc, float(eps),
         _p(_flat(mean, "mean")), _p(_flat(rstd, "rstd")), _stream())


def bn_apply(x2d, y2d, groups, mean, rstd, gamma, beta, act_slope):
    _flat(x2d, "x"); _flat(y2d, "y"); rows, Cc = x2d.shape
    assert y2d.shape == x2d.shape and mean.numel() == groups * Cc and gamma.numel() == Cc == beta.numel()
    call("tg_bn_apply", _p(x2d), _p(y2d), rows, Cc, groups, _p(mean), _p(rstd), _p(_flat(gamma, "gamma")),
         _p(_flat(beta, "beta")), float(act_slope), _stream())
    return y2d


def bn_backward(dy2d, x2d, dx2d, mean, rstd, gamma, beta, act_slope, ws, dgamma, dbeta):
    _flat(dy2d, "dy"); _flat(x2d, "x"); _flat(dx2d, "dx"); rows, Cc = x2d.shape
    assert dy2d.shape == x2d.shape == dx2d.shape and mean.numel() >= Cc and ws.dtype == torch.float64 and ws.numel() >= 2 * Cc
    call("tg_bn_backward", _p(dy2d), _p(x2d), _p(dx2d), rows, Cc, _p(mean), _p(rstd), _p(gamma), _p(beta),
         float(act_slope), _p(ws), _p(dgamma), _p(dbeta), _stream())
    return dx2d


# ------------------------------------------------------------------------------------------------- WavEncoder front end
# Conv1d(1, 16, 15) -> BatchNorm1d(16) -> LeakyReLU on raw audio without the pre-BatchNorm tensor (csrc/audio.hip); other shapes
# take the generic window-GEMM + BatchNorm launches
WAV_FUSED = True
WAV_FUSED_DGRAD = True      # backward also forms conv2's input gradient inside the reduction
_wav_ws = {}


def _wav_scratch(dev):
    """Per-device scratch of the front-end reductions (partials of one launch, consumed by the finalize launch right behind it)."""
    ws = _wav_ws.get(dev)
    if ws is None:
        ws = torch.empty(_lib.load().tg_wav_front_ws_doubles(), dtype=torch.float64, device=dev)
        _wav_ws[dev] = ws
    return ws


def _wav_geom(audio, w, bias, stride, pad):
    _f32(audio, "audio"); _flat(w, "w"); _flat(bias, "bias")
    assert audio.dim() == 2 and audio.stride(1) == 1, (audio.shape, audio.stride())
    B, L = audio.shape
    assert tuple(w.shape) == (16, 1, 15) and bias.numel() == 16, w.shape
    if (B - 1) * audio.stride(0) + L - 1 >= _room(audio):
        raise ValueError("wav_front: audio exceeds its tensor")
    T1 = (L + 2 * pad - 15) // stride + 1
    return B, L, T1


def wav_front_stats(audio, w, bias, stride, pad, mean, rstd, running_mean, running_var, nbt, fstat, eps=1e-5, momentum=0.1, repeats=1):
    B, L, T1 = _wav_geom(audio, w, bias, stride, pad)
    ws = _wav_scratch(audio.device)
    assert mean.numel() == 16 == rstd.numel() and fstat.dtype == torch.float64 and fstat.numel() >= _lib.load().tg_wav_front_fstat_doubles()
    assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda)
    call("tg_wav_front_stats", _p(audio), audio.stride(0), B, L, _p(w), _p(bias), int(stride), int(pad), T1, _p(ws), ws.numel(),
         _p(_flat(mean, "mean")), _p(_flat(rstd, "rstd")), _p(running_mean), _p(running_var), _p(nbt), _p(fstat), float(eps), float(momentum),
         int(repeats), _stream())


def wav_front_apply(audio, w, bias, stride, pad, mean, rstd, gamma, beta, act_slope, y, gate=None):
    B, L, T1 = _wav_geom(audio, w, bias, stride, pad)
    _flat(y, "y"); assert tuple(y.shape) == (B, T1, 16), (y.shape, B, T1)
    assert mean.numel() == 16 == rstd.numel() == gamma.numel() == beta.numel()
    if gate is not None:
        assert gate.dtype == torch.int64 and gate.is_cuda and gate.is_contiguous() and gate.numel() >= _lib.load().tg_wav_front_gate_words(B, T1)
    call("tg_wav_front_apply", _p(audio), audio.stride(0), B, L, _p(w), _p(bias), int(stride), int(pad), T1, _p(_flat(mean, "mean")),
         _p(_flat(rstd, "rstd")), _p(_flat(gamma, "gamma")), _p(_flat(beta, "beta")), float(act_slope), _p(y), _p(gate), _stream())
    return y


def wav_front_backward(dact, gate, audio, w, bias, stride, pad, mean, rstd, gamma, fstat, act_slope, dW, dbias, dgamma, dbeta):
    B, L, T1 = _wav_geom(audio, w, bias, stride, pad)
    _flat(dact, "dact"); assert tuple(dact.shape) == (B, T1, 16), (dact.shape, B, T1)
    assert gate.dtype == torch.int64 and gate.is_cuda and gate.is_contiguous() and gate.numel() >= _lib.load().tg_wav_front_gate_words(B, T1)
    assert fstat.dtype == torch.float64 and fstat.is_cuda and fstat.numel() >= _lib.load().tg_wav_front_fstat_doubles()
    for t, n in ((dW, 240), (dbias, 16), (dgamma, 16), (dbeta, 16)):
        assert t is None or (_flat(t, "grad").numel() == n)
    ws = _wav_scratch(audio.device)
    call("tg_wav_front_backward", _p(dact), _p(gate), _p(audio), audio.stride(0), B, L, _p(w), _p(bias), int(stride), int(pad), T1,
         _p(_flat(mean, "mean")), _p(_flat(rstd, "rstd")), _p(_flat(gamma, "gamma")), _p(fstat), float(act_slope), _p(ws), ws.numel(),
         _p(dW), _p(dbias), _p(dgamma), _p(dbeta), _stream())


def wav_front_backward_fused(dc2, w2, gate, audio, w, bias, stride, pad, mean, rstd, gamma, fstat, act_slope, dW, dbias, dgamma, dbeta):
    """wav_front_backward with d act = conv_dgrad(dc2, w2) (Conv1d(16, 32, 15, stride 6)) formed inside the kernel."""
    B, L, T1 = _wav_geom(audio, w, bias, stride, pad)
    _flat(dc2, "dc2"); _flat(w2, "w2")
    T2 = (T1 - 15) // 6 + 1
    assert tuple(dc2.shape) == (B, T2, 32) and tuple(w2.shape) == (32, 16, 15), (dc2.shape, w2.shape, B, T2)
    assert gate.dtype == torch.int64 and gate.is_cuda and gate.is_contiguous() and gate.numel() >= _lib.load().tg_wav_front_gate_words(B, T1)
    assert fstat.dtype == torch.float64 and fstat.is_cuda and fstat.numel() >= _lib.load().tg_wav_front_fstat_doubles()
    for t, n in ((dW, 240), (dbias, 16), (dgamma, 16), (dbeta, 16)):
        assert t is None or (_flat(t, "grad").numel() == n)
    ws = _wav_scratch(audio.device)
    call("tg_wav_front_backward_fused", _p(dc2), T2, _p(w2), _p(gate), _p(audio), audio.stride(0), B, L, _p(w), _p(bias), int(stride), int(pad), T1,
         _p(_flat(mean, "mean")), _p(_flat(rstd, "rstd")), _p(_flat(gamma, "gamma")), _p(fstat), float(act_slope), _p(ws), ws.numel(),
         _p(dW), _p(dbias), _p(dgamma), _p(dbeta), _stream())


_w2_ws = {}


def wav_conv2_wgrad(dc2, act, dW2, db2):
    """dW2 (32, 16, 15) / db2 (32) += the gradients of Conv1d(16, 32, 15, stride 6) for dc2 (B, T2, 32) over act (B, T1, 16): one pass, deterministic."""
    _flat(dc2, "dc2"); _flat(act, "act")
    B, T1, _ = act.shape
    T2 = (T1 - 15) // 6 + 1
    assert act.shape[2] == 16 and tuple(dc2.shape) == (B, T2, 32), (act.shape, dc2.shape)
    assert dW2 is None or (_flat(dW2, "dW2").numel() == 32 * 16 * 15)
    assert db2 is None or (_flat(db2, "db2").numel() == 32)
    ws = _w2_ws.get(act.device)
    if ws is None:
        ws = _w2_ws[act.device] = torch.empty(_lib.load().tg_wav_conv2_wgrad_ws_floats(), device=act.device)
    call("tg_wav_conv2_wgrad", _p(dc2), _p(act), B, T1, T2, _p(ws), ws.numel(), _p(dW2), _p(db2), _stream())


# ------------------------------------------------------------------------------------------------- element-wise
def _same(*ts):
    n = ts[0].numel()
    for t in ts:
        _flat(t, "operand"); assert t.numel() == n, [tuple(x.shape) for x in ts]
    return n


def add_relu(a, b, y):
    call("tg_add_relu", _p(a), _p(b), _p(y), _same(a, b, y), _stream()); return y


def act_mask_bwd(dy, y, mask, slope, dx):
    if isinstance(mask, Drop):
        n = _same(dy, y, dx)
        assert mask.numel() == n
        call("tg_act_mask_bwd_drop", _p(dy), _p(y), mask.p, _p(mask.state), mask.site, mask.index0, float(slope), _p(dx), n, _stream()); return dx
    n = _same(dy, y, dx) if mask is None else _same(dy, y, mask, dx)
    call("tg_act_mask_bwd", _p(dy), _p(y), _p(mask), float(slope), _p(dx), n, _stream()); return dx


def act_mask_bwd2(dy, y, o, mask, slope, dsum, dc):
    """(dsum, dc): dsum = dy * (y > 0), dc = dsum * (o > 0 ? 1 : slope) * mask -- both gates of a residual block's backward in one pass."""
    if isinstance(mask, Drop):
        n = _same(dy, y, o, dsum, dc)
        assert mask.numel() == n
        call("tg_act_mask_bwd2_drop", _p(dy), _p(y), _p(o), mask.p, _p(mask.state), mask.site, mask.index0, float(slope), _p(dsum), _p(dc), n, _stream())
        return dsum, dc
    n = _same(dy, y, o, dsum, dc) if mask is None else _same(dy, y, o, mask, dsum, dc)
    call("tg_act_mask_bwd2", _p(dy), _p(y), _p(o), _p(mask), float(slope), _p(dsum), _p(dc), n, _stream()); return dsum, dc


def mul(x, mask, y):
    if isinstance(mask, Drop):                 # x * regenerated mask = act_mask_bwd with slope 1 (the gate factor is then 1 everywhere)
        return act_mask_bwd(x, x, mask, 1.0, y)
    call("tg_mul", _p(x), _p(mask), _p(y), _same(x, mask, y), _stream()); return y


def axpy(x, y, alpha=1.0, accumulate=True):
    call("tg_axpy", _p(x), _p(y), float(alpha), int(accumulate), _same(x, y), _stream()); return y


def _chk2d(t, rows, cols, name):
    _f32(t, name)
    assert t.dim() == 2 and t.stride(1) == 1 and t.shape[0] >= rows and t.shape[1] >= cols, (name, t.shape, rows, cols)
    if (rows - 1) * t.stride(0) + cols - 1 >= _room(t):
        raise ValueError(f"{name}: exceeds its tensor")


def copy2d(src, dst, *, accumulate=False):
    """dst[r, c] (+)= src[r, c] for 2-D views with unit inner stride."""
    rows, cols = src.shape
    _chk2d(src, rows, cols, "src"); _chk2d(dst, rows, cols, "dst"); assert tuple(dst.shape) == (rows, cols)
    call("tg_copy2d", _p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, int(accumulate), _stream()); return dst


def repeat_rows(src, dst, B, T):
    """dst[(b*T + t), :] = src[b, :]; src [B, cols], dst 2-D view [B*T, cols]."""
    cols = src.shape[1]
    _chk2d(src, B, cols, "src"); _chk2d(dst, B * T, cols, "dst")
    call("tg_repeat_rows", _p(src), src.stride(0), _p(dst), dst.stride(0), B, T, cols, _stream()); return dst


def sum_rows(src, dst, B, T, *, accumulate=False):
    cols = dst.shape[1]
    _chk2d(src, B * T, cols, "src"); _chk2d(dst, B, cols, "dst")
    call("tg_sum_rows", _p(src), src.stride(0), _p(dst), dst.stride(0), B, T, cols, int(accumulate), _stream()); return dst


def sum_parts(parts, out):
    """out = parts[0] + parts[1] + ..: parts (n_parts, ...) contiguous, out of the shape of one part (tg_sum_parts)."""
    _flat(parts, "parts"); _flat(out, "out")
    assert parts.numel() == parts.shape[0] * out.numel()
    call("tg_sum_parts", _p(parts), out.numel(), parts.shape[0], _p(out), out.numel(), _stream()); return out


def narrow8_pair(a0, a1, w0, w1, out):
    """out [M, 8] = a0 [M, K] @ w0 [K, 8] + a1 [M, K] @ w1 [K, 8] (tg_narrow8_pair)."""
    for t_ in (a0, a1, w0, w1, out):
        _flat(t_, "operand")
    M, K = a0.shape
    assert a1.shape == a0.shape and tuple(w0.shape) == (K, 8) == tuple(w1.shape) and tuple(out.shape) == (M, 8) and K % 4 == 0
    call("tg_narrow8_pair", _p(a0), _p(a1), _p(w0), _p(w1), _p(out), M, K, _stream()); return out


def add_halves(y, o):
    _flat(y, "y"); _flat(o, "o"); H = o.shape[-1]; M = o.numel() // H
    assert y.numel() == 2 * o.numel()
    call("tg_add_halves", _p(y), _p(o), M, H, _stream()); return o


def dup_halves(d_o, dy):
    _flat(d_o, "do"); _flat(dy, "dy"); H = d_o.shape[-1]; M = d_o.numel() // H
    assert dy.numel() == 2 * d_o.numel()
    call("tg_dup_halves", _p(d_o), _p(dy), M, H, _stream()); return dy


def _i64(t, name):
    if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous CUDA int64 tensor")
    return t


def make_pre_seq(target, pre, n_pre):
    _flat(target, "target"); _flat(pre, "pre"); B, T, D = target.shape
    assert tuple(pre.shape) == (B, T, D + 1)
    call("tg_make_pre_seq", _p(target), _p(pre), B, T, D, int(n_pre), _stream()); return pre


def embed_gather(table, idx, out):
    _flat(table, "table"); _i64(idx, "idx"); _flat(out, "out")
    n_rows, D = table.shape
    assert out.numel() == idx.numel() * D
    call("tg_embed_gather", _p(table), _p(idx), _p(out), idx.numel(), D, n_rows, _stream()); return out


def assemble_batch(rec, out_text, out_audio, out_vec, out_vid, *, remove_word_timing=False):
    """Raw per-clip records on the device (data.RecordLayout views of one flat buffer) -> the training step's input tensors, in place:
    SpeechMotionDataset.__getitem__ + default_collate_fn for the whole batch in one launch (tg_assemble_batch)."""
    B, n_poses = out_text.shape
    A = out_audio.shape[1]
    Dp = out_vec.shape[2]
    assert out_text.dtype == torch.int64 and out_vid.dtype == torch.int64 and out_text.is_contiguous() and out_audio.is_contiguous() and out_vec.is_contiguous()
    assert tuple(out_audio.shape) == (B, A) and tuple(out_vec.shape) == (B, n_poses, Dp) and out_vid.numel() == B
    _f32(out_audio, "out_audio"); _f32(out_vec, "out_vec"); _f32(rec["audio"], "audio"); _f32(rec["vec"], "vec")
    Wmax = rec["word_idx"].shape[1]
    assert rec["audio_off"].numel() == B + 1 and rec["vec_off"].numel() == B + 1 and tuple(rec["word_idx"].shape) == (B, Wmax) == tuple(rec["word_onset"].shape)
    assert rec["word_idx"].dtype == torch.int64 and rec["word_onset"].dtype == torch.float64 and rec["times"].dtype == torch.float64
    assert rec["n_words"].dtype == torch.int32 and rec["n_ext"].dtype == torch.int32 and rec["vid"].dtype == torch.int64
    assert rec["audio"].numel() >= 1 and rec["vec"].numel() >= B * n_poses * Dp
    call("tg_assemble_batch", _p(rec["audio"]), _p(rec["audio_off"]), _p(rec["vec"]), _p(rec["vec_off"]), _p(rec["word_idx"]), _p(rec["word_onset"]),
         _p(rec["n_words"]), _p(rec["times"]), _p(rec["n_ext"]), _p(rec["vid"]), B, Wmax, n_poses, Dp, A, int(bool(remove_word_timing)),
         _p(out_text), _p(out_audio), _p(out_vec), _p(out_vid), _stream())


def embed_gather_drop(table, idx, out, p, state, site):
    """(out, Drop): out = table[idx] * dropout mask (F.dropout after the look-up, one pass; the mask is not stored)."""
    _flat(table, "table"); _i64(idx, "idx"); _flat(out, "out")
    n_rows, D = table.shape
    assert out.numel() == idx.numel() * D and D % 4 == 0
    call("tg_embed_gather_drop", _p(table), _p(idx), _p(out), idx.numel(), D, n_rows, float(p), _p(_i64(state, "rng_state")), int(site), _stream())
    return out, Drop(state, site, p, out.shape)


def embed_scatter_add(dout, idx, dtable):
    _flat(dout, "dout"); _i64(idx, "idx"); _flat(dtable, "dtable")
    n_rows, D = dtable.shape
    assert dout.numel() == idx.numel() * D
    call("tg_embed_scatter_add", _p(dout), _p(idx), _p(dtable), idx.numel(), D, n_rows, _stream()); return dtable


def permute3(x, out, perm):
    _flat(x, "in"); _flat(out, "out"); assert x.dim() == 3 and out.numel() == x.numel()
    call("tg_permute3", _p(x), _p(out), *x.shape, *perm, _stream()); return out


def permute3_batch(desc, n_jobs, total_workgroups):
    """desc: int64 device tensor [n_jobs, 10] built by layers.WeightPrep (pointers of live tensors it owns / was given)."""
    assert desc.dtype == torch.int64 and desc.is_cuda and desc.is_contiguous() and tuple(desc.shape) == (n_jobs, 10)
    call("tg_permute3_batch", C.c_void_p(desc.data_ptr()), int(n_jobs), int(total_workgroups), _stream())


def conv_dgrad_pack(w, out, stride):
    _flat(w, "w"); _flat(out, "out"); Co, Ci, kw = w.shape
    J = (kw + stride - 1) // stride
    assert out.numel() == stride * Ci * J * Co
    call("tg_conv_dgrad_pack", _p(w), _p(out), Co, Ci, kw, stride, _stream()); return out


def weight_norm_fwd(v, g, w_packed):
    _flat(v, "v"); _flat(g, "g"); _flat(w_packed, "w"); Co, Ci, kw = v.shape
    assert g.numel() == Co and w_packed.numel() == v.numel()
    call("tg_weight_norm_fwd", _p(v), _p(g), _p(w_packed), Co, Ci, kw, _stream()); return w_packed


def weight_norm_fwd_batch(vs, gs, want_t=True):
    """Every weight-normed conv of a network in one launch.  vs / gs: lists of (Co, Ci, kw) / (Co, 1, 1) parameters of equal shape.
    Returns (w_packed [n, Co, kw*Ci], w_t [n, Ci, kw*Co] or None)."""
    n = len(vs)
    Co, Ci, kw = vs[0].shape
    for v, g in zip(vs, gs):
        _flat(v, "v"); _flat(g, "g"); assert tuple(v.shape) == (Co, Ci, kw) and g.numel() == Co
    wp = torch.empty(n, Co, kw * Ci, device=vs[0].device)
    wt = torch.empty(n, Ci, kw * Co, device=vs[0].device) if want_t else None
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    for j0 in range(0, n, 8):                       # the entry point takes 1..8 convs per launch; n_layers is a free hyper-parameter
        j1 = min(n, j0 + 8)
        call("tg_weight_norm_fwd_batch", j1 - j0, arr(vs[j0:j1]), arr(gs[j0:j1]), arr([wp[i] for i in range(j0, j1)]),
             arr([wt[i] for i in range(j0, j1)]) if want_t else None, Co, Ci, kw, _stream())
    return wp, wt


def weight_norm_bwd(dw_packed, v, g, dg, dv):
    Co, Ci, kw = v.shape
    assert _same(dw_packed, v, dv) and dg.numel() == Co == g.numel()
    call("tg_weight_norm_bwd", _p(dw_packed), _p(v), _p(g), _p(_flat(dg, "dg")), _p(dv), Co, Ci, kw, _stream())


def weight_norm_bwd_batch(dws, vs, gs, dgs, dvs):
    """weight_norm_bwd for up to 8 convs of equal shape in one launch (lists of tensors, one entry per conv)."""
    n = len(vs)
    Co, Ci, kw = vs[0].shape
    for dw, v, g, dg, dv in zip(dws, vs, gs, dgs, dvs):
        assert tuple(v.shape) == (Co, Ci, kw) and _same(dw, v, dv) and dg.numel() == Co == g.numel()
        _flat(dg, "dg"); _flat(g, "g")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    call("tg_weight_norm_bwd_batch", n, arr(dws), arr(vs), arr(gs), arr(dgs), arr(dvs), Co, Ci, kw, _stream())


# ------------------------------------------------------------------------------------------------- RNG
def new_rng_state(seed, device):
    return torch.tensor([int(seed) & (2 ** 63 - 1), 0], dtype=torch.int64, device=device)


def rng_advance(state):
    call("tg_rng_advance", _p(_i64(state, "rng_state")), _stream())


def iter_head(rng_a, rng_b, step_a, step_b, target, n_pre, copies, text=None, vid=None, permute_last=False, perm_in=None, perm_site=0, row_floats=None,
              target_copy=None):
    """tg_iter_head: counters + stacked seed poses / word ids / speaker ids of a GAN iteration, one launch.  Returns (pre_s, text_s, vid_s).
    row_floats > D + 1: pre_s is the [:, :, :D + 1] view of a fresh (copies * B, T, row_floats) buffer -- the generator's GRU input rows, whose
    pose columns are then already in place (GeneratorEngine.forward recognises the view and skips its copy).
    target_copy ((B, T, D), optional): receives a copy of target (the real half of the discriminator's stacked input)."""
    for r in (rng_a, rng_b):
        assert r is None or _i64(r, "rng_state") is r
    for c in (step_a, step_b):
        assert c is None or (c.is_cuda and c.dtype == torch.int32)
    _flat(target, "target"); B, T, D = target.shape
    ld = D + 1 if row_floats is None else int(row_floats)
    assert ld >= D + 1
    pre = torch.empty(copies * B, T, ld, device=target.device)
    text_s = vid_s = None
    if text is not None:
        _i64(text, "text"); assert tuple(text.shape) == (B, T)
        text_s = torch.empty(copies * B, T, dtype=torch.int64, device=target.device)
    if vid is not None:
        _i64(vid, "vid"); assert vid.numel() == B
        vid_s = torch.empty(copies * B, dtype=torch.int64, device=target.device)
    if perm_in is not None:
        _i64(perm_in, "perm"); assert perm_in.numel() == B
    if target_copy is not None:
        _flat(target_copy, "target_copy"); assert tuple(target_copy.shape) == (B, T, D)
    call("tg_iter_head", _p(rng_a), _p(rng_b), _p(step_a), _p(step_b), _p(target), _p(pre), ld, B, T, D, int(n_pre), int(copies), _p(text), _p(text_s),
         _p(vid), _p(vid_s), int(bool(permute_last)), _p(perm_in), int(perm_site), None, _p(target_copy), _stream())
    return pre[:, :, :D + 1], text_s, vid_s


def iter_begin(rng_a, rng_b, step_a, step_b):
    """Advance up to two RNG states and up to two Adam step counters in one launch (None = skip)."""
    for r in (rng_a, rng_b):
        assert r is None or _i64(r, "rng_state") is r
    for c in (step_a, step_b):
        assert c is None or (c.is_cuda and c.dtype == torch.int32)
    call("tg_iter_begin", _p(rng_a), _p(rng_b), _p(step_a), _p(step_b), _stream())


def dropout_mask(mask, p, state, site):
    call("tg_dropout_mask", _p(_flat(mask, "mask")), mask.numel(), float(p), _p(_i64(state, "rng_state")), int(site), _stream())
    return mask


def dropout_apply(x, p, state, site, store_mask=True):
    """(y, mask) with mask drawn as dropout_mask does and y = x * mask, one pass.  store_mask=False: the mask is not written; the second
    return value is the Drop its backward consumers regenerate it from (x.numel() % 4 == 0)."""
    _flat(x, "x")
    y = torch.empty_like(x)
    if not store_mask and x.numel() % 4 == 0:
        call("tg_dropout_apply", _p(x), _p(y), None, x.numel(), float(p), _p(_i64(state, "rng_state")), int(site), _stream())
        return y, Drop(state, site, p, x.shape)
    mask = torch.empty_like(x)
    call("tg_dropout_apply", _p(x), _p(y), _p(mask), x.numel(), float(p), _p(_i64(state, "rng_state")), int(site), _stream())
    return y, mask


def normal(out, state, site):
    call("tg_normal", _p(_flat(out, "out")), out.numel(), _p(_i64(state, "rng_state")), int(site), _stream()); return out


def randperm(out, state, site):
    call("tg_randperm", _p(_i64(out, "out")), out.numel(), _p(_i64(state, "rng_state")), int(site), _stream()); return out


def gather_i64(src, perm, out):
    _i64(src, "src"); _i64(perm, "perm"); _i64(out, "out"); assert src.numel() == perm.numel() == out.numel()
    call("tg_gather_i64", _p(src), _p(perm), _p(out), src.numel(), _stream()); return out


# ------------------------------------------------------------------------------------------------- speaker path / losses
def reparam_fwd(mu, logvar, eps, z):
    call("tg_reparam_fwd", _p(mu), _p(logvar), _p(eps), _p(z), _same(mu, logvar, eps, z), _stream()); return z


def reparam_bwd(dz, logvar, eps, dmu, dlogvar):
    call("tg_reparam_bwd", _p(dz), _p(logvar), _p(eps), _p(dmu), _p(dlogvar), _same(dz, logvar, eps, dmu, dlogvar), _stream())


SPEAKER_FUSED = True


def speaker_fwd(table, vid, w1, b1, wmu, bmu, wlv, blv, eps, rep=None, T=0, draw=None):
    """Fused speaker path forward -> (se, zc, mu, logvar, z), each [B, 16]; rep: 2-D view [B * T, 16] (row stride free) that receives z per frame.
    draw = (rng_state, site): eps is an OUTPUT, drawn in the same launch exactly as normal(eps, state, site) would."""
    _flat(table, "table"); _i64(vid, "vid"); _flat(eps, "eps")
    B = vid.numel()
    assert table.shape[1] == 16 and tuple(eps.shape) == (B, 16)
    for w in (w1, wmu, wlv):
        _flat(w, "w"); assert tuple(w.shape) == (16, 16)
    for b in (b1, bmu, blv):
        _flat(b, "b"); assert b.numel() == 16
    outs = [torch.empty(B, 16, device=table.device) for _ in range(5)]
    rep_ld = 0
    if rep is not None:
        _chk2d(rep, B * T, 16, "rep"); rep_ld = rep.stride(0)
    call("tg_speaker_fwd", _p(table), _p(vid), table.shape[0], _p(w1), _p(b1), _p(wmu), _p(bmu), _p(wlv), _p(blv), _p(eps), *[_p(o) for o in outs], B,
         _p(rep), rep_ld, int(T), _p(_i64(draw[0], "rng_state")) if draw is not None else None, int(draw[1]) if draw is not None else 0, _stream())
    return outs


def speaker_bwd_supported(nb):
    # (deterministic mode: the fused kernel scatters the speaker-embedding gradient with float atomics -> the generic chain of launches)
    return SPEAKER_FUSED and nb <= _lib.load().tg_speaker_bwd_max_rows() and not deterministic()


def speaker_bwd(dz, d_mu, d_logvar, logvar, eps, zc, se, vid, w1, wmu, wlv, dw1, db1, dwmu, dbmu, dwlv, dblv, dtable):
    nb = dz.shape[0]
    for t in (dz, logvar, eps, zc, se):
        _flat(t, "operand"); assert tuple(t.shape) == (nb, 16), t.shape
    for t in (d_mu, d_logvar):
        assert t is None or (_flat(t, "direct gradient").shape == dz.shape)
    _i64(vid, "vid"); assert vid.numel() == nb
    for t in (w1, wmu, wlv, dw1, dwmu, dwlv):
        _flat(t, "w"); assert t.numel() == 256
    for t in (db1, dbmu, dblv):
        _flat(t, "b"); assert t.numel() == 16
    _flat(dtable, "dtable"); assert dtable.shape[1] == 16
    call("tg_speaker_bwd", _p(dz), _p(d_mu), _p(d_logvar), _p(logvar), _p(eps), _p(zc), _p(se), _p(vid), dtable.shape[0], _p(w1), _p(wmu), _p(wlv),
         _p(dw1), _p(db1), _p(dwmu), _p(dbmu), _p(dwlv), _p(dblv), _p(dtable), nb, _stream())


OUT_MLP_COMPOSED = True


def out_mlp_compose(w1, b1, w2, b2, dup=1):
    """(w21 [D, dup H], w21t [dup H, D], b21 [D]) of Linear(H, Hm) -> identity -> Linear(Hm, D); dup = 2: the composed weight side by side
    twice, acting on a bidirectional GRU output [fwd | rev] without the direction sum."""
    for t in (w1, b1, w2, b2):
        _flat(t, "parameter")
    Hm, H = w1.shape; D = w2.shape[0]
    assert tuple(w2.shape) == (D, Hm) and b1.numel() == Hm and b2.numel() == D and dup in (1, 2)
    w21, w21t, b21 = torch.empty(D, dup * H, device=w1.device), torch.empty(dup * H, D, device=w1.device), torch.empty(D, device=w1.device)
    call("tg_out_mlp_compose", _p(w1), _p(b1), _p(w2), _p(b2), H, Hm, D, dup, _p(w21), _p(w21t), _p(b21), _stream())
    return w21, w21t, b21


def out_mlp_param_grads(Pm, s, w1, b1, w2, dw1, db1, dw2, db2, dup=1):
    Hm, H = w1.shape; D = w2.shape[0]
    for t in (Pm, s, w1, b1, w2, dw1, db1, dw2, db2):
        _flat(t, "operand")
    assert tuple(Pm.shape) == (D, dup * H) and s.numel() == D and dw1.numel() == Hm * H and dw2.numel() == D * Hm and db1.numel() == Hm and db2.numel() == D
    call("tg_out_mlp_param_grads", _p(Pm), _p(s), _p(w1), _p(b1), _p(w2), H, Hm, D, dup, _p(dw1), _p(db1), _p(dw2), _p(db2), _stream())


def gan_d_loss(logit_real, logit_fake, out, d_real, d_fake):
    B = _same(logit_real, logit_fake, d_real, d_fake); _flat(out, "out")
    call("tg_gan_d_loss", _p(logit_real), _p(logit_fake), B, _p(out), _p(d_real), _p(d_fake), _stream())


def gan_g_loss(out_pose, target, out_rand, z, z_rand, mu, logvar, logit_out, weights, use_gan, ws, scalars, d_out, d_mu,
               d_logvar, d_logit):
    B = out_pose.shape[0]
    TD = _same(out_pose, target, out_rand, d_out) // B
    Z = _same(z, z_rand, mu, logvar, d_mu, d_logvar) // B
    assert _same(logit_out, d_logit) == B and ws.numel() >= 3 * B and scalars.numel() >= 5
    call("tg_gan_g_loss", _p(out_pose), _p(target), _p(out_rand), _p(z), _p(z_rand), _p(mu), _p(logvar), _p(logit_out),
         B, TD, Z, *[float(w) for w in weights], int(bool(use_gan)), _p(_flat(ws, "ws")), _p(_flat(scalars, "scalars")),
         _p(d_out), _p(d_mu), _p(d_logvar), _p(d_logit), _stream())


def d_head_fwd(y, w1, b1, w2, b2):
    """ConvDiscriminator head: y [B, T, 2H] -> (l1 [B, T], logit [B, 1], prob [B, 1]) in one launch."""
    _flat(y, "y"); B, T, H2 = y.shape; H = H2 // 2
    assert w1.numel() == H and b1.numel() == 1 and w2.numel() == T and b2.numel() == 1
    l1, logit, prob = (torch.empty(B, T, device=y.device), torch.empty(B, 1, device=y.device), torch.empty(B, 1, device=y.device))
    call("tg_d_head_fwd", _p(y), _p(_flat(w1, "w1")), _p(b1), _p(_flat(w2, "w2")), _p(b2), _p(l1), _p(logit), _p(prob), B, T, H, _stream())
    return l1, logit, prob


def d_head_bwd(d_logit, y, l1, w1, w2, dy, grads=None):
    """Backward of the head: d_logit [nb] -> dy [nb, T, 2H]; grads = (dw1, db1, dw2, db2) accumulate, or None."""
    _flat(d_logit, "d_logit"); _flat(y, "y"); _flat(l1, "l1"); _flat(dy, "dy")
    nb, T, H2 = y.shape; H = H2 // 2
    assert d_logit.numel() == nb and tuple(l1.shape) == (nb, T) and tuple(dy.shape) == tuple(y.shape) and H <= 64 and T <= 64
    g = (None,) * 4 if grads is None else tuple(_flat(t, "grad") for t in grads)
    call("tg_d_head_bwd", _p(d_logit), _p(y), _p(l1), _p(_flat(w1, "w1")), _p(_flat(w2, "w2")), _p(dy), _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]),
         nb, T, H, _stream())
    return dy


_head_step_counter = {}        # device -> the zero word tg_d_head_step counts its workgroups' arrivals in (the kernel leaves it at zero)


def d_head_step(y, w1, b1, w2, b2, n_real, scale_real, scale_fake, out=None, grads=None):
    """ConvDiscriminator head forward + per-clip GAN loss terms + head backward in ONE launch (tg_d_head_step).  y [n_rows, T, 2H]: rows
    [0, n_real) scored as real, the rest as fake.  Discriminator step (train_gan.py:36-41): y = [real ; fake], n_real = B, both scales 1 / B.
    Generator step (:55-57, 86-88): n_real = n_rows = B, scale_real = loss_gan_weight / B, grads None.
    Returns dict(l1, logit, prob, d_logit, terms, dy); the loss is -terms.sum() / n_real -- written to out[0] when out is given (serial
    last-workgroup tail in the kernel), else left to the caller.  grads = (dw1, db1, dw2, db2) accumulate, or None."""
    _flat(y, "y"); n_rows, T, H2 = y.shape; H = H2 // 2
    assert 0 < n_real <= n_rows and w1.numel() == H and b1.numel() == 1 and w2.numel() == T and b2.numel() == 1 and H <= 64 and T <= 32
    dev = y.device
    key = (dev.type, dev.index)
    if key not in _head_step_counter:
        _head_step_counter[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    e = lambda *shape: torch.empty(*shape, device=dev)
    o = dict(l1=e(n_rows, T), logit=e(n_rows, 1), prob=e(n_rows, 1), d_logit=e(n_rows), terms=e(n_rows), dy=e(n_rows, T, H2))
    g = (None,) * 4 if grads is None else tuple(_flat(t, "grad") for t in grads)
    call("tg_d_head_step", _p(y), _p(_flat(w1, "w1")), _p(b1), _p(_flat(w2, "w2")), _p(b2), _p(o["l1"]), _p(o["logit"]), _p(o["prob"]),
         _p(o["d_logit"]), _p(o["terms"]), _p(None if out is None else _flat(out, "out")), _p(_head_step_counter[key]), _p(o["dy"]), _p(g[0]),
         _p(g[1]), _p(g[2]), _p(g[3]), n_rows, int(n_real), float(scale_real), float(scale_fake), T, H, _stream())
    return o


def l1_mean(a, b, out):
    call("tg_l1_mean", _p(a), _p(b), _same(a, b), _p(_flat(out, "out")), _stream()); return out


def sigmoid(x, y):
    call("tg_sigmoid", _p(x), _p(y), _same(x, y), _stream()); return y


def sigmoid_bwd(dy, y, dx):
    call("tg_sigmoid_bwd", _p(dy), _p(y), _p(dx), _same(dy, y, dx), _stream()); return dx


def window_blend(prev_tail, nxt):
    """nxt (B,T,D) in place: first n frames cross-faded with prev_tail (B,n,D)."""
    _flat(prev_tail, "prev_tail"); _flat(nxt, "next")
    B, n, D = prev_tail.shape
    assert nxt.shape[0] == B and nxt.shape[2] == D and n <= nxt.shape[1]
    call("tg_window_blend", _p(prev_tail), _p(nxt), B, nxt.shape[1], D, n, _stream()); return nxt


def pose_metrics(out, target, mean_dir_vec, n_pre, sums):
    _same(out, target); _flat(mean_dir_vec, "mean"); B, T, D = out.shape
    assert D == 27 and mean_dir_vec.numel() == 27 and sums.dtype == torch.float64 and sums.numel() >= 3 and sums.is_cuda
    call("tg_pose_metrics", _p(out), _p(target), _p(mean_dir_vec), B, T, int(n_pre), _p(sums), _stream()); return sums


def ae_loss(recon, target, out, d_recon):
    B, T, D = recon.shape; _same(recon, target, d_recon)
    call("tg_ae_loss", _p(recon), _p(target), B, T, D, _p(_flat(out, "out")), _p(d_recon), _stream())


# ------------------------------------------------------------------------------------------------- fused autoencoder training step
# csrc/ae_step.hip: the FGD autoencoder's training step up to the gradients in 18 launches (train_feature_extractor.py:54-97)
_AE_E, _AE_D = "pose_encoder", "decoder"
AE_PARAMS = ([(f"{_AE_E}.net.{i}.{j}.{w}", s) for i, (co, ci, kw) in enumerate(((32, 27, 3), (64, 32, 3), (64, 64, 4)))
              for j, w, s in ((0, "weight", (co, ci, kw)), (0, "bias", (co,)), (1, "weight", (co,)), (1, "bias", (co,)))] +
             [(f"{_AE_E}.net.3.weight", (32, 64, 3)), (f"{_AE_E}.net.3.bias", (32,)),
              (f"{_AE_E}.out_net.0.weight", (256, 384)), (f"{_AE_E}.out_net.0.bias", (256,)), (f"{_AE_E}.out_net.1.weight", (256,)), (f"{_AE_E}.out_net.1.bias", (256,)),
              (f"{_AE_E}.out_net.3.weight", (128, 256)), (f"{_AE_E}.out_net.3.bias", (128,)), (f"{_AE_E}.out_net.4.weight", (128,)), (f"{_AE_E}.out_net.4.bias", (128,)),
              (f"{_AE_E}.out_net.6.weight", (32, 128)), (f"{_AE_E}.out_net.6.bias", (32,)), (f"{_AE_E}.fc_mu.weight", (32, 32)), (f"{_AE_E}.fc_mu.bias", (32,)),
              (f"{_AE_D}.pre_net.0.weight", (64, 32)), (f"{_AE_D}.pre_net.0.bias", (64,)), (f"{_AE_D}.pre_net.1.weight", (64,)), (f"{_AE_D}.pre_net.1.bias", (64,)),
              (f"{_AE_D}.pre_net.3.weight", (136, 64)), (f"{_AE_D}.pre_net.3.bias", (136,)),
              (f"{_AE_D}.net.0.weight", (4, 32, 3)), (f"{_AE_D}.net.0.bias", (32,)), (f"{_AE_D}.net.1.weight", (32,)), (f"{_AE_D}.net.1.bias", (32,)),
              (f"{_AE_D}.net.3.weight", (32, 32, 3)), (f"{_AE_D}.net.3.bias", (32,)), (f"{_AE_D}.net.4.weight", (32,)), (f"{_AE_D}.net.4.bias", (32,)),
              (f"{_AE_D}.net.6.weight", (32, 32, 3)), (f"{_AE_D}.net.6.bias", (32,)), (f"{_AE_D}.net.7.weight", (27, 32, 3)), (f"{_AE_D}.net.7.bias", (27,))])
AE_BNS = [f"{_AE_E}.net.0.1", f"{_AE_E}.net.1.1", f"{_AE_E}.net.2.1", f"{_AE_E}.out_net.1", f"{_AE_E}.out_net.4", f"{_AE_D}.pre_net.1", f"{_AE_D}.net.1", f"{_AE_D}.net.4"]
assert len(AE_PARAMS) == 44


class AeStep:
    """The argument block and workspace of tg_ae_train_step for one parameter slab and batch size (built once; safe to capture)."""

    @staticmethod
    def supported(slab, B, poses_shape):
        if tuple(poses_shape[1:]) != (34, 27) or not _lib.load().tg_ae_step_supported(int(B)):
            return False
        shapes = {n: tuple(p.shape) for n, p in zip(slab.names, slab.params)}
        known = {n for n, _ in AE_PARAMS} | {f"{_AE_E}.fc_logvar.weight", f"{_AE_E}.fc_logvar.bias"}
        return all(shapes.get(n) == s for n, s in AE_PARAMS) and set(shapes) <= known and not slab.frozen

    @staticmethod
    def key(slab, buffers, B):
        """Every raw device pointer the argument block holds (+ the batch size): the plan is valid exactly while this is unchanged -- a
        re-allocated slab, gradient slab, step counter or BatchNorm buffer (load_state_dict(assign=True), module re-materialisation) gives a
        new key instead of a step that writes through stale pointers."""
        bufs = tuple(buffers[bn + sfx].data_ptr() for bn in AE_BNS for sfx in (".running_mean", ".running_var", ".num_batches_tracked"))
        return (int(B), slab.flat.data_ptr(), slab.grad.data_ptr(), slab.step.data_ptr()) + bufs

    def __init__(self, slab, buffers, B):
        dev = slab.flat.device
        self.slab_ptr, self.B = slab.flat.data_ptr(), int(B)
        self.cache_key = AeStep.key(slab, buffers, B)
        off = dict(zip(slab.names, slab.offsets))
        q = _lib.AeStepArgs()
        q.params, q.grads = slab.flat.data_ptr(), slab.grad.data_ptr()
        for i, (n, _) in enumerate(AE_PARAMS):
            q.off[i] = off[n]
        self._keep = []
        for i, bn in enumerate(AE_BNS):
            rm, rv, nbt = buffers[bn + ".running_mean"], buffers[bn + ".running_var"], buffers[bn + ".num_batches_tracked"]
            assert rm.is_cuda and rm.dtype == torch.float32 and nbt.dtype == torch.int64
            q.running_mean[i], q.running_var[i], q.num_batches_tracked[i] = rm.data_ptr(), rv.data_ptr(), nbt.data_ptr()
            self._keep += [rm, rv, nbt]
        nbytes = _lib.load().tg_ae_step_ws_bytes(self.B)
        self.ws = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=dev)
        q.ws, q.ws_bytes = self.ws.data_ptr(), self.ws.numel() * 4
        q.step = slab.step.data_ptr()
        q.B, q.bn_eps, q.momentum = self.B, 1e-5, 0.1
        self.q = q

    def run(self, x, loss, recon=None, feat=None, last_phase=0, adam=None):
        """adam = (m slab, v slab, lr, beta1, beta2, eps): the last launch also takes the optimiser step (torch.optim.Adam) -- the step is then
        complete; None: gradients only (the caller runs FusedAdam.step(counter_advanced=True))."""
        _flat(x, "poses"); assert tuple(x.shape) == (self.B, 34, 27)
        q = self.q
        if adam is None:
            q.adam_m = q.adam_v = None
        else:
            m, v, lr, b1, b2, eps = adam
            assert m.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0 and m.numel() == v.numel()
            q.adam_m, q.adam_v, q.lr, q.beta1, q.beta2, q.adam_eps = m.data_ptr(), v.data_ptr(), float(lr), float(b1), float(b2), float(eps)
        q.x, q.loss = x.data_ptr(), _flat(loss, "loss").data_ptr()
        q.recon = None if recon is None else _flat(recon, "recon").data_ptr()
        q.feat = None if feat is None else _flat(feat, "feat").data_ptr()
        q.last_phase = int(last_phase)
        call("tg_ae_train_step", C.byref(q), _stream())

    def workspace_views(self):
        """(sums [17, 2, 256] fp64, act [B, AE_ACT], part [B, AE_PART]) views of the workspace (tests)."""
        n_act, n_part = 10256, 40700
        assert _lib.load().tg_ae_step_ws_bytes(self.B) == 17 * 512 * 8 + (self.B * (n_act + n_part) + 146944) * 4      # (+ the transposed linear weights)
        f = self.ws.view(torch.float32)
        sums = self.ws[:17 * 512 * 2].view(torch.float64).view(17, 2, 256)
        act = f[17 * 512 * 2:17 * 512 * 2 + self.B * n_act].view(self.B, n_act)
        part = f[17 * 512 * 2 + self.B * n_act:17 * 512 * 2 + self.B * (n_act + n_part)].view(self.B, n_part)
        return sums, act, part


# ------------------------------------------------------------------------------------------------- optimiser
def counter_inc(counter):
    assert counter.is_cuda and counter.dtype == torch.int32
    call("tg_counter_inc", _p(counter), _stream())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step_dev):
    n = _same(p, g, m, v)
    assert step_dev.is_cuda and step_dev.dtype == torch.int32
    call("tg_adam_step", _p(p), _p(g), _p(m), _p(v), n, float(lr), float(beta1), float(beta2), float(eps), _p(step_dev), _stream())
