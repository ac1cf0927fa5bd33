/*
 * trimodal_hip.h -- C ABI of libtrimodal_hip.so (gfx950 / MI355X kernels for the trimodal gesture GAN path).
 *
 * The reference (ai4r/Gesture-Generation-from-Trimodal-Context) has no FFI: its hot path is PyTorch
 * nn.Modules whose arithmetic runs in ATen.  Each entry point below replaces the ATen
 * kernel behind one reference call site; the call site is cited per function as file:line relative to
 * /root/reference/scripts/.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*.
 *   - every function returns 0 on success, non-zero on error; tg_last_error() gives the message
 *     (thread-local).  No C++ exception crosses the ABI.
 *   - all device memory is owned by the caller.  Functions never allocate, never synchronise the device
 *     and are safe to call during hipGraph stream capture.
 *   - all tensors are fp32, row-major, "channel-last" for sequences: (batch, time, channels).
 *   - functions documented "accumulates" add into their output; the caller zeroes it when needed.
 */
#ifndef TRIMODAL_HIP_H
#define TRIMODAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 8

int tg_version(void);

/* Deterministic mode (ABI 5).  Scope: the GAN TRAINING ITERATION (train_eval/train_gan.py:13-103 as train_gan.GanTrainer issues it).
 * on != 0: every combine across workgroups on that path runs in a fixed order -- weight-gradient splits through the two-pass
 * fp64 reduce (the caller passes a workspace for every problem and takes bias gradients through tg_colsum), embedding scatters with one
 * writer per table row, the discriminator head's parameter gradients and tg_colsum by one workgroup, BatchNorm partial sums in thread order --
 * so two runs from the same state give bit-identical results (the reference on CPU is reproducible given a seed; float atomics are not).
 * The fused discriminator front-end backward (tg_d_preconv_bwd) and the fused speaker backward (tg_speaker_bwd) combine by float atomics and
 * are not to be called in this mode (the host mirror takes their generic forms).  Process-wide, like tg_set_math_mode.
 * NOT covered (they keep their atomics whatever the mode): the autoencoder path (tg_ae_train_step's fp64 BatchNorm / loss sums, tg_ae_loss),
 * tg_l1_mean and the single-launch BatchNorm statistics kernels for tensors under 16 K elements (tg_bn_train_stats / tg_bn_backward: fp64
 * atomics, whose order-dependence is below fp32 resolution but not zero).  tests/test_trajectory_gpu.py pins the guarantee at B = 64. */
int tg_set_deterministic(int32_t on);
int tg_get_deterministic(void);
/* Workgroup cap of the persistent weight-gradient kernel (tg_gemm_tn / tg_gemm_tn_group on the mover-wave kernel; process-wide, 0 = none):
 * with n > 0 those launches are planned for, and occupy, at most n CUs -- for a launch issued on a second stream BESIDE a kernel that needs
 * the other CUs to itself (the cluster-synchronised GRU backward keeps 160 of 256; train_eval/train_gan.py:89 is the backward it belongs
 * to).  A group the capped plan cannot take is refused by tg_gemm_tn_kernel_plan (code 0) exactly as without the cap; callers check that
 * before they fork. */
int tg_set_tn_workgroup_cap(int32_t n);
int tg_get_tn_workgroup_cap(void);
const char* tg_last_error(void);

/* Math mode of the GEMM-shaped kernels (process-wide, like cublasSetMathMode):
 *   0 (default): fp32-accurate products, parity with the reference to ~1e-6.  Small products run on the f32 matrix cores
 *      (v_mfma_f32_16x16x4_f32, exact fp32 fma chains); the big forward / input-gradient products (tg_gemm_nt, M >= 1024) split
 *      each fp32 operand exactly into three bf16 terms and accumulate the six significant partial products in fp32 on the
 *      bf16 matrix cores (csrc/gemm_split.hip): same error as the f32 path, 1.2-1.8x its speed.  Environment TG_GEMM_X3=0
 *      keeps everything on the f32 matrix cores.
 *   1: those big products round their operands to bf16 (RNE) instead, one MFMA per product, fp32 accumulate; everything in HBM
 *      stays fp32, weight gradients and the recurrences stay fp32.  The precision BASELINE.json configs[1] names; tolerance
 *      2e-2 forward / 5e-2 gradients. */
int tg_set_math_mode(int32_t mode);
int tg_get_math_mode(void);

/*
 * A "row window" view of a channel-last activation buffer: the A operand of every GEMM-shaped op.
 * Logical matrix element (m, k), m in [0, M), k in [0, K):
 *     b  = m / rows_out,  r = m % rows_out            (clip b, output row r inside the clip)
 *     kk = k / cw,        c = k % cw                  (tap kk, channel c)
 *     sr = r * row_step + shift + kk * dil            (source row inside the clip)
 *     value = (0 <= sr < rows_in) ? ptr[b * batch_stride + sr * row_stride + c] : 0
 * A plain matrix is {rows_out = rows_in = M, row_step = 1, shift = 0, cw = K}.  A Conv1d(kernel kw, stride s,
 * padding p, dilation d) over (B, L, C) is {cw = C, K = kw*C, row_step = s, shift = -p, dil = d}: im2col is never
 * materialised.
 */
typedef struct tg_window {
    const float* ptr;
    int64_t batch_stride;
    int64_t row_stride;
    int32_t rows_in;
    int32_t rows_out;
    int32_t row_step;
    int32_t shift;
    int32_t dil;
    int32_t cw;
    int32_t K;
} tg_window;

/* ---- GEMM-shaped ops (f32 MFMA v_mfma_f32_16x16x4_f32, exact fp32 accumulate) ------------------------------
 * tg_gemm_nt: C(m, n) = act( sum_k A(m,k) * Bw[n*ldb + k] + bias[n] ) (+ C(m,n) when accumulate != 0)
 *   act(x) = x >= 0 ? x : act_slope * x   (1.0 = identity, 0.0 = ReLU, 0.3/0.2 = LeakyReLU)
 *   C row m is stored at C + (m / c_rows_out) * c_batch_stride + (m % c_rows_out) * c_row_stride.
 *   Replaces nn.Linear / nn.Conv1d / nn.ConvTranspose1d forward and their input-gradients:
 *   model/multimodal_context_net.py:13-22,51,89-93,100-104,214-220,225-226; model/tcn.py:19,25;
 *   the x @ W_ih^T half of nn.GRU (:98,223); model/embedding_net.py:24,48-62,187-206. */
typedef struct tg_gemm_nt_problem {
    tg_window A;
    const float* Bw;
    int64_t ldb;
    int32_t b_seg_k;           /* 0: one weight matrix.  > 0 (a divisor of K): K-concatenated weights, k in [s*b_seg_k, (s+1)*b_seg_k) */
    int32_t reserved;          /*    reads Bw + s*b_seg_stride + n*ldb + (k - s*b_seg_k) -- e.g. both GRU directions' W_ih^T side by side */
    int64_t b_seg_stride;
    const float* bias;
    float* C;
    int64_t c_batch_stride, c_row_stride;
    int32_t c_rows_out, M, N;
    float act_slope;
    int32_t accumulate;
    const float* out_scale;    /* NULL, or an element-wise multiplier applied after the activation, addressed like C: the inverted-dropout
                                  scale mask of F.dropout(relu(conv(x))) (model/tcn.py:22-29) rides in the epilogue */
    const void* b_planes;      /* NULL, or Bw PRE-SPLIT: plane 0 of a slab-tiled bf16 x 3 plane buffer (see tg_split3_planes below) of b_rows rows */
    int64_t b_plane_stride;    /*   and K columns whose rows b_row0 .. b_row0 + N - 1 are Bw's (planes b_plane_stride elements apart; one weight */
    int32_t b_rows, b_row0;    /*   matrix only: b_seg_k == 0).  Many-row products then run on the mover-wave kernel (csrc/gemm_mw.hip): the weights go
                                    global -> LDS by DMA, only the activation is split while staged.  Ignored by the other kernels (Bw is still read). */
    /* Epilogue extensions (big-product path only, tg_gemm_nt_ext_supported): what the reference computes right after the product in
     * model/tcn.py:27-45 without another pass over the tensor.  All three are addressed like C.
     *   gate : the result (after act / out_scale) is kept where gate > 0 and zeroed elsewhere -- ReLU backward through the tensor the
     *          product's consumer saw (the chain rule of relu + dropout of the PREVIOUS conv, applied to this input gradient);
     *   res, C2 (together): second output C2 = act2(C_new + res), act2 = leaky-ReLU of slope res_slope (0: ReLU) -- the residual
     *          block's relu(out + x) written next to out, which the backward still needs. */
    const float* gate;
    const float* res;
    float* C2;
    float res_slope;
    /* ABI 4 -- the dropout scale REGENERATED instead of read (no mask tensor, no draw launch): with drop_state != NULL (out_scale must be
     * NULL) the multiplier of the element at offset o from C (in elements, o as C is addressed) is element drop_index0 + o of the draw
     * tg_dropout_mask(mask, n, drop_p, drop_state, drop_site) would write -- Philox4x32-10 keyed by the element index, so forward and
     * backward consumers of one F.dropout (model/tcn.py:22-29) see the same mask without storing it.  Needs the big-product path
     * (tg_gemm_nt_ext_supported), a vectorisable C and drop_index0 % 4 == 0; refused with an error elsewhere. */
    uint32_t drop_site;
    const uint64_t* drop_state;
    int64_t drop_index0;
    float drop_p;
    int32_t reserved4;
    /* ABI 7 -- fp16 x 2 operands (three matrix instructions per product instead of bf16 x 3's six, csrc/common.hpp "two-term fp16 split"):
     * b_planes_kind == 1 says b_planes is the TWO-plane fp16 buffer tg_split2h_planes writes (hi / lo of every row scaled by its own power of
     * two) and b_inv_scale its per-row inverse scales (b_rows + 1 floats, 16-byte aligned, b_row0 % 4 == 0); a_row_scale (or a_rowmax, below) then holds the
     * power-of-two scale of every product row m < M of the window A (M floats, tg_h2_row_scales: the row's largest magnitude over its K values
     * scaled into [2^14, 2^15)) by which the kernel multiplies the row before splitting it.  b_planes_kind == 0: bf16 x 3
     * planes as before, both pointers ignored.  Same arithmetic contract as before: fp32 nn.Linear / nn.Conv1d / nn.GRU input projections
     * (model/multimodal_context_net.py:98-104, model/tcn.py:19-25) within the fp32 tolerance. */
    int32_t b_planes_kind;
    int32_t reserved5;
    const float* b_inv_scale;
    const float* a_row_scale;
    /* ... or, for windows of one or two taps, a_rowmax: the largest magnitude of every SOURCE row of A's tensor (index batch * rows_in + source row,
     * non-negative floats: tg_win_row_absmax, or the product that wrote the tensor through c_rowmax below); the kernel derives the product rows'
     * scales from it.  Exactly one of a_row_scale / a_rowmax with fp16 x 2 planes.
     * c_rowmax / c2_rowmax (optional, mover-wave kernel only -- refused elsewhere): M floats each; row m's entry is raised (atomic unsigned max on the
     * float's bits) to the largest magnitude the product wrote into row m of C / C2.  The caller zeroes them before the first product of a pass;
     * a chain of convs (model/tcn.py:27-46) then needs no separate pass over an activation to scale it. */
    const float* a_rowmax;
    float* c_rowmax;
    float* c2_rowmax;
    int32_t a_rowmax_rows;     /* 0 / 1: a_rowmax has one entry per source row; n > 1: one entry per n consecutive source rows (entry (batch * rows_in +
                                  source row) / n) -- e.g. one per clip of T frames, as tg_gru_backward_cluster_stats leaves them */
    int32_t reserved6;
} tg_gemm_nt_problem;
/* tg_gemm_nt_group: up to 8 independent tg_gemm_nt products in ONE launch (both GRU directions' input projections, the stride
 * phases of a conv input-gradient ...).  All problems must fall into the same kernel family as problem 0 (big / narrow / small);
 * outputs must not overlap.  The table is copied into the kernel arguments: nothing has to outlive the call. */
int tg_gemm_nt_group(const tg_gemm_nt_problem* problems, int32_t n, void* stream);
/* kernel family a problem would run in (0 big, 1 narrow N <= 32, 2 small; -1 invalid): problems of one group must agree */
int32_t tg_gemm_nt_family(const tg_gemm_nt_problem* problem);
/* 1 when this problem would run on a kernel that implements gate / res / C2 (family 0 on the bf16 x 3 or bf16 path), else 0 */
int32_t tg_gemm_nt_ext_supported(const tg_gemm_nt_problem* problem);
/* Which kernel a group of n problems would run on, without launching: 0 = f32-MFMA / narrow / small kernels (gemm.hip),
 * 1 = bf16 x 3 staged-slab kernel (gemm_split.hip), 2 = bf16 x 3 mover-wave kernel (gemm_mw.hip: 512-thread workgroups, big tiles,
 * chosen when its tiles fill the chip); -1 invalid.  *tile_m / *tile_n (may be NULL) receive the workgroup tile for plan 2.
 * Replaces nothing in the reference (torch picks its own GEMM there, multimodal_context_net.py:98-99, model/tcn.py:19-46): test
 * and profiling aid, so that a parity test can assert WHICH kernel it covered. */
int32_t tg_gemm_nt_kernel_plan(const tg_gemm_nt_problem* problems, int32_t n, int32_t* tile_m, int32_t* tile_n);
/* Process-wide switch of the mover-wave kernel: 1 on, 0 off (every big product on gemm_split.hip), -1 back to the environment default
 * (TG_NT_MW, on).  For same-process A/B timing (tools/nt_mw_probe.py) and tests; results are equal within the fp32 tolerance either way. */
int tg_set_nt_mover_waves(int32_t on);
int tg_gemm_nt(const tg_window* A, const float* Bw, int64_t ldb, const float* bias, float* C,
               int64_t c_batch_stride, int64_t c_row_stride, int32_t c_rows_out, int32_t M, int32_t N,
               float act_slope, int32_t accumulate, void* stream);

/* ---- pre-split operand planes (csrc/planes.hip).  Plane buffer of an fp32 matrix [rows][cw]: three bf16 planes (x = hi + mid + lo exactly)
 * `plane_stride` ELEMENTS apart, each SLAB-TILED: [cwp / 32][rows + 1][32] (cwp = cw rounded up to a multiple of 32; element (r, c) at
 * ((c / 32) * (rows + 1) + r) * 32 + c % 32; zero past cw and in row `rows` of every slab), so that one 32-deep K slab of 16 consecutive rows
 * is 1 KB of contiguous memory.  tg_gemm_nt_problem.b_planes takes weights in this form (mover-wave kernel, csrc/gemm_mw.hip). */
int tg_split3_planes(const float* x, int64_t ldx, int32_t rows, int32_t cw, void* planes, int32_t cwp, int64_t plane_stride, void* stream);
/* fp16 x 2 form of the same buffer (ABI 7): TWO fp16 planes, same slab tiling, holding hi = fp16(x s_r), lo = fp16(x s_r - hi) with s_r the
 * power of two that puts row r's largest magnitude into [2^14, 2^15); inv_scale[r] = 1 / s_r for r < rows, 0 for the zero row (rows + 1 floats).
 * Weights of nn.GRU / weight-normed nn.Conv1d as the matrix cores' fp16 operand (multimodal_context_net.py:98-99, model/tcn.py:19-25). */
int tg_split2h_planes(const float* x, int64_t ldx, int32_t rows, int32_t cw, void* planes, int32_t cwp, int64_t plane_stride, float* inv_scale, void* stream);
/* The same planes for the K-CONCATENATED TRANSPOSE of two [rows][cols] matrices: plane row n < cols, column k < 2 rows holds w{k / rows}[k % rows][n]
 * (cwp >= 2 rows; inv_scale: cols + 1 floats) -- the weight operand of the GRU layer's input gradient dx = [dgi_fwd | dgi_rev] @ [W_ih_fwd ; W_ih_rev]
 * as one product over K = 6H, straight from the two weight_ih parameters (nn.GRU backward, model/multimodal_context_net.py:98-99). */
int tg_split2h_planes_tcat(const float* w0, const float* w1, int32_t rows, int32_t cols, void* planes, int32_t cwp, int64_t plane_stride, float* inv_scale,
                           void* stream);
/* rowmax[b * A->rows_in + r] = max_c |A->ptr[b * batch_stride + r * row_stride + c]|, c < A->cw, for b < batches, r < A->rows_in: the largest magnitude
 * of every SOURCE row of a window operand (one pass over the tensor; several windows over one tensor share it). */
int tg_win_row_absmax(const tg_window* A, int32_t batches, float* rowmax, void* stream);
/* row_scale[m] = 2^(141 - e_m), e_m the biased fp32 exponent (clamped to [32, 250]) of the largest magnitude among the K values of product row m of
 * the window A -- tg_gemm_nt_problem.a_row_scale.  src_rowmax == NULL: read from the tensor itself (a row with t taps is read t times);
 * else from tg_win_row_absmax's output for that tensor (batch b = m / rows_out, source rows (m % rows_out) * row_step + shift + tap * dil). */
int tg_h2_row_scales(const tg_window* A, int32_t M, const float* src_rowmax, float* row_scale, void* stream);
/* One pass over an fp32 matrix x[M][C] (row stride ldx): rowmax[m] = largest magnitude of row m (M floats, or NULL); colmax[g * C + c] = largest
 * magnitude of column c within row group g (the M rows cut into `groups` equal consecutive parts, e.g. the two directions of a GRU layer's gate
 * gradients [2][B * T][3H]; groups * C floats, or NULL).  colmax is zeroed and then raised by atomic unsigned max.  Feeds a_rowmax of
 * tg_gemm_nt_problem and y_colmax / a_colmax of tg_gemm_tn_problem when the tensor's producer does not supply them. */
int tg_absmax_rows_cols(const float* x, int64_t ldx, int32_t M, int32_t C, int32_t groups, float* rowmax, float* colmax, void* stream);

/* tg_gemm_tn (weight gradient, accumulates): dW[n*ldw + perm(k)] += sum_m dY[m*ldy + n] * A(m, k).
 *   out_kw == 0: perm(k) = k.  out_kw == K/cw: perm(k) = (k % cw) * out_kw + k / cw, i.e. the gradient lands in
 *   the (Cout, Cin, kw) layout of nn.Conv1d weights.
 *   The sum over m is split across workgroups.  ws == NULL: partial tiles are combined with f32 atomics (order
 *   varies run to run).  ws != NULL (>= tg_gemm_tn_ws_floats(M, N, K) floats): partials of <= 512 rows go to ws and are
 *   combined in split order in fp64 -- bitwise reproducible, and accurate for the ~1e6-row sums of the audio encoder.
 *   Replaces the weight-gradient half of aten::convolution_backward / addmm backward for the same call sites. */
int64_t tg_gemm_tn_ws_floats(int32_t M, int32_t N, int32_t K);
/*   dbias != NULL: also dbias[n] += sum_m dY[m*ldy + n] (the bias gradient rides along; no separate pass over dY). */
typedef struct tg_gemm_tn_problem {
    const float* dY;
    int64_t ldy;
    tg_window A;
    float* dW;
    int64_t ldw;
    int32_t M, N, out_kw, reserved;
    float* dbias;
    float* ws;
    int64_t ws_floats;
    /* ABI 7 -- fp16 x 2 operands on the mover-wave kernel (both non-NULL, 16-byte aligned; ignored by the other kernels, which read the fp32
     * operands): y_colmax[n] = largest magnitude of column n of dY over the M rows (N floats), a_colmax[c] = largest magnitude of channel c of A's
     * tensor (cw floats, valid for every tap) -- tg_absmax_rows_cols or the kernels that wrote the tensors.  The sum runs over rows, so each
     * COLUMN is scaled by its own power of two; the combine scales the sums back.  Same contract: the weight-gradient half of
     * aten::convolution_backward / addmm backward within the fp32 tolerance. */
    const float* y_colmax;
    const float* a_colmax;
} tg_gemm_tn_problem;
/* tg_gemm_tn_group: up to 8 independent weight gradients in ONE launch (the four of a GRU layer: W_ih / W_hh of both directions). */
int tg_gemm_tn_group(const tg_gemm_tn_problem* problems, int32_t n, void* stream);
/* which kernel the group would run on, without launching: 0 = f32-MFMA tiles, 1 = bf16 x 3 staged slabs (gemm_split.hip), 2 = bf16 x 3 mover
 * waves (gemm_tn_mw.hip: persistent 768-thread workgroups, 192 x 160 tiles); -1 invalid.  Test / profiling aid like tg_gemm_nt_kernel_plan. */
int32_t tg_gemm_tn_kernel_plan(const tg_gemm_tn_problem* problems, int32_t n);
int tg_gemm_tn(const float* dY, int64_t ldy, const tg_window* A, float* dW, int64_t ldw, int32_t M, int32_t N,
               int32_t out_kw, float* dbias, float* ws, int64_t ws_floats, void* stream);

/* tg_colsum (bias gradient): out[n] (+)= sum_m X[m*ldx + n]. */
int tg_colsum(const float* X, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate, void* stream);

/* ---- GRU recurrence (nn.GRU, batch_first, bidirectional; model/multimodal_context_net.py:98-99,155,223-224,241)
 * gi   : [2][B][T][3H] input projections x@W_ih^T + b_ih per direction (dir stride in floats given explicitly)
 * y    : [B][T][2H]    layer output; direction d writes columns [d*H, (d+1)*H); also the recurrent state store
 * save : [2][B][T][4H] r, z, n, (W_hn h + b_hn) per step for the backward pass, or NULL (inference / no-grad)
 * Gate order and maths are PyTorch's: r,z = sigmoid, n = tanh(gi_n + r*(W_hn h + b_hn)), h' = (1-z) n + z h.
 * H != 64: one launch per time step (both directions, all batch rows); the launch boundary is the grid-wide dependency.
 * H == 64: one persistent launch for the whole sequence (tg_gru_h64_forward without the fused dropout). */
int tg_gru_forward(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                   const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                   int32_t B, int32_t T, int32_t H, void* stream);

/* Backward through time.  dy: [B][T][2H] gradient w.r.t. y.  w_hh_t_*: TRANSPOSED recurrent weights [H][3H].
 * Outputs dgi, dgh: [2][B][T][3H] gradients w.r.t. the input-side and hidden-side gate pre-activations
 * (feed tg_gemm_tn / tg_gemm_nt / tg_colsum for dW_ih, dW_hh, dx, db).  dh_scratch: 4*B*H floats. */
int tg_gru_backward(const float* dy, const float* y, const float* save, int64_t save_dir_stride,
                    const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                    float* dh_scratch, int32_t B, int32_t T, int32_t H, void* stream);

/* H = 64 recurrence (the discriminator's GRU): one persistent launch per layer, recurrent product on the bf16 matrix cores at
 * fp32 accuracy (csrc/gru_h64.hip).  Same tensors as tg_gru_forward / tg_gru_backward plus the fused inter-layer dropout of
 * nn.GRU(dropout=p):
 *   forward : drop_mask [B][T][128] (inverted-dropout scale mask, 0 or 1/(1-p)) and y_drop [B][T][128] -- both or neither;
 *             y_drop = y * drop_mask is the next layer's input (y itself stays the recurrent state / backward operand).
 *   backward: dy_mask [B][T][128] or NULL: the incoming gradient is multiplied by it while it is loaded (the gradient w.r.t.
 *             y_drop becomes the gradient w.r.t. y).  No dh_scratch: the carried dh stays in registers. */
int tg_gru_h64_forward(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                       const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                       const float* drop_mask, float* y_drop, int32_t B, int32_t T, void* stream);
int tg_gru_h64_backward(const float* dy, const float* dy_mask, const float* y, const float* save, int64_t save_dir_stride,
                        const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                        int32_t B, int32_t T, void* stream);

/* Persistent, cluster-synchronised variant of tg_gru_forward for H <= 320 (the generator): ONE launch walks all T
 * steps of both directions; the workgroups that share a batch tile exchange h_t through `ws` with write-through stores and
 * per-workgroup flag words, no grid-wide barrier (csrc/gru_cluster.hip).  Same arguments and results as tg_gru_forward.
 * ws: tg_gru_cluster_ws_bytes(B, H) bytes of device memory, 16-byte aligned, ZERO-FILLED ONCE by the caller when allocated and
 * then left to the library: the flag words behind the timeout block are numbered by a per-cluster generation that persists from
 * launch to launch (no fill kernel per launch; re-zeroing the WHOLE workspace between launches is allowed, part of it is not).  Its
 * first 16 words are a sticky TIMEOUT block: word 0 is set by the kernel if a bounded spin expires (results are then invalid),
 * words 1.. hold diagnostics.  No launch ever clears them -- a timeout in any launch sharing the workspace stays visible until
 * the caller reads word 0 back (after synchronising) and clears it itself.
 * tg_gru_cluster_supported(B, H) != 0 iff the launch fits co-resident at one workgroup per CU on the current device (its CU
 * count is queried: B <= 384 at H = 300 on the 256 CUs of an unpartitioned MI355X; 0 under CPX/DPX partitions that are too small). */
int32_t tg_gru_cluster_supported(int32_t B, int32_t H);
int64_t tg_gru_cluster_ws_bytes(int32_t B, int32_t H);
/* drop_mask / y_drop ([B][T][2H], both or neither; need tg_gru_cluster_fused_dropout() != 0): the inter-layer dropout of
 * nn.GRU(dropout=p) rides in the kernel's output stage, y_drop = y * drop_mask is the next layer's input. */
int32_t tg_gru_cluster_fused_dropout(void);
int tg_gru_forward_cluster(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                           const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                           const float* drop_mask, float* y_drop, void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, void* stream);
/* The same, saving the gates only for batch rows [save_row0, save_row0 + save_rows): of the stacked generator calls of a GAN iteration
 * (train_gan.py:30,50,67) only one is differentiated; the other rows of `save` are left untouched. */
int tg_gru_forward_cluster_rows(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                                const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                                const float* drop_mask, float* y_drop, void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H,
                                int32_t save_row0, int32_t save_rows, void* stream);

/* Inference recurrence for a handful of sequences (1 <= B <= 4, 64 < H <= 320): the single-utterance synthesis window of
 * scripts/synthesize.py:131-160 (one 34-frame window per generator call; multimodal_context_net.py:155, nn.GRU bidirectional, eval mode: no
 * saved gates, no dropout).  h_t is exchanged as fp32 words that are their own flags (sentinel-polled), the product is fp32 FMAs on resident
 * fp32 weights: no operand split at all, results at fp32 rounding of the reference's (csrc/gru_vec.hip).  Same operand layout as
 * tg_gru_forward.  Workspace: tg_gru_vec_ws_bytes(H) bytes whose first tg_gru_vec_ws_header_bytes() are ZERO and every byte behind them 0xFF
 * before the first use (and again after a timeout: word 0 is the sticky timeout word of tg_gru_forward_cluster's convention). */
int32_t tg_gru_vec_supported(int32_t B, int32_t H);
int64_t tg_gru_vec_ws_bytes(int32_t H);
int32_t tg_gru_vec_ws_header_bytes(void);
int tg_gru_forward_vec(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev, const float* b_hh_fwd,
                       const float* b_hh_rev, float* y, void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, void* stream);

/* Persistent cluster-synchronised variant of tg_gru_backward (B <= 192 at H = 300; no dh_scratch: the carried dh stays in
 * registers).  Same workspace / timeout-word convention as tg_gru_forward_cluster. */
int32_t tg_gru_cluster_bwd_supported(int32_t B, int32_t H);
int64_t tg_gru_cluster_bwd_ws_bytes(int32_t B, int32_t H);
/* dy_mask ([B][T][2H] or NULL): multiplied into dy while it is loaded (the backward of the fused dropout). */
int tg_gru_backward_cluster(const float* dy, const float* dy_mask, const float* y, const float* save, int64_t save_dir_stride,
                            const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                            void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, void* stream);
/* The same launch, also leaving the magnitudes of what it wrote for the fp16 x 2 products that read dgi / dgh next (ABI 7; all three or none, zeroed
 * by the caller, raised by atomic unsigned max when the kernel leaves, so row chunks of one pass accumulate): gi_rowmax[dir * rowmax_dir_stride + b] =
 * largest |dgi| of batch row b over its T steps (a_rowmax of the input-gradient product dx = dgi @ W_ih with a_rowmax_rows = T: one scale per
 * clip), gi_colmax / gh_colmax[dir * 3H + c] = largest |dgi| / |dgh| of column c
 * (y_colmax of the weight-gradient products).  Replaces nothing in the reference beyond tg_gru_backward_cluster's nn.GRU backward
 * (model/multimodal_context_net.py:98-99). */
int tg_gru_backward_cluster_stats(const float* dy, const float* dy_mask, const float* y, const float* save, int64_t save_dir_stride,
                                  const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                                  void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, float* gi_rowmax, int64_t rowmax_dir_stride,
                                  float* gi_colmax, float* gh_colmax, void* stream);

/* ---- BatchNorm1d, channel-last [rows][C] (model/multimodal_context_net.py:14,17,20,215,218) -------------
 * Training statistics per group: the rows are split into `groups` equal consecutive slabs, each normalised with
 * its own batch statistics (several reference forward calls stacked into one launch); running stats are updated
 * once per group in order (momentum 0.1, unbiased variance), num_batches_tracked += groups * repeats.
 * repeats >= 1: the running update of each group is applied that many times (the same batch seen by `repeats`
 * identical reference forward calls: the three generator forwards of one GAN iteration share their audio input).
 * ws: 2*groups*C doubles of scratch.  mean/rstd: [groups][C] outputs. */
int tg_bn_train_stats(const float* x, int32_t rows, int32_t C, int32_t groups, double* ws, float* mean, float* rstd,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                      float momentum, int32_t repeats, void* stream);
/* eval mode: mean/rstd from running stats. */
/* Small tensors (the discriminator's and the autoencoder's BatchNorms): statistics, running-stat update AND y = act(gamma * xhat +
 * beta) in ONE single-workgroup launch; y == NULL computes the statistics only.  tg_bn_fused_supported: rows * C <= 2^19, C a
 * multiple of 4 that divides 4096 (16-byte accesses with fixed channels per thread); x / y 16-byte aligned. */
int32_t tg_bn_fused_supported(int32_t rows, int32_t C, int32_t groups);
int tg_bn_train_fused(const float* x, float* y, int32_t rows, int32_t C, int32_t groups, float* mean, float* rstd,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                      const float* beta, float act_slope, float eps, float momentum, int32_t repeats, void* stream);
int tg_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps, float* mean,
                     float* rstd, void* stream);
/* y = act((x - mean[g]) * rstd[g] * gamma + beta). */
int tg_bn_apply(const float* x, float* y, int32_t rows, int32_t C, int32_t groups, const float* mean,
                const float* rstd, const float* gamma, const float* beta, float act_slope, void* stream);
/* backward of act(BN(x)) for one group: dx, and dgamma/dbeta (accumulate).  ws: 2*C doubles.
 * mean==NULL selects eval-mode backward is not supported (training only). */
int tg_bn_backward(const float* dy, const float* x, float* dx, int32_t rows, int32_t C, const float* mean,
                   const float* rstd, const float* gamma, const float* beta, float act_slope, double* ws,
                   float* dgamma, float* dbeta, void* stream);

/* Two-launch forms for tensors past the single-workgroup size: x [groups * rows_per_group][C] holds `groups` stacked forward calls, each
 * normalised with its own batch statistics (mean / rstd [groups][C]); no zero fill, no atomics, deterministic.  ws: tg_bn2_ws_doubles
 * doubles of scratch.  tg_bn2_train: statistics, running-stat updates in call order (each `repeats` times) and y = act(BN(x)) (y may be
 * NULL: statistics only); tg_bn2_backward: dx for all groups and dgamma / dbeta += the groups' sums (either may be NULL).
 * Supported: C a multiple of 4 that divides 256, rows_per_group * C a multiple of 4, 16-byte aligned pointers. */
int32_t tg_bn2_supported(int32_t rows_per_group, int32_t C);
int64_t tg_bn2_ws_doubles(int32_t rows_per_group, int32_t C, int32_t groups);
int tg_bn2_train(const float* x, float* y, int32_t rows_per_group, int32_t C, int32_t groups, double* ws, int64_t ws_doubles, float* mean,
                 float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                 const float* beta, float act_slope, float eps, float momentum, int32_t repeats, void* stream);
int tg_bn2_backward(const float* dy, const float* x, float* dx, int32_t rows_per_group, int32_t C, int32_t groups, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, float act_slope, double* ws, int64_t ws_doubles,
                    float* dgamma, float* dbeta, void* stream);

/* ---- ConvDiscriminator.pre_conv, train-mode forward in ONE launch (model/multimodal_context_net.py:214-220: Conv1d(27,16,3) -> BN(16) ->
 * LeakyReLU(True) = identity -> Conv1d(16,8,3) -> BN(8) -> identity -> Conv1d(8,8,3); poses [Bs][34][27] channel-last, `groups` stacked
 * forward calls with their own batch statistics).  Writes what the separate launches write: the conv outputs c1 [Bs][32][16], c2 [Bs][30][8],
 * c3 [Bs][28][8], the BatchNorm outputs y1 / y2 (same shapes as c1 / c2), mean / rstd [groups][C], and updates the running statistics in call
 * order (either pair may be NULL).  ws: tg_d_preconv_ws_bytes(Bs) bytes, 16-byte aligned, ZERO before the first use and left zero by every
 * launch (workgroups of one launch meet at two device-wide barriers: one launch at a time per workspace; a timeout sets ws[0], sticky). */
int32_t tg_d_preconv_fwd_supported(int32_t Bs, int32_t groups);
int64_t tg_d_preconv_ws_bytes(int32_t Bs);
int tg_d_preconv_fwd(const float* poses, const float* w1, const float* b1, const float* gamma1, const float* beta1, const float* w2,
                     const float* b2, const float* gamma2, const float* beta2, const float* w3, const float* b3, float* c1, float* y1,
                     float* c2, float* y2, float* c3, float* mean1, float* rstd1, float* mean2, float* rstd2, float* running_mean1,
                     float* running_var1, int64_t* nbt1, float* running_mean2, float* running_var2, int64_t* nbt2, void* ws,
                     int64_t ws_bytes, int32_t Bs, int32_t groups, float eps, float momentum, void* stream);

/* The same block backwards in ONE launch: dc3 [nb][28][8] = gradient at the GRU input; the forward's tensors of the same nb clips (whole
 * statistics groups; mean / rstd of exactly those groups); parameter gradients ACCUMULATE (float atomics) -- the ten pointers are all given
 * or all NULL; dposes (NULL, or [nb][34][27]) receives the pose gradient, added to its content when dposes_accumulate.  Same workspace rule. */
int tg_d_preconv_bwd(const float* dc3, const float* poses, const float* c1, const float* y1, const float* c2, const float* y2,
                     const float* mean1, const float* rstd1, const float* mean2, const float* rstd2, const float* w1, const float* w2,
                     const float* w3, const float* gamma1, const float* gamma2, float* dw1, float* db1, float* dgamma1, float* dbeta1,
                     float* dw2, float* db2, float* dgamma2, float* dbeta2, float* dw3, float* db3, float* dposes,
                     int32_t dposes_accumulate, void* ws, int64_t ws_bytes, int32_t nb, int32_t groups, void* stream);

/* ---- speaker / style path (model/multimodal_context_net.py:83-95,125-137; embedding_net.py:10-13), fused ------------------------------
 * forward: se = table[vid], zc = W1 se + b1, mu = Wmu zc + bmu, logvar = Wlv zc + blv, z = mu + eps * exp(0.5 logvar) (all [B][16]); with
 * rep != NULL also rep[(b * T + t) * rep_ld + j] = z[b][j] (the style columns of the GRU input).  eps is read when rng_state == NULL and
 * WRITTEN otherwise (ABI 4): drawn as tg_normal(eps, B * 16, rng_state, site) would, in the same launch (torch.randn_like of :92).
 * backward (nb <= tg_speaker_bwd_max_rows()): dz [nb][16] = gradient w.r.t. z; d_mu_in / d_logvar_in: direct gradients or NULL; every
 * parameter gradient accumulates; dtable rows meet in float atomics. */
int tg_speaker_fwd(const float* table, const int64_t* vid, int32_t n_rows, const float* w1, const float* b1, const float* wmu, const float* bmu,
                   const float* wlv, const float* blv, float* eps, float* se, float* zc, float* mu, float* logvar, float* z,
                   int32_t B, float* rep, int64_t rep_ld, int32_t T, const uint64_t* rng_state, uint32_t site, void* stream);
int32_t tg_speaker_bwd_max_rows(void);
int tg_speaker_bwd(const float* dz, const float* d_mu_in, const float* d_logvar_in, const float* logvar, const float* eps, const float* zc,
                   const float* se, const int64_t* vid, int32_t n_rows, const float* w1, const float* wmu, const float* wlv, float* dw1,
                   float* db1, float* dwmu, float* dbmu, float* dwlv, float* dblv, float* dtable, int32_t nb, void* stream);

/* ---- output MLP Linear(H, Hm) -> LeakyReLU(True) (== identity, reference README.md:122) -> Linear(Hm, D), as ONE linear map -------------
 * (model/multimodal_context_net.py:100-104,157-158).  _compose: w21 [D][dup H] = W2 W1 (written dup = 1 or 2 times side by side), w21t
 * [dup H][D] its transpose (may be NULL), b21 = W2 b1 + b2.  dup = 2 lets the map act on the bidirectional GRU output [fwd | rev] without the
 * direction sum (:155-156) being formed.  _param_grads: from P [D][dup H] = d_out^T o (dup = 2: its halves are added) and s [D] =
 * colsum(d_out): dW1 += W2^T P, db1 += W2^T s, dW2 += P W1^T + s b1^T, db2 += s. */
int tg_out_mlp_compose(const float* w1, const float* b1, const float* w2, const float* b2, int32_t H, int32_t Hm, int32_t D, int32_t dup,
                       float* w21, float* w21t, float* b21, void* stream);
int tg_out_mlp_param_grads(const float* P, const float* s, const float* w1, const float* b1, const float* w2, int32_t H, int32_t Hm, int32_t D,
                           int32_t dup, float* dw1, float* db1, float* dw2, float* db2, void* stream);

/* ---- WavEncoder front end: Conv1d(1, 16, 15, stride, padding) -> BatchNorm1d(16) -> LeakyReLU, fused ------------------------
 * Replaces feat_extractor[0..2] of model/multimodal_context_net.py:13-15 (and their autograd backward) without materialising the
 * pre-BatchNorm tensor: the convolution is recomputed from the raw audio wherever it is needed (csrc/audio.hip).
 * audio: B clips of L samples, `audio_stride` floats apart; w [16][15], bias [16]; T1 = (L + 2 pad - 15) / stride + 1 output frames.
 * ws: scratch of tg_wav_front_ws_doubles() doubles (no initialisation needed); fstat: tg_wav_front_fstat_doubles() doubles written by
 * _stats and read by _backward (sums of the forward pass); gate: tg_wav_front_gate_words(B, T1) 64-bit words written by _apply
 * (one bit per output element: pre-activation >= 0) and read by _backward. */
int64_t tg_wav_front_ws_doubles(void);
int64_t tg_wav_front_fstat_doubles(void);
int64_t tg_wav_front_gate_words(int32_t B, int32_t T1);
/* train-mode batch statistics of the conv output over all B * T1 frames: mean / rstd [16], running statistics updated `repeats` times
 * (momentum form of nn.BatchNorm1d), num_batches_tracked += repeats.  running_* / num_batches_tracked / fstat may be NULL. */
int tg_wav_front_stats(const float* audio, int64_t audio_stride, int32_t B, int32_t L, const float* w, const float* bias, int32_t stride,
                       int32_t pad, int32_t T1, double* ws, int64_t ws_doubles, float* mean, float* rstd, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, double* fstat, float eps, float momentum, int32_t repeats,
                       void* stream);
/* y [B][T1][16] = act((conv - mean) * rstd * gamma + beta) (mean / rstd from _stats, or tg_bn_eval_stats in eval mode); gate may be NULL. */
int tg_wav_front_apply(const float* audio, int64_t audio_stride, int32_t B, int32_t L, const float* w, const float* bias, int32_t stride,
                       int32_t pad, int32_t T1, const float* mean, const float* rstd, const float* gamma, const float* beta, float act_slope,
                       float* y, uint64_t* gate, void* stream);
/* backward of the block for d y = dact [B][T1][16]: dW [16][15], dbias, dgamma, dbeta accumulate (each may be NULL); the input is raw
 * audio, so no input gradient. */
int tg_wav_front_backward(const float* dact, const uint64_t* gate, const float* audio, int64_t audio_stride, int32_t B, int32_t L, const float* w,
                          const float* bias, int32_t stride, int32_t pad, int32_t T1, const float* mean, const float* rstd, const float* gamma,
                          const double* fstat, float act_slope, double* ws, int64_t ws_doubles, float* dW, float* dbias, float* dgamma,
                          float* dbeta, void* stream);

/* The same backward with the input gradient of the NEXT layer, Conv1d(16, 32, 15, stride 6) (feat_extractor[3], weight w2 [32][16][15]),
 * computed on the fly from dc2 [B][T2][32] = the gradient w.r.t. that conv's output: d act is never materialised. */
int tg_wav_front_backward_fused(const float* dc2, int32_t T2, const float* w2, const uint64_t* gate, const float* audio, int64_t audio_stride,
                                int32_t B, int32_t L, const float* w, const float* bias, int32_t stride, int32_t pad, int32_t T1,
                                const float* mean, const float* rstd, const float* gamma, const double* fstat, float act_slope, double* ws,
                                int64_t ws_doubles, float* dW, float* dbias, float* dgamma, float* dbeta, void* stream);

/* Weight and bias gradient of feat_extractor[3] = Conv1d(16, 32, 15, stride 6): dW2 [32][16][15] += sum dc2[b, q, co] act[b, 6 q + k, ci],
 * db2 [32] += sum dc2 (either may be NULL); act [B][T1][16], dc2 [B][T2][32], T2 = (T1 - 15) / 6 + 1.  ws: tg_wav_conv2_wgrad_ws_floats()
 * floats of scratch.  One pass over the activation, deterministic. */
int64_t tg_wav_conv2_wgrad_ws_floats(void);
int tg_wav_conv2_wgrad(const float* dc2, const float* act, int32_t B, int32_t T1, int32_t T2, float* ws, int64_t ws_floats, float* dW2,
                       float* db2, void* stream);

/* ---- element-wise / data movement ------------------------------------------------------------------- */
/* y = max(a + b, 0)  (model/tcn.py:46);  dx = dy * (y > 0). */
int tg_add_relu(const float* a, const float* b, float* y, int64_t n, void* stream);
/* both gates of a residual block's backward in one pass: dsum = dy * (y > 0); dc = dsum * (o > 0 ? 1 : slope) * mask (mask may be NULL) */
int tg_act_mask_bwd2(const float* dy, const float* y, const float* o, const float* mask, float slope, float* dsum, float* dc, int64_t n, void* stream);
/* dx = dy * mask * (y > 0 ? 1 : slope); mask may be NULL (=1).  Backward of act() followed by dropout. */
int tg_act_mask_bwd(const float* dy, const float* y, const float* mask, float slope, float* dx, int64_t n,
                    void* stream);
/* The same two with the dropout scale REGENERATED: the multiplier of element i is element index0 + i of the draw
 * tg_dropout_mask(mask, n, p, rng_state, site) would write (see tg_gemm_nt_problem.drop_state).  n and index0 multiples of 4, 16-byte
 * aligned pointers.  model/tcn.py:22-29 backward without a stored mask. */
int tg_act_mask_bwd_drop(const float* dy, const float* y, float p, const uint64_t* rng_state, uint32_t site, int64_t index0, float slope,
                         float* dx, int64_t n, void* stream);
int tg_act_mask_bwd2_drop(const float* dy, const float* y, const float* o, float p, const uint64_t* rng_state, uint32_t site, int64_t index0,
                          float slope, float* dsum, float* dc, int64_t n, void* stream);
/* y = x * mask  (nn.Dropout with a materialised inverted-dropout mask: 0 or 1/(1-p)). */
int tg_mul(const float* x, const float* mask, float* y, int64_t n, void* stream);
/* y (+)= alpha * x. */
/* Zero `bytes` (multiple of 4) at p with a kernel of this library (graph-capture safe; the library never uses hipMemsetAsync:
 * captured memset nodes were observed to replay with a clobbered fill pattern on ROCm 7.2, see csrc/common.hpp). */
int tg_zero(void* p, int64_t bytes, void* stream);
int tg_axpy(const float* x, float* y, float alpha, int32_t accumulate, int64_t n, void* stream);
/* dst[r*ldd + c] (+)= src[r*lds + c], r < rows, c < cols. */
int tg_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t rows, int32_t cols,
              int32_t accumulate, void* stream);
/* dst[(b*T + t)*ldd + c] = src[b*lds + c]  (z repeated over time, multimodal_context_net.py:151-153). */
int tg_repeat_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t B, int32_t T, int32_t cols,
                   void* stream);
/* dst[b*ldd + c] (+)= sum_t src[(b*T + t)*lds + c]. */
int tg_sum_rows(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t B, int32_t T, int32_t cols,
                int32_t accumulate, void* stream);
/* o[m*H + j] = y[m*2H + j] + y[m*2H + H + j]  (multimodal_context_net.py:156,242). */
int tg_add_halves(const float* y, float* o, int32_t M, int32_t H, void* stream);
/* out[i] = sum_q parts[q * stride + i], i < n (n, stride multiples of 4; 16-byte aligned): K-split partial products combined in a fixed order. */
int tg_sum_parts(const float* parts, int64_t stride, int32_t n_parts, float* out, int64_t n, void* stream);
/* out [M][8] = a0 [M][K] . w0 [K][8] + a1 [M][K] . w1 [K][8] (K % 4 == 0, contiguous operands, 16-byte aligned a0 / a1): the input gradient of a
 * bidirectional GRU layer with 8 input channels, dx = dgi_fwd W_ih_fwd + dgi_rev W_ih_rev (the discriminator's nn.GRU(8, 64), :222-223). */
int tg_narrow8_pair(const float* a0, const float* a1, const float* w0, const float* w1, float* out, int32_t M, int32_t K, void* stream);
/* dy[m*2H + j] = dy[m*2H + H + j] = do[m*H + j]. */
int tg_dup_halves(const float* d_o, float* dy, int32_t M, int32_t H, void* stream);
/* pre[b][t][:D] = t < n_pre ? target[b][t][:] : 0 ; pre[b][t][D] = t < n_pre  (train_eval/train_gan.py:20-22). */
int tg_make_pre_seq(const float* target, float* pre, int32_t B, int32_t T, int32_t D, int32_t n_pre, void* stream);
/* Batch assembly on the device (ABI 5): SpeechMotionDataset.__getitem__ after the LMDB read (data_loader/lmdb_data_loader.py:107-171) and
 * default_collate_fn's stacking (:43-53) for B clips in one launch, from RAW per-clip records.
 *   audio_raw / audio_off [B + 1]: the stored audio of the clips back to back, clip b = [audio_off[b], audio_off[b + 1]) (any length);
 *   vec_raw / vec_off [B + 1] (offsets in floats): the stored direction vectors, at least n_poses frames per clip; n_ext [B] or NULL: the stored
 *   frame count vec_seq.shape[0] of each sample (NULL: (vec_off[b + 1] - vec_off[b]) / pose_floats, i.e. the whole sequence was shipped);
 *   word_idx / word_onset [B][Wmax], n_words [B]: vocabulary index (lang_model.get_word_index, resolved on the host) and onset time of each word;
 *   times [B][2]: aux_info start_time, end_time; vid_in [B] or NULL: speaker index (train.py:178-183).
 * Writes in_text [B][n_poses] (extend_word_seq :115-140, fp64 as the reference's doubles; remove_word_timing != 0: words evenly spread),
 * in_audio [B][audio_len] (utils/data_utils.py:68-74: truncate or numpy 'symmetric' padding), target [B][n_poses][pose_floats]
 * (vec_seq[0:n_poses]) and vid [B] -- bit for bit what the host path produces. */
int tg_assemble_batch(const float* audio_raw, const int64_t* audio_off, const float* vec_raw, const int64_t* vec_off, const int64_t* word_idx,
                      const double* word_onset, const int32_t* n_words, const double* times, const int32_t* n_ext, const int64_t* vid_in, int32_t B, int32_t Wmax,
                      int32_t n_poses, int32_t pose_floats, int32_t audio_len, int32_t remove_word_timing, int64_t* out_text, float* out_audio,
                      float* out_vec, int64_t* out_vid, void* stream);
/* out[i*D + :] = table[idx[i]*D + :]  (nn.Embedding, multimodal_context_net.py:40,89). */
int tg_embed_gather(const float* table, const int64_t* idx, float* out, int32_t n_idx, int32_t D, int32_t n_rows,
                    void* stream);
/* The look-up followed by F.dropout(p) (multimodal_context_net.py:47-52, train mode) in one pass: out = table[idx] * mask, mask = the draw
 * tg_dropout_mask(mask, n_idx * D, p, rng_state, site) would write; it is not stored -- the backward regenerates it (tg_act_mask_bwd_drop with
 * slope 1).  D % 4 == 0, 16-byte aligned table and out. */
int tg_embed_gather_drop(const float* table, const int64_t* idx, float* out, int32_t n_idx, int32_t D, int32_t n_rows, float p,
                         const uint64_t* rng_state, uint32_t site, void* stream);
/* dtable[idx[i]*D + :] += dout[i*D + :]  (dense embedding gradient; accumulates). */
int tg_embed_scatter_add(const float* dout, const int64_t* idx, float* dtable, int32_t n_idx, int32_t D,
                         int32_t n_rows, void* stream);
/* out[perm-ed] = in: generic 3-D permute, out[i0][i1][i2] with out dims (d[p0], d[p1], d[p2]) of in dims d. */
int tg_permute3(const float* in, float* out, int32_t d0, int32_t d1, int32_t d2, int32_t p0, int32_t p1,
                int32_t p2, void* stream);

/* One launch for a table of independent tg_permute3 jobs (all weight transposes / conv packs of a network after an optimiser
 * step).  desc: device array of n_jobs x 10 int64 = {src, dst, d0, d1, d2, p0, p1, p2, first_workgroup, workgroup_count}; jobs
 * sorted by first_workgroup, total_workgroups = sum of the counts.  Same element mapping as tg_permute3. */
int tg_permute3_batch(const int64_t* desc, int32_t n_jobs, int32_t total_workgroups, void* stream);
/* Weight pack for the input-gradient of Conv1d(stride s) (= forward of ConvTranspose1d):
 * w: [Co][Ci][kw] -> out: [s][Ci][J][Co], J = ceil(kw/s), out[r][ci][j][co] = (r + s*j < kw) ? w[co][ci][r + s*j] : 0.
 * Phase r serves the input positions p with (p % s) == r:  dx[6q + r] = sum_j dy[q - j] . out[r][:, j, :]. */
int tg_conv_dgrad_pack(const float* w, float* out, int32_t Co, int32_t Ci, int32_t kw, int32_t s, void* stream);

/* ---- weight norm (torch.nn.utils.weight_norm dim=0; model/tcn.py:19,25) ------------------------------
 * v: [Co][Ci][kw], g: [Co].  w_packed: [Co][kw][Ci] = g * v / ||v||  (tap-major, the layout tg_gemm_nt wants). */
int tg_weight_norm_fwd(const float* v, const float* g, float* w_packed, int32_t Co, int32_t Ci, int32_t kw,
                       void* stream);
/* All weight-normed convs of a network in ONE launch (same Co, Ci, kw): w_packed[i] as tg_weight_norm_fwd, and -- when w_t != NULL --
 * w_t[i] = the same weight as [Ci][kw*Co] (w_t[ci][tap*Co + co]), the B operand of the conv's input gradient. */
int tg_weight_norm_fwd_batch(int32_t n, const float* const* v, const float* const* g, float* const* w_packed, float* const* w_t,
                             int32_t Co, int32_t Ci, int32_t kw, void* stream);
/* dw_packed: [Co][kw][Ci] -> dg[Co], dv[Co][Ci][kw] (both accumulate). */
int tg_weight_norm_bwd(const float* dw_packed, const float* v, const float* g, float* dg, float* dv, int32_t Co,
                       int32_t Ci, int32_t kw, void* stream);
/* The same for n <= 8 convs of equal shape in one launch (tables of n pointers). */
int tg_weight_norm_bwd_batch(int32_t n, const float* const* dw_packed, const float* const* v, const float* const* g, float* const* dg,
                             float* const* dv, int32_t Co, int32_t Ci, int32_t kw, void* stream);

/* ---- randomness: Philox4x32-10 counter RNG, graph-replay safe -----------------------------------------
 * rng_state: device uint64[2] = {seed, step}.  tg_rng_advance bumps step by one (launch once per iteration).
 * Every draw site passes its own `site` id so streams never collide. */
int tg_rng_advance(uint64_t* rng_state, void* stream);
/* start of a training iteration in one launch: rng_state[1] += 1 for up to two RNG states and += 1 for up to two Adam step counters
 * (tg_adam_step then runs with the counter already advanced); any pointer may be NULL, not all. */
int tg_iter_begin(uint64_t* rng_a, uint64_t* rng_b, int32_t* adam_step_a, int32_t* adam_step_b, void* stream);
/* The head of one train_iter_gan call (scripts/train_eval/train_gan.py:13-30,50,67-72) in ONE launch: tg_iter_begin's counters; the seed
 * poses of the `copies` stacked generator calls (tg_make_pre_seq, pre_stacked [copies][B][T] rows of D + 1 floats, pre_ld floats apart:
 * pre_ld > D + 1 writes them straight into the pose columns of the GRU input rows); the word ids copied `copies` times
 * (text [B][T] -> text_stacked, both may be NULL); the speaker ids [vid] * (copies - 1) + [last] (vid [B] -> vid_stacked [copies][B], both
 * may be NULL) where last = vid[perm] if permute_last -- perm = torch.randperm(B) of :69 drawn as tg_randperm(.., rng_a at its NEW step,
 * perm_site) or given (perm_in, tests), also written to perm_out when non-NULL -- and vid otherwise.  B <= 1024 with speaker ids.
 * target_copy (ABI 5; NULL, or [B][T][D]): receives a copy of target -- the "real" half of the discriminator's stacked input, so that
 * torch.cat((target, out_dir_vec.detach())) of :30-31 needs no launch of its own (the generator writes its half behind it). */
int tg_iter_head(uint64_t* rng_a, uint64_t* rng_b, int32_t* adam_step_a, int32_t* adam_step_b, const float* target, float* pre_stacked, int64_t pre_ld,
                 int32_t B, int32_t T, int32_t D, int32_t n_pre, int32_t copies, const int64_t* text, int64_t* text_stacked, const int64_t* vid,
                 int64_t* vid_stacked, int32_t permute_last, const int64_t* perm_in, uint32_t perm_site, int64_t* perm_out, float* target_copy,
                 void* stream);
int tg_dropout_mask(float* mask, int64_t n, float p, const uint64_t* rng_state, uint32_t site, void* stream);
/* Draw the same mask and apply it in one pass: mask as tg_dropout_mask, y[i] = x[i] * mask[i] (F.dropout, train mode).  mask may be NULL
 * (ABI 4): the mask is not stored, its later consumers regenerate it (tg_gemm_nt_problem.drop_state, tg_act_mask_bwd_drop). */
int tg_dropout_apply(const float* x, float* y, float* mask, int64_t n, float p, const uint64_t* rng_state, uint32_t site,
                     void* stream);
int tg_normal(float* out, int64_t n, const uint64_t* rng_state, uint32_t site, void* stream);
/* out = random permutation of [0, n), n <= 1024 (torch.randperm at train_eval/train_gan.py:62). */
int tg_randperm(int64_t* out, int32_t n, const uint64_t* rng_state, uint32_t site, void* stream);
/* out[i] = src[perm[i]] for int64 vectors (vid_indices[rand_idx], train_gan.py:63). */
int tg_gather_i64(const int64_t* src, const int64_t* perm, int64_t* out, int32_t n, void* stream);

/* ---- speaker path (multimodal_context_net.py:128-131; model/embedding_net.py:10-13) -------------------- */
/* z = mu + eps * exp(0.5 * logvar) */
int tg_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream);
/* dmu += dz ; dlogvar += dz * eps * 0.5 * exp(0.5*logvar) */
int tg_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar, int64_t n,
                   void* stream);

/* ---- losses (train_eval/train_gan.py:41,53-89) ----------------------------------------------------------
 * Discriminator step.  logit_*: pre-sigmoid outputs [B].  out[0] = dis_error; d_logit_* = d dis_error / d logit. */
int tg_gan_d_loss(const float* logit_real, const float* logit_fake, int32_t B, float* out, float* d_logit_real,
                  float* d_logit_fake, void* stream);
/* Generator step.  out_pose/target/out_rand: [B][T*D]; z, z_rand, mu, logvar: [B][Z]; logit_out: [B] (pre-sigmoid
 * D(out)).  scalars[0..4] = huber, kld, div_reg, gen_error, total loss (unweighted terms; total is weighted).
 * Gradients of the TOTAL loss: d_out [B][T*D], d_mu, d_logvar [B][Z] (written), d_logit_out [B] (0 when
 * use_gan == 0).  ws: 3*B floats. */
int tg_gan_g_loss(const float* out_pose, const float* target, const float* out_rand, const float* z,
                  const float* z_rand, const float* mu, const float* logvar, const float* logit_out, int32_t B,
                  int32_t TD, int32_t Z, float w_huber, float w_kld, float w_div, float w_gan, int32_t use_gan,
                  float* ws, float* scalars, float* d_out, float* d_mu, float* d_logvar, float* d_logit_out,
                  void* stream);
/* ConvDiscriminator head (multimodal_context_net.py:243-252) in one launch each way.  y: [B][T][2H] last GRU layer output;
 * w1/b1: out (Linear H -> 1), w2/b2: out2 (Linear T -> 1).  forward: l1 [B][T] per-frame logits (kept for the backward), logit [B]
 * (pre-sigmoid), prob [B].  backward: d_logit [B] -> dy [B][T][2H]; dw1/db1/dw2/db2 accumulate (all NULL: input gradient only). */
int tg_d_head_fwd(const float* y, const float* w1, const float* b1, const float* w2, const float* b2, float* l1, float* logit, float* prob,
                  int32_t B, int32_t T, int32_t H, void* stream);
int tg_d_head_bwd(const float* d_logit, const float* y, const float* l1, const float* w1, const float* w2, float* dy, float* dw1, float* db1,
                  float* dw2, float* db2, int32_t B, int32_t T, int32_t H, void* stream);

/* Head forward, per-clip GAN loss terms and head backward in ONE launch: what tg_d_head_fwd + (tg_gan_d_loss | the d_logit part of
 * tg_gan_g_loss) + tg_d_head_bwd compute in three.  y [n_rows][T][2H]; rows [0, n_real) are scored as real -- term = log(s + 1e-8),
 * d_logit = -scale_real s (1 - s) / (s + 1e-8) -- and rows [n_real, n_rows) as fake -- log(1 - s + 1e-8), +scale_fake s (1 - s) / (1 - s + 1e-8).
 * Discriminator step (train_gan.py:36-41): the stacked batch [real ; fake], n_rows = 2 n_real, both scales 1 / n_real, dis_error =
 * -sum(terms) / n_real.  Generator step (:55-57, 86-88): n_rows = n_real = B, scale_real = loss_gan_weight / B (0 in warm-up), no parameter
 * gradients, gen_error = -sum(terms) / B.  terms [n_rows]; out[0] = -sum(terms) / n_real summed by the last workgroup in the order of
 * tg_gan_d_loss, or out == NULL: the caller sums terms itself (saves the serial tail); counter (needed with out): one zero-initialised
 * device word the kernel leaves at zero (never shared by launches that may run concurrently).  dw1/db1/dw2/db2 accumulate (all NULL: none). */
int tg_d_head_step(const float* y, const float* w1, const float* b1, const float* w2, const float* b2, float* l1, float* logit, float* prob,
                   float* d_logit, float* terms, float* out, uint32_t* counter, float* dy, float* dw1, float* db1, float* dw2, float* db2,
                   int32_t n_rows, int32_t n_real, float scale_real, float scale_fake, int32_t T, int32_t H, void* stream);

/* out[0] = mean |a - b| over n elements (F.l1_loss, train.py:282). */
int tg_l1_mean(const float* a, const float* b, int64_t n, float* out, void* stream);
/* y = 1 / (1 + exp(-x)) */
int tg_sigmoid(const float* x, float* y, int64_t n, void* stream);
/* dx = dy * y * (1 - y), y = sigmoid(x)  (torch.sigmoid backward, multimodal_context_net.py:250). */
int tg_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* Cross-fade of consecutive synthesis windows (synthesize.py:145-153), in place on the new window:
 * next[b][j][:] = prev_tail[b][j][:] * (n-j)/(n+1) + next[b][j][:] * (j+1)/(n+1), j < n.  next: [B][T][D], prev_tail: [B][n][D]. */
int tg_window_blend(const float* prev_tail, float* next, int32_t B, int32_t T, int32_t D, int32_t n, void* stream);

/* evaluate_testset metrics of one batch (train.py:282-310; utils/data_utils.py:77-98): out/target direction vectors [B][T][27],
 * mean_dir_vec [27].  sums[0] = sum |joint error| over frames >= n_pre and 10x3 joints, sums[1] = sum |second-difference error| over
 * T-2 frames, sums[2] = sum |out - target|.  (joint_mae = sums[0]/(B*(T-n_pre)*30), accel = sums[1]/(B*(T-2)*30), l1 = sums[2]/(B*T*27).) */
int tg_pose_metrics(const float* out_dir_vec, const float* target_dir_vec, const float* mean_dir_vec, int32_t B, int32_t T,
                    int32_t n_pre, double* sums, void* stream);

/* FGD autoencoder loss (train_feature_extractor.py:64-72): loss = sum_b [mean|r-t| + mean|dr-dt|];
 * out[0] = loss, d_recon = d loss / d recon. */
int tg_ae_loss(const float* recon, const float* target, int32_t B, int32_t T, int32_t D, float* out, float* d_recon,
               void* stream);

/* One FGD autoencoder training step up to the gradients (scripts/train_feature_extractor.py:54-97 on model/embedding_net.py:42-82,165-217:
 * PoseEncoderConv + PoseDecoderConv, 34 frames x 27, variational_encoding = False) in 18 launches (ABI 5; csrc/ae_step.hip): forward in
 * train mode (the eight BatchNorms' running statistics advance), loss = sum_b [mean |recon - x| + mean |d recon - d x|], and every parameter
 * gradient WRITTEN (not accumulated) into `grads` at the parameter's offset; fc_logvar receives no gradient (:58) and is not touched.
 * With adam_m / adam_v (the optimiser's moment slabs) the last launch also applies torch.optim.Adam(lr, betas, eps) to every parameter it has a
 * gradient for -- the step is then complete (fc_logvar is skipped like a parameter whose .grad is None); without them tg_adam_step over the slab
 * completes it.  *step (device Adam counter, may be NULL) is advanced by one by the first launch.
 * off[44]: offsets in floats (multiples of 4) of, in this order: pose_encoder.net.{0,1,2}: {0.weight, 0.bias, 1.weight, 1.bias} each; net.3.{weight,
 * bias}; out_net.0.{weight, bias}, out_net.1.{weight, bias}, out_net.3.{..}, out_net.4.{..}, out_net.6.{..}; fc_mu.{..}; decoder.pre_net.0.{..},
 * pre_net.1.{..}, pre_net.3.{..}; decoder.net.0.{..}, net.1.{..}, net.3.{..}, net.4.{..}, net.6.{..}, net.7.{..}.
 * running_mean / running_var / num_batches_tracked[8]: the BatchNorms in forward order (encoder net.0-2, out_net.1, out_net.4, pre_net.1, decoder
 * net.1, net.4); a NULL pair skips that update.  ws: tg_ae_step_ws_bytes(B) bytes, 16-byte aligned, ZERO before the first use; a complete step
 * leaves its statistics block zero again (after an aborted step: zero it).  recon [B][34][27] and feat [B][32] (= mu) are optional outputs.
 * last_phase: 0 = the whole step; 1..18 = stop after that launch (tests).  2 <= B <= 256 (tg_ae_step_supported). */
typedef struct {
    const float* x;                       /* [B][34][27] */
    float* params;
    float* grads;
    int32_t off[44];
    float* running_mean[8];
    float* running_var[8];
    int64_t* num_batches_tracked[8];
    void* ws;
    int64_t ws_bytes;
    float* loss;                          /* [1] */
    float* recon;
    float* feat;
    int32_t* step;
    int32_t B;
    float bn_eps, momentum;
    int32_t last_phase;
    float* adam_m;                        /* Adam moments, slab images like params (both or neither) */
    float* adam_v;
    float lr, beta1, beta2, adam_eps;
} tg_ae_step_args;
int32_t tg_ae_step_supported(int32_t B);
int64_t tg_ae_step_ws_bytes(int32_t B);
int tg_ae_train_step(const tg_ae_step_args* args, void* stream);

/* ---- optimiser (torch.optim.Adam, train.py:104-109): one fused launch over a flat parameter slab ------
 * step_dev: device int32 step counter, incremented by tg_counter_inc BEFORE the update (graph-replay safe). */
int tg_counter_inc(int32_t* counter, void* stream);
int tg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, const int32_t* step_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIMODAL_HIP_H */
