// Shared helpers for the gfx950 kernels of libtrimodal_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/trimodal_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace tg {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

#define TG_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            tg::set_error(__VA_ARGS__);    \
            return 2;                      \
        }                                  \
    } while (0)

extern "C" int tg_get_deterministic(void);
inline bool deterministic() { return tg_get_deterministic() != 0; }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Zero `bytes` (a multiple of 4) at p with a kernel of this library.  Used instead of hipMemsetAsync everywhere: a memset
// captured into a hipGraph replays as a runtime fill kernel that reads its pattern from a staging area of the runtime, and on
// ROCm 7.2 that area is reused by eager blit copies issued between replays -- the "zero" fill then wrote copy arguments
// (source / destination pointers) over the flag words of the persistent GRU kernels (tools/graph_memset_hazard_probe.py).
int zero_async(void* p, size_t bytes, hipStream_t s);

// grid for a grid-stride element-wise kernel: enough blocks to fill 256 CUs, capped (guide: Guideline 11)
inline int ew_grid(int64_t n, int block = 256, int per_thread = 4) {
    int64_t b = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// device copy of tg_window
struct Win {
    const float* ptr;
    long bs, rs;
    int rows_in, rows_out, step, shift, dil, cw, K;
};

inline Win to_win(const tg_window* w) {
    Win d;
    d.ptr = w->ptr; d.bs = w->batch_stride; d.rs = w->row_stride; d.rows_in = w->rows_in; d.rows_out = w->rows_out;
    d.step = w->row_step; d.shift = w->shift; d.dil = w->dil; d.cw = w->cw; d.K = w->K;
    return d;
}

// ---- grouped launches ---------------------------------------------------------------------------------------------------------
// Several independent products of one layer (both GRU directions' input projections, the four weight gradients of a GRU layer, the
// stride phases of a conv input-gradient ...) run as ONE launch: the problem table travels BY VALUE in the kernel arguments (safe
// under hipGraph capture), workgroup ranges [wg_begin[i], wg_begin[i+1]) belong to problem i.
constexpr int TG_MAX_GROUP = 8;

struct NtProb {
    Win A;
    const float* Bw;
    long ldb;
    long b_seg_stride;       // K-concatenated weights: k in [s * b_seg_k, (s+1) * b_seg_k) reads Bw + s * b_seg_stride + n * ldb + (k - s * b_seg_k)
    int b_seg_k;             // (== K: one weight matrix)
    const float* bias;
    const float* mul;        // optional element-wise multiplier applied after the activation (a dropout scale mask), addressed like C
    const uint64_t* drop_state;  // ... or that mask REGENERATED (dropout_scale4) from element index drop_index0 + offset in C; big-product kernels, vec_c
    long drop_index0;
    unsigned drop_site;
    float drop_p;
    float* C;
    long cbs, crs;
    int cR, M, N;
    float slope;
    int accumulate;
    int n_nt;
    int vec_c;               // C / mul rows can be accessed as 16-byte pieces (N % 4 == 0, strides % 4 == 0, aligned pointers)
    const float* gate;       // optional epilogue extensions (gemm_split.hip only): keep the result where gate > 0 ...
    const float* res;        // ... and / or a second output C2 = act2(C + res), slope res_slope; all addressed like C
    float* C2;
    float res_slope;
    unsigned a_bytes, b_bytes;   // byte extents of the A tensor / the weights from their base pointers (gemm_mw.hip buffer descriptors; filled by its launcher)
    const __bf16* Bpl;           // optional pre-split weights: plane 0 of a slab-tiled plane buffer (plane_tiled_off) that holds Bw's N rows of K columns
    long bpl_plane;              //   from buffer row b_row0 on; planes bpl_plane elements apart; b_slab_rows = rows of the buffer + 1 (its zero row last)
    int b_slab_rows, b_row0;
    // fp16 x 2 (h2 != 0): Bpl holds TWO fp16 planes (hi / lo of the scaled rows, tg_split2h_planes), b_inv[buffer row] = 1 / that row's scale,
    // a_scale[m] = the power-of-two scale of product row m (tg_h2_row_scales: 2^(141 - e) for the largest magnitude over the row's K values)
    const float* a_scale;
    // ... or (a_rmax != nullptr, windows of at most two taps) the largest magnitude of every SOURCE row, index batch * rows_in + source row, written by
    // the activation's producer (non-negative floats; combined with atomic unsigned max): the kernel derives the product rows' scales itself
    const float* a_rmax;
    int a_rmax_div;          // source rows per a_rmax entry: entry (batch * rows_in + source row) / a_rmax_div (1: per row; T: one per clip of T rows)
    const float* b_inv;
    int h2;
    // optional outputs for the NEXT product's a_rmax: c_rmax[m] / c2_rmax[m] = max(old, largest magnitude of row m of C / C2) by atomic unsigned
    // max (the caller zeroes them once per pass; rows are product rows: M floats)
    float* c_rmax;
    float* c2_rmax;
};

struct NtGroup {
    int n;
    int wg_begin[TG_MAX_GROUP + 1];
    NtProb p[TG_MAX_GROUP];
};

struct TnProb {
    const float* dY;
    long ldy;
    Win A;
    float* dW;
    long ldw;
    int M, N, rows_per_split, out_kw;
    float* partial;
    float* dbias;
    int vec_y, vec_a, n_nt, n_kt;
    unsigned a_bytes, y_bytes;   // byte extents of the A tensor / dY from their base pointers (gemm_tn_mw.hip buffer descriptors; filled by its planner)
    // fp16 x 2 (both non-null; mover-wave kernel): largest magnitude of every COLUMN of dY (N floats, 16-byte aligned) and of every channel of A's
    // tensor (cw floats, shared by the taps) -- the reduction runs over rows, so the power-of-two scales belong to the columns
    const float* y_cmax;
    const float* a_cmax;
};

struct TnGroup {
    int n;
    int wg_begin[TG_MAX_GROUP + 1];
    TnProb p[TG_MAX_GROUP];
};

// index of the problem that owns workgroup `bid` (wave-uniform)
template <typename G>
__device__ __forceinline__ int group_find(const G& g, int bid) {
    int i = 0;
#pragma unroll
    for (int q = 1; q < TG_MAX_GROUP; ++q) i += (q < g.n && bid >= g.wg_begin[q]) ? 1 : 0;
    return i;
}

// XCD-aware workgroup order.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with a private
// 4 MB L2), so neighbours in id space -- which here share a weight slice or an activation panel -- land on eight
// different L2s and every one of them re-fetches the shared operand from Infinity Cache / HBM.  This bijection hands each
// XCD one CONTIGUOUS chunk of the logical id space instead (guide T1, non-divisible-safe form).  Speed only: any
// placement gives the same results.
__device__ __forceinline__ int xcd_chunked_id(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

__device__ __forceinline__ float act_fn(float x, float slope) { return x >= 0.f ? x : x * slope; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// GRU gate non-linearities (inside the dependent chain of every recurrence step; the gate epilogue is VALU-bound -- in the H = 64
// kernel it is two thirds of a step): the hardware exponential v_exp_f32 (2^y, one ulp) on y = x * log2(e), the one-ulp hardware
// reciprocal instead of the IEEE division sequence, and tanh through the same exponential, (1 - e) / (1 + e) with
// e = exp(-2|x|) in (0, 1].  Error budget for |x| <= 16: the rounding of y moves e by <= |y| * 2^-24 * ln 2 relative, v_exp and v_rcp
// add one ulp each; through d sigmoid / d e = -s^2 (<= 1/4 where e matters) the result is within ~1e-7 absolute of the exact value --
// the same class as the libm expf version (4 instructions instead of ~16 per sigmoid, 7 instead of ~20 per tanh).
__device__ __forceinline__ float gate_exp_neg(float x) { return __builtin_amdgcn_exp2f(x * -1.44269504088896341f); }     // exp(-x)
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + gate_exp_neg(x)); }
__device__ __forceinline__ float gate_tanh(float x) {
    const float e = gate_exp_neg(2.f * fabsf(x));
    return copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), x);
}

// ---- exact three-way bf16 split of an fp32 value (the "bf16 x 3" matrix-core feed, see gemm_split.hip) -----------------------
// x = hi + mid + lo exactly: hi = the top 8 significand bits of x (truncation), mid = the top 8 bits of x - hi, lo = the rest.
// Each term is returned as an fp32 bit pattern whose low 16 bits are zero (its upper half IS the bf16 value).
__device__ __forceinline__ void split3_bits(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(hi);
    mid = __float_as_uint(r) & 0xffff0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));
}
// upper halves of two fp32 words -> one dword holding two bf16 (first element in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned first, unsigned second) { return __builtin_amdgcn_perm(second, first, 0x07060302u); }

// Plane buffer of an fp32 matrix [rows][cw] (gemm_planes.hip): three bf16 planes (hi / mid / lo), each SLAB-TILED -- [cwp / 32 slabs][rows + 1]
// [32 columns], cwp = cw rounded up to 32, zero past cw and in the extra row `rows` of every slab.  One 32-deep K slab of 16 consecutive
// rows is then 1 KB of contiguous memory: a single LDS-DMA instruction fetches it as eight whole cache lines (with row-major planes the
// same instruction touched 16 half lines 2 cwp bytes apart and cost the CU's address unit ~60 cycles: profiles/r3_g_nt_mw_probe.txt).
__device__ __host__ __forceinline__ long plane_tiled_off(long r, int c, int rows) { return ((long)(c >> 5) * (rows + 1) + r) * 32 + (c & 31); }
// one 8-column piece of such a buffer: columns c .. c + 7 (c % 8 == 0) of row r of the fp32 matrix (row stride ldx)
typedef unsigned tg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_write_piece(const float* __restrict__ x, long ldx, int rows, int cw, int cwp, __bf16* __restrict__ planes,
                                                   long plane_stride, long r, int c, bool vec) {
    float v[8];
    if (r < rows && vec && c + 8 <= cw) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * ldx + c), b = *reinterpret_cast<const f32x4*>(x + r * ldx + c + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = a[q]; v[4 + q] = b[q]; }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (r < rows && c + q < cw) ? x[r * ldx + c + q] : 0.f;
    }
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) split3_bits(v[q], h[q], m[q], l[q]);
    const long o = plane_tiled_off(r, c, rows);
    *reinterpret_cast<tg_u32x4*>(planes + o) = tg_u32x4{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7])};
    *reinterpret_cast<tg_u32x4*>(planes + plane_stride + o) = tg_u32x4{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7])};
    *reinterpret_cast<tg_u32x4*>(planes + 2 * plane_stride + o) = tg_u32x4{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7])};
}

// ---- two-term fp16 split of an fp32 value ("fp16 x 2": three matrix instructions per product instead of bf16 x 3's six) -------------------
// x * s = hi + lo + eps: hi = fp16(x * s) (round to nearest even), lo = fp16(x * s - hi) (the residual is exact in fp32), |eps| <= 2^-23 |x s|
// while lo is a normal fp16.  s is an exact power of two PER ROW of the operand (per output channel for weights): with e the biased fp32
// exponent of the row's largest magnitude, s = 2^(141 - e) puts that magnitude into [2^14, 2^15) -- nothing overflows fp16's 65 504, and
// every element within 2^-17 of the row's largest keeps a normal lo (22-23 significand bits against fp32's 24; smaller elements degrade
// gracefully: their absolute error stays below 2^-40 of the row's largest even where the matrix cores flush fp16 subnormals).  The product
// hi_a hi_b + hi_a lo_b + lo_a hi_b (lo lo dropped: 2^-22 relative) accumulates in fp32 and is scaled back by the exact 2^(e_a - 141) 2^(e_b - 141)
// in the epilogue.  Exponents are clamped to [32, 250]: rows whose largest magnitude is below 2^-95 (or zero) use the scale of 2^-95, an
// infinity / NaN in a row reaches the output as one.
typedef _Float16 tg_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 tg_f16x8 __attribute__((ext_vector_type(8)));
__device__ __host__ __forceinline__ int h2_exp_of_bits(unsigned absmax_bits) {
    const int e = (int)((absmax_bits & 0x7fffffffu) >> 23);
    return e < 32 ? 32 : (e > 250 ? 250 : e);
}
__device__ __forceinline__ float h2_scale_of_exp(int e) { return __uint_as_float((unsigned)(268 - e) << 23); }     // 2^(141 - e)
__device__ __forceinline__ float h2_inv_of_exp(int e) { return __uint_as_float((unsigned)(e - 14) << 23); }        // 2^(e - 141)
__device__ __forceinline__ float h2_inv_of_scale(float s) { return __uint_as_float(0x7f000000u - __float_as_uint(s)); }   // 1 / s for a power of two s
// two already scaled values -> one dword of their hi terms, one of their lo terms (first value in the low half)
__device__ __forceinline__ void h2_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const tg_f16x2 h = {(_Float16)a, (_Float16)b};
    const float ha = (float)h[0], hb = (float)h[1];
    const tg_f16x2 l = {(_Float16)(a - ha), (_Float16)(b - hb)};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// wave-wide maximum of non-negative float bit patterns (unsigned compare == float compare for them)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
    return v;
}

// One WAVE writes row r (r <= rows; r == rows is the all-zero row) of the fp16 x 2 plane buffer of an fp32 matrix [rows][cw] (row stride ldx):
// two fp16 planes (hi / lo of x * s_r) in the slab-tiled layout of plane_tiled_off, `plane_stride` elements apart, and inv[r] = 1 / s_r.
__device__ __forceinline__ void h2_write_row(const float* __restrict__ x, long ldx, int rows, int cw, int cwp, _Float16* __restrict__ planes,
                                             long plane_stride, float* __restrict__ inv, long r, int lane, bool vec) {
    const int c8n = cwp / 8;
    const bool live = r < rows;
    unsigned mx = 0;
    if (live) {
        for (int c = lane * 4; c < cw; c += 256) {
            if (vec && c + 4 <= cw) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float f = v[q]; const unsigned b = __float_as_uint(f) & 0x7fffffffu; mx = mx > b ? mx : b; }
            } else {
                for (int q = 0; q < 4 && c + q < cw; ++q) { const unsigned b = __float_as_uint(x[r * ldx + c + q]) & 0x7fffffffu; mx = mx > b ? mx : b; }
            }
        }
        mx = wave_max_u32(mx);
    }
    const int e = h2_exp_of_bits(mx);
    const float s = h2_scale_of_exp(e);
    if (lane == 0) inv[r] = live ? h2_inv_of_exp(e) : 0.f;
    for (int p8 = lane; p8 < c8n; p8 += 64) {
        const int c = p8 * 8;
        float v[8];
        if (live && vec && c + 8 <= cw) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * ldx + c), b = *reinterpret_cast<const f32x4*>(x + r * ldx + c + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = a[q]; v[4 + q] = b[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (live && c + q < cw) ? x[r * ldx + c + q] : 0.f;
        }
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h2_split2(v[2 * q] * s, v[2 * q + 1] * s, h[q], l[q]);
        const long o = plane_tiled_off(r, c, rows);
        *reinterpret_cast<tg_u32x4*>(planes + o) = tg_u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<tg_u32x4*>(planes + plane_stride + o) = tg_u32x4{l[0], l[1], l[2], l[3]};
    }
}

// fp16 x 2 planes of the K-CONCATENATED TRANSPOSE of two matrices: out row n (n < cols), column k (k < 2 rows) = w{k / rows}[k % rows][n], w0 / w1
// [rows][cols] fp32 contiguous -- the weight operand of dx = [dgi_fwd | dgi_rev] @ [W_ih_fwd ; W_ih_rev] (one product over K = 6H) straight from
// the two nn.GRU parameters.  One 256-thread workgroup per 8 output rows: thread (n_l = t % 8, kg = t / 8): a read instruction fetches 32-byte
// pieces of eight source rows, and a thread walks only 2 rows / 32 columns per pass (round 6, first form: 32 rows per workgroup -- 19 workgroups
// per matrix, 225 dependent-latency loads per thread: 100 us for three layers).  `wg` of `nwg` workgroups walk the row blocks; planes as
// h2_write_row writes them (zero row `cols` included).
constexpr int H2_TCAT_ROWS = 8;
__device__ __forceinline__ void h2_planes_tcat_block(const float* __restrict__ w0, const float* __restrict__ w1, int rows, int cols, int cwp,
                                                     _Float16* __restrict__ planes, long plane_stride, float* __restrict__ inv, int wg, int nwg,
                                                     unsigned (&smax)[32][H2_TCAT_ROWS]) {
    const int t = threadIdx.x, n_l = t & 7, kg = t >> 3;
    const int K = 2 * rows;
    for (int n0 = wg * H2_TCAT_ROWS; n0 <= cols; n0 += nwg * H2_TCAT_ROWS) {
        const int n = n0 + n_l;
        const bool live = n < cols;
        unsigned mx = 0u;
        if (live)
            for (int k = kg; k < K; k += 32) {
                const float v = (k < rows ? w0 : w1)[(long)(k < rows ? k : k - rows) * cols + n];
                const unsigned b = __float_as_uint(v) & 0x7fffffffu;
                mx = mx > b ? mx : b;
            }
        smax[kg][n_l] = mx;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) { const unsigned b = smax[q][n_l]; mx = mx > b ? mx : b; }
        __syncthreads();
        const int e = h2_exp_of_bits(mx);
        const float sc = h2_scale_of_exp(e);
        if (kg == 0 && n <= cols) inv[n] = live ? h2_inv_of_exp(e) : 0.f;
        if (n <= cols)
            for (int p8 = kg; p8 < cwp / 8; p8 += 32) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = 8 * p8 + q;
                    v[q] = (live && k < K) ? (k < rows ? w0 : w1)[(long)(k < rows ? k : k - rows) * cols + n] : 0.f;
                }
                unsigned h[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) h2_split2(v[2 * q] * sc, v[2 * q + 1] * sc, h[q], l[q]);
                const long o = plane_tiled_off(n, 8 * p8, cols);
                *reinterpret_cast<tg_u32x4*>(planes + o) = tg_u32x4{h[0], h[1], h[2], h[3]};
                *reinterpret_cast<tg_u32x4*>(planes + plane_stride + o) = tg_u32x4{l[0], l[1], l[2], l[3]};
            }
    }
}

// ---- Philox4x32-10 ------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32(uint64_t seed, uint64_t idx, uint32_t site, uint32_t step, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), site, step};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)
// elements 4 idx4 .. 4 idx4 + 3 of the inverted-dropout scale mask tg_dropout_mask(.., p, st, site) writes: the consumers of a dropout
// regenerate it from the element index instead of reading a stored mask (forward and backward see the same draw: st is only advanced by
// tg_iter_begin)
__device__ __forceinline__ f32x4 dropout_scale4(const uint64_t* __restrict__ st, uint32_t site, float p, uint64_t idx4) {
    uint32_t r[4];
    philox4x32(st[0], idx4, site, (uint32_t)st[1], r);
    const float keep = 1.f / (1.f - p);
    f32x4 m;
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = u01(r[q]) >= p ? keep : 0.f;
    return m;
}

}  // namespace tg
