// GEMM-shaped ops on the f32 matrix cores (v_mfma_f32_16x16x4_f32): exact fp32 fmaf chains, so results agree with
// the reference's fp32 ATen kernels to rounding.  One "row window" descriptor (tg_window) covers nn.Linear, every
// Conv1d / ConvTranspose1d of the path (channel-last, im2col never materialised) and the shifted h_{t-1} view the
// GRU weight gradient needs.
//
// tg_gemm_nt: no LDS.  The MFMA k index is permuted so that lane group kq = lane>>4 owns k = 16u + 4kq + {0..3}
// of every 16-deep super-step: each lane then feeds four MFMAs from ONE contiguous 16-byte global load per operand
// (both operands use the same permutation, so the sum over k is unchanged).  Operands of this path are L2-resident
// (weights <= 2 MB, activations a few MB per layer); L1/L2 serve the 2x intra-workgroup reuse.
//
// tg_gemm_tn (weight gradients): the reduction runs over ROWS, so MFMA fragments are column-strided; tiles of 16 rows
// are staged through LDS with coalesced 16-byte loads, double-buffered, one barrier per tile.
#include "common.hpp"

namespace tg {

template <bool VEC>
__device__ __forceinline__ void load_nt_frags(const Win& A, const long (&a_off)[2], const int (&a_r)[2], const bool (&a_ok)[2],
                                              const float* const (&b_ptr)[2], const bool (&b_ok)[2], int k, int kk, int c,
                                              f32x4 (&fa)[2], f32x4 (&fb)[2]) {
    const int K = A.K;
    if (VEC) {
        const bool inb = k < K;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int sr = a_r[mt] + kk * A.dil;
            const bool ok = a_ok[mt] && inb && sr >= 0 && sr < A.rows_in;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            fa[mt] = ok ? *reinterpret_cast<const f32x4*>(A.ptr + a_off[mt] + (long)sr * A.rs + c) : z;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            fb[nt] = (b_ok[nt] && inb) ? *reinterpret_cast<const f32x4*>(b_ptr[nt] + k) : z;
        }
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int kv = k + v;
            const bool inb = kv < K;
            const int kkv = inb ? kv / A.cw : 0;
            const int cv = kv - kkv * A.cw;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int sr = a_r[mt] + kkv * A.dil;
                const bool ok = a_ok[mt] && inb && sr >= 0 && sr < A.rows_in;
                fa[mt][v] = ok ? A.ptr[a_off[mt] + (long)sr * A.rs + cv] : 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) fb[nt][v] = (b_ok[nt] && inb) ? b_ptr[nt][kv] : 0.f;
        }
    }
}

// Workgroup = 4 waves arranged WM x WN, each wave a 32x32 output tile (2x2 MFMA tiles of 16x16).
template <bool VEC, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(Win A, const float* __restrict__ Bw, long ldb,
                                                      const float* __restrict__ bias, float* __restrict__ C, long cbs,
                                                      long crs, int cR, int M, int N, float slope, int accumulate, int n_nt) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;
    // logical order: output-column tile fastest -> the workgroups that re-read one A row panel sit on ONE XCD's L2
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m_base = (lid / n_nt) * (WM * 32) + wm * 32;
    const int n_base = (lid % n_nt) * (WN * 32) + wn * 32;
    if (m_base >= M || n_base >= N) return;   // no LDS, no barriers: a whole wave may leave
    const int K = A.K;

    long a_off[2];
    int a_r[2];
    bool a_ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m_base + mt * 16 + r16;
        a_ok[mt] = m < M;
        const int mm = a_ok[mt] ? m : 0;
        const int b = mm / A.rows_out;
        const int r = mm - b * A.rows_out;
        a_off[mt] = (long)b * A.bs;
        a_r[mt] = r * A.step + A.shift;
    }
    const float* b_ptr[2];
    bool b_ok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n_base + nt * 16 + r16;
        b_ok[nt] = n < N;
        b_ptr[nt] = Bw + (long)(b_ok[nt] ? n : 0) * ldb;
    }

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // tap / channel of this lane's first k, advanced incrementally (VEC path: cw % 4 == 0, no division in the loop)
    int k = 4 * kq;
    int kk = 0, c = k;
    if (VEC) {
        kk = k / A.cw;
        c = k - kk * A.cw;
    }
    f32x4 fa[2], fb[2], na[2], nb[2];
    load_nt_frags<VEC>(A, a_off, a_r, a_ok, b_ptr, b_ok, k, kk, c, fa, fb);
    for (int k0 = 0; k0 < K; k0 += 16) {
        const bool more = k0 + 16 < K;
        if (more) {
            k += 16;
            if (VEC) {
                c += 16;
                while (c >= A.cw) { c -= A.cw; ++kk; }
            }
            load_nt_frags<VEC>(A, a_off, a_r, a_ok, b_ptr, b_ok, k, kk, c, na, nb);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][v], fb[nt][v], acc[mt][nt], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
        }
    }

    // C/D layout of the 16x16 tile: row = (lane>>4)*4 + i, col = lane&15
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m_base + mt * 16 + kq * 4 + i;
            if (row >= M) continue;
            const int cb = row / cR;
            const int cr = row - cb * cR;
            float* crow = C + (long)cb * cbs + (long)cr * crs;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = n_base + nt * 16 + r16;
                if (col >= N) continue;
                float v = acc[mt][nt][i];
                if (bias) v += bias[col];
                v = act_fn(v, slope);
                if (accumulate) v += crow[col];
                crow[col] = v;
            }
        }
    }
}

// Large-tile path of tg_gemm_nt for the big products of the step (GRU input projections, TCN convs, their input
// gradients): workgroup tile 128 x (32*TN), 4 waves as 2 x 2, each wave 64 x (16*TN) = 4 x TN MFMA tiles, so every
// operand fragment read from LDS feeds 4 (B) or TN (A) MFMAs instead of 2.  16-deep K slabs of A and B are staged through
// LDS with coalesced 16-byte global loads ([row][16 k] rows padded to 20 floats: the k-permuted 16-byte fragment reads
// stay 16-byte aligned and spread over the banks), double-buffered: the next slab's global loads are in flight while
// the current slab's 16*TN MFMAs per wave run.  Needs the vectorisable layout (cw % 4 == 0 etc., checked on the host).
constexpr int BG_LD = 20;

template <int TN>
__global__ __launch_bounds__(256) void gemm_nt_big_kernel(Win A, const float* __restrict__ Bw, long ldb,
                                                          const float* __restrict__ bias, float* __restrict__ C, long cbs,
                                                          long crs, int cR, int M, int N, float slope, int accumulate, int n_nt) {
    constexpr int BN = 32 * TN;                       // workgroup tile width
    constexpr int BROWS = BN / 64;                    // B rows staged per thread (1 or 2)
    __shared__ __attribute__((aligned(16))) float as[2][128][BG_LD];
    __shared__ __attribute__((aligned(16))) float bs[2][BN][BG_LD];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m0 = (lid / n_nt) * 128, n0 = (lid % n_nt) * BN;
    const int K = A.K;

    // staging role: row (t >> 2) [+64], 16-byte piece (t & 3) of the 16-deep slab
    const int srow = t >> 2, sk = 4 * (t & 3);
    long a_off[2];
    int a_r[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + srow + 64 * i;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / A.rows_out;
        const int r = mm - b * A.rows_out;
        a_off[i] = (long)b * A.bs;
        a_r[i] = r * A.step + A.shift;
    }
    const float* b_ptr[BROWS];
    bool b_ok[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + srow + 64 * i;
        b_ok[i] = n < N;
        b_ptr[i] = Bw + (long)(b_ok[i] ? n : 0) * ldb;
    }
    int kk = sk / A.cw, c = sk - (sk / A.cw) * A.cw;   // tap / channel of this thread's piece, advanced by 16 per slab

    f32x4 ga[2], gb[BROWS];
    auto fetch = [&](int k0) {
        const int k = k0 + sk;
        const bool inb = k < K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sr = a_r[i] + kk * A.dil;
            const bool ok = a_ok[i] && inb && sr >= 0 && sr < A.rows_in;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ga[i] = ok ? *reinterpret_cast<const f32x4*>(A.ptr + a_off[i] + (long)sr * A.rs + c) : z;
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            gb[i] = (b_ok[i] && inb) ? *reinterpret_cast<const f32x4*>(b_ptr[i] + k) : z;
        }
        c += 16;
        while (c >= A.cw) { c -= A.cw; ++kk; }
    };

    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&as[buf][srow + 64 * i][sk]) = ga[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *reinterpret_cast<f32x4*>(&bs[buf][srow + 64 * i][sk]) = gb[i];
        __syncthreads();
        if (k0 + 16 < K) fetch(k0 + 16);
        f32x4 fa[4], fb[TN];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&as[buf][wm * 64 + i * 16 + r16][4 * kq]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&bs[buf][wn * (16 * TN) + j * 16 + r16][4 * kq]);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][v], fb[j][v], acc[i][j], 0, 0, 0);
        buf ^= 1;
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * 64 + i * 16 + kq * 4 + q;
            if (row >= M) continue;
            const int cb = row / cR;
            const int cr = row - cb * cR;
            float* crow = C + (long)cb * cbs + (long)cr * crs;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + r16;
                if (col >= N) continue;
                float v = acc[i][j][q];
                if (bias) v += bias[col];
                v = act_fn(v, slope);
                if (accumulate) v += crow[col];
                crow[col] = v;
            }
        }
    }
}

// dW[n][k] += sum_m dY[m][n] * A(m,k)  (and dbias[n] += sum_m dY[m][n] when asked).
// Workgroup = 4 waves as 2 (n) x 2 (k), each 32x32 of dW, i.e. a 64 x 64 tile of dW per workgroup; the m range is split
// over blockIdx.z.  Per 16-row tile every thread fetches one 16-byte piece of dY and one of A (coalesced rows), the
// pieces go to LDS ([row][col], row stride 68 floats: the column-strided MFMA fragment reads are conflict-free), the
// next tile's global loads are in flight while the current tile's 16 MFMAs per wave run.
constexpr int TN_LD = 68;

__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dY, long ldy, Win A, float* __restrict__ dW,
                                                      long ldw, int M, int N, int rows_per_split, int out_kw,
                                                      float* __restrict__ partial, float* __restrict__ dbias, int vec_y, int vec_a,
                                                      int n_nt, int n_kt) {
    __shared__ __attribute__((aligned(16))) float ys[2][16][TN_LD];
    __shared__ __attribute__((aligned(16))) float xs[2][16][TN_LD];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, mq = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    // logical order: (n tile, k tile) fastest, split slowest -> the tiles that re-read one chunk of rows share an XCD
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tn_n = lid % n_nt, tn_k = (lid / n_nt) % n_kt, tn_s = lid / (n_nt * n_kt);
    const int n0 = tn_n * 64, k0 = tn_k * 64;
    const int K = A.K;
    const int m_begin = tn_s * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);

    // staging role of this thread: row (t>>4) of the 16-row tile, 4 consecutive columns starting at 4*(t&15)
    const int srow = t >> 4, scol = 4 * (t & 15);
    const int yn = n0 + scol;                 // first dY column of this thread's piece
    const int ak = k0 + scol;                 // first k of this thread's piece
    int a_roff[4], a_ch[4];
    bool a_kok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a_kok[q] = ak + q < K;
        const int kc = a_kok[q] ? ak + q : 0;
        const int kk = kc / A.cw;
        a_roff[q] = kk * A.dil;
        a_ch[q] = kc - kk * A.cw;
    }
    int mb = 0, mr = 0;                       // (clip, row in clip) of this thread's staging row, advanced incrementally
    {
        const int m = m_begin + srow;
        mb = m / A.rows_out;
        mr = m - mb * A.rows_out;
    }
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = dbias != nullptr && tn_k == 0;

    auto fetch = [&](int m0, f32x4& yv, f32x4& xv) {
        const int m = m0 + srow;
        const bool ok = m < m_end;
        yv = f32x4{0.f, 0.f, 0.f, 0.f};
        xv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float* yp = dY + (long)m * ldy + yn;
            if (vec_y && yn + 3 < N) {
                yv = *reinterpret_cast<const f32x4*>(yp);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) yv[q] = (yn + q < N) ? yp[q] : 0.f;
            }
            const long base = (long)mb * A.bs;
            const int sr0 = mr * A.step + A.shift;
            if (vec_a && a_kok[3]) {            // cw % 4 == 0: the four k share one tap
                const int sr = sr0 + a_roff[0];
                if (sr >= 0 && sr < A.rows_in) xv = *reinterpret_cast<const f32x4*>(A.ptr + base + (long)sr * A.rs + a_ch[0]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sr = sr0 + a_roff[q];
                    xv[q] = (a_kok[q] && sr >= 0 && sr < A.rows_in) ? A.ptr[base + (long)sr * A.rs + a_ch[q]] : 0.f;
                }
            }
        }
        mr += 16;
        while (mr >= A.rows_out) { mr -= A.rows_out; ++mb; }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 yv, xv;
    fetch(m_begin, yv, xv);
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += 16) {
        *reinterpret_cast<f32x4*>(&ys[buf][srow][scol]) = yv;
        *reinterpret_cast<f32x4*>(&xs[buf][srow][scol]) = xv;
        if (want_bias) bsum += yv;
        __syncthreads();                                   // tile `buf` complete; the other buffer is free again
        if (m0 + 16 < m_end) fetch(m0 + 16, yv, xv);       // next tile's loads fly during the MFMAs
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ml = 4 * mq + v;
            float ya[2], xa[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ya[q] = ys[buf][ml][wn * 32 + q * 16 + r16];
                xa[q] = xs[buf][ml][wk * 32 + q * 16 + r16];
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[nt], xa[kt], acc[nt][kt], 0, 0, 0);
        }
        buf ^= 1;
    }

#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 32 + nt * 16 + mq * 4 + i;
                const int kc = k0 + wk * 32 + kt * 16 + r16;
                if (n < N && kc < K && partial) {
                    partial[((long)tn_s * N + n) * K + kc] = acc[nt][kt][i];     // combined in fp64 by tn_reduce_kernel
                } else if (n < N && kc < K) {
                    // out_kw > 0: k = (tap, channel) is stored channel-major, tap-minor: the (Co, Ci, kw) layout of
                    // nn.Conv1d / ConvTranspose1d weights, so conv weight gradients need no separate permute pass
                    const long off = out_kw > 0 ? (long)(kc % A.cw) * out_kw + kc / A.cw : (long)kc;
                    atomicAdd(&dW[(long)n * ldw + off], acc[nt][kt][i]);
                }
            }
    if (want_bias) {       // column sums of dY over this split: 16 staging rows -> one value per column
        __syncthreads();
        *reinterpret_cast<f32x4*>(&ys[0][srow][scol]) = bsum;
        __syncthreads();
        if (t < 64 && n0 + t < N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += ys[0][r][t];
            atomicAdd(&dbias[n0 + t], s);
        }
    }
}

// Deterministic combine of the split-M partial tiles: fp64 sum over the splits in a fixed order, one rounding, += into dW.
// One (n, k) entry per 16 threads: the split axis is strided over them, then summed in lane order.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, int splits, int N, int K, int cw, int out_kw,
                                                        float* __restrict__ dW, long ldw) {
    __shared__ double sh[256];
    const long total = (long)N * K;
    const int sub = threadIdx.x & 15;
    const long n_iter = (total + 15) / 16;          // every thread of a workgroup runs the same number of iterations
    for (long it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const long i = it * 16 + (threadIdx.x >> 4);
        double s = 0.0;
        if (i < total)
            for (int q = sub; q < splits; q += 16) s += (double)partial[(long)q * total + i];
        sh[threadIdx.x] = s;
        __syncthreads();
        if (sub == 0 && i < total) {
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += sh[threadIdx.x + q];
            const int n = (int)(i / K), kc = (int)(i - (long)n * K);
            const long off = out_kw > 0 ? (long)(kc % cw) * out_kw + kc / cw : (long)kc;
            dW[(long)n * ldw + off] += (float)tot;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ldx, int M, int N, float* __restrict__ out,
                                                     int rows_per_split) {
    __shared__ float red[4][64];
    const int cidx = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cidx;
    const int m_begin = blockIdx.y * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    float s = 0.f;
    if (n < N)
        for (int m = m_begin + rp; m < m_end; m += 4) s += X[(long)m * ldx + n];
    red[rp][cidx] = s;
    __syncthreads();
    if (rp == 0 && n < N) atomicAdd(&out[n], red[0][cidx] + red[1][cidx] + red[2][cidx] + red[3][cidx]);
}

}  // namespace tg

using namespace tg;

// bf16-operand tier (gemm_bf16.hip), selected by tg_set_math_mode(1)
extern "C" int tg_get_math_mode(void);
int tg_gemm_nt_bf16_launch(const Win& w, const float* Bw, long ldb, const float* bias, float* C, long cbs, long crs, int cR, int M,
                           int N, float slope, int accumulate, hipStream_t s);
int tg_gemm_tn_bf16_launch(const float* dY, long ldy, const Win& w, float* dW, long ldw, int M, int N, int out_kw, float* dbias,
                           void* ws, int64_t ws_bytes, hipStream_t s);

static int check_window(const tg_window* w, const char* who) {
    TG_REQUIRE(w && w->ptr, "%s: null window", who);
    TG_REQUIRE(w->K > 0 && w->cw > 0 && w->K % w->cw == 0, "%s: K=%d must be a positive multiple of cw=%d", who, w->K, w->cw);
    TG_REQUIRE(w->rows_out > 0 && w->rows_in > 0, "%s: rows_out/rows_in must be positive", who);
    return 0;
}

extern "C" int tg_gemm_nt(const tg_window* A, const float* Bw, int64_t ldb, const float* bias, float* C,
                          int64_t c_batch_stride, int64_t c_row_stride, int32_t c_rows_out, int32_t M, int32_t N,
                          float act_slope, int32_t accumulate, void* stream) {
    if (int e = check_window(A, "tg_gemm_nt")) return e;
    TG_REQUIRE(Bw && C, "tg_gemm_nt: null pointer");
    TG_REQUIRE(M > 0 && N > 0 && c_rows_out > 0 && ldb >= A->K, "tg_gemm_nt: bad sizes M=%d N=%d ldb=%ld K=%d", M, N, (long)ldb, A->K);
    Win w = to_win(A);
    const bool vec = (w.cw % 4 == 0) && (w.K % 4 == 0) && (w.bs % 4 == 0) && (w.rs % 4 == 0) && aligned16(w.ptr) &&
                     (ldb % 4 == 0) && aligned16(Bw);
    hipStream_t s = (hipStream_t)stream;
    if (tg_get_math_mode() >= 1 && vec && M >= 256 && N >= 32 && w.K >= 32)
        return tg_gemm_nt_bf16_launch(w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope,
                                      accumulate, s);
    if (vec && N >= 96 && M >= 1024 && w.K >= 64) {
        // big products: 128-row tiles; 128 columns per tile unless that leaves the last column tile mostly empty or
        // too few workgroups to fill 256 CUs, then 64
        const int waste128 = cdiv(N, 128) * 128 - N, waste64 = cdiv(N, 64) * 64 - N;
        const bool wide = (waste128 <= waste64 + 32) && ((long)cdiv(M, 128) * cdiv(N, 128) >= 384);
        if (wide) {
            const int n_nt = cdiv(N, 128);
            hipLaunchKernelGGL((gemm_nt_big_kernel<4>), dim3(cdiv(M, 128) * n_nt), dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate, n_nt);
        } else {
            const int n_nt = cdiv(N, 64);
            hipLaunchKernelGGL((gemm_nt_big_kernel<2>), dim3(cdiv(M, 128) * n_nt), dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate, n_nt);
        }
    } else if (N <= 32) {
        const int n_nt = cdiv(N, 32);
        dim3 grid(cdiv(M, 128) * n_nt);
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<true, 4, 1>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate, n_nt);
        else     hipLaunchKernelGGL((gemm_nt_kernel<false, 4, 1>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate, n_nt);
    } else {
        const int n_nt = cdiv(N, 64);
        dim3 grid(cdiv(M, 64) * n_nt);
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<true, 2, 2>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate, n_nt);
        else     hipLaunchKernelGGL((gemm_nt_kernel<false, 2, 2>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate, n_nt);
    }
    return check_launch("tg_gemm_nt");
}

static void tn_plan(int M, int N, int K, bool two_pass, int* splits_out, int* rows_out) {
    const int tiles = cdiv(N, 64) * cdiv(K, 64);
    int splits = 1024 / tiles;
    if (two_pass) {                       // short fp32 chains per split: at most 512 rows, fp64 across splits
        const int by_len = cdiv(M, 512);
        if (splits < by_len) splits = by_len;
    }
    const int max_splits = cdiv(M, 64);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    int rows = cdiv(M, splits);
    rows = ((rows + 15) / 16) * 16;
    *rows_out = rows;
    *splits_out = cdiv(M, rows);
}

extern "C" int64_t tg_gemm_tn_ws_floats(int32_t M, int32_t N, int32_t K) {
    int splits, rows;
    tn_plan(M, N, K, true, &splits, &rows);
    return (int64_t)splits * N * K;
}

extern "C" int tg_gemm_tn(const float* dY, int64_t ldy, const tg_window* A, float* dW, int64_t ldw, int32_t M, int32_t N,
                          int32_t out_kw, float* dbias, float* ws, int64_t ws_floats, void* stream) {
    if (int e = check_window(A, "tg_gemm_tn")) return e;
    TG_REQUIRE(dY && dW && M > 0 && N > 0 && ldy >= N && ldw >= A->K, "tg_gemm_tn: bad arguments");
    TG_REQUIRE(out_kw == 0 || out_kw * A->cw == A->K, "tg_gemm_tn: out_kw=%d must be 0 or K/cw", out_kw);
    Win w = to_win(A);
    if (tg_get_math_mode() == 2 && ws != nullptr)      // math mode 2: ws is the byte workspace of tg_gemm_tn_bf16_ws_bytes()
        return tg_gemm_tn_bf16_launch(dY, (long)ldy, w, dW, (long)ldw, M, N, out_kw, dbias, ws, ws_floats * 4, (hipStream_t)stream);
    int splits, rows_per_split;
    tn_plan(M, N, w.K, ws != nullptr, &splits, &rows_per_split);
    TG_REQUIRE(ws == nullptr || ws_floats >= (int64_t)splits * N * w.K, "tg_gemm_tn: workspace too small (%ld < %ld floats)",
               (long)ws_floats, (long)splits * N * w.K);
    const int vec_y = (ldy % 4 == 0) && aligned16(dY);
    const int vec_a = (w.cw % 4 == 0) && (w.bs % 4 == 0) && (w.rs % 4 == 0) && aligned16(w.ptr);
    const int n_nt = cdiv(N, 64), n_kt = cdiv(w.K, 64);
    dim3 grid(n_nt * n_kt * splits);
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, dY, (long)ldy, w, dW, (long)ldw, M, N, rows_per_split, out_kw,
                       ws, dbias, vec_y, vec_a, n_nt, n_kt);
    if (ws) {
        int blocks = cdiv((long)N * w.K, 16);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws, splits, N, w.K, w.cw, out_kw, dW, (long)ldw);
    }
    return check_launch("tg_gemm_tn");
}

extern "C" int tg_colsum(const float* X, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate, void* stream) {
    TG_REQUIRE(X && out && M > 0 && N > 0 && ldx >= N, "tg_colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s) != hipSuccess) { set_error("tg_colsum: memset failed"); return 1; }
    }
    int splits = 512 / cdiv(N, 64);
    const int max_splits = cdiv(M, 64);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int rows_per_split = cdiv(M, splits);
    splits = cdiv(M, rows_per_split);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), splits), dim3(256), 0, s, X, (long)ldx, M, N, out, rows_per_split);
    return check_launch("tg_colsum");
}
