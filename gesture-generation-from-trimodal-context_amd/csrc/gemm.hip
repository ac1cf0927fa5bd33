// GEMM-shaped ops on the f32 matrix cores (v_mfma_f32_16x16x4_f32): exact fp32 fmaf chains, so results agree with
// the reference's fp32 ATen kernels to rounding.  One "row window" descriptor (tg_window) covers nn.Linear, every
// Conv1d / ConvTranspose1d of the path (channel-last, im2col never materialised) and the shifted h_{t-1} view the
// GRU weight gradient needs.
//
// Operand staging: no LDS.  The MFMA k index is permuted so that lane group kq = lane>>4 owns k = 16u + 4kq + {0..3}
// of every 16-deep super-step: each lane then feeds four MFMAs from ONE contiguous 16-byte global load per operand
// (both operands use the same permutation, so the sum over k is unchanged).  Operands of this path are L2-resident
// (weights <= 2 MB, activations a few MB per layer); L1/L2 serve the 2x intra-workgroup reuse.
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
                                                      long crs, int cR, int M, int N, float slope, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;
    const int m_base = blockIdx.x * (WM * 32) + wm * 32;
    const int n_base = blockIdx.y * (WN * 32) + wn * 32;
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

// dW[n][k] += sum_m dY[m][n] * A(m,k).  Workgroup = 4 waves as 2 (n) x 2 (k), each 32x32 of dW; the m range is
// split over blockIdx.z and partial tiles are combined with f32 atomics (one global_atomic_add_f32 per element).
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dY, long ldy, Win A, float* __restrict__ dW,
                                                      long ldw, int M, int N, int rows_per_split, int out_kw,
                                                      float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r16 = lane & 15, mq = lane >> 4;
    const int n_base = blockIdx.x * 64 + (wave >> 1) * 32;
    const int k_base = blockIdx.y * 64 + (wave & 1) * 32;
    const int K = A.K;
    if (n_base >= N || k_base >= K) return;
    const int m_begin = blockIdx.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    if (m_begin >= m_end) return;

    int n_idx[2], k_col[2], k_roff[2], k_ch[2];
    bool n_ok[2], k_ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        n_idx[t] = n_base + t * 16 + r16;
        n_ok[t] = n_idx[t] < N;
        k_col[t] = k_base + t * 16 + r16;
        k_ok[t] = k_col[t] < K;
        const int kc = k_ok[t] ? k_col[t] : 0;
        const int kk = kc / A.cw;
        k_roff[t] = kk * A.dil;
        k_ch[t] = kc - kk * A.cw;
    }
    // (clip, row) of the four m values this lane owns per 16-row super-step, advanced incrementally
    int mb[4], mr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int m = m_begin + 4 * mq + v;
        mb[v] = m / A.rows_out;
        mr[v] = m - mb[v] * A.rows_out;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int m0 = m_begin; m0 < m_end; m0 += 16) {
        f32x4 ya[2], xa[2];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = m0 + 4 * mq + v;
            const bool ok = m < m_end;
            const long base = (long)mb[v] * A.bs;
            const int sr0 = mr[v] * A.step + A.shift;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ya[t][v] = (ok && n_ok[t]) ? dY[(long)m * ldy + n_idx[t]] : 0.f;
                const int sr = sr0 + k_roff[t];
                const bool okk = ok && k_ok[t] && sr >= 0 && sr < A.rows_in;
                xa[t][v] = okk ? A.ptr[base + (long)sr * A.rs + k_ch[t]] : 0.f;
            }
            mr[v] += 16;
            while (mr[v] >= A.rows_out) { mr[v] -= A.rows_out; ++mb[v]; }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[nt][v], xa[kt][v], acc[nt][kt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n_base + nt * 16 + mq * 4 + i;
                const int kc = k_base + kt * 16 + r16;
                if (n < N && kc < K && partial) {
                    partial[((long)blockIdx.z * N + n) * K + kc] = acc[nt][kt][i];     // combined in fp64 by tn_reduce_kernel
                } else if (n < N && kc < K) {
                    // out_kw > 0: k = (tap, channel) is stored channel-major, tap-minor: the (Co, Ci, kw) layout of
                    // nn.Conv1d / ConvTranspose1d weights, so conv weight gradients need no separate permute pass
                    const long off = out_kw > 0 ? (long)(kc % A.cw) * out_kw + kc / A.cw : (long)kc;
                    atomicAdd(&dW[(long)n * ldw + off], acc[nt][kt][i]);
                }
            }
}

// Deterministic combine of the split-M partial tiles: fp64 sum in split order, one rounding, then += into dW.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, int splits, int N, int K, int cw, int out_kw,
                                                        float* __restrict__ dW, long ldw) {
    const long total = (long)N * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int q = 0; q < splits; ++q) s += (double)partial[(long)q * total + i];
        const int n = (int)(i / K), kc = (int)(i - (long)n * K);
        const long off = out_kw > 0 ? (long)(kc % cw) * out_kw + kc / cw : (long)kc;
        dW[(long)n * ldw + off] += (float)s;
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
    if (N <= 32) {
        dim3 grid(cdiv(M, 128), cdiv(N, 32));
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<true, 4, 1>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate);
        else     hipLaunchKernelGGL((gemm_nt_kernel<false, 4, 1>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate);
    } else {
        dim3 grid(cdiv(M, 64), cdiv(N, 64));
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<true, 2, 2>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate);
        else     hipLaunchKernelGGL((gemm_nt_kernel<false, 2, 2>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate);
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
                          int32_t out_kw, float* ws, int64_t ws_floats, void* stream) {
    if (int e = check_window(A, "tg_gemm_tn")) return e;
    TG_REQUIRE(dY && dW && M > 0 && N > 0 && ldy >= N && ldw >= A->K, "tg_gemm_tn: bad arguments");
    TG_REQUIRE(out_kw == 0 || out_kw * A->cw == A->K, "tg_gemm_tn: out_kw=%d must be 0 or K/cw", out_kw);
    Win w = to_win(A);
    int splits, rows_per_split;
    tn_plan(M, N, w.K, ws != nullptr, &splits, &rows_per_split);
    TG_REQUIRE(ws == nullptr || ws_floats >= (int64_t)splits * N * w.K, "tg_gemm_tn: workspace too small (%ld < %ld floats)",
               (long)ws_floats, (long)splits * N * w.K);
    dim3 grid(cdiv(N, 64), cdiv(w.K, 64), splits);
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, dY, (long)ldy, w, dW, (long)ldw, M, N, rows_per_split, out_kw, ws);
    if (ws)
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(ew_grid((long)N * w.K, 256, 1)), dim3(256), 0, (hipStream_t)stream, ws, splits, N, w.K, w.cw,
                           out_kw, dW, (long)ldw);
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
