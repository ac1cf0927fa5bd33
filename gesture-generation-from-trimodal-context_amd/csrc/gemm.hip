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
#include <stdlib.h>

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

// LDS-staged path of tg_gemm_nt for the big products of the step (GRU input projections, TCN convs, their input
// gradients): workgroup tile (32*TM) x (32*TN), 4 waves as 2 x 2, each wave (16*TM) x (16*TN) = TM x TN MFMA tiles, so every
// operand fragment read from LDS feeds TM (B) or TN (A) MFMAs.  16-deep K slabs of A and B are staged through
// LDS with coalesced 16-byte global loads ([row][16 k] rows padded to 20 floats: the k-permuted 16-byte fragment reads
// stay 16-byte aligned and spread over the banks), double-buffered: the next slab's global loads are in flight while
// the current slab's 4*TM*TN MFMAs per wave run.  Needs the vectorisable layout (cw % 4 == 0 etc., checked on the host).
// BK (slab depth) is a template parameter; 64-deep slabs were tried for the short-K discriminator products (K = 128 / 192: 2-3
// slabs instead of 8-12) and measured within +-8 % of BK = 16 under graph replay (tools/nt_small_probe.py), so only 16 is used.
// The tile is chosen per shape by nt_pick_tile(): the f32 pipe sustains ~100 TFLOP/s on real data whatever the tile
// (tools/gemm_lab.hip), so what matters is tile quantisation and filling 256 CUs x 3 workgroups.
template <int TM, int TN, int BK = 16>
__global__ __launch_bounds__(256) void gemm_nt_big_kernel(Win A, const float* __restrict__ Bw, long ldb,
                                                          const float* __restrict__ bias, float* __restrict__ C, long cbs,
                                                          long crs, int cR, int M, int N, float slope, int accumulate, int n_nt) {
    constexpr int BM = 32 * TM, BN = 32 * TN;         // workgroup tile
    constexpr int PPR = BK / 4;                       // 16-byte pieces per slab row
    constexpr int RPP = 256 / PPR;                    // slab rows staged per pass of the 256 threads
    constexpr int NPA = (BM + RPP - 1) / RPP;         // pieces staged per thread
    constexpr int NPB = (BN + RPP - 1) / RPP;
    constexpr int LD = BK + 4;                        // padded slab row (BG_LD for BK = 16)
    __shared__ __attribute__((aligned(16))) float as[2][BM][LD];
    __shared__ __attribute__((aligned(16))) float bs[2][BN][LD];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    const int K = A.K;

    // staging role: row (t / PPR) [+RPP per extra piece], 16-byte piece (t % PPR) of the BK-deep slab
    const int srow = t / PPR, sk = 4 * (t % PPR);
    long a_off[NPA];
    int a_r[NPA];
    bool a_ok[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + srow + RPP * i;
        a_ok[i] = (srow + RPP * i < BM) && m < M;
        const int mm = a_ok[i] ? m : 0;
        const int b = mm / A.rows_out;
        const int r = mm - b * A.rows_out;
        a_off[i] = (long)b * A.bs;
        a_r[i] = r * A.step + A.shift;
    }
    const float* b_ptr[NPB];
    bool b_ok[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int n = n0 + srow + RPP * i;
        b_ok[i] = (srow + RPP * i < BN) && n < N;
        b_ptr[i] = Bw + (long)(b_ok[i] ? n : 0) * ldb;
    }
    int kk = sk / A.cw, c = sk - (sk / A.cw) * A.cw;   // tap / channel of this thread's piece, advanced by BK per slab

    f32x4 ga[NPA], gb[NPB];
    auto fetch = [&](int k0) {
        const int k = k0 + sk;
        const bool inb = k < K;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int sr = a_r[i] + kk * A.dil;
            const bool ok = a_ok[i] && inb && sr >= 0 && sr < A.rows_in;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ga[i] = ok ? *reinterpret_cast<const f32x4*>(A.ptr + a_off[i] + (long)sr * A.rs + c) : z;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            gb[i] = (b_ok[i] && inb) ? *reinterpret_cast<const f32x4*>(b_ptr[i] + k) : z;
        }
        c += BK;
        while (c >= A.cw) { c -= A.cw; ++kk; }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int i = 0; i < NPA; ++i)
            if (BM % RPP == 0 || srow + RPP * i < BM) *reinterpret_cast<f32x4*>(&as[buf][srow + RPP * i][sk]) = ga[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i)
            if (BN % RPP == 0 || srow + RPP * i < BN) *reinterpret_cast<f32x4*>(&bs[buf][srow + RPP * i][sk]) = gb[i];
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
#pragma unroll
        for (int u = 0; u < BK / 16; ++u) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&as[buf][wm * (16 * TM) + i * 16 + r16][16 * u + 4 * kq]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&bs[buf][wn * (16 * TN) + j * 16 + r16][16 * u + 4 * kq]);
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][v], fb[j][v], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }

#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * (16 * TM) + i * 16 + kq * 4 + q;
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

// Tile menu of the LDS-staged path.  Rules read off tools/gemm_lab.hip runs on the shapes of the training step (the f32 pipe
// sustains ~100 TFLOP/s on real data whatever the tile; what matters is tile quantisation and having >= 2-3 workgroups per CU
// in flight): take the largest tile that still yields enough workgroups, smaller tiles below that.
struct NtTile { int tm, tn; };
static NtTile nt_pick_tile(int M, int N) {
    auto wgs = [&](int bm, int bn) { return (long)cdiv(M, bm) * cdiv(N, bn); };
    auto waste = [&](int bn) { return cdiv(N, bn) * bn - N; };
    if (wgs(128, 128) >= 1400 && waste(128) <= waste(64) + 32) return {4, 4};
    if (wgs(128, 96) >= 700 && waste(96) <= waste(64) + 16) return {4, 3};
    if (wgs(128, 64) >= 500) return {4, 2};
    if (wgs(64, 64) >= 640) return {2, 2};
    return {2, 1};
}

// dW[n][k] += sum_m dY[m][n] * A(m,k)  (and dbias[n] += sum_m dY[m][n] when asked).
// Workgroup = 4 waves as 2 (n) x 2 (k), each (16*WTN) x (16*WTK) of dW, i.e. a (32*WTN) x (32*WTK) tile of dW per workgroup;
// the m range is split over workgroups.  Per MR-row slab every thread fetches 16-byte pieces of dY and of A (coalesced rows),
// the pieces go to LDS ([row][col], row stride = width + 4 floats: 4 * stride % 32 == 16, so the column-strided ds_read_b32
// MFMA fragment reads of the two row groups of a 32-lane half hit disjoint banks), the next slab's global loads are in
// flight while the current slab's MR/4 * WTN * WTK MFMAs per wave run: one barrier per slab.  <2, 2, 16> (64 x 64 tile) is the
// configuration in use; larger ones (<4, 2, 32>: 128 x 64, 64 MFMAs per wave per barrier) compile but measured slower here.
template <int WTN, int WTK, int MR>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dY, long ldy, Win A, float* __restrict__ dW,
                                                      long ldw, int M, int N, int rows_per_split, int out_kw,
                                                      float* __restrict__ partial, float* __restrict__ dbias, int vec_y, int vec_a,
                                                      int n_nt, int n_kt) {
    constexpr int BN = 32 * WTN, BC = 32 * WTK;            // tile of dW: BN rows (n) x BC columns (k)
    constexpr int LDY = BN + 4, LDX = BC + 4;
    constexpr int PY = BN / 4, PX = BC / 4;                // 16-byte pieces per slab row
    constexpr int NY = MR * PY / 256, NX = MR * PX / 256;  // pieces per thread
    static_assert(MR * PY % 256 == 0 && MR * PX % 256 == 0 && 256 % PY == 0 && 256 % PX == 0, "staging map");
    __shared__ __attribute__((aligned(16))) float ys[2][MR][LDY];
    __shared__ __attribute__((aligned(16))) float xs[2][MR][LDX];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, mq = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    // logical order: (n tile, k tile) fastest, split slowest -> the tiles that re-read one chunk of rows share an XCD
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tn_n = lid % n_nt, tn_k = (lid / n_nt) % n_kt, tn_s = lid / (n_nt * n_kt);
    const int n0 = tn_n * BN, k0 = tn_k * BC;
    const int K = A.K;
    const int m_begin = tn_s * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);

    // staging roles: dY piece i = row (t / PY) + i * (256 / PY), columns 4 * (t % PY) .. +3; A piece likewise with PX
    const int yrow = t / PY, ycol = 4 * (t % PY);
    const int xrow = t / PX, xcol = 4 * (t % PX);
    const int yn = n0 + ycol;                 // first dY column of this thread's pieces
    const int ak = k0 + xcol;                 // first k of this thread's pieces
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
    int mb[NX], mr[NX];                       // (clip, row in clip) of each A staging row, advanced incrementally
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int m = m_begin + xrow + i * (256 / PX);
        mb[i] = m / A.rows_out;
        mr[i] = m - mb[i] * A.rows_out;
    }
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = dbias != nullptr && tn_k == 0;

    f32x4 yv[NY], xv[NX];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int m = m0 + yrow + i * (256 / PY);
            yv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                const float* yp = dY + (long)m * ldy + yn;
                if (vec_y && yn + 3 < N) {
                    yv[i] = *reinterpret_cast<const f32x4*>(yp);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) yv[i][q] = (yn + q < N) ? yp[q] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int m = m0 + xrow + i * (256 / PX);
            xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                const long base = (long)mb[i] * A.bs;
                const int sr0 = mr[i] * A.step + A.shift;
                if (vec_a && a_kok[3]) {            // cw % 4 == 0: the four k share one tap
                    const int sr = sr0 + a_roff[0];
                    if (sr >= 0 && sr < A.rows_in) xv[i] = *reinterpret_cast<const f32x4*>(A.ptr + base + (long)sr * A.rs + a_ch[0]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sr = sr0 + a_roff[q];
                        xv[i][q] = (a_kok[q] && sr >= 0 && sr < A.rows_in) ? A.ptr[base + (long)sr * A.rs + a_ch[q]] : 0.f;
                    }
                }
            }
            mr[i] += MR;
            while (mr[i] >= A.rows_out) { mr[i] -= A.rows_out; ++mb[i]; }
        }
    };

    f32x4 acc[WTN][WTK];
#pragma unroll
    for (int i = 0; i < WTN; ++i)
#pragma unroll
        for (int j = 0; j < WTK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(m_begin);
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += MR) {
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            *reinterpret_cast<f32x4*>(&ys[buf][yrow + i * (256 / PY)][ycol]) = yv[i];
            if (want_bias) bsum += yv[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) *reinterpret_cast<f32x4*>(&xs[buf][xrow + i * (256 / PX)][xcol]) = xv[i];
        __syncthreads();                                   // slab `buf` complete; the other buffer is free again
        if (m0 + MR < m_end) fetch(m0 + MR);               // next slab's loads fly during the MFMAs
#pragma unroll
        for (int u = 0; u < MR / 16; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ml = 16 * u + 4 * mq + v;
                float ya[WTN], xa[WTK];
#pragma unroll
                for (int q = 0; q < WTN; ++q) ya[q] = ys[buf][ml][wn * 16 * WTN + q * 16 + r16];
#pragma unroll
                for (int q = 0; q < WTK; ++q) xa[q] = xs[buf][ml][wk * 16 * WTK + q * 16 + r16];
#pragma unroll
                for (int nt = 0; nt < WTN; ++nt)
#pragma unroll
                    for (int kt = 0; kt < WTK; ++kt)
                        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[nt], xa[kt], acc[nt][kt], 0, 0, 0);
            }
        buf ^= 1;
    }

#pragma unroll
    for (int nt = 0; nt < WTN; ++nt)
#pragma unroll
        for (int kt = 0; kt < WTK; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 16 * WTN + nt * 16 + mq * 4 + i;
                const int kc = k0 + wk * 16 * WTK + kt * 16 + r16;
                if (n < N && kc < K && partial) {
                    partial[((long)tn_s * N + n) * K + kc] = acc[nt][kt][i];     // combined in fp64 by tn_reduce_kernel
                } else if (n < N && kc < K) {
                    // out_kw > 0: k = (tap, channel) is stored channel-major, tap-minor: the (Co, Ci, kw) layout of
                    // nn.Conv1d / ConvTranspose1d weights, so conv weight gradients need no separate permute pass
                    const long off = out_kw > 0 ? (long)(kc % A.cw) * out_kw + kc / A.cw : (long)kc;
                    atomicAdd(&dW[(long)n * ldw + off], acc[nt][kt][i]);
                }
            }
    if (want_bias) {       // column sums of dY over this split: 256 / PY staging rows -> one value per column
        __syncthreads();
        *reinterpret_cast<f32x4*>(&ys[0][yrow][ycol]) = bsum;
        __syncthreads();
        if (t < BN && n0 + t < N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 256 / PY; ++r) s += ys[0][r][t];
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

// bf16 matrix-core path of the big products (gemm_split.hip): three-way operand split at fp32 accuracy (math mode 0, default) or
// plain bf16 operands (math mode 1).  TG_GEMM_X3=0 in the environment keeps math mode 0 on the f32-MFMA kernels below.
extern "C" int tg_get_math_mode(void);
int tg_gemm_nt_split_launch(const Win& w, const float* Bw, long ldb, const float* bias, float* C, long cbs, long crs, int cR, int M,
                            int N, float slope, int accumulate, hipStream_t s);
static bool use_split_path() {
    static int x3 = -1;
    if (x3 < 0) {
        const char* e = getenv("TG_GEMM_X3");
        x3 = (e && e[0] == '0') ? 0 : 1;
    }
    return x3 == 1 || tg_get_math_mode() == 1;
}

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
    if (vec && N >= 48 && M >= 1024 && w.K >= 64 && use_split_path())
        return tg_gemm_nt_split_launch(w, Bw, (long)ldb, bias, C, (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope,
                                       accumulate, s);
    if (vec && N >= 48 && M >= 1024 && w.K >= 64) {
        const NtTile tl = nt_pick_tile(M, N);
        const int n_nt = cdiv(N, 32 * tl.tn);
        const dim3 grid(cdiv(M, 32 * tl.tm) * n_nt);
#define TG_NT_BIG(TM_, TN_) hipLaunchKernelGGL((gemm_nt_big_kernel<TM_, TN_>), grid, dim3(256), 0, s, w, Bw, (long)ldb, bias, C, \
                                               (long)c_batch_stride, (long)c_row_stride, c_rows_out, M, N, act_slope, accumulate, n_nt)
        if (tl.tm == 4 && tl.tn == 4) TG_NT_BIG(4, 4);
        else if (tl.tm == 4 && tl.tn == 3) TG_NT_BIG(4, 3);
        else if (tl.tm == 4 && tl.tn == 2) TG_NT_BIG(4, 2);
        else if (tl.tm == 2 && tl.tn == 2) TG_NT_BIG(2, 2);
        else TG_NT_BIG(2, 1);
#undef TG_NT_BIG
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

// Split plan of a weight gradient: 64 x 64 tiles of dW (the 128 x 64 / 32-row-slab configuration <4, 2, 32> of the kernel measured
// slower on every shape of the training step, tools/tn_probe.py), the m range cut into ~320-row pieces (20 slabs: enough to
// amortise the prologue and the atomic epilogue) but at least ~640 workgroups; measured optimum on the M = 4352 weight
// gradients: 10-16 splits whatever the tile count.
static void tn_plan(int M, int N, int K, bool two_pass, int* splits_out, int* rows_out) {
    const int tiles = cdiv(N, 64) * cdiv(K, 64);
    int splits = cdiv(M, 320);
    if (splits < cdiv(640, tiles)) splits = cdiv(640, tiles);
    if (two_pass) {                       // short fp32 chains per split: at most 512 rows, fp64 across splits
        const int by_len = cdiv(M, 512);
        if (splits < by_len) splits = by_len;
    }
    const int max_splits = cdiv(M, 64);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    int rows = cdiv(M, splits);
    rows = ((rows + 31) / 32) * 32;
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
    int splits, rows_per_split;
    tn_plan(M, N, w.K, ws != nullptr, &splits, &rows_per_split);
    TG_REQUIRE(ws == nullptr || ws_floats >= (int64_t)splits * N * w.K, "tg_gemm_tn: workspace too small (%ld < %ld floats)",
               (long)ws_floats, (long)splits * N * w.K);
    const int vec_y = (ldy % 4 == 0) && aligned16(dY);
    const int vec_a = (w.cw % 4 == 0) && (w.bs % 4 == 0) && (w.rs % 4 == 0) && aligned16(w.ptr);
    const int n_nt = cdiv(N, 64), n_kt = cdiv(w.K, 64);
    dim3 grid(n_nt * n_kt * splits);
    hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 16>), grid, dim3(256), 0, (hipStream_t)stream, dY, (long)ldy, w, dW, (long)ldw, M, N,
                       rows_per_split, out_kw, ws, dbias, vec_y, vec_a, n_nt, n_kt);
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
        if (zero_async(out, sizeof(float) * (size_t)N, s)) return 1;
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
