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

#include <type_traits>
#include <utility>

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

template <int N, typename F, int... I>
__device__ __forceinline__ void nt_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void nt_static_for(F&& f) { nt_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// Workgroup = 4 waves arranged WM x WN, each wave a 32x32 output tile (2x2 MFMA tiles of 16x16).
// KS = 4 (with WM = WN = 1, VEC): the four waves share ONE 32 x 32 tile and split its K range; the partial tiles meet in LDS and wave 0
// runs the epilogue.  For narrow products with a long reduction and too few rows to fill the chip (audio conv4: M = 4352, N = 32,
// K = 960 -> 34 workgroups of 128 rows, each wave walking 60 dependent k-steps): four times the workgroups, a quarter of the chain.
template <bool VEC, int WM, int WN, int NT_RING, int KS = 1>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const NtGroup g) {
    static_assert(KS == 1 || (KS == 4 && WM == 1 && WN == 1 && VEC), "K split: one tile per workgroup, vectorised path");
    const int pi = group_find(g, blockIdx.x);
    const NtProb& pr = g.p[pi];
    const Win A = pr.A;
    const float* __restrict__ Bw = pr.Bw;
    const long ldb = pr.ldb;
    const float* __restrict__ bias = pr.bias;
    const float* __restrict__ mul = pr.mul;
    float* __restrict__ C = pr.C;
    const long cbs = pr.cbs, crs = pr.crs;
    const int cR = pr.cR, M = pr.M, N = pr.N, accumulate = pr.accumulate, n_nt = pr.n_nt;
    const float slope = pr.slope;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = KS > 1 ? 0 : wave / WN, wn = KS > 1 ? 0 : wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;
    // logical order: output-column tile fastest -> the workgroups that re-read one A row panel sit on ONE XCD's L2
    // (every problem's workgroup range starts at a multiple of 8, so bid & 7 still names the XCD)
    const int lid = xcd_chunked_id(blockIdx.x - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
    const int m_base = (lid / n_nt) * (WM * 32) + wm * 32;
    const int n_base = (lid % n_nt) * (WN * 32) + wn * 32;
    if (m_base >= M || n_base >= N) return;   // no LDS, no barriers: a whole wave may leave (KS > 1: the whole workgroup does)
    // this wave's K range: all of it, or its quarter rounded up to whole 16-deep k-steps
    const int kslice = KS > 1 ? ((A.K + 16 * KS - 1) / (16 * KS)) * 16 : A.K;
    const int K_beg = KS > 1 ? wave * kslice : 0;
    const int K = KS > 1 ? (A.K < K_beg + kslice ? A.K : K_beg + kslice) : A.K;       // end of the range

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

    if constexpr (VEC) {
        // Ring of NT_RING k-steps of operands in flight (16 VGPRs each; 4 for small grids with a long K, where nothing else hides the
        // latency, 1 for chip-filling grids, where occupancy does and the registers cost more than they give: conv1-sized launches
        // went 95 -> 128 us with a ring of 4).  The loads are issued UNCONDITIONALLY from an always-valid address and zeroed
        // when the set is used (mask bits): a predicated load makes the count of outstanding loads dynamic and hipcc then drains
        // everything (vmcnt(0)) at the next use -- with the one-deep prefetch of the first version every 16-deep k-step of a small grid
        // cost a full memory latency (audio conv4, 40 workgroups: 56 us for 0.27 GFLOP).
        constexpr int R = NT_RING;
        int k = K_beg + 4 * kq;
        int kk = k / A.cw, c = k - (k / A.cw) * A.cw;      // tap / channel of this lane's next k, advanced by 16 per load (cw % 4 == 0)
        f32x4 ra[R][2], rb[R][2];
        unsigned rm[R];
        auto load = [&](auto set_c) {
            constexpr int j = decltype(set_c)::value;
            const bool inb = k < K;
            unsigned m = 0u;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int sr = a_r[mt] + kk * A.dil;
                const bool ok = a_ok[mt] && inb && sr >= 0 && sr < A.rows_in;
                ra[j][mt] = *reinterpret_cast<const f32x4*>(ok ? A.ptr + a_off[mt] + (long)sr * A.rs + c : A.ptr);
                m |= ok ? (1u << mt) : 0u;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bool ok = b_ok[nt] && inb;
                rb[j][nt] = *reinterpret_cast<const f32x4*>(ok ? b_ptr[nt] + k : Bw);
                m |= ok ? (4u << nt) : 0u;
            }
            rm[j] = m;
            k += 16;
            c += 16;
            while (c >= A.cw) { c -= A.cw; ++kk; }
        };
        nt_static_for<R>([&](auto j) { load(j); });
        for (int k0 = K_beg; k0 < K; k0 += 16 * R) {
            nt_static_for<R>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                {                                           // (k-steps past K multiply masked zeros: no branch, so the counts stay static)
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    f32x4 fa[2], fb[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        fa[i] = (rm[j] >> i) & 1u ? ra[j][i] : z;
                        fb[i] = (rm[j] >> (2 + i)) & 1u ? rb[j][i] : z;
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][v], fb[nt][v], acc[mt][nt], 0, 0, 0);
                    // the k-step R ahead (past K: a valid address, masked); a ring sized to hold a whole K range (R >= 8, the few-row decode
                    // products) skips the refill once nothing is left -- its single pass then ends without a second memory latency
                    if constexpr (R >= 8) { if (k0 + 16 * R < K) load(jc); }
                    else load(jc);
                }
            });
        }
    } else {
        int k = 4 * kq;
        f32x4 fa[2], fb[2], na[2], nb[2];
        load_nt_frags<VEC>(A, a_off, a_r, a_ok, b_ptr, b_ok, k, 0, k, fa, fb);
        for (int k0 = 0; k0 < K; k0 += 16) {
            const bool more = k0 + 16 < K;
            if (more) {
                k += 16;
                load_nt_frags<VEC>(A, a_off, a_r, a_ok, b_ptr, b_ok, k, 0, k, na, nb);
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
    }

    if constexpr (KS > 1) {
        // partial tiles of waves 1 .. KS - 1 through LDS (lane-private slots), summed by wave 0 in wave order
        __shared__ f32x4 red[KS - 1][4][64];
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) red[wave - 1][mt * 2 + nt][lane] = acc[mt][nt];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int q = 0; q < KS - 1; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] += red[q][mt * 2 + nt][lane];
    }

    // C/D layout of the 16x16 tile: row = (lane>>4)*4 + i, col = lane&15.  Every global read of the epilogue (bias, dropout mask,
    // accumulate operand: up to 34 per lane) is issued before the first use, from always-valid addresses; element by element they
    // each waited out an L2 round trip in turn.
    long off[2][4][2];
    bool okc[2][4][2];
    float bv[2] = {0.f, 0.f}, mv[2][4][2], cv[2][4][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n_base + nt * 16 + r16;
        if (bias) bv[nt] = bias[col < N ? col : 0];
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m_base + mt * 16 + kq * 4 + i;
            const int rr = row < M ? row : 0;
            const int cb = rr / cR;
            const int cr = rr - cb * cR;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = n_base + nt * 16 + r16;
                okc[mt][i][nt] = row < M && col < N;
                off[mt][i][nt] = okc[mt][i][nt] ? (long)cb * cbs + (long)cr * crs + col : 0;
                if (mul) mv[mt][i][nt] = mul[off[mt][i][nt]];
                if (accumulate) cv[mt][i][nt] = C[off[mt][i][nt]];
            }
        }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float v = act_fn(acc[mt][nt][i] + bv[nt], slope);
                if (mul) v *= mv[mt][i][nt];
                if (accumulate) v += cv[mt][i][nt];
                if (okc[mt][i][nt]) C[off[mt][i][nt]] = v;
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
__global__ __launch_bounds__(256) void gemm_nt_big_kernel(const NtGroup g) {
    const int pi = group_find(g, blockIdx.x);
    const NtProb& pr = g.p[pi];
    const Win A = pr.A;
    const float* __restrict__ Bw = pr.Bw;
    const long ldb = pr.ldb, b_seg_stride = pr.b_seg_stride;
    const int b_seg_k = pr.b_seg_k;
    const float* __restrict__ bias = pr.bias;
    const float* __restrict__ mul = pr.mul;
    float* __restrict__ C = pr.C;
    const long cbs = pr.cbs, crs = pr.crs;
    const int cR = pr.cR, M = pr.M, N = pr.N, accumulate = pr.accumulate, n_nt = pr.n_nt;
    const float slope = pr.slope;
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
    const int lid = xcd_chunked_id(blockIdx.x - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    if (m0 >= M) return;                               // padding workgroup of a grouped launch (uniform: before any barrier)
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
    int bsg = sk / b_seg_k, bc = sk - (sk / b_seg_k) * b_seg_k;   // weight segment / column inside it, likewise

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
            gb[i] = (b_ok[i] && inb) ? *reinterpret_cast<const f32x4*>(b_ptr[i] + bsg * b_seg_stride + bc) : z;
        }
        c += BK;
        while (c >= A.cw) { c -= A.cw; ++kk; }
        bc += BK;
        while (bc >= b_seg_k) { bc -= b_seg_k; ++bsg; }
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
            const float* mrow = mul ? mul + (long)cb * cbs + (long)cr * crs : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + r16;
                if (col >= N) continue;
                float v = acc[i][j][q];
                if (bias) v += bias[col];
                v = act_fn(v, slope);
                if (mrow) v *= mrow[col];
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
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnGroup g) {
    const int pi = group_find(g, blockIdx.x);
    const TnProb& pr = g.p[pi];
    const float* __restrict__ dY = pr.dY;
    const long ldy = pr.ldy, ldw = pr.ldw;
    const Win A = pr.A;
    float* __restrict__ dW = pr.dW;
    const int M = pr.M, N = pr.N, rows_per_split = pr.rows_per_split, out_kw = pr.out_kw;
    float* __restrict__ partial = pr.partial;
    float* __restrict__ dbias = pr.dbias;
    const int vec_y = pr.vec_y, vec_a = pr.vec_a, n_nt = pr.n_nt, n_kt = pr.n_kt;
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
    const int lid = xcd_chunked_id(blockIdx.x - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
    const int tn_n = lid % n_nt, tn_k = (lid / n_nt) % n_kt, tn_s = lid / (n_nt * n_kt);
    const int n0 = tn_n * BN, k0 = tn_k * BC;
    const int K = A.K;
    const int m_begin = tn_s * rows_per_split;
    if (m_begin >= M) return;                              // padding workgroup of a grouped launch (uniform: before any barrier)
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
// One (n, k) entry per SUB consecutive threads (SUB = 16 or 256: the ~2000 splits of the audio conv1 gradient, whose output has only
// 240 entries, need a whole workgroup per entry -- 16 threads per entry took 65 us there): the split axis is strided over them, the
// partial sums are then added in lane order.
template <int SUB>
__global__ __launch_bounds__(256) void tn_reduce_kernel_t(const float* __restrict__ partial, int splits, int N, int K, int cw, int out_kw,
                                                          float* __restrict__ dW, long ldw) {
    __shared__ double sh[256];
    constexpr int EPW = 256 / SUB;                  // entries per workgroup and iteration
    const long total = (long)N * K;
    const int sub = threadIdx.x % SUB;
    const long n_iter = (total + EPW - 1) / EPW;    // every thread of a workgroup runs the same number of iterations
    for (long it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const long i = it * EPW + threadIdx.x / SUB;
        double s = 0.0;
        if (i < total)
            for (int q = sub; q < splits; q += SUB) s += (double)partial[(long)q * total + i];
        sh[threadIdx.x] = s;
        __syncthreads();
#pragma unroll
        for (int w = SUB / 2; w >= 16; w >>= 1) {   // tree down to 16 partials per entry (fixed order)
            if (sub < w) sh[threadIdx.x] += sh[threadIdx.x + w];
            __syncthreads();
        }
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

// Deterministic column sums (tg_set_deterministic): one workgroup per CW columns walks ALL rows (1024 / CW row lanes per column, fp64: lane r adds
// rows r, r + RL, .. in order), the lanes' sums meet in LDS in lane order; no split over workgroups, no atomics.  CW = 16 where 64-column
// workgroups would leave the chip empty (N = 900: 15 workgroups took 23 us per launch, 57 take a quarter of that; 64-byte row segments).
template <int CW>
__global__ __launch_bounds__(1024) void colsum_det_kernel(const float* __restrict__ X, long ldx, int M, int N, float* __restrict__ out, int accumulate) {
    constexpr int RL = 1024 / CW;
    __shared__ double red[RL][CW];
    const int cidx = threadIdx.x % CW, rp = threadIdx.x / CW;
    const int n = blockIdx.x * CW + cidx;
    double s = 0.0;
    if (n < N) {
        int m = rp;
        for (; m + 7 * RL < M; m += 8 * RL) {             // eight independent loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = X[(long)(m + RL * u) * ldx + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; m < M; m += RL) s += X[(long)m * ldx + n];
    }
    red[rp][cidx] = s;
    __syncthreads();
    if (rp == 0 && n < N) {
        double t = 0.0;
        for (int q = 0; q < RL; ++q) t += red[q][cidx];
        out[n] = (accumulate ? out[n] : 0.f) + (float)t;
    }
}

}  // namespace tg

using namespace tg;

// bf16 matrix-core path of the big products (gemm_split.hip): three-way operand split at fp32 accuracy (math mode 0, default) or
// plain bf16 operands (math mode 1).  TG_GEMM_X3=0 in the environment keeps math mode 0 on the f32-MFMA kernels below.
extern "C" int tg_get_math_mode(void);
int tg_gemm_nt_split_launch(NtGroup& g, hipStream_t s);
int tg_gemm_tn_split_launch(const TnGroup& g, int total_wgs, int tnw, int tkw, hipStream_t s);
bool tg_gemm_tn_mw_plan(TnGroup& g, int* splits_out, int* grid, const long* ws_floats);
int tg_gemm_tn_mw_launch(const TnGroup& g, int grid, int splits, hipStream_t s);
int tg_gemm_tn_mw_reduce_launch(const TnGroup& g, const int* splits, hipStream_t s);
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

// kernel family of one product: 0 = big (LDS-staged: split-bf16 or f32 tiles), 1 = narrow (N <= 32), 2 = small 64 x 64
static bool nt_vec(const NtProb& p) {
    const Win& w = p.A;
    return (w.cw % 4 == 0) && (w.K % 4 == 0) && (w.bs % 4 == 0) && (w.rs % 4 == 0) && aligned16(w.ptr) && (p.ldb % 4 == 0) && aligned16(p.Bw) &&
           (p.b_seg_k % 4 == 0) && (p.b_seg_stride % 4 == 0);
}
static bool nt_has_ext(const NtProb& p) { return p.gate != nullptr || p.res != nullptr || p.drop_state != nullptr; }
static int nt_family(const NtProb& p) {
    if (nt_vec(p) && p.N >= 48 && p.M >= 1024 && p.A.K >= 64) return 0;
    return p.N <= 32 ? 1 : 2;
}

// Lay the problems of a group out over one grid: every problem's workgroup range starts at a multiple of 8 (the XCD round-robin
// then still maps bid & 7 to the XCD inside each problem); the padding workgroups find m0 >= M and leave.
static int nt_layout(NtGroup& g, int bm, int bn) {
    int wg = 0;
    for (int i = 0; i < g.n; ++i) {
        g.p[i].n_nt = cdiv(g.p[i].N, bn);
        g.wg_begin[i] = wg;
        wg += (cdiv(g.p[i].M, bm) * g.p[i].n_nt + 7) / 8 * 8;
    }
    for (int i = g.n; i <= TG_MAX_GROUP; ++i) g.wg_begin[i] = wg;
    return wg;
}

static int nt_launch(NtGroup& g, hipStream_t s) {
    const int fam = nt_family(g.p[0]);
    bool vec = true;
    for (int i = 0; i < g.n; ++i) {
        TG_REQUIRE(nt_family(g.p[i]) == fam, "tg_gemm_nt_group: problem %d does not fit the kernel family of problem 0 (M=%d N=%d K=%d)", i,
                   g.p[i].M, g.p[i].N, g.p[i].A.K);
        vec = vec && nt_vec(g.p[i]);
        TG_REQUIRE(fam == 0 || g.p[i].b_seg_k == g.p[i].A.K, "tg_gemm_nt: K-concatenated weights need the big-product path (problem %d)", i);
    }
    if (fam == 0 && use_split_path()) return tg_gemm_nt_split_launch(g, s);
    for (int i = 0; i < g.n; ++i)
        TG_REQUIRE(!g.p[i].c_rmax && !g.p[i].c2_rmax, "tg_gemm_nt: c_rowmax / c2_rowmax are outputs of the mover-wave kernel only (tg_gemm_nt_kernel_plan == 2; problem %d)", i);
    for (int i = 0; i < g.n; ++i)
        TG_REQUIRE(!nt_has_ext(g.p[i]), "tg_gemm_nt: gate / res / C2 / regenerated dropout need the big-product split path (tg_gemm_nt_ext_supported; problem %d)", i);
    if (fam == 0) {
        int Mx = 0, Nx = 0;
        for (int i = 0; i < g.n; ++i) { Mx = Mx > g.p[i].M ? Mx : g.p[i].M; Nx = Nx > g.p[i].N ? Nx : g.p[i].N; }
        const NtTile tl = nt_pick_tile(Mx * g.n, Nx);
        const dim3 grid(nt_layout(g, 32 * tl.tm, 32 * tl.tn));
#define TG_NT_BIG(TM_, TN_) hipLaunchKernelGGL((gemm_nt_big_kernel<TM_, TN_>), grid, dim3(256), 0, s, g)
        if (tl.tm == 4 && tl.tn == 4) TG_NT_BIG(4, 4);
        else if (tl.tm == 4 && tl.tn == 3) TG_NT_BIG(4, 3);
        else if (tl.tm == 4 && tl.tn == 2) TG_NT_BIG(4, 2);
        else if (tl.tm == 2 && tl.tn == 2) TG_NT_BIG(2, 2);
        else TG_NT_BIG(2, 1);
#undef TG_NT_BIG
    } else {
        int Kx = 0;
        for (int i = 0; i < g.n; ++i) Kx = Kx > g.p[i].A.K ? Kx : g.p[i].A.K;
        const int wgs = fam == 1 ? nt_layout(g, 128, 32) : nt_layout(g, 64, 64);
        const dim3 grid(wgs);
        const bool deep = wgs < 1024 && Kx >= 256;        // operand ring of 4 k-steps: small grid, long reduction (see the kernel)
        if (fam == 1 && vec && wgs < 256 && Kx >= 256) {
            // few rows, long reduction: one 32 x 32 tile per workgroup, its four waves split K (see the kernel)
            const dim3 grid4(nt_layout(g, 32, 32));
            hipLaunchKernelGGL((gemm_nt_kernel<true, 1, 1, 2, 4>), grid4, dim3(256), 0, s, g);
            return check_launch("tg_gemm_nt");
        }
        int Mmax = 0;
        for (int i = 0; i < g.n; ++i) Mmax = Mmax > g.p[i].M ? Mmax : g.p[i].M;
        if (fam == 2 && vec && wgs <= 64 && Kx >= 256 && Mmax <= 576) {
            // a handful of 64 x 64 tiles with a long reduction and at most 576 rows (a synthesis window of up to 16 utterances: [34 x 300 x 600]
            // text-encoder convs are five workgroups walking 150 dependent k-steps each, 25 us): 32 x 32 tiles whose four waves split K --
            // four times the workgroups, an eighth of the chain: ~11 us (round 6: one utterance 883 -> 723 us per window, profiles/r6_z_decode_b1.txt)
            // Ring of ten k-steps when a wave's K range fits it (K <= 640): every operand of the wave is requested before its first MFMA, one
            // memory latency per launch instead of five (the ring of two held 32 of a wave's 150 k: 11.4 -> see profiles/r6_bg_decode_ring.txt)
            const dim3 grid4(nt_layout(g, 32, 32));
            if (Kx <= 640) hipLaunchKernelGGL((gemm_nt_kernel<true, 1, 1, 10, 4>), grid4, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((gemm_nt_kernel<true, 1, 1, 2, 4>), grid4, dim3(256), 0, s, g);
            return check_launch("tg_gemm_nt");
        }
        if (fam == 1) {
            if (vec && deep) hipLaunchKernelGGL((gemm_nt_kernel<true, 4, 1, 4>), grid, dim3(256), 0, s, g);
            else if (vec)    hipLaunchKernelGGL((gemm_nt_kernel<true, 4, 1, 1>), grid, dim3(256), 0, s, g);
            else             hipLaunchKernelGGL((gemm_nt_kernel<false, 4, 1, 1>), grid, dim3(256), 0, s, g);
        } else {
            if (vec && deep) hipLaunchKernelGGL((gemm_nt_kernel<true, 2, 2, 4>), grid, dim3(256), 0, s, g);
            else if (vec)    hipLaunchKernelGGL((gemm_nt_kernel<true, 2, 2, 1>), grid, dim3(256), 0, s, g);
            else             hipLaunchKernelGGL((gemm_nt_kernel<false, 2, 2, 1>), grid, dim3(256), 0, s, g);
        }
    }
    return check_launch("tg_gemm_nt");
}

static int nt_fill(NtProb& p, const tg_gemm_nt_problem& q, int idx) {
    if (int e = check_window(&q.A, "tg_gemm_nt")) return e;
    TG_REQUIRE(q.Bw && q.C, "tg_gemm_nt: null pointer (problem %d)", idx);
    TG_REQUIRE(q.M > 0 && q.N > 0 && q.c_rows_out > 0, "tg_gemm_nt: bad sizes M=%d N=%d (problem %d)", q.M, q.N, idx);
    const int seg = q.b_seg_k > 0 ? q.b_seg_k : q.A.K;
    TG_REQUIRE(q.A.K % seg == 0 && q.ldb >= seg, "tg_gemm_nt: weight segments of %d do not tile K=%d / ldb=%ld (problem %d)", seg, q.A.K,
               (long)q.ldb, idx);
    p.A = to_win(&q.A);
    p.Bw = q.Bw; p.ldb = (long)q.ldb; p.b_seg_k = seg; p.b_seg_stride = q.b_seg_k > 0 ? (long)q.b_seg_stride : 0;
    TG_REQUIRE(q.drop_state == nullptr || (q.out_scale == nullptr && q.drop_index0 >= 0 && q.drop_index0 % 4 == 0 && q.drop_p >= 0.f && q.drop_p < 1.f),
               "tg_gemm_nt: regenerated dropout (problem %d) excludes out_scale and needs drop_index0 %% 4 == 0, 0 <= drop_p < 1", idx);
    p.drop_state = q.drop_state; p.drop_index0 = (long)q.drop_index0; p.drop_site = q.drop_site; p.drop_p = q.drop_p;
    p.bias = q.bias; p.mul = q.out_scale; p.C = q.C; p.cbs = (long)q.c_batch_stride; p.crs = (long)q.c_row_stride; p.cR = q.c_rows_out;
    p.M = q.M; p.N = q.N; p.slope = q.act_slope; p.accumulate = q.accumulate; p.n_nt = 0;
    p.a_bytes = p.b_bytes = 0;
    p.Bpl = nullptr; p.bpl_plane = 0; p.b_slab_rows = 0; p.b_row0 = 0;
    p.h2 = 0; p.a_scale = nullptr; p.a_rmax = nullptr; p.b_inv = nullptr; p.a_rmax_div = 1;
    p.c_rmax = q.c_rowmax; p.c2_rmax = q.c2_rowmax;
    TG_REQUIRE(!q.c2_rowmax || q.C2, "tg_gemm_nt: c2_rowmax without C2 (problem %d)", idx);
    if (q.b_planes) {
        const int64_t kp = ((int64_t)q.A.K + 31) / 32 * 32;
        // (K-concatenated weight segments may bring planes too: the planes then hold the concatenation, all K columns of every row)
        TG_REQUIRE(q.b_rows >= q.N && q.b_row0 >= 0 && q.b_row0 + (int64_t)q.N <= q.b_rows && aligned16(q.b_planes) &&
                       q.b_plane_stride >= ((int64_t)q.b_rows + 1) * kp && q.b_plane_stride % 8 == 0 && ((int64_t)q.b_rows + 1) * kp < (1LL << 30),
                   "tg_gemm_nt: bad weight planes (problem %d): one weight matrix, its N=%d rows inside the buffer's %d, plane stride >= (rows + 1) * K "
                   "rounded up to 32, 16-byte aligned", idx, q.N, q.b_rows);
        p.Bpl = reinterpret_cast<const __bf16*>(q.b_planes); p.bpl_plane = (long)q.b_plane_stride; p.b_slab_rows = q.b_rows + 1; p.b_row0 = q.b_row0;
        if (q.b_planes_kind == 1) {
            TG_REQUIRE(q.b_inv_scale && ((q.a_row_scale != nullptr) != (q.a_rowmax != nullptr)) && aligned16(q.b_inv_scale) && q.b_row0 % 4 == 0,
                       "tg_gemm_nt: fp16 x 2 weight planes (problem %d) need b_inv_scale (16-byte aligned), exactly one of a_row_scale / a_rowmax and b_row0 %% 4 == 0", idx);
            TG_REQUIRE(!q.a_rowmax || q.A.K <= 2 * q.A.cw, "tg_gemm_nt: a_rowmax serves windows of at most two taps (problem %d: %d); pass a_row_scale (tg_h2_row_scales)",
                       idx, q.A.K / q.A.cw);
            TG_REQUIRE(q.a_rowmax_rows >= 0 && (int64_t)cdiv(q.M, q.A.rows_out) * q.A.rows_in < (1LL << 31), "tg_gemm_nt: bad a_rowmax_rows=%d (problem %d)", q.a_rowmax_rows, idx);
            p.h2 = 1; p.a_scale = q.a_row_scale; p.a_rmax = q.a_rowmax; p.b_inv = q.b_inv_scale; p.a_rmax_div = q.a_rowmax_rows > 0 ? q.a_rowmax_rows : 1;
        } else TG_REQUIRE(q.b_planes_kind == 0, "tg_gemm_nt: b_planes_kind=%d (problem %d): 0 (bf16 x 3) or 1 (fp16 x 2)", q.b_planes_kind, idx);
    }
    TG_REQUIRE((q.res == nullptr) == (q.C2 == nullptr), "tg_gemm_nt: res and C2 go together (problem %d)", idx);
    p.gate = q.gate; p.res = q.res; p.C2 = q.C2; p.res_slope = q.res_slope;
    p.vec_c = (q.N % 4 == 0) && (q.c_batch_stride % 4 == 0) && (q.c_row_stride % 4 == 0) && aligned16(q.C) &&
              (q.bias == nullptr || aligned16(q.bias)) && (q.out_scale == nullptr || aligned16(q.out_scale)) &&
              (q.gate == nullptr || aligned16(q.gate)) && (q.res == nullptr || (aligned16(q.res) && aligned16(q.C2)));
    return 0;
}


extern "C" int32_t tg_gemm_nt_family(const tg_gemm_nt_problem* problem) {
    NtProb p;
    if (!problem || nt_fill(p, *problem, 0)) return -1;
    return nt_family(p);
}

extern "C" int32_t tg_gemm_nt_ext_supported(const tg_gemm_nt_problem* problem) {
    NtProb p;
    if (!problem || nt_fill(p, *problem, 0)) return 0;
    return nt_family(p) == 0 && use_split_path() ? 1 : 0;
}

bool tg_gemm_nt_mw_eligible(NtGroup& g, int* tm, int* tn);
extern "C" int32_t tg_gemm_nt_kernel_plan(const tg_gemm_nt_problem* problems, int32_t n, int32_t* tile_m, int32_t* tile_n) {
    if (!problems || n < 1 || n > TG_MAX_GROUP) return -1;
    NtGroup g;
    g.n = n;
    for (int i = 0; i < n; ++i)
        if (nt_fill(g.p[i], problems[i], i)) return -1;
    for (int i = n; i < TG_MAX_GROUP; ++i) g.p[i] = g.p[0];
    const int fam = nt_family(g.p[0]);
    for (int i = 0; i < n; ++i)
        if (nt_family(g.p[i]) != fam) return -1;
    if (fam != 0 || !use_split_path()) return 0;
    int tm = 0, tn = 0;
    if (!tg_gemm_nt_mw_eligible(g, &tm, &tn)) return 1;
    if (tile_m) *tile_m = 32 * tm;
    if (tile_n) *tile_n = 32 * tn;
    return 2;
}

extern "C" int tg_gemm_nt_group(const tg_gemm_nt_problem* problems, int32_t n, void* stream) {
    TG_REQUIRE(problems && n >= 1 && n <= TG_MAX_GROUP, "tg_gemm_nt_group: 1..%d problems", TG_MAX_GROUP);
    NtGroup g;
    g.n = n;
    for (int i = 0; i < n; ++i)
        if (int e = nt_fill(g.p[i], problems[i], i)) return e;
    for (int i = n; i < TG_MAX_GROUP; ++i) g.p[i] = g.p[0];
    return nt_launch(g, (hipStream_t)stream);
}

extern "C" int tg_gemm_nt(const tg_window* A, const float* Bw, int64_t ldb, const float* bias, float* C,
                          int64_t c_batch_stride, int64_t c_row_stride, int32_t c_rows_out, int32_t M, int32_t N,
                          float act_slope, int32_t accumulate, void* stream) {
    TG_REQUIRE(A, "tg_gemm_nt: null window");
    tg_gemm_nt_problem q = {};
    q.A = *A; q.Bw = Bw; q.ldb = ldb; q.b_seg_k = 0; q.b_seg_stride = 0; q.bias = bias; q.C = C; q.c_batch_stride = c_batch_stride;
    q.c_row_stride = c_row_stride; q.c_rows_out = c_rows_out; q.M = M; q.N = N; q.act_slope = act_slope; q.accumulate = accumulate;
    q.out_scale = nullptr; q.reserved = 0;
    TG_REQUIRE(ldb >= A->K, "tg_gemm_nt: ldb=%ld < K=%d", (long)ldb, A->K);
    return tg_gemm_nt_group(&q, 1, stream);
}

// Split plan of a weight gradient: 64 x 64 tiles of dW (the 128 x 64 / 32-row-slab configuration <4, 2, 32> of the kernel measured
// slower on every shape of the training step, tools/tn_probe.py), the m range cut into ~320-row pieces (20 slabs: enough to
// amortise the prologue and the atomic epilogue) but at least ~640 workgroups; measured optimum on the M = 4352 weight
// gradients: 10-16 splits whatever the tile count.
static void tn_plan(int M, int tiles, bool two_pass, int* splits_out, int* rows_out) {
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

static void launch_tn_reduce(const float* partial, int splits, int N, int K, int cw, int out_kw, float* dW, long ldw, hipStream_t s) {
    const long total = (long)N * K;
    if (splits >= 512) {
        const long blocks = total > 4096 ? 4096 : total;
        hipLaunchKernelGGL(tn_reduce_kernel_t<256>, dim3((int)blocks), dim3(256), 0, s, partial, splits, N, K, cw, out_kw, dW, ldw);
    } else {
        long blocks = (total + 15) / 16;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(tn_reduce_kernel_t<16>, dim3((int)blocks), dim3(256), 0, s, partial, splits, N, K, cw, out_kw, dW, ldw);
    }
}

extern "C" int64_t tg_gemm_tn_ws_floats(int32_t M, int32_t N, int32_t K) {
    int splits, rows;
    tn_plan(M, cdiv(N, 64) * cdiv(K, 64), true, &splits, &rows);       // 64 x 64 tiles: the upper bound over the tile menu
    return (int64_t)splits * N * K;
}

static int tn_fill(TnProb& p, const tg_gemm_tn_problem& q, int idx, int group_tiles, int bn, int bk, int* splits_out) {
    if (int e = check_window(&q.A, "tg_gemm_tn")) return e;
    TG_REQUIRE(q.dY && q.dW && q.M > 0 && q.N > 0 && q.ldy >= q.N && q.ldw >= q.A.K, "tg_gemm_tn: bad arguments (problem %d)", idx);
    TG_REQUIRE(q.out_kw == 0 || q.out_kw * q.A.cw == q.A.K, "tg_gemm_tn: out_kw=%d must be 0 or K/cw (problem %d)", q.out_kw, idx);
    p.A = to_win(&q.A);
    int splits, rows_per_split;
    tn_plan(q.M, cdiv(q.N, bn) * cdiv(p.A.K, bk), q.ws != nullptr, &splits, &rows_per_split);
    if (q.ws == nullptr && group_tiles > cdiv(q.N, bn) * cdiv(p.A.K, bk)) {
        // grouped launch, atomic combine: the group as a whole fills the chip, so each problem needs fewer row splits -- and every split
        // costs one float atomic per output element (memory-side, ~1.3 TB/s chip-wide: 14 splits of the four GRU weight gradients
        // were 91 MB of atomics, a third of the launch).  Aim at ~768 workgroups for the group, at least 256 rows per split.
        constexpr int target_wgs = 768;      // workgroups aimed at per grouped launch (sweep of the final round-2 build: 768 beats 512 / 1024 / 1536 / 2048 by ~0.5 % of the iteration)
        constexpr int target_wgs22 = 2000;   // the 64 x 64 tile's own target (two workgroups per CU by LDS; same-box sweep 768 / 1200 / 2000 / 3200 -> 5.467 / 5.444 / 5.434 / 5.453 ms per iteration)
        int s2 = cdiv(bn == 64 ? target_wgs22 : target_wgs, group_tiles);
        const int cap = cdiv(q.M, 256);
        if (s2 > cap) s2 = cap;
        if (s2 < 1) s2 = 1;
        if (s2 < splits) {
            int rows = cdiv(q.M, s2);
            rows = ((rows + 31) / 32) * 32;
            rows_per_split = rows;
            splits = cdiv(q.M, rows);
        }
    }
    TG_REQUIRE(q.ws == nullptr || q.ws_floats >= (int64_t)splits * q.N * p.A.K, "tg_gemm_tn: workspace too small (%ld < %ld floats, problem %d)",
               (long)q.ws_floats, (long)splits * q.N * p.A.K, idx);
    p.dY = q.dY; p.ldy = (long)q.ldy; p.dW = q.dW; p.ldw = (long)q.ldw; p.M = q.M; p.N = q.N; p.rows_per_split = rows_per_split;
    p.out_kw = q.out_kw; p.partial = q.ws; p.dbias = q.dbias;
    TG_REQUIRE((q.y_colmax == nullptr) == (q.a_colmax == nullptr) && aligned16(q.y_colmax) && aligned16(q.a_colmax),
               "tg_gemm_tn: y_colmax and a_colmax go together, 16-byte aligned (problem %d)", idx);
    p.y_cmax = q.y_colmax; p.a_cmax = q.a_colmax;
    p.vec_y = (q.ldy % 4 == 0) && aligned16(q.dY);
    p.vec_a = (p.A.cw % 4 == 0) && (p.A.bs % 4 == 0) && (p.A.rs % 4 == 0) && aligned16(p.A.ptr);
    p.n_nt = cdiv(q.N, bn); p.n_kt = cdiv(p.A.K, bk);
    *splits_out = splits;
    return 0;
}

// plan_only != nullptr: decide the kernel (0 f32-MFMA, 1 bf16 x 3 staged slabs, 2 bf16 x 3 mover waves) and return without launching
extern "C" int tg_get_tn_workgroup_cap(void);
static int tn_group_impl(const tg_gemm_tn_problem* problems, int32_t n, void* stream, int32_t* plan_only) {
    TG_REQUIRE(problems && n >= 1 && n <= TG_MAX_GROUP, "tg_gemm_tn_group: 1..%d problems", TG_MAX_GROUP);
    TnGroup g;
    g.n = n;
    // bf16 x 3 kernel (gemm_split.hip) when every problem is on the vectorisable layout and long enough to amortise its 32-row slabs
    bool x3 = use_split_path();                                        // weight gradients without a mover-wave plan stay bf16 x 3 in both math modes
    bool mw_ok = use_split_path();                                     // the mover-wave kernel also has the plain-bf16 form (math mode 1)
    bool two_pass = false;            // workspaces are sized for the 64 x 64 tile's split plan (tg_gemm_tn_ws_floats)
    for (int i = 0; i < n; ++i) {
        const tg_gemm_tn_problem& q = problems[i];
        TG_REQUIRE(q.A.ptr && q.A.cw > 0 && q.A.K > 0 && q.N > 0, "tg_gemm_tn: bad arguments (problem %d)", i);
        const bool vec = (q.ldy % 4 == 0) && aligned16(q.dY) && (q.A.cw % 4 == 0) && (q.A.batch_stride % 4 == 0) && (q.A.row_stride % 4 == 0) && aligned16(q.A.ptr);
        x3 = x3 && vec && q.N % 4 == 0 && q.A.K % 4 == 0 && q.M >= 1024 && q.N >= 48 && q.A.K >= 48;
        mw_ok = mw_ok && vec;
        two_pass = two_pass || q.ws != nullptr;
    }
    // tile (tools/tn_tile_lab.py, profiles/r2_tn_tile_lab.txt): 64 x 64, or 128 x 64 when every problem has >= 512 output rows; 128 x 128
    // and wider run at one workgroup per CU and lose 15-30 % on every shape of the step
    int tnw = 2, tkw = 2;
    if (x3 && !two_pass) {
        static int forced = -1;
        if (forced < 0) { const char* e = getenv("TG_TN_TILE"); forced = e ? atoi(e) : 0; }
        int min_n = 1 << 30;
        for (int i = 0; i < n; ++i) min_n = problems[i].N < min_n ? problems[i].N : min_n;
        if (forced >= 22) { tnw = forced / 10; tkw = forced % 10; }       // lab switch
        else if (min_n >= 512) { tnw = 4; tkw = 2; }                      // 128 x 64: the GRU weight gradients (N = 900) 208 -> 185 us per group
    }
    const int bn = 32 * tnw, bk = 32 * tkw;
    int splits[TG_MAX_GROUP], wg = 0, group_tiles = 0;
    for (int i = 0; i < n; ++i)
        if (problems[i].N > 0 && problems[i].A.K > 0) group_tiles += cdiv(problems[i].N, bn) * cdiv(problems[i].A.K, bk);
    for (int i = 0; i < n; ++i) {
        if (int e = tn_fill(g.p[i], problems[i], i, group_tiles, bn, bk, &splits[i])) return e;
        g.wg_begin[i] = wg;
        wg += (g.p[i].n_nt * g.p[i].n_kt * splits[i] + 7) / 8 * 8;
    }
    for (int i = n; i <= TG_MAX_GROUP; ++i) g.wg_begin[i] = wg;
    for (int i = n; i < TG_MAX_GROUP; ++i) g.p[i] = g.p[0];
    hipStream_t s = (hipStream_t)stream;
    bool all_ws = true;
    for (int i = 0; i < n; ++i) all_ws = all_ws && problems[i].ws != nullptr;
    // The mover-wave kernel's workspace combine reduces ALL problems of the group in one launch with plain read-modify-writes (one writer per
    // output element): two problems that accumulate into the same dW or dbias (the output MLP's pair does, below this kernel's size gate) would
    // race there.  Such a group keeps the per-problem combines of the other kernels, which run one after the other.
    bool aliased = false;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            aliased = aliased || problems[i].dW == problems[j].dW || (problems[i].dbias && problems[i].dbias == problems[j].dbias);
    if (aliased && all_ws) mw_ok = false;
    if (mw_ok && (!two_pass || all_ws)) {
        // big gradients whose 192 x 160 tiles fill the chip: mover-wave kernel (gemm_tn_mw.hip), bf16 x 3 or plain bf16.  Row splits are
        // combined by float atomics, or -- when EVERY problem brings a workspace -- through tile-sized partial images and a fixed-order second
        // pass (deterministic, and faster: the atomics all land at the end of the launch)
        int grid = 0, mw_splits[TG_MAX_GROUP];
        long wsf[TG_MAX_GROUP];
        TnGroup g2 = g;
        for (int i = 0; i < n; ++i) wsf[i] = (long)problems[i].ws_floats;
        if (tg_gemm_tn_mw_plan(g2, mw_splits, &grid, wsf)) {
            if (plan_only) { *plan_only = 2; return 0; }
            if (tg_get_math_mode() == 1)
                for (int i = 0; i < TG_MAX_GROUP; ++i) g2.p[i].y_cmax = g2.p[i].a_cmax = nullptr;       // the bf16 tier has its own operand format
            if (int e = tg_gemm_tn_mw_launch(g2, grid, tg_get_math_mode() == 1 ? 1 : 3, s)) return e;
            return all_ws ? tg_gemm_tn_mw_reduce_launch(g2, mw_splits, s) : 0;
        }
    }
    if (plan_only) { *plan_only = x3 ? 1 : 0; return 0; }
    // tg_set_tn_workgroup_cap promises a launch that occupies at most that many CUs (it runs beside a cluster-synchronised recurrence whose
    // workgroups must all stay resident): only the persistent mover-wave kernel keeps that promise -- the kernels below flood the chip with short
    // workgroups.  A capped group they would take is refused; the caller probes the plan with the problems it will launch (ops.tn_kernel_plan).
    TG_REQUIRE(tg_get_tn_workgroup_cap() == 0, "tg_gemm_tn_group: a workgroup cap of %d is set but this group does not run on the persistent "
               "mover-wave kernel (tg_gemm_tn_kernel_plan != 2): launch it without the cap", tg_get_tn_workgroup_cap());
    if (x3) {
        if (int e = tg_gemm_tn_split_launch(g, wg, tnw, tkw, s)) return e;
    } else {
        hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 16>), dim3(wg), dim3(256), 0, s, g);
    }
    for (int i = 0; i < n; ++i)
        if (g.p[i].partial) {       // deterministic fp64 combine of this problem's split partials
            const TnProb& p = g.p[i];
            launch_tn_reduce(p.partial, splits[i], p.N, p.A.K, p.A.cw, p.out_kw, p.dW, p.ldw, s);
        }
    return check_launch("tg_gemm_tn");
}

extern "C" int tg_gemm_tn_group(const tg_gemm_tn_problem* problems, int32_t n, void* stream) { return tn_group_impl(problems, n, stream, nullptr); }

extern "C" int32_t tg_gemm_tn_kernel_plan(const tg_gemm_tn_problem* problems, int32_t n) {
    int32_t plan = -1;
    if (!problems || n < 1 || n > TG_MAX_GROUP) return -1;
    for (int i = 0; i < n; ++i)
        if (!problems[i].dY || !problems[i].dW || !problems[i].A.ptr) return -1;
    return tn_group_impl(problems, n, nullptr, &plan) == 0 ? plan : -1;
}

extern "C" int tg_gemm_tn(const float* dY, int64_t ldy, const tg_window* A, float* dW, int64_t ldw, int32_t M, int32_t N,
                          int32_t out_kw, float* dbias, float* ws, int64_t ws_floats, void* stream) {
    TG_REQUIRE(A, "tg_gemm_tn: null window");
    tg_gemm_tn_problem q;
    q.dY = dY; q.ldy = ldy; q.A = *A; q.dW = dW; q.ldw = ldw; q.M = M; q.N = N; q.out_kw = out_kw; q.dbias = dbias; q.ws = ws; q.ws_floats = ws_floats;
    return tg_gemm_tn_group(&q, 1, stream);
}

extern "C" int tg_colsum(const float* X, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate, void* stream) {
    TG_REQUIRE(X && out && M > 0 && N > 0 && ldx >= N, "tg_colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (deterministic()) {
        if (cdiv(N, 64) >= 64 || M < 1024) hipLaunchKernelGGL(colsum_det_kernel<64>, dim3(cdiv(N, 64)), dim3(1024), 0, s, X, (long)ldx, M, N, out, accumulate);
        else hipLaunchKernelGGL(colsum_det_kernel<16>, dim3(cdiv(N, 16)), dim3(1024), 0, s, X, (long)ldx, M, N, out, accumulate);
        return check_launch("tg_colsum(deterministic)");
    }
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
