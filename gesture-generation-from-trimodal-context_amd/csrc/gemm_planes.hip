// tg_gemm_nt on PRE-SPLIT operands: the bf16 x 3 product of gemm_split.hip without the splitting in its main loop.
//
// PMC of gemm_nt_split_kernel (profiles/r2_q_pmc_gemm_split_waves.txt, M = 13056, N = 900, K = 600): 4.8 vector instructions per MFMA -- the
// three-way split of every operand element while its slab is staged -- occupy the SIMDs 52 % of the time against 44 % for the matrix pipe,
// and the two overlap in only 28 % of the matrix cycles: the kernel is bound by the sum of its split arithmetic and its MFMAs.  The split of
// an A element is repeated by every column tile that uses it (19 times for N = 1800) and of a weight by every row tile (102 times).
// Here both operands arrive as three bf16 planes (hi / mid / lo, exact: x = hi + mid + lo), written ONCE: weights by the per-optimiser-step
// refresh (layers.WeightPrep), activations by tg_split3_planes or by the kernel that produces them.  The main loop is then 16-byte global
// loads -> 16-byte LDS stores -> fragment reads -> MFMAs.
//
// Plane buffer of a row-major fp32 matrix [rows][cw]: three bf16 planes, each slab-tiled [cwp / 32][rows + 1][32] (common.hpp
// plane_tiled_off; cwp = cw rounded up to 32, zero columns past cw, row `rows` of every slab all zero).
// A conv window (taps of time-shifted rows) addresses source rows of that buffer; a row outside [0, rows_in) is redirected to the zero row,
// so the loop has no predicated loads and no masks.  K' = taps * cwp; a 32-deep slab never crosses a tap (cwp % 32 == 0).  The weights are
// the planes of the [taps * N rows][cw] matrix with row = tap * N + n.
// Rows past M (N) of the last tile are clamped to the last valid row: they only feed outputs the epilogue drops.
#include "common.hpp"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace tg {

struct NpProb {
    const __bf16* A;
    long a_plane;             // elements between planes
    long a_batch_rows;        // rows between batches of the window
    int a_zero_row;           // index of the all-zero row
    int cwp, rows_in, rows_out, step, shift, dil, taps;
    const __bf16* B;          // planes of [taps * N rows][cw], row = tap * N + n
    long b_plane;
    const float* bias;
    const float* mul;
    float* C;
    long cbs, crs;
    int cR, M, N;
    float slope;
    int accumulate;
    int n_nt;
    int vec_c;
};

struct NpGroup {
    int n;
    int wg_begin[TG_MAX_GROUP + 1];
    NpProb p[TG_MAX_GROUP];
};

constexpr int NP_LD = 32;
__device__ __forceinline__ int np_swz(int row) { return ((row >> 3) & 1) << 4; }      // XOR for a bf16 column index (gemm_split.hip sp_swz)

// Workgroup tile (32 TM) x (32 TN), 4 waves as 2 x 2; K slab 32.  One LDS buffer [3][BM + BN][32] bf16 (two barriers per slab), the next
// slab's global loads in flight (registers) during the MFMA section.
template <int TM, int TN, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_nt_planes_kernel(const NpGroup g) {
    const int pi = group_find(g, blockIdx.x);
    const NpProb& pr = g.p[pi];
    constexpr int BM = 32 * TM, BN = 32 * TN, ROWS = BM + BN;
    static_assert(BM % 64 == 0 && BN % 32 == 0, "row groups of 64 (A) / 64 + 32 (B)");
    constexpr int CLD = BN + 4;
    constexpr int OPER_BYTES = 3 * ROWS * NP_LD * 2, CT_BYTES = BM * CLD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OPER_BYTES > CT_BYTES ? OPER_BYTES : CT_BYTES];
    typedef __bf16 (*lds_t)[ROWS][NP_LD];
    const lds_t lds = reinterpret_cast<lds_t>(smem);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int M = pr.M, N = pr.N, n_nt = pr.n_nt;
    const int lid = xcd_chunked_id(blockIdx.x - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    if (m0 >= M) return;                                       // padding workgroup of a grouped launch (uniform: before any barrier)
    const int cwp = pr.cwp;
    const int Kp = pr.taps * cwp;

    // staging map: thread t owns 16-byte slot (t & 3) of image rows (t >> 2) + 64 h, in all three planes (four consecutive lanes cover one
    // 64-byte slab row).  Per row group the only per-thread state is one 32-bit element offset; plane bases are scalar.
    constexpr int GA = BM / 64;                                // A row groups
    constexpr int GBF = BN / 64;                               // full B row groups
    constexpr bool GBH = (BN % 64) != 0;                       // + a half group (32 rows: threads 0..127)
    constexpr int GB = GBF + (GBH ? 1 : 0);
    const int srow = t >> 2, slot = t & 3;
    const bool half_on = t < 128;                              // wave-uniform
    int a_row[GA], a_boff[GA], b_off[GB], dst_a[GA], dst_b[GB];
#pragma unroll
    for (int h = 0; h < GA; ++h) {
        const int row = srow + 64 * h;
        int m = m0 + row;
        m = m < M ? m : M - 1;
        const int b = m / pr.rows_out;
        a_row[h] = (m - b * pr.rows_out) * pr.step + pr.shift;
        a_boff[h] = (int)(b * pr.a_batch_rows) * 32 + 8 * slot;            // element offset of (batch row 0, slot) inside a slab
        dst_a[h] = row * NP_LD + ((8 * slot) ^ np_swz(row));
    }
#pragma unroll
    for (int h = 0; h < GB; ++h) {
        const int row = (h < GBF || half_on) ? srow + 64 * h : srow;          // inactive threads of the half group: any valid row
        int n = n0 + row;
        n = n < N ? n : N - 1;
        b_off[h] = n * 32 + 8 * slot;
        dst_b[h] = (BM + row) * NP_LD + ((8 * slot) ^ np_swz(BM + row));
    }
    const int zero_off = pr.a_zero_row * 32 + 8 * slot;
    const int a_slab = (pr.a_zero_row + 1) * 32, b_slab = (pr.N * pr.taps + 1) * 32;      // elements per 32-column slab of a plane

    u32x4 ga[3][GA], gb[3][GB];
    auto fetch = [&](int k0) {
        const int tap = k0 / cwp, c0 = k0 - tap * cwp;         // uniform
        const int roff = tap * pr.dil;
        int offa[GA];
#pragma unroll
        for (int h = 0; h < GA; ++h) {
            const int sr = a_row[h] + roff;
            // a row outside the window reads the buffer's zero row (no predicated load, no mask)
            offa[h] = (sr >= 0 && sr < pr.rows_in) ? a_boff[h] + sr * 32 : zero_off;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const __bf16* ap = pr.A + s * pr.a_plane + (long)(c0 >> 5) * a_slab;                          // scalar bases
            const __bf16* bp = pr.B + s * pr.b_plane + (long)(c0 >> 5) * b_slab + (long)tap * pr.N * 32;
#pragma unroll
            for (int h = 0; h < GA; ++h) ga[s][h] = *reinterpret_cast<const u32x4*>(ap + offa[h]);
#pragma unroll
            for (int h = 0; h < GB; ++h) gb[s][h] = *reinterpret_cast<const u32x4*>(bp + b_off[h]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fcol = (8 * kq) ^ np_swz(r16);
    __bf16* const lbase = &lds[0][0][0];
    fetch(0);
    for (int k0 = 0; k0 < Kp; k0 += 32) {
        if (k0 > 0) __syncthreads();                           // everybody has read the previous slab
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int h = 0; h < GA; ++h) *reinterpret_cast<u32x4*>(lbase + s * ROWS * NP_LD + dst_a[h]) = ga[s][h];
#pragma unroll
            for (int h = 0; h < GBF; ++h) *reinterpret_cast<u32x4*>(lbase + s * ROWS * NP_LD + dst_b[h]) = gb[s][h];
        }
        if constexpr (GBH) {
            if (half_on) {
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(lbase + s * ROWS * NP_LD + dst_b[GB - 1]) = gb[s][GB - 1];
            }
        }
        __syncthreads();
        if (k0 + 32 < Kp) fetch(k0 + 32);
        // B fragments of the wave's TN column tiles stay for the slab; A fragments come one 16-row tile at a time (the next tile's three
        // reads are issued before this tile's MFMAs): 12 (TN + 2) fragment registers live instead of 12 (TM + TN) -- with all 21 fragments
        // live beside the prefetch set the kernel needs > 168 registers and hipcc, held to three waves per SIMD, serialised the MFMAs of
        // one accumulator back to back.  Within a tile the TN accumulators are interleaved term by term (dependent MFMAs TN issues apart).
        bf16x8 fb[3][TN], fa[2][3];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(&lds[s][BM + wn * (16 * TN) + j * 16 + r16][fcol]);
#pragma unroll
        for (int s = 0; s < 3; ++s) fa[0][s] = *reinterpret_cast<const bf16x8*>(&lds[s][wm * (16 * TM) + r16][fcol]);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i + 1 < TM) {
#pragma unroll
                for (int s = 0; s < 3; ++s) fa[(i + 1) & 1][s] = *reinterpret_cast<const bf16x8*>(&lds[s][wm * (16 * TM) + (i + 1) * 16 + r16][fcol]);
            }
            const bf16x8 (&a)[3] = fa[i & 1];
            // the six significant partial products, smallest first
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], fb[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], fb[2][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], fb[1][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], fb[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], fb[1][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue (as gemm_nt_split_kernel): accumulator tile row-major through LDS, then 16-byte pieces of C / bias / mask / accumulate
    const float* __restrict__ bias = pr.bias;
    const float* __restrict__ mul = pr.mul;
    float* __restrict__ C = pr.C;
    const long cbs = pr.cbs, crs = pr.crs;
    const int cR = pr.cR, accumulate = pr.accumulate;
    const float slope = pr.slope;
    if (pr.vec_c) {
        float* ct = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    ct[(wm * (16 * TM) + i * 16 + kq * 4 + q) * CLD + wn * (16 * TN) + j * 16 + r16] = acc[i][j][q];
        __syncthreads();
        constexpr int C4 = BN / 4, NP = BM * C4 / 256, CH = 4;
        static_assert(BM * C4 % 256 == 0, "whole pieces per thread");
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += CH) {
            f32x4 bv[CH], mv[CH], cv[CH];
            long o[CH];
            bool ok[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (p0 + u >= NP) continue;
                const int idx = t + 256 * (p0 + u);
                const int rl = idx / C4, c4 = idx - rl * C4;
                const int row = m0 + rl, col = n0 + 4 * c4;
                ok[u] = row < M && col < N;
                const int cb = row / cR;
                const int cr = row - cb * cR;
                o[u] = ok[u] ? (long)cb * cbs + (long)cr * crs + col : 0;
                bv[u] = bias ? *reinterpret_cast<const f32x4*>(bias + (ok[u] ? col : 0)) : z4;
                if (mul) mv[u] = *reinterpret_cast<const f32x4*>(mul + o[u]);
                if (accumulate) cv[u] = *reinterpret_cast<const f32x4*>(C + o[u]);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (p0 + u >= NP) continue;
                const int idx = t + 256 * (p0 + u);
                const int rl = idx / C4, c4 = idx - rl * C4;
                f32x4 v = *reinterpret_cast<const f32x4*>(&ct[rl * CLD + 4 * c4]) + bv[u];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = act_fn(v[q], slope);
                if (mul) v *= mv[u];
                if (accumulate) v += cv[u];
                if (ok[u]) *reinterpret_cast<f32x4*>(C + o[u]) = v;
            }
        }
        return;
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


// ------------------------------------------------------------------------------------------------------------------------------
// The same product with MOVER WAVES and LDS-DMA (the many-row shapes of the stacked forward; one persistent 512-thread workgroup per CU).
//
// gemm_mw.hip showed what bounds a big-tile bf16 x 3 product on fp32 operands: the movers' split arithmetic (~300 vector instructions per
// slab) shares each SIMD's issue port with a matrix wave whose 144 MFMAs hold that port half of the time -- both roles end up waiting
// for each other (profiles/r3_f_mw_roles.txt: 4 500 cycles per slab against 2 304 of matrix pipe).  With BOTH operands pre-split the
// movers have nothing left to compute: a slab is 3 planes x (BM + BN) rows x 64 bytes that go global -> LDS by DMA
// (buffer_load_dwordx4 ... lds: one wave-instruction lands 16 rows x 64 B, no VGPR, no ds_write), 15 instructions per mover wave and slab.
// The LDS image is lane-linear per DMA (row = lane >> 2, 16-byte chunk = lane & 3), so the bank swizzle of the fragment reads
// (chunk ^= 2 for rows 8-15 of every 16) is applied to the SOURCE chunk each lane fetches (cdna_hip_programming.md rule 21).
// Window rows outside [0, rows_in) and nothing else are redirected to the plane buffer's zero row; rows past M / N are clamped (their
// products are dropped by the epilogue); the K tail is zero columns of the planes.
// Matrix waves: as gemm_mw.hip (product taken transposed, epilogue straight from the accumulators, persistent over tiles).
constexpr unsigned NPM_RSRC3 = 0x00020000u;
typedef __attribute__((address_space(3))) void lds_void;
// one DMA: 64 lanes x 16 bytes from the buffer (per-lane byte offset `voff`, uniform `soff`) to 1 KB of LDS at byte offset `lds_off` of smem.
// (The address-space cast only exists in the device pass: the host pass of this template would otherwise drop the kernel's launch stub.)
// a buffer descriptor whose every input is PROVABLY wave-uniform to the compiler (cdna_hip_programming.md T20: otherwise each buffer
// operation is wrapped in a readfirstlane / saveexec "waterfall" loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t npm_rsrc(const __bf16* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), NPM_RSRC3);
}
__device__ __forceinline__ void npm_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* smem, unsigned lds_off, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + lds_off), 16, voff, soff, 0, 0);
#endif
}

template <int TM, int TN>
__global__ __launch_bounds__(512, 2) void gemm_np_mw_kernel(const NpGroup g) {
    constexpr int BM = 32 * TM, BN = 32 * TN, ROWS = BM + BN;
    constexpr int PLANE_B = ROWS * 64, BUF_B = 3 * PLANE_B;    // bytes
    constexpr int GA = BM / 16, GALL = ROWS / 16;              // 16-row DMA groups: A first, then B
    constexpr int MAXG = (GALL + 3) / 4;                       // groups per mover wave (group = mover + 4 u)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF_B];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int total_tiles = g.wg_begin[TG_MAX_GROUP];
    const int G = gridDim.x;
    auto decode = [&](int vb, int& pi, int& m0, int& n0, int& nslab) -> bool {
        pi = group_find(g, vb);
        const NpProb& pr = g.p[pi];
        const int lid = xcd_chunked_id(vb - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
        m0 = (lid / pr.n_nt) * BM;
        n0 = (lid % pr.n_nt) * BN;
        nslab = (pr.taps * pr.cwp) >> 5;
        return m0 < pr.M;
    };
    int total = 0;
    for (int vb = blockIdx.x; vb < total_tiles; vb += G) {
        int pi, m0, n0, ns;
        if (decode(vb, pi, m0, n0, ns)) total += ns;
    }

    if (wave >= 4) {
        // ============================================================================================ movers (waves 4-7)
        const int mw = wave - 4;
        const int grow = lane >> 2;                                            // row inside a 16-row group
        const unsigned chunk_b = (unsigned)(((lane & 3) ^ (((grow >> 3) & 1) << 1)) * 16);   // source chunk (bytes) of this lane's LDS slot
        int vb_f = blockIdx.x - G, s_f = 0, nslab_f = 0, tap_f = 0, c0_f = 0;
        bool live = true;
        unsigned voff[MAXG];                                                   // B groups: final byte offset; A groups: batch offset + chunk
        int a_row[MAXG];
        unsigned zero_off = 0, a_slab_b = 0, b_slab_b = 0, b_tap_b = 0;      // bytes: per 32-column slab of a plane, per tap of the weights
        int cwp = 32, dil = 0, rows_in = 0;
        const __bf16* a_base = g.p[0].A;
        const __bf16* b_base = g.p[0].B;
        long a_plane = 0, b_plane = 0;
        unsigned a_bytes = 0, b_bytes = 0;
        auto next_tile = [&]() {
            int pi = 0, m0 = 0, n0 = 0;
            do {
                vb_f += G;
                if (vb_f >= total_tiles) { live = false; break; }
            } while (!decode(vb_f, pi, m0, n0, nslab_f));
            s_f = 0; tap_f = 0; c0_f = 0;
            if (!live) return;
            const NpProb& pr = g.p[pi];
            cwp = pr.cwp; dil = pr.dil; rows_in = pr.rows_in;
            a_slab_b = (unsigned)(pr.a_zero_row + 1) * 64u;
            b_slab_b = (unsigned)(pr.N * pr.taps + 1) * 64u;
            b_tap_b = (unsigned)pr.N * 64u;
            zero_off = (unsigned)pr.a_zero_row * 64u + chunk_b;
#pragma unroll
            for (int u = 0; u < MAXG; ++u) {
                const int grp = mw + 4 * u;                                    // wave-uniform
                if (grp < GA) {
                    int m = m0 + 16 * grp + grow;
                    m = m < pr.M ? m : pr.M - 1;
                    const int b = m / pr.rows_out;
                    a_row[u] = (m - b * pr.rows_out) * pr.step + pr.shift;
                    voff[u] = (unsigned)(b * pr.a_batch_rows) * 64u + chunk_b;
                } else {
                    int n = n0 + 16 * (grp - GA) + grow;
                    n = n < pr.N ? n : pr.N - 1;
                    a_row[u] = 0;
                    voff[u] = (unsigned)n * 64u + chunk_b;
                }
            }
            a_bytes = (unsigned)(pr.a_zero_row + 1) * (unsigned)(cwp * 2); b_bytes = (unsigned)(pr.N * pr.taps + 1) * (unsigned)(cwp * 2);
            a_base = pr.A; b_base = pr.B; a_plane = pr.a_plane; b_plane = pr.b_plane;
        };
        // DMA of the cursor's slab into LDS buffer `buf`, then advance the cursor
        auto fill = [&](int buf) {
            const unsigned c0b = (unsigned)__builtin_amdgcn_readfirstlane((c0_f >> 5) * (int)a_slab_b);                                   // the slab's
            const unsigned k0b = (unsigned)__builtin_amdgcn_readfirstlane((c0_f >> 5) * (int)b_slab_b + tap_f * (int)b_tap_b);            // byte bases
            const int roff = tap_f * dil;
            __amdgpu_buffer_rsrc_t ar[3], br[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) { ar[s] = npm_rsrc(a_base + s * a_plane, a_bytes); br[s] = npm_rsrc(b_base + s * b_plane, b_bytes); }
#pragma unroll
            for (int u = 0; u < MAXG; ++u) {
                const int grp = mw + 4 * u;
                if (grp >= GALL) continue;                                     // wave-uniform
                const unsigned lbase = (unsigned)(buf * BUF_B + grp * 1024);
                if (grp < GA) {
                    const int sr = a_row[u] + roff;
                    const unsigned vo = ((unsigned)sr < (unsigned)rows_in) ? voff[u] + (unsigned)sr * 64u : zero_off;
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        npm_dma16(ar[s], smem, lbase + s * PLANE_B, vo, c0b);
                } else {
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        npm_dma16(br[s], smem, lbase + s * PLANE_B, voff[u], k0b);
                }
            }
            c0_f += 32;
            if (c0_f >= cwp) { c0_f = 0; ++tap_f; }                            // (cwp % 32 == 0: a slab never crosses a tap)
            if (++s_f >= nslab_f) next_tile();
        };
#define NPM_BARRIER_MOVER() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
        next_tile();
        if (live) fill(0);                                                     // slab 0
        NPM_BARRIER_MOVER();
        // step n: the matrix waves multiply slab n out of buffer n & 1 while slab n + 1 lands in the other one (its last readers passed
        // the previous barrier); the DMAs of a step are waited for before the step's barrier
        for (int n = 0; n < total; ++n) {
            if (n + 1 < total) fill((n + 1) & 1);
            NPM_BARRIER_MOVER();
        }
#undef NPM_BARRIER_MOVER
    } else {
        // ============================================================================================ matrix waves (0-3)
        const int wm = (wave >> 1) & 1, wn = wave & 1;
        const int r16 = lane & 15, kq = lane >> 4;
        const int fcol_b = ((8 * kq) ^ np_swz(r16)) * 2;                      // byte offset of the lane's 16-byte fragment inside its 64-byte row
        f32x4 acc[TM][TN];
        auto multiply = [&](int buf) {
            const unsigned char* const lb = smem + buf * BUF_B;
            bf16x8 fa[3][TM];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(lb + s * PLANE_B + (wm * (16 * TM) + i * 16 + r16) * 64 + fcol_b);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8 fb[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(lb + s * PLANE_B + (BM + wn * (16 * TN) + j * 16 + r16) * 64 + fcol_b);
#pragma unroll
                for (int i = 0; i < TM; ++i) {                                 // weight fragment as the A operand: lane = output row, 4 columns
                    f32x4 cc = acc[i][j];
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[2][i], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[0][i], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[1][i], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[1][i], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[0][i], cc, 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0][i], cc, 0, 0, 0);
                }
            }
        };
#define NPM_BARRIER_MATRIX() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
        NPM_BARRIER_MATRIX();
        int n = 0;
        for (int vb = blockIdx.x; vb < total_tiles; vb += G) {
            int pi, m0, n0, nslab;
            if (!decode(vb, pi, m0, n0, nslab)) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < nslab; ++s) {
                multiply(n & 1);
                NPM_BARRIER_MATRIX();
                ++n;
            }
            const NpProb& pr = g.p[pi];
            const float* __restrict__ bias = pr.bias;
            const float* __restrict__ mul = pr.mul;
            float* __restrict__ C = pr.C;
            const float slope = pr.slope;
            const int cR = pr.cR, accumulate = pr.accumulate, M = pr.M, N = pr.N;
            long ro[TM];
            bool rok[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = m0 + wm * (16 * TM) + i * 16 + r16;
                rok[i] = row < M;
                const int rr = rok[i] ? row : 0;
                const int cb = rr / cR;
                ro[i] = (long)cb * pr.cbs + (long)(rr - cb * cR) * pr.crs;
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + 4 * kq;
                const bool cok = col < N;                                      // N % 4 == 0: a piece is inside or outside as a whole
                const int cc0 = cok ? col : 0;
                const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + cc0) : z4;
                f32x4 mv[TM], cv[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const long o = ro[i] + cc0;
                    if (mul) mv[i] = *reinterpret_cast<const f32x4*>(mul + o);
                    if (accumulate) cv[i] = *reinterpret_cast<const f32x4*>(C + o);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    f32x4 v = acc[i][j] + bv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = act_fn(v[q], slope);
                    if (mul) v *= mv[i];
                    if (accumulate) v += cv[i];
                    if (rok[i] & cok) *reinterpret_cast<f32x4*>(C + ro[i] + cc0) = v;
                }
            }
        }
#undef NPM_BARRIER_MATRIX
    }
}

// fp32 [rows][cw] (row stride ldx) -> planes [3][rows + 1][cwp] bf16: exact three-way split, zero columns past cw, zero row `rows`.
// One thread per 8 output columns of a row: two 16-byte loads where the source allows, three 16-byte stores.
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ x, long ldx, int rows, int cw, int cwp, __bf16* __restrict__ planes,
                                                            long plane_stride, int vec) {
    const int c8 = cwp / 8;
    const long total = (long)(rows + 1) * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        split3_write_piece(x, ldx, rows, cw, cwp, planes, plane_stride, r, c, vec != 0);
    }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_split3_planes(const float* x, int64_t ldx, int32_t rows, int32_t cw, void* planes, int32_t cwp, int64_t plane_stride, void* stream) {
    TG_REQUIRE(x && planes && rows > 0 && cw > 0 && ldx >= cw, "tg_split3_planes: bad arguments");
    TG_REQUIRE(cwp >= cw && cwp % 32 == 0 && plane_stride >= (int64_t)(rows + 1) * cwp && plane_stride % 8 == 0 && aligned16(planes),
               "tg_split3_planes: cwp=%d must be a multiple of 32 >= cw=%d, plane_stride >= (rows + 1) * cwp and a multiple of 8, planes 16-byte aligned", cwp, cw);
    const int vec = (ldx % 4 == 0) && aligned16(x);
    const long total = (long)(rows + 1) * (cwp / 8);
    hipLaunchKernelGGL(split3_planes_kernel, dim3(ew_grid(total, 256, 2)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, cw, cwp,
                       reinterpret_cast<__bf16*>(planes), (long)plane_stride, vec);
    return check_launch("tg_split3_planes");
}

extern "C" int tg_gemm_nt_planes_group(const tg_gemm_nt_planes_problem* problems, int32_t n, void* stream) {
    TG_REQUIRE(problems && n >= 1 && n <= TG_MAX_GROUP, "tg_gemm_nt_planes_group: 1..%d problems", TG_MAX_GROUP);
    NpGroup g;
    g.n = n;
    int Mx = 0, Nx = 0;
    for (int i = 0; i < n; ++i) {
        const tg_gemm_nt_planes_problem& q = problems[i];
        TG_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0, "tg_gemm_nt_planes: null pointer / empty problem %d", i);
        TG_REQUIRE(q.cwp > 0 && q.cwp % 32 == 0 && q.taps >= 1 && q.a_rows > 0 && q.rows_in > 0 && q.rows_out > 0 && q.a_batch_rows >= 0,
                   "tg_gemm_nt_planes: cwp=%d must be a positive multiple of 32, taps >= 1, rows > 0 (problem %d)", q.cwp, i);
        TG_REQUIRE(q.a_plane_stride >= (int64_t)(q.a_rows + 1) * q.cwp && q.a_plane_stride % 8 == 0 && q.b_plane_stride >= ((int64_t)q.N * q.taps + 1) * q.cwp &&
                       q.b_plane_stride % 8 == 0 && aligned16(q.A) && aligned16(q.B),
                   "tg_gemm_nt_planes: plane strides too small / unaligned (problem %d)", i);
        const long nb = ((long)q.M + q.rows_out - 1) / q.rows_out;
        TG_REQUIRE((nb - 1) * q.a_batch_rows + q.rows_in <= q.a_rows, "tg_gemm_nt_planes: window exceeds the plane buffer (problem %d)", i);
        TG_REQUIRE((int64_t)(q.a_rows + 1) * q.cwp < (1LL << 30) && ((int64_t)q.N * q.taps + 1) * q.cwp < (1LL << 30), "tg_gemm_nt_planes: operand too large for 32-bit offsets (problem %d)", i);
        TG_REQUIRE(q.c_rows_out > 0 && q.c_row_stride >= q.N, "tg_gemm_nt_planes: bad C addressing (problem %d)", i);
        NpProb& p = g.p[i];
        p.A = reinterpret_cast<const __bf16*>(q.A); p.a_plane = q.a_plane_stride; p.a_batch_rows = q.a_batch_rows; p.a_zero_row = q.a_rows;
        p.cwp = q.cwp; p.rows_in = q.rows_in; p.rows_out = q.rows_out; p.step = q.row_step; p.shift = q.shift; p.dil = q.dil; p.taps = q.taps;
        p.B = reinterpret_cast<const __bf16*>(q.B); p.b_plane = q.b_plane_stride;
        p.bias = q.bias; p.mul = q.out_scale; p.C = q.C; p.cbs = q.c_batch_stride; p.crs = q.c_row_stride; p.cR = q.c_rows_out;
        p.M = q.M; p.N = q.N; p.slope = q.act_slope; p.accumulate = q.accumulate;
        p.vec_c = (q.N % 4 == 0) && (q.c_row_stride % 4 == 0) && (q.c_batch_stride % 4 == 0) && aligned16(q.C) && (!q.bias || aligned16(q.bias)) &&
                  (!q.out_scale || aligned16(q.out_scale));
        Mx = Mx > q.M ? Mx : q.M; Nx = Nx > q.N ? Nx : q.N;
    }
    for (int i = n; i < TG_MAX_GROUP; ++i) g.p[i] = g.p[0];
    {   // mover waves + LDS-DMA when big tiles fill the chip (persistent workgroups, one per CU; see gemm_np_mw_kernel)
        static const int env_on = [] { const char* e = getenv("TG_NP_MW"); return e ? atoi(e) : 1; }();
        bool vec_all = true;
        for (int i = 0; i < n; ++i) vec_all = vec_all && g.p[i].vec_c;
        int best_tn = 0;
        double best = 0.0;
        if (env_on && vec_all) {
            for (int tn : {6, 5}) {
                long tiles = 0;
                double work = 0.0;
                for (int i = 0; i < n; ++i) {
                    const long ti = (long)cdiv(g.p[i].M, 128) * cdiv(g.p[i].N, 32 * tn);
                    tiles += ti;
                    work += ti * ((g.p[i].taps * g.p[i].cwp / 32) * (6.0 * 4 * tn * 16.0) + 3000.0);
                }
                if (tiles < 150) continue;
                const double cost = (double)cdiv(tiles, 256) * work / tiles;
                if (!best_tn || cost < best) { best = cost; best_tn = tn; }
            }
        }
        if (best_tn) {
            const int bm = 128, bn = 32 * best_tn;
            int wg = 0;
            for (int i = 0; i < n; ++i) {
                g.p[i].n_nt = cdiv(g.p[i].N, bn);
                g.wg_begin[i] = wg;
                wg += (cdiv(g.p[i].M, bm) * g.p[i].n_nt + 7) / 8 * 8;
            }
            for (int i = n; i <= TG_MAX_GROUP; ++i) g.wg_begin[i] = wg;
            static const int n_cu = [] {
                int dev = 0, cus = 256;
                if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                return cus >= 8 ? cus / 8 * 8 : 8;
            }();
            const dim3 grid(wg < n_cu ? wg : n_cu);
            if (best_tn == 6) hipLaunchKernelGGL((gemm_np_mw_kernel<4, 6>), grid, dim3(512), 0, (hipStream_t)stream, g);
            else hipLaunchKernelGGL((gemm_np_mw_kernel<4, 5>), grid, dim3(512), 0, (hipStream_t)stream, g);
            return check_launch("tg_gemm_nt_planes(mover waves)");
        }
    }
    // tile menu: 128 x 96 where the grid has >= 2 workgroups per CU, else 64 x 96 / 64 x 64 (more, smaller workgroups for the backward shapes)
    auto wgs = [&](int bm, int bn) { return (long)n * cdiv(Mx, bm) * cdiv(Nx, bn); };
    auto waste = [&](int bn) { return cdiv(Nx, bn) * bn - Nx; };
    int tm, tn;
    if (wgs(128, 96) >= 512 && waste(96) <= waste(64) + 32) { tm = 4; tn = 3; }
    else if (wgs(128, 64) >= 384) { tm = 4; tn = 2; }
    else if (waste(96) <= waste(64) + 32 && wgs(64, 96) >= 256) { tm = 2; tn = 3; }
    else { tm = 2; tn = 2; }
    const int bm = 32 * tm, bn = 32 * tn;
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i].n_nt = cdiv(g.p[i].N, bn);
        g.wg_begin[i] = wg;
        wg += (cdiv(g.p[i].M, bm) * g.p[i].n_nt + 7) / 8 * 8;
    }
    for (int i = n; i <= TG_MAX_GROUP; ++i) g.wg_begin[i] = wg;
    hipStream_t s = (hipStream_t)stream;
    static const int occ2 = [] { const char* e = getenv("TG_NP_OCC"); return e ? atoi(e) == 2 : 0; }();      // lab switch: the 128 x 96 tile at two waves per SIMD
    if (tm == 4 && tn == 3 && occ2) hipLaunchKernelGGL((gemm_nt_planes_kernel<4, 3, 2>), dim3(wg), dim3(256), 0, s, g);
    else if (tm == 4 && tn == 3) hipLaunchKernelGGL((gemm_nt_planes_kernel<4, 3, 3>), dim3(wg), dim3(256), 0, s, g);
    else if (tm == 4 && tn == 2) hipLaunchKernelGGL((gemm_nt_planes_kernel<4, 2, 3>), dim3(wg), dim3(256), 0, s, g);
    else if (tm == 2 && tn == 3) hipLaunchKernelGGL((gemm_nt_planes_kernel<2, 3, 3>), dim3(wg), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_nt_planes_kernel<2, 2, 4>), dim3(wg), dim3(256), 0, s, g);
    return check_launch("tg_gemm_nt_planes");
}
