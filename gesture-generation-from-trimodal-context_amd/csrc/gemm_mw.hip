// tg_gemm_nt, big products with many rows: fp32-accurate bf16 x 3 (or plain bf16) product with MOVER WAVES.
//
// gemm_split.hip's kernels stage a slab, barrier, multiply, barrier: their own ablation (profiles/r2_nt_ablate.txt) shows the matrix
// section ADDING its time to the staging skeleton (182 = 111 + 71 us on [13056 x 1800 x 600]) -- every wave alternates between a
// load / split / LDS-store phase and a 72-MFMA phase, and the three resident workgroups of a CU fall into step.  Here the two jobs
// belong to different waves of one 512-thread workgroup (what the H = 64 recurrence's mover waves proved, gru_h64.hip):
//   * waves 4-7 (movers, one per SIMD) own ALL global traffic of the main loop: they fetch the fp32 operand slabs two slabs ahead into
//     two register sets, split every value exactly into its three bf16 terms and store the planes of slab s + 1 into the other half of a
//     double-buffered LDS image while
//   * waves 0-3 (one per SIMD) read their fragments of slab s from LDS and issue nothing but ds_read_b128 and MFMAs: 6 TM TN
//     v_mfma_f32_16x16x32_bf16 per slab and wave (144 for the 128 x 192 tile = 2304 cycles of matrix pipe).
// One workgroup barrier per slab.  A mover shares its SIMD's issue port with one matrix wave: an MFMA holds the port for 8 of its 16
// cycles (MI355X_MICROARCH.md, vector-instruction ISSUE cost), which leaves the mover ~2 vector instructions per MFMA -- the 128 x 192
// tile needs ~270 per slab (10 loads, 40 values x 5.5 split instructions, 30 LDS stores) against 288 such slots.  A bigger tile is what
// makes that fit: staging work grows with BM + BN, matrix work with BM x BN, and the L2 -> CU bytes per flop fall by 1.6 x against the
// 128 x 96 tile.  LDS: 2 x 3 planes x (BM + BN) rows x 64 B = 120 KB (128 x 192): one workgroup per CU.
//
// Addressing: every operand piece is a bounds-checked BUFFER load (raw_buffer_load_b128) whose voffset is either the piece's byte
// offset or a value past num_records -- padding rows of a conv window, rows past M / N and the K tail all read as zero with no select
// and no predicated load (static vector-memory counts), so one instantiation serves plain and padded windows.
// Precondition (host): both operands are addressable with 31-bit byte offsets from their base pointers, pieces never straddle a tap
// or a weight segment (cw % 4 == 0, b_seg_k % 4 == 0), C on the vectorisable layout.  Everything else stays on gemm_split.hip.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace tg {

constexpr unsigned MW_RSRC3 = 0x00020000u;
constexpr unsigned MW_OOB = 0x80000000u;            // voffset of a piece that must read as zero (>= num_records: extents are < 2^31)

template <int NS>
__device__ __forceinline__ void mw_split4(const f32x4 v, u32x2 (&out)[NS]) {
    if constexpr (NS == 1) {                         // plain bf16 tier: round to nearest even
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 r;
        r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
        out[0] = __builtin_bit_cast(u32x2, r);
    } else {
        static_assert(NS == 3, "1 or 3 terms");
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xf = v[i];                   // (bit_cast on an ext-vector ELEMENT is miscompiled by hipcc 7.2: scalar copy first)
            split3_bits(xf, h[i], m[i], l[i]);
        }
        out[0] = u32x2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
        out[1] = u32x2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
        out[2] = u32x2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
    }
}

#ifdef TG_LAB_ABLATE
// lab timeline: s_memrealtime (100 MHz) stamps per workgroup and role -- [blockIdx][role 0 matrix / 1 mover][8]: 0 entry, 1 set-up done,
// 2 first barrier passed, 3 main loop done, 4 accumulator tile laid out (barrier passed), 5 end; 6 = XCC id, 7 = CU id bits
__device__ unsigned long long mw_stamps[2048 * 16];
#define MW_STAMP(i) do { if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 2048) mw_stamps[blockIdx.x * 16 + (wave >> 2) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MW_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int mw_swz(int row) { return ((row >> 3) & 1) << 4; }      // as gemm_split.hip: XOR for a bf16 column index

// Workgroup tile (32 TM) x (32 TN); matrix waves 0-3 as 2 x 2, wave tile (16 TM) x (16 TN).
// ABL (lab builds only, -DTG_LAB_ABLATE, tools/mw_ablate.py; results are WRONG by construction): bit 0 drops the MFMAs, bit 1 the movers'
// split arithmetic + LDS stores, bit 2 the global operand loads, bit 3 the matrix waves' fragment reads, bit 4 the epilogue's global traffic
template <int TM, int TN, int NS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_mw_kernel(const NtGroup g) {
    const int pi = group_find(g, blockIdx.x);
    const NtProb& pr = g.p[pi];
    const Win A = pr.A;
    const int M = pr.M, N = pr.N, n_nt = pr.n_nt, K = A.K;
    constexpr int BM = 32 * TM, BN = 32 * TN, ROWS = BM + BN;
    constexpr int PLANE = ROWS * 32;                           // bf16 elements of one plane of one slab
    constexpr int BUF = NS * PLANE;
    constexpr int CLD = BN + 4;
    constexpr int OPER_BYTES = 2 * BUF * 2, CT_BYTES = BM * CLD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OPER_BYTES > CT_BYTES ? OPER_BYTES : CT_BYTES];
    __bf16* const lds = reinterpret_cast<__bf16*>(smem);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lid = xcd_chunked_id(blockIdx.x - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    if (m0 >= M) return;                                       // padding workgroup of a grouped launch (uniform: before any barrier)
    const int nslab = (K + 31) >> 5;
    MW_STAMP(0);
#ifdef TG_LAB_ABLATE
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 2048) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        mw_stamps[blockIdx.x * 16 + (wave >> 2) * 8 + 6] = xcc;
        mw_stamps[blockIdx.x * 16 + (wave >> 2) * 8 + 7] = hwid;
    }
#endif

    // ------------------------------------------------------------------------------------------------ mover state (waves 4-7)
    // thread mt owns the 16-byte piece (mt & 7) of slab rows (mt >> 3) + 32 q: 8 consecutive lanes cover one 128-byte row piece
    const int mt = t & 255;
    const int sp = 4 * (mt & 7), sr0 = mt >> 3;
    const int sp_w = sp ^ mw_swz(sr0);
    unsigned a_boff[TM], b_boff[TN];
    int a_r[TM];
    bool a_ok[TM], b_ok[TN];
#pragma unroll
    for (int q = 0; q < TM; ++q) {
        const int m = m0 + sr0 + 32 * q;
        a_ok[q] = m < M;
        const int mm = a_ok[q] ? m : 0;
        const int b = mm / A.rows_out;
        a_boff[q] = (unsigned)(((long)b * A.bs) * 4);
        a_r[q] = (mm - b * A.rows_out) * A.step + A.shift;
    }
#pragma unroll
    for (int q = 0; q < TN; ++q) {
        const int n = n0 + sr0 + 32 * q;
        b_ok[q] = n < N;
        b_boff[q] = (unsigned)(((long)(b_ok[q] ? n : 0) * pr.ldb) * 4);
    }
    const unsigned rs4 = (unsigned)(A.rs * 4);
    // tap / channel of this thread's piece in the A window and weight segment / column in B, advanced by one slab per fetch
    int kk = sp / A.cw, c = sp - (sp / A.cw) * A.cw;
    int bc = sp - (sp / pr.b_seg_k) * pr.b_seg_k;
    unsigned kb = (unsigned)(((long)(sp / pr.b_seg_k) * pr.b_seg_stride + bc) * 4);
    const unsigned kb_wrap = (unsigned)((pr.b_seg_stride - pr.b_seg_k) * 4);
    int kcur = sp;                                             // k of this thread's piece in the slab the next fetch loads
    __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.ptr), 0, pr.a_bytes, MW_RSRC3);
    __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.Bw), 0, pr.b_bytes, MW_RSRC3);

    u32x4 ga[2][TM], gb[2][TN];
    auto fetch = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
        const bool inb = kcur < K;
        const unsigned c4 = (unsigned)(c * 4);
#pragma unroll
        for (int q = 0; q < TM; ++q) {
            const int sr = a_r[q] + kk * A.dil;
            const bool ok = a_ok[q] & inb & ((unsigned)sr < (unsigned)A.rows_in);
            if constexpr (ABL & 4) ga[set][q] = u32x4{0x3f800000u + (unsigned)lane, 0x40000000u, 0x3fc00000u + (unsigned)kcur, ok ? 0x3e800000u : 0u};
            else ga[set][q] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, ok ? a_boff[q] + (unsigned)sr * rs4 + c4 : MW_OOB, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            if constexpr (ABL & 4) gb[set][q] = u32x4{0x3f000000u + (unsigned)lane, 0x3e000000u, 0x3f400000u + kb, (b_ok[q] & inb) ? 0x3e800000u : 0u};
            else gb[set][q] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (b_ok[q] & inb) ? b_boff[q] + kb : MW_OOB, 0, 0);
        }
        kcur += 32;
        c += 32;
        while (c >= A.cw) { c -= A.cw; ++kk; }
        bc += 32; kb += 128u;
        while (bc >= pr.b_seg_k) { bc -= pr.b_seg_k; kb += kb_wrap; }
    };
    auto stage = [&](auto set_c, int buf) {
        constexpr int set = decltype(set_c)::value;
        __bf16* const lb = lds + buf * BUF;
        if constexpr (ABL & 2) {                              // loaded values stay live (the loads must still be waited for), nothing else
#pragma unroll
            for (int q = 0; q < TM; ++q) asm volatile("" :: "v"(ga[set][q]));
#pragma unroll
            for (int q = 0; q < TN; ++q) asm volatile("" :: "v"(gb[set][q]));
            return;
        }
#pragma unroll
        for (int q = 0; q < TM; ++q) {
            u32x2 o[NS];
            mw_split4<NS>(__builtin_bit_cast(f32x4, ga[set][q]), o);
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(lb + s * PLANE + (sr0 + 32 * q) * 32 + sp_w) = o[s];
        }
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            u32x2 o[NS];
            mw_split4<NS>(__builtin_bit_cast(f32x4, gb[set][q]), o);
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(lb + s * PLANE + (BM + sr0 + 32 * q) * 32 + sp_w) = o[s];
        }
    };

    // ------------------------------------------------------------------------------------------------ matrix state (waves 0-3)
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int fcol = (8 * kq) ^ mw_swz(r16);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](int buf) {
        const __bf16* const lb = lds + buf * BUF;
        bf16x8 fa[NS][TM];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (ABL & 8) fa[s][i] = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u + (unsigned)(lane + i), 0x3f003f00u + (unsigned)s, 0x3f803f80u, 0x3e803e80u + (unsigned)buf});
                else fa[s][i] = *reinterpret_cast<const bf16x8*>(lb + s * PLANE + (wm * (16 * TM) + i * 16 + r16) * 32 + fcol);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf16x8 fb[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if constexpr (ABL & 8) fb[s] = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u + (unsigned)(lane + j), 0x3e803e80u + (unsigned)s, 0x3f003f00u, 0x3f803f80u + (unsigned)buf});
                else fb[s] = *reinterpret_cast<const bf16x8*>(lb + s * PLANE + (BM + wn * (16 * TN) + j * 16 + r16) * 32 + fcol);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4 cc = acc[i][j];
                if constexpr (ABL & 1) {               // fragments stay live, no matrix instruction
                    asm volatile("" :: "v"(fb[0]), "v"(fb[NS - 1]), "v"(fb[NS / 2]), "v"(fa[0][i]), "v"(fa[NS - 1][i]), "v"(fa[NS / 2][i]));
                    continue;
                }
                if constexpr (NS == 3) {               // smallest terms first
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][i], fb[0], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[2], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[1], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1], cc, 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0], cc, 0, 0, 0);
            }
        }
    };

    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    const bool mover = wave >= 4;
    float* const ct = reinterpret_cast<float*>(smem);
    // The two roles run SEPARATE loops that meet at the same sequence of s_barrier instructions (one per slab + the prologue's + the
    // epilogue's): a shared loop body would keep both roles' registers live in every wave (accumulators + fragments beside two operand
    // register sets: 250 spilled VGPRs).  Raw barriers: a mover drains its LDS stores (lgkmcnt) before arriving, its global loads
    // stay in flight across the barrier; a matrix wave's fragment reads have been consumed by its MFMAs when it arrives.
#define MW_BARRIER_MOVER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define MW_BARRIER_MATRIX() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
    MW_STAMP(1);
    if (mover) {
        fetch(set0{});                                         // slab 0
        fetch(set1{});                                         // slab 1 (past K: every piece out of range, zeros, no traffic)
        stage(set0{}, 0);
        fetch(set0{});                                         // slab 2
        MW_BARRIER_MOVER();
        MW_STAMP(2);
        for (int s = 0; s < nslab; s += 2) {
            if (s + 1 < nslab) stage(set1{}, 1);               // slab s + 1 -> buffer 1 (read last during step s - 1)
            fetch(set1{});                                     // slab s + 3
            MW_BARRIER_MOVER();
            if (s + 1 >= nslab) break;
            if (s + 2 < nslab) stage(set0{}, 0);               // slab s + 2 -> buffer 0
            fetch(set0{});                                     // slab s + 4
            MW_BARRIER_MOVER();
        }
        MW_STAMP(3);
        MW_BARRIER_MOVER();                                    // the matrix waves have laid out the accumulator tile
        MW_STAMP(4);
    } else {
        MW_BARRIER_MATRIX();
        MW_STAMP(2);
        for (int s = 0; s < nslab; s += 2) {
            multiply(0);
            MW_BARRIER_MATRIX();
            if (s + 1 >= nslab) break;
            multiply(1);
            MW_BARRIER_MATRIX();
        }
        MW_STAMP(3);
        // the operand image is idle now (every wave has passed the loop's last barrier): accumulator tile row-major into LDS
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    ct[(wm * (16 * TM) + i * 16 + kq * 4 + q) * CLD + wn * (16 * TN) + j * 16 + r16] = acc[i][j][q];
        MW_BARRIER_MOVER();                                    // (drains this wave's LDS stores first)
        MW_STAMP(4);
    }
#undef MW_BARRIER_MOVER
#undef MW_BARRIER_MATRIX

    // ---- epilogue: all 512 threads handle 16-byte row pieces of the tile: bias, activation, dropout scale, gate, accumulate, second output
    const float* __restrict__ bias = pr.bias;
    const float* __restrict__ mul = pr.mul;
    const float* __restrict__ gate = pr.gate;
    const float* __restrict__ res = pr.res;
    float* __restrict__ C2 = pr.C2;
    float* __restrict__ C = pr.C;
    const float slope = pr.slope, slope2 = pr.res_slope;
    const long cbs = pr.cbs, crs = pr.crs;
    const int cR = pr.cR, accumulate = pr.accumulate;
    constexpr int C4 = BN / 4, NP = BM * C4 / 512, CH = 4;
    static_assert(BM * C4 % 512 == 0, "whole pieces per thread");
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += CH) {
        f32x4 bv[CH], mv[CH], cv[CH], gv[CH], rv[CH];
        long o[CH];
        bool ok[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (p0 + u >= NP) continue;                      // compile-time
            const int idx = t + 512 * (p0 + u);
            const int rl = idx / C4, c4 = idx - rl * C4;
            const int row = m0 + rl, col = n0 + 4 * c4;
            ok[u] = row < M && col < N;                      // N % 4 == 0: a piece is inside or outside as a whole
            const int cb = row / cR;
            const int cr = row - cb * cR;
            o[u] = ok[u] ? (long)cb * cbs + (long)cr * crs + col : 0;
            bv[u] = bias ? *reinterpret_cast<const f32x4*>(bias + (ok[u] ? col : 0)) : z4;
            if (mul) mv[u] = *reinterpret_cast<const f32x4*>(mul + o[u]);
            if (gate) gv[u] = *reinterpret_cast<const f32x4*>(gate + o[u]);
            if (res) rv[u] = *reinterpret_cast<const f32x4*>(res + o[u]);
            if (accumulate) cv[u] = *reinterpret_cast<const f32x4*>(C + o[u]);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (p0 + u >= NP) continue;
            const int idx = t + 512 * (p0 + u);
            const int rl = idx / C4, c4 = idx - rl * C4;
            f32x4 v = *reinterpret_cast<const f32x4*>(&ct[rl * CLD + 4 * c4]) + bv[u];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = act_fn(v[q], slope);
            if (mul) v *= mv[u];
            if (gate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gv[u][q] > 0.f ? v[q] : 0.f;
            }
            if (accumulate) v += cv[u];
            if constexpr (ABL & 16) { if (v[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(C + o[u]) = v; continue; }
            if (ok[u]) *reinterpret_cast<f32x4*>(C + o[u]) = v;
            if (res) {
                f32x4 w = v + rv[u];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = act_fn(w[q], slope2);
                if (ok[u]) *reinterpret_cast<f32x4*>(C2 + o[u]) = w;
            }
        }
    }
    MW_STAMP(5);
}

}  // namespace tg

using namespace tg;

#ifdef TG_LAB_ABLATE
extern "C" int tg_lab_mw_stamps(void* host_out, int64_t n_words) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mw_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

// byte extents of the two operands as seen from their base pointers (what the buffer descriptors bound); false when an operand is not
// addressable that way (negative strides, >= 2 GB)
static bool mw_extents(NtProb& p) {
    const Win& A = p.A;
    if (A.bs < 0 || A.rs < 0 || p.ldb < 0 || p.b_seg_stride < 0 || A.rows_out <= 0) return false;
    const long batches = cdiv(p.M, A.rows_out);
    const long a_el = (batches - 1) * A.bs + (long)(A.rows_in - 1) * A.rs + A.cw;
    const long segs = A.K / p.b_seg_k;
    const long b_el = (long)(p.N - 1) * p.ldb + (segs - 1) * p.b_seg_stride + p.b_seg_k;
    if (a_el <= 0 || b_el <= 0 || a_el >= (1l << 29) || b_el >= (1l << 29)) return false;
    p.a_bytes = (unsigned)(a_el * 4);
    p.b_bytes = (unsigned)(b_el * 4);
    return true;
}

struct MwTile { int tm, tn; };

// Tile choice.  The mover-wave kernel runs ONE workgroup per CU; it pays when its tiles fill most of the 256 CUs at least once:
//   * 128 x 192: N = 900 -> 5 column tiles (6.7 % padding); the stacked forward's two GRU projections are 1020 tiles = 3.98 rounds
//   * 128 x 160: N = 300 / 600 -> 2 / 4 column tiles (6.7 % padding)
// cost model: rounds x (slab count x matrix cycles per slab + fixed prologue / epilogue), smallest wins; not eligible below ~0.6 rounds.
static int g_mw_on = -1;                 // -1: environment TG_NT_MW (default on); 0 / 1: set by tg_set_nt_mover_waves
extern "C" int tg_set_nt_mover_waves(int32_t on) {
    TG_REQUIRE(on >= -1 && on <= 1, "tg_set_nt_mover_waves: -1 (environment default), 0 or 1");
    g_mw_on = on;
    return 0;
}

static bool mw_pick_tile(const NtGroup& g, MwTile* out) {
    static const int env_on = [] { const char* e = getenv("TG_NT_MW"); return e ? atoi(e) : 1; }();
    if (!(g_mw_on < 0 ? env_on : g_mw_on)) return false;
    const MwTile menu[2] = {{4, 6}, {4, 5}};
    double best = 0.0;
    bool found = false;
    for (const MwTile& tl : menu) {
        const int bm = 32 * tl.tm, bn = 32 * tl.tn;
        long tiles = 0;
        double work = 0.0;
        for (int i = 0; i < g.n; ++i) {
            const long ti = (long)cdiv(g.p[i].M, bm) * cdiv(g.p[i].N, bn);
            tiles += ti;
            const double per_tile = cdiv(g.p[i].A.K, 32) * (6.0 * tl.tm * tl.tn * 16.0) + 6000.0;        // cycles
            work += ti * per_tile;
        }
        if (tiles < 150) continue;
        const double rounds = (double)cdiv(tiles, 256);
        const double cost = rounds * work / tiles;
        if (!found || cost < best) { best = cost; *out = tl; found = true; }
    }
    return found;
}

// true when the group can (and should) run on the mover-wave kernel; fills the extents
bool tg_gemm_nt_mw_eligible(NtGroup& g, int* tm, int* tn) {
    for (int i = 0; i < g.n; ++i) {
        NtProb& p = g.p[i];
        if (!p.vec_c || p.Bpl != nullptr || p.A.cw % 4 != 0 || p.b_seg_k % 4 != 0 || p.A.K % 4 != 0) return false;
        if (!mw_extents(p)) return false;
    }
    MwTile tl;
    if (!mw_pick_tile(g, &tl)) return false;
    *tm = tl.tm; *tn = tl.tn;
    return true;
}

int tg_gemm_nt_mw_launch(NtGroup& g, int tm, int tn, int splits, hipStream_t s) {
    const int bm = 32 * tm, bn = 32 * tn;
    int wg = 0;
    for (int i = 0; i < g.n; ++i) {             // every problem's range starts at a multiple of 8 (XCD mapping, see gemm.hip nt_layout)
        g.p[i].n_nt = cdiv(g.p[i].N, bn);
        g.wg_begin[i] = wg;
        wg += (cdiv(g.p[i].M, bm) * g.p[i].n_nt + 7) / 8 * 8;
    }
    for (int i = g.n; i <= TG_MAX_GROUP; ++i) g.wg_begin[i] = wg;
    const dim3 grid(wg);
#ifdef TG_LAB_ABLATE
    {
        const char* e = getenv("TG_MW_ABL");
        const int abl = e ? atoi(e) : 0;
        if (abl && tm == 4 && tn == 6 && splits == 3) {
#define TG_ABL(A_) case A_: hipLaunchKernelGGL((gemm_nt_mw_kernel<4, 6, 3, A_>), grid, dim3(512), 0, s, g); return check_launch("tg_gemm_nt(mover waves, ablated)")
            switch (abl) { TG_ABL(1); TG_ABL(2); TG_ABL(3); TG_ABL(4); TG_ABL(6); TG_ABL(7); TG_ABL(8); TG_ABL(9); TG_ABL(15); TG_ABL(16); TG_ABL(31); default: break; }
#undef TG_ABL
        }
    }
#endif
#define TG_MW(TM_, TN_)                                                                                        \
    do {                                                                                                       \
        if (splits == 3) hipLaunchKernelGGL((gemm_nt_mw_kernel<TM_, TN_, 3>), grid, dim3(512), 0, s, g);       \
        else hipLaunchKernelGGL((gemm_nt_mw_kernel<TM_, TN_, 1>), grid, dim3(512), 0, s, g);                   \
    } while (0)
    if (tm == 4 && tn == 6) TG_MW(4, 6);
    else if (tm == 4 && tn == 5) TG_MW(4, 5);
    else TG_REQUIRE(false, "tg_gemm_nt(mover waves): no %d x %d tile", bm, bn);
#undef TG_MW
    return check_launch("tg_gemm_nt(mover waves)");
}
