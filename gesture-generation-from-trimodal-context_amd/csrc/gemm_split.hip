// tg_gemm_nt on the bf16 matrix cores at fp32 accuracy: operand splitting ("bf16 x 3").
//
// gfx950 has no TF32-style fast path for fp32 GEMMs: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA
// rate.  Here every fp32 operand x is split EXACTLY into three bf16 terms while its 32-deep K slab is staged into LDS,
//     x = hi + mid + lo        hi = top 8 significand bits of x (truncation), mid = top 8 bits of x - hi, lo = the rest (<= 8 bits),
// and the product keeps the six partial products whose weight is >= 2^-16 of the leading one,
//     a.b ~= hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi            (dropped: mid.lo, lo.mid, lo.lo <= 2^-23 |a.b|),
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Six bf16 MFMAs replace sixteen f32-MFMA
// issue slots of the same tile (2.67x the matrix rate); measured error vs fp64 is the f32 MFMA path's (max 6e-7 of max|C| at
// K = 600, tools/gemm_split_lab.hip) -- the only roundings are fp32 accumulations, as there.  Measured 1.15-1.85x faster than the
// f32-MFMA kernel on the shapes of the training step (the VALU split work and the LDS traffic of three planes eat the rest).
//
// SPLITS = 1 is the plain bf16-operand tier (math mode 1): operands rounded to nearest even, one MFMA per product, fp32 accumulate.
//
// Workgroup tile (32*TM) x (32*TN), 4 waves as 2 x 2, wave tile (16*TM) x (16*TN) of 16x16x32 MFMAs.  LDS rows hold 32 bf16
// (64 B) padded to 80 B so a lane's fragment -- A[row l&15][k = 8*(l>>4) .. +7] -- is one conflict-light 16-byte ds_read.  The
// next slab's global loads are in flight (registers) while the current slab is split, stored and multiplied.  DB = 0 keeps ONE
// LDS buffer (two barriers per slab, <= 54 KB: two workgroups per CU, the better choice when the grid has >= 2 workgroups per
// CU); DB = 1 double-buffers (one barrier per slab): used with the 64-row tiles of the small grids of the backward shapes.
#include "common.hpp"
#include "tr_image.hpp"
#include <stdlib.h>
#include <type_traits>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace tg {

// LDS rows hold one 32-deep slab row = 64 bytes = four 16-byte slots (slot kq = k 8 kq .. 8 kq + 7), unpadded; the slot index is XORed with
// 2 * bit 3 of the row.  ds_read_b128 is served in four fixed 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... --
// MI355X_MICROARCH.md, LDS): with lane = (kq, row r16) this XOR puts every group on 16 distinct slots of the 256-byte bank row.  The
// former 80-byte padded rows were 2-way conflicted on every fragment read (SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE,
// profiles/r2_pmc_gemm_split.txt); the 8-byte staging stores stay conflict-free (16 consecutive lanes = two whole rows).
constexpr int SP_LD = 32;
__device__ __forceinline__ int sp_swz(int row) { return ((row >> 3) & 1) << 4; }      // XOR for a bf16 column index

template <int SPLITS>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&out)[SPLITS]) {
    if constexpr (SPLITS == 1) {          // plain bf16 tier: round to nearest even (v_cvt_pk_bf16_f32)
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 r;
        r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
        out[0] = __builtin_bit_cast(u32x2, r);
    } else {
        static_assert(SPLITS == 3, "1 or 3 terms");
        unsigned t[4], u[4], s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (__builtin_bit_cast(unsigned, v[i]) on an ext_vector ELEMENT is miscompiled by hipcc 7.2 -- every i reads element 0:
            // copy the element to a scalar first)
            const float xf = v[i];
            split3_bits(xf, t[i], u[i], s[i]);
        }
        out[0] = u32x2{pack_hi16(t[0], t[1]), pack_hi16(t[2], t[3])};
        out[1] = u32x2{pack_hi16(u[0], u[1]), pack_hi16(u[2], u[3])};
        out[2] = u32x2{pack_hi16(s[0], s[1]), pack_hi16(s[2], s[3])};
    }
}

// C(m, n) = act(sum_k A(m,k) * Bw[n][k] + bias[n]) (+ C).  A: fp32 row window (vectorisable layout: checked by the caller),
// Bw: fp32 [N][ldb].  Same contract as gemm_nt_big_kernel (gemm.hip).
// RING = register sets of global loads in flight (slabs fetched ahead): 1 = the next slab only (the only depth still instantiated: two slabs
// ahead measured no gain on any shape of the iteration and was dropped from the menu in round 4), 2 = two slabs ahead (+ 4 (TM + TN)
// VGPRs; the loads of a slab then have two MFMA sections to land instead of one)
// OCC = waves per SIMD the register allocation is held to (HIP's second launch-bounds argument; 1 = unconstrained)
// ABL (lab builds only, -DTG_LAB_ABLATE, tools/nt_ablate.py; results are WRONG by construction): bit 0 drops the MFMAs, bit 1 the split
// arithmetic, bit 2 the LDS fragment reads, bit 3 the global operand loads -- what each phase costs in situ and how much of it overlaps
// FAST: every problem of the group has a window WITHOUT padding (every (row, tap) inside the tensor: nt_fast_ok below), operands addressable
// with 32-bit byte offsets, cw >= 32 and b_seg_k >= 32.  The slab loop then carries no row masks and no 64-bit address arithmetic: rows past
// M / N are CLAMPED to the last one (their products are computed and dropped by the epilogue), a piece's address is one 32-bit add to an
// SGPR base, the tap / segment walk is one compare-and-select per thread and slab, and only the K tail slab selects zeros.  The generic
// loop spends ~110 of its 284 vector instructions per slab on exactly that (ISA count, 128 x 96 tile) and is vector-issue bound next to its
// 72 MFMAs (tools/nt_ablate.py: every 160 vector instructions removed = 30 us of the 182 us launch).
template <int TM, int TN, int SPLITS, int DB, int RING = 1, int OCC = 1, int ABL = 0, bool FAST = false>
__global__ __launch_bounds__(256, OCC) void gemm_nt_split_kernel(const NtGroup g) {
    const int pi = group_find(g, blockIdx.x);
    const NtProb& pr = g.p[pi];
    const Win A = pr.A;
    const float* __restrict__ Bw = pr.Bw;
    const long ldb = pr.ldb, b_seg_stride = pr.b_seg_stride;
    const int b_seg_k = pr.b_seg_k;
    const float* __restrict__ bias = pr.bias;
    const float* __restrict__ mul = pr.mul;
    const uint64_t* __restrict__ drop = pr.drop_state;       // (vec_c only: checked by the launcher)
    const float* __restrict__ gate = pr.gate;
    const float* __restrict__ res = pr.res;
    float* __restrict__ C2 = pr.C2;
    const float slope2 = pr.res_slope;
    float* __restrict__ C = pr.C;
    const long cbs = pr.cbs, crs = pr.crs;
    const int cR = pr.cR, M = pr.M, N = pr.N, accumulate = pr.accumulate, n_nt = pr.n_nt;
    const float slope = pr.slope;
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int NPA = BM / 32, NPB = BN / 32;               // f32x4 pieces per thread per slab (8 pieces per 32-deep row)
    constexpr int NS = SPLITS;
    constexpr int NB = DB ? 2 : 1;
    constexpr int CLD = BN + 4;                                // row stride (floats) of the epilogue's accumulator tile
    constexpr int OPER_BYTES = NB * NS * (BM + BN) * SP_LD * 2, CT_BYTES = BM * CLD * 4;
    // ONE array: operand slabs [NB][NS][BM + BN][SP_LD] bf16 (A rows first, then B rows); reused by the epilogue as [BM][CLD] fp32
    __shared__ __attribute__((aligned(16))) unsigned char smem[OPER_BYTES > CT_BYTES ? OPER_BYTES : CT_BYTES];
    typedef __bf16 (*lds_t)[NS][BM + BN][SP_LD];
    const lds_t lds = reinterpret_cast<lds_t>(smem);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    if (m0 >= M) return;                                       // padding workgroup of a grouped launch (uniform: before any barrier)
    const int K = A.K;

    // staging map: 8 consecutive lanes cover one 128-byte row piece of the slab (32 fp32): whole cache lines per load instruction;
    // thread t owns piece (t & 7) of rows (t >> 3) + 32 q
    const int sp = 4 * (t & 7), sr0 = t >> 3;
    const int sp_w = sp ^ sp_swz(sr0), fcol = (8 * kq) ^ sp_swz(r16);     // swizzled store / fragment columns (row bases are multiples of 16)
    long a_off[NPA];
    int a_r[NPA];
    bool a_ok[NPA];
    unsigned a_base[NPA], b_base[NPB];                        // FAST: byte offsets of (row, tap 0, column 0) from A.ptr / Bw
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
        const int m = m0 + sr0 + 32 * q;
        a_ok[q] = m < M;
        const int mm = a_ok[q] ? m : (FAST ? M - 1 : 0);
        const int b = mm / A.rows_out;
        a_off[q] = (long)b * A.bs;
        a_r[q] = (mm - b * A.rows_out) * A.step + A.shift;
        a_base[q] = (unsigned)((a_off[q] + (long)a_r[q] * A.rs) * 4);
    }
    const float* b_ptr[NPB];
    bool b_ok[NPB];
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        const int n = n0 + sr0 + 32 * q;
        b_ok[q] = n < N;
        b_ptr[q] = Bw + (long)(b_ok[q] ? n : (FAST ? N - 1 : 0)) * ldb;
        b_base[q] = (unsigned)((long)(b_ok[q] ? n : N - 1) * ldb * 4);
    }
    int kk = sp / A.cw, c = sp - (sp / A.cw) * A.cw;          // tap / channel of this thread's piece, advanced by 32 per slab
    int bsg = sp / b_seg_k, bc = sp - (sp / b_seg_k) * b_seg_k;   // weight segment / column inside it, likewise
    // FAST: byte offset of this thread's piece inside its row (tap walk folded in), and what a tap / segment wrap adds to it
    unsigned ka = (unsigned)(((long)kk * A.dil * A.rs + c) * 4), kb = (unsigned)(((long)bsg * b_seg_stride + bc) * 4);
    const unsigned ka_wrap = (unsigned)(((long)A.dil * A.rs - A.cw) * 4), kb_wrap = (unsigned)((b_seg_stride - b_seg_k) * 4);
    const char* const a_bytes = reinterpret_cast<const char*>(A.ptr);
    const char* const b_bytes = reinterpret_cast<const char*>(Bw);

    f32x4 ga[RING][NPA], gb[RING][NPB];
    unsigned ga_ok[RING], gb_ok[RING];
    auto fetch = [&](auto set_c, int k0) {
        constexpr int set = decltype(set_c)::value;
        // loads are issued UNCONDITIONALLY from an always-valid address: a predicated load makes the number of outstanding loads
        // dynamic and hipcc then drains everything (vmcnt(0)) at the next use.  Out-of-range pieces are zeroed when the slab is STAGED
        // (masks ga_ok / gb_ok), not here: a select placed next to its load is scheduled right behind it and waits for the loads
        // just issued -- before the MFMA section they were meant to overlap.
        const int k = k0 + sp;
        const bool inb = k < K;
        unsigned ma = 0u, mb = 0u;
        if constexpr (FAST) {
            // pieces past K (tail slab only) re-read the row start; the staging of the tail slab selects zeros for them
            const unsigned oa = inb ? ka : 0u, ob = inb ? kb : 0u;
#pragma unroll
            for (int q = 0; q < NPA; ++q) ga[set][q] = *reinterpret_cast<const f32x4*>(a_bytes + (a_base[q] + oa));
#pragma unroll
            for (int q = 0; q < NPB; ++q) gb[set][q] = *reinterpret_cast<const f32x4*>(b_bytes + (b_base[q] + ob));
            ga_ok[set] = gb_ok[set] = inb ? ~0u : 0u;
            c += 32; ka += 128u;
            if (c >= A.cw) { c -= A.cw; ka += ka_wrap; }
            bc += 32; kb += 128u;
            if (bc >= b_seg_k) { bc -= b_seg_k; kb += kb_wrap; }
            return;
        }
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            const int sr = a_r[q] + kk * A.dil;
            const bool ok = a_ok[q] && inb && sr >= 0 && sr < A.rows_in;
            if constexpr (ABL & 8) ga[set][q] = f32x4{1.f + lane, 2.f, 3.f + k0, 4.f};
            else ga[set][q] = *reinterpret_cast<const f32x4*>(ok ? A.ptr + a_off[q] + (long)sr * A.rs + c : A.ptr);
            ma |= ok ? (1u << q) : 0u;
        }
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            const bool ok = b_ok[q] && inb;
            if constexpr (ABL & 8) gb[set][q] = f32x4{.5f + lane, .25f, .125f + k0, 1.f};
            else gb[set][q] = *reinterpret_cast<const f32x4*>(ok ? b_ptr[q] + bsg * b_seg_stride + bc : Bw);
            mb |= ok ? (1u << q) : 0u;
        }
        ga_ok[set] = ma; gb_ok[set] = mb;
        c += 32;
        while (c >= A.cw) { c -= A.cw; ++kk; }
        bc += 32;
        while (bc >= b_seg_k) { bc -= b_seg_k; ++bsg; }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, RING - 1>;
    int buf = 0;
    // one slab: stage register set `set` into LDS, refill the set with the slab RING ahead, multiply
    // tail_c: this slab may hold pieces past K or past the operand's rows -> select zeros while staging (always, on the generic path)
    auto slab = [&](auto set_c, int k0, auto tail_c) {
        constexpr int set = decltype(set_c)::value;
        constexpr bool MASKED = !FAST || decltype(tail_c)::value;
        if (!DB && k0 > 0) __syncthreads();                     // single buffer: everybody has read the previous slab
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            u32x2 o[NS];
            if constexpr (ABL & 2) {
#pragma unroll
                for (int s = 0; s < NS; ++s) o[s] = u32x2{__float_as_uint(ga[set][q][0]), __float_as_uint(ga[set][q][2])};
            } else if constexpr (!MASKED) split4<SPLITS>(ga[set][q], o);
            else
            split4<SPLITS>((ga_ok[set] >> q) & 1u ? ga[set][q] : f32x4{0.f, 0.f, 0.f, 0.f}, o);
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(&lds[buf][s][sr0 + 32 * q][sp_w]) = o[s];
        }
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            u32x2 o[NS];
            if constexpr (ABL & 2) {
#pragma unroll
                for (int s = 0; s < NS; ++s) o[s] = u32x2{__float_as_uint(gb[set][q][1]), __float_as_uint(gb[set][q][3])};
            } else if constexpr (!MASKED) split4<SPLITS>(gb[set][q], o);
            else
            split4<SPLITS>((gb_ok[set] >> q) & 1u ? gb[set][q] : f32x4{0.f, 0.f, 0.f, 0.f}, o);
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(&lds[buf][s][BM + sr0 + 32 * q][sp_w]) = o[s];
        }
        __syncthreads();
        if (k0 + 32 * RING < K) fetch(set_c, k0 + 32 * RING);
        bf16x8 fa[NS][TM], fb[NS][TN];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (ABL & 4) { u32x2 c2 = {0x3f803f80u + lane + i, 0x3f003f00u + s}; fa[s][i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(c2, c2, 0, 1, 0, 1)); }
                else fa[s][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][s][wm * (16 * TM) + i * 16 + r16][fcol]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (ABL & 4) { u32x2 c2 = {0x3f803f80u + lane + j, 0x3e803e80u + s}; fb[s][j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(c2, c2, 0, 1, 0, 1)); }
                else fb[s][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][s][BM + wn * (16 * TN) + j * 16 + r16][fcol]);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 cc = acc[i][j];
                if constexpr (ABL & 1) {               // fragments stay live, no matrix instruction
                    if (i == 0) { asm volatile("" :: "v"(fb[0][j]), "v"(fb[NS - 1][j]), "v"(fb[NS / 2][j])); }
                    if (j == 0) { asm volatile("" :: "v"(fa[0][i]), "v"(fa[NS - 1][i]), "v"(fa[NS / 2][i])); }
                    continue;
                }
                if constexpr (SPLITS == 3) {           // smallest terms first
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][i], fb[0][j], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[2][j], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[1][j], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0][j], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1][j], cc, 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0][j], cc, 0, 0, 0);
            }
        if (DB) buf ^= 1;
    };
    using tail_t = std::true_type;
    using full_t = std::false_type;
    fetch(set0{}, 0);
    if constexpr (RING == 2) {
        if (32 < K) fetch(set1{}, 32);
        int k0 = 0;
        if constexpr (FAST) {
            for (; k0 + 64 <= K; k0 += 64) {
                slab(set0{}, k0, full_t{});
                slab(set1{}, k0 + 32, full_t{});
            }
            if (k0 + 32 <= K) {
                slab(set0{}, k0, full_t{});
                if (k0 + 32 < K) slab(set1{}, k0 + 32, tail_t{});
            } else if (k0 < K) slab(set0{}, k0, tail_t{});
        } else {
            for (; k0 < K; k0 += 64) {
                slab(set0{}, k0, tail_t{});
                if (k0 + 32 < K) slab(set1{}, k0 + 32, tail_t{});
            }
        }
    } else {
        int k0 = 0;
        if constexpr (FAST) {
            for (; k0 + 32 <= K; k0 += 32) slab(set0{}, k0, full_t{});
            if (k0 < K) slab(set0{}, k0, tail_t{});
        } else {
            for (; k0 < K; k0 += 32) slab(set0{}, k0, tail_t{});
        }
    }

    const int vec_c = pr.vec_c;
    if (vec_c) {
        // ---- epilogue through LDS: the accumulator tile goes row-major into the (now idle) slab memory, then every thread handles
        // 16-byte row pieces -- C, the bias, the dropout mask and the accumulate operand are read / written as coalesced float4
        // instead of one scalar per (lane, tile) with 64-byte segments
        float* ct = reinterpret_cast<float*>(smem);
        __syncthreads();                                        // every wave is done with the last operand slab
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    ct[(wm * (16 * TM) + i * 16 + kq * 4 + q) * CLD + wn * (16 * TN) + j * 16 + r16] = acc[i][j][q];
        __syncthreads();
        // pieces are handled in chunks of four with every global read of a chunk issued before the first use, from always-valid
        // addresses: the straightforward loop (predicated `continue`, bias -> mask -> C one after the other) made each piece wait
        // out two or three L2 round trips in turn, 12 pieces per thread
        constexpr int C4 = BN / 4, NP = BM * C4 / 256, CH = 4;
        static_assert(BM * C4 % 256 == 0, "whole pieces per thread");
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += CH) {
            f32x4 bv[CH], mv[CH], cv[CH], gv[CH], rv[CH];
            long o[CH];
            bool ok[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (p0 + u >= NP) continue;                      // compile-time
                const int idx = t + 256 * (p0 + u);
                const int rl = idx / C4, c4 = idx - rl * C4;
                const int row = m0 + rl, col = n0 + 4 * c4;
                ok[u] = row < M && col < N;                      // N % 4 == 0: a piece is inside or outside as a whole
                const int cb = row / cR;
                const int cr = row - cb * cR;
                o[u] = ok[u] ? (long)cb * cbs + (long)cr * crs + col : 0;
                bv[u] = bias ? *reinterpret_cast<const f32x4*>(bias + (ok[u] ? col : 0)) : z4;
                if (mul) mv[u] = *reinterpret_cast<const f32x4*>(mul + o[u]);
                else if (drop) mv[u] = dropout_scale4(drop, pr.drop_site, pr.drop_p, (unsigned long)(pr.drop_index0 + o[u]) >> 2);
                if (gate) gv[u] = *reinterpret_cast<const f32x4*>(gate + o[u]);
                if (res) rv[u] = *reinterpret_cast<const f32x4*>(res + o[u]);
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
                if (mul || drop) v *= mv[u];
                if (gate) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = gv[u][q] > 0.f ? v[q] : 0.f;
                }
                if (accumulate) v += cv[u];
                if (ok[u]) *reinterpret_cast<f32x4*>(C + o[u]) = v;
                if (res) {
                    f32x4 w = v + rv[u];
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] = act_fn(w[q], slope2);
                    if (ok[u]) *reinterpret_cast<f32x4*>(C2 + o[u]) = w;
                }
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
            const long ro = (long)cb * cbs + (long)cr * crs;
            float* crow = C + ro;
            const float* mrow = mul ? mul + ro : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + r16;
                if (col >= N) continue;
                float v = acc[i][j][q];
                if (bias) v += bias[col];
                v = act_fn(v, slope);
                if (mrow) v *= mrow[col];
                if (gate) v = gate[ro + col] > 0.f ? v : 0.f;
                if (accumulate) v += crow[col];
                crow[col] = v;
                if (res) C2[ro + col] = act_fn(v + res[ro + col], slope2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradients on the same bf16 x 3 scheme:  dW[n][k] += sum_m dY[m][n] * A(m, k)  (+ dbias[n] += sum_m dY[m][n]).
// The reduction runs over ROWS, so both MFMA operands are column-strided in memory.  32-row slabs of dY and of the A window are
// staged row-major into LDS exactly as they are loaded (coalesced 16-byte pieces, split into three bf16 planes, 8-byte stores);
// the fragments -- eight consecutive rows m of one column per lane -- come out of LDS through gfx950's transposing read,
// ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and every lane receives one COLUMN of it, so two reads give
// a lane its 8 x bf16 fragment with no shuffle and no transposed staging (which would need 2-byte scattered stores).  Plain row-major
// rows make these reads 2-way bank-conflicted (rows 8 apart share banks whatever the padding: cdna_hip_programming.md T10); at 24
// reads against 24 MFMAs of 16 cycles per wave and slab that is hidden.
// Workgroup = 64 x 64 tile of dW, 4 waves as 2 (n) x 2 (k), each 32 x 32 = 2 x 2 MFMA tiles; the m range is split over workgroups
// exactly as in gemm_tn_kernel (gemm.hip: same split plan, same epilogue: float atomics or per-split partial tiles).
// Workgroup tile = (32 TNW) x (32 TKW) of dW: 4 waves as 2 (n) x 2 (k), each (16 TNW) x (16 TKW) = TNW x TKW MFMA tiles.  <2, 2> is the
// 64 x 64 tile of gemm_tn_kernel; the wide tiles halve the LDS fragment bytes and the L2 bytes per flop (the 64 x 64 tile moves
// 72 KB through LDS per slab for 24 MFMAs per wave: LDS-bound at the f32 kernel's speed), at one workgroup per CU.
template <int TNW, int TKW>
__global__ __launch_bounds__(256) void gemm_tn_split_kernel(const TnGroup g) {
    const int pi = group_find(g, blockIdx.x);
    const TnProb& pr = g.p[pi];
    const float* __restrict__ dY = pr.dY;
    const long ldy = pr.ldy, ldw = pr.ldw;
    const Win A = pr.A;
    float* __restrict__ dW = pr.dW;
    const int M = pr.M, N = pr.N, rows_per_split = pr.rows_per_split, out_kw = pr.out_kw;
    float* __restrict__ partial = pr.partial;
    float* __restrict__ dbias = pr.dbias;
    const int n_nt = pr.n_nt, n_kt = pr.n_kt;
    constexpr int MR = 32;                                   // slab depth = the MFMA's K
    constexpr int BN = 32 * TNW, BK = 32 * TKW;
    using YI = TrImage<BN>;
    using XI = TrImage<BK>;
    constexpr int LDN = YI::LD, LDK = XI::LD;
    constexpr int PY = BN / 4, PX = BK / 4;                  // 4-column pieces per slab row
    __shared__ __attribute__((aligned(16))) __bf16 ys[2][3][MR][LDN];
    __shared__ __attribute__((aligned(16))) __bf16 xs[2][3][MR][LDK];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    const int lid = xcd_chunked_id(blockIdx.x - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
    const int tn_n = lid % n_nt, tn_k = (lid / n_nt) % n_kt, tn_s = lid / (n_nt * n_kt);
    const int n0 = tn_n * BN, k0 = tn_k * BK;
    const int K = A.K;
    const int m_begin = tn_s * rows_per_split;
    if (m_begin >= M) return;                              // padding workgroup of a grouped launch (uniform: before any barrier)
    const int m_end = min(M, m_begin + rows_per_split);

    // staging: thread t owns pieces t, t + 256, ... of the slab's dY image (TNW of them) and of its A image (TKW); N % 4 == 0 and
    // K % 4 == 0 on this path, so a piece is inside or outside as a whole
    int yrow[TNW], ycol[TNW], ypos[TNW];
    bool y_ok[TNW];
#pragma unroll
    for (int i = 0; i < TNW; ++i) {
        const int p = t + 256 * i;
        yrow[i] = p / PY;
        ycol[i] = 4 * (p - yrow[i] * PY);
        y_ok[i] = n0 + ycol[i] < N;
        ypos[i] = YI::at(yrow[i], ycol[i]);
    }
    int xrow[TKW], xcol[TKW], xpos[TKW], a_off[TKW], a_tapd[TKW], mb[TKW], mr[TKW];
    bool a_ok[TKW];
#pragma unroll
    for (int i = 0; i < TKW; ++i) {
        const int p = t + 256 * i;
        xrow[i] = p / PX;
        xcol[i] = 4 * (p - xrow[i] * PX);
        xpos[i] = XI::at(xrow[i], xcol[i]);
        const int ak = k0 + xcol[i];
        a_ok[i] = ak < K;
        const int kc = a_ok[i] ? ak : 0;
        const int tap = kc / A.cw;
        a_off[i] = kc - tap * A.cw;
        a_tapd[i] = A.shift + tap * A.dil;
        const int m = m_begin + xrow[i];
        mb[i] = m / A.rows_out;
        mr[i] = m - mb[i] * A.rows_out;
    }
    const bool want_bias = dbias != nullptr && tn_k == 0;
    f32x4 bsum[TNW];
#pragma unroll
    for (int i = 0; i < TNW; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Two slabs of loads in flight (register sets 0 / 1, used alternately): with one, every 32-row slab cost a full memory latency
    // (~3900 cycles per slab against 384 cycles of MFMA at two workgroups per CU).  Loads are issued UNCONDITIONALLY from an
    // always-valid address and zeroed afterwards: a predicated load makes the number of outstanding loads dynamic and hipcc then
    // drains everything (vmcnt(0)) at the next use.  The slab count is rounded up to even; a slab past m_end stages zeros.
    // (the zeroing happens when a set is STAGED, not here: a select placed next to its load gets scheduled into the current slab's
    // MFMA section and waits for the loads just issued)
    f32x4 yv[2][TNW], xv[2][TKW];
    unsigned ymask[2] = {0u, 0u}, xmask[2] = {0u, 0u};
    auto fetch = [&](auto set_c, int m0) {
        constexpr int set = decltype(set_c)::value;
        unsigned ym = 0u, xm = 0u;
#pragma unroll
        for (int i = 0; i < TNW; ++i) {
            const int m = m0 + yrow[i];
            const bool ok = m < m_end && y_ok[i];
            yv[set][i] = *reinterpret_cast<const f32x4*>(ok ? dY + (long)m * ldy + n0 + ycol[i] : dY);
            ym |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < TKW; ++i) {
            const int sr = mr[i] * A.step + a_tapd[i];
            const bool ok = m0 + xrow[i] < m_end && a_ok[i] && sr >= 0 && sr < A.rows_in;
            xv[set][i] = *reinterpret_cast<const f32x4*>(ok ? A.ptr + (long)mb[i] * A.bs + (long)sr * A.rs + a_off[i] : A.ptr);
            xm |= ok ? (1u << i) : 0u;
            mr[i] += MR;
            while (mr[i] >= A.rows_out) { mr[i] -= A.rows_out; ++mb[i]; }
        }
        ymask[set] = ym; xmask[set] = xm;
    };

    f32x4 acc[TNW][TKW];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TKW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one slab: stage register set `set` into LDS buffer `set`, refill the set with the slab two ahead, multiply
    auto slab = [&](auto set_c, int m_next) {
        constexpr int set = decltype(set_c)::value;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TNW; ++i) {
            u32x2 o[3];
            const f32x4 v = (ymask[set] >> i) & 1u ? yv[set][i] : z;
            split4<3>(v, o);
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x2*>(&ys[set][s][0][0] + ypos[i]) = o[s];
            if (want_bias) bsum[i] += v;
        }
#pragma unroll
        for (int i = 0; i < TKW; ++i) {
            u32x2 o[3];
            const f32x4 v = (xmask[set] >> i) & 1u ? xv[set][i] : z;
            split4<3>(v, o);
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x2*>(&xs[set][s][0][0] + xpos[i]) = o[s];
        }
        __syncthreads();                                   // slab complete in buffer `set`; the other buffer is free again
        fetch(set_c, m_next);
        bf16x8 fb[3][TKW];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int q = 0; q < TKW; ++q) fb[s][q] = XI::frag(&xs[set][s][0][0], wk * (16 * TKW) + q * 16, r16, kq);
#pragma unroll
        for (int nt = 0; nt < TNW; ++nt) {
            bf16x8 fa[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) fa[s] = YI::frag(&ys[set][s][0][0], wn * (16 * TNW) + nt * 16, r16, kq);
#pragma unroll
            for (int kt = 0; kt < TKW; ++kt) {             // smallest terms first
                f32x4 cc = acc[nt][kt];
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1][kt], cc, 0, 0, 0);
                acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0][kt], cc, 0, 0, 0);
            }
        }
    };
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    fetch(set0{}, m_begin);
    fetch(set1{}, m_begin + MR);
    for (int m0 = m_begin; m0 < m_end; m0 += 2 * MR) {
        slab(set0{}, m0 + 2 * MR);
        slab(set1{}, m0 + 3 * MR);
    }

#pragma unroll
    for (int nt = 0; nt < TNW; ++nt)
#pragma unroll
        for (int kt = 0; kt < TKW; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * (16 * TNW) + nt * 16 + kq * 4 + i;
                const int kcol = k0 + wk * (16 * TKW) + kt * 16 + r16;
                if (n < N && kcol < K && partial) {
                    partial[((long)tn_s * N + n) * K + kcol] = acc[nt][kt][i];     // combined in fp64 by the reduce kernel
                } else if (n < N && kcol < K) {
                    const long off = out_kw > 0 ? (long)(kcol % A.cw) * out_kw + kcol / A.cw : (long)kcol;
                    atomicAdd(&dW[(long)n * ldw + off], acc[nt][kt][i]);
                }
            }
    if (want_bias) {       // column sums of dY over this split: the staging rows' partial sums meet in LDS (the dY image is done with)
        float (*bs)[BN + 4] = reinterpret_cast<float (*)[BN + 4]>(&ys[0][0][0][0]);
        static_assert(sizeof(float) * MR * (BN + 4) <= sizeof(ys), "bias scratch fits the dY image");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TNW; ++i) *reinterpret_cast<f32x4*>(&bs[yrow[i]][ycol[i]]) = bsum[i];
        __syncthreads();
        if (t < BN && n0 + t < N) {
            float sacc = 0.f;
#pragma unroll
            for (int r = 0; r < MR; ++r) sacc += bs[r][t];
            atomicAdd(&dbias[n0 + t], sacc);
        }
    }
}

}  // namespace tg

using namespace tg;

static int g_math_mode = 0;

extern "C" int tg_set_math_mode(int32_t mode) {
    TG_REQUIRE(mode == 0 || mode == 1, "tg_set_math_mode: mode must be 0 (fp32-accurate) or 1 (bf16 operands)");
    g_math_mode = mode;
    return 0;
}

extern "C" int tg_get_math_mode(void) { return g_math_mode; }

// Deterministic mode: every cross-workgroup combine in a fixed order (no float atomics) -- two runs from the same state are bit-identical.
static int g_deterministic = 0;
extern "C" int tg_set_deterministic(int32_t on) { g_deterministic = on ? 1 : 0; return 0; }
extern "C" int tg_get_deterministic(void) { return g_deterministic; }

// Tile menu (tools/gemm_split_lab.hip on the shapes of the training step).  Single-buffered 128-row tiles run two workgroups per CU
// and win when the grid has at least ~2 workgroups per CU (the stacked forward, M = 13056); the backward shapes (M = 4352) have too
// few 128-row tiles for that and run 64-row double-buffered tiles instead (476 x [64 x 96] for N = 600: 50 us against 59 for 238 x
// [128 x 96] and 68 for the f32-MFMA kernel).
// FAST addressing precondition for one problem (host): no padding anywhere in the window, 32-bit byte offsets, taps / weight segments at
// least one slab wide
static bool nt_fast_ok(const NtProb& p) {
    const Win& A = p.A;
    if (A.cw < 32 || p.b_seg_k < 32 || A.K < 64 || A.cw % 4 != 0 || p.b_seg_k % 4 != 0 || A.K % A.cw != 0 || A.K % p.b_seg_k != 0) return false;
    const int taps = A.K / A.cw, segs = A.K / p.b_seg_k;
    if (A.rows_out <= 0 || p.M % A.rows_out != 0) return false;                 // whole batches
    // source rows r * step + shift + kk * dil over r in [0, rows_out), kk in [0, taps): both extremes inside [0, rows_in)
    const long r_lo = (A.step >= 0 ? 0 : (long)(A.rows_out - 1) * A.step) + A.shift + (A.dil >= 0 ? 0 : (long)(taps - 1) * A.dil);
    const long r_hi = (A.step >= 0 ? (long)(A.rows_out - 1) * A.step : 0) + A.shift + (A.dil >= 0 ? (long)(taps - 1) * A.dil : 0);
    if (r_lo < 0 || r_hi >= A.rows_in) return false;
    if (A.bs < 0 || A.rs < 0 || p.ldb < 0 || p.b_seg_stride < 0) return false;
    const long batches = p.M / A.rows_out;
    const long a_max = (batches - 1) * A.bs + r_hi * A.rs + A.cw;                // elements
    const long b_max = (long)(p.N - 1) * p.ldb + (long)(segs - 1) * p.b_seg_stride + p.b_seg_k;
    return a_max < (1l << 29) && b_max < (1l << 29);
}

struct SplitTile { int tm, tn, db; };
static SplitTile split_pick_tile(int M, int N) {
    auto wgs = [&](int bm, int bn) { return (long)cdiv(M, bm) * cdiv(N, bn); };
    auto waste = [&](int bn) { return cdiv(N, bn) * bn - N; };
    if (wgs(128, 96) >= 512 && waste(96) <= waste(64) + 32) return {4, 3, 0};
    if (wgs(128, 64) >= 384) return {4, 2, 0};
    if (wgs(64, 96) >= 256 && waste(96) <= waste(64) + 32) return {2, 3, 1};
    return {2, 2, 1};
}

bool tg_gemm_nt_mw_eligible(NtGroup& g, int* tm, int* tn);
int tg_gemm_nt_mw_launch(NtGroup& g, int tm, int tn, int splits, hipStream_t s);

int tg_gemm_nt_split_launch(NtGroup& g, hipStream_t s) {
    {   // many-row products whose big tiles fill the chip: mover-wave kernel (gemm_mw.hip)
        int tm = 0, tn = 0;
        if (tg_gemm_nt_mw_eligible(g, &tm, &tn)) {
            TG_REQUIRE(!(g.p[0].h2 && g_math_mode == 1), "tg_gemm_nt: fp16 x 2 weight planes belong to the fp32-accurate mode, not to the bf16 tier");
            return tg_gemm_nt_mw_launch(g, tm, tn, g_math_mode == 1 ? 1 : 3, s);
        }
    }
    for (int i = 0; i < g.n; ++i)
        TG_REQUIRE(!g.p[i].c_rmax && !g.p[i].c2_rmax, "tg_gemm_nt: c_rowmax / c2_rowmax are outputs of the mover-wave kernel only (tg_gemm_nt_kernel_plan == 2; problem %d)", i);
    for (int i = 0; i < g.n; ++i)
        TG_REQUIRE(g.p[i].drop_state == nullptr || g.p[i].vec_c, "tg_gemm_nt: regenerated dropout needs a vectorisable C (N %% 4 == 0, strides %% 4 == 0, "
                   "16-byte aligned; problem %d)", i);
    int Mx = 0, Nx = 0;
    for (int i = 0; i < g.n; ++i) { Mx = Mx > g.p[i].M ? Mx : g.p[i].M; Nx = Nx > g.p[i].N ? Nx : g.p[i].N; }
    const SplitTile tl = split_pick_tile(Mx * g.n, Nx);
    const int bm = 32 * tl.tm, bn = 32 * tl.tn;
    int wg = 0;
    for (int i = 0; i < g.n; ++i) {             // every problem's range starts at a multiple of 8 (XCD mapping, see gemm.hip nt_layout)
        g.p[i].n_nt = cdiv(g.p[i].N, bn);
        g.wg_begin[i] = wg;
        wg += (cdiv(g.p[i].M, bm) * g.p[i].n_nt + 7) / 8 * 8;
    }
    for (int i = g.n; i <= TG_MAX_GROUP; ++i) g.wg_begin[i] = wg;
    const dim3 grid(wg);
    // the 128-row tiles held to 3 (128 x 96: 180 -> 162 VGPRs) / 4 (128 x 64) waves per SIMD: three workgroups per CU instead of two, 119.7 ->
    // 116.5 us on [13056 x 900 x 600], 6.29 -> 6.24 ms per iteration
#define TG_SPLIT(TM_, TN_, SP_, DB_)                                                                                   \
    do {                                                                                                               \
        if (TM_ == 4 && SP_ == 3) hipLaunchKernelGGL((gemm_nt_split_kernel<TM_, TN_, SP_, DB_, 1, (TM_ == 4 && SP_ == 3) ? (TN_ == 3 ? 3 : 4) : 1>), grid, dim3(256), 0, s, g); \
        else hipLaunchKernelGGL((gemm_nt_split_kernel<TM_, TN_, SP_, DB_, 1>), grid, dim3(256), 0, s, g);              \
    } while (0)
#define TG_SPLIT_MENU(SP_)                                              \
    do {                                                                \
        if (tl.tm == 4 && tl.tn == 3) TG_SPLIT(4, 3, SP_, 0);           \
        else if (tl.tm == 4 && tl.tn == 2) TG_SPLIT(4, 2, SP_, 0);      \
        else if (tl.tm == 2 && tl.tn == 3) TG_SPLIT(2, 3, SP_, 1);      \
        else TG_SPLIT(2, 2, SP_, 1);                                    \
    } while (0)
    // fast addressing (see the kernel's FAST note): decided for the group as a whole
    static const int fast_on = [] { const char* e = getenv("TG_NT_FAST"); return e ? atoi(e) : 1; }();
    bool fast = fast_on != 0 && g_math_mode == 0;
    for (int i = 0; i < g.n && fast; ++i) fast = nt_fast_ok(g.p[i]);
    if (fast) {
#define TG_FAST(TM_, TN_, DB_, R_, O_) hipLaunchKernelGGL((gemm_nt_split_kernel<TM_, TN_, 3, DB_, R_, O_, 0, true>), grid, dim3(256), 0, s, g)
        if (tl.tm == 4 && tl.tn == 3) TG_FAST(4, 3, 0, 1, 3);
        else if (tl.tm == 4 && tl.tn == 2) TG_FAST(4, 2, 0, 1, 4);
        else if (tl.tm == 2 && tl.tn == 3) TG_FAST(2, 3, 1, 1, 1);
        else TG_FAST(2, 2, 1, 1, 1);
#undef TG_FAST
        return check_launch("tg_gemm_nt(split, fast addressing)");
    }
#ifdef TG_LAB_ABLATE
    {
        const char* e = getenv("TG_NT_ABL");
        const int abl = e ? atoi(e) : 0;
        if (abl && tl.tm == 4 && tl.tn == 3 && g_math_mode == 0) {
#define TG_ABL(A_) case A_: hipLaunchKernelGGL((gemm_nt_split_kernel<4, 3, 3, 0, 1, 3, A_>), grid, dim3(256), 0, s, g); return check_launch("tg_gemm_nt(split, ablated)")
            switch (abl) { TG_ABL(1); TG_ABL(2); TG_ABL(3); TG_ABL(4); TG_ABL(5); TG_ABL(7); TG_ABL(8); TG_ABL(9); TG_ABL(11); TG_ABL(15); default: break; }
#undef TG_ABL
        }
    }
#endif
    if (g_math_mode == 1) TG_SPLIT_MENU(1);
    else TG_SPLIT_MENU(3);
#undef TG_SPLIT_MENU
#undef TG_SPLIT
    return check_launch("tg_gemm_nt(split)");
}

// every problem of the group must be on the vectorisable layout (checked by the caller: TnProb.vec_y && vec_a, N % 4 == 0, K % 4 == 0);
// n_nt / n_kt of the problems count (32 tnw) x (32 tkw) tiles
int tg_gemm_tn_split_launch(const TnGroup& g, int total_wgs, int tnw, int tkw, hipStream_t s) {
#define TG_TN(TNW_, TKW_) hipLaunchKernelGGL((gemm_tn_split_kernel<TNW_, TKW_>), dim3(total_wgs), dim3(256), 0, s, g)
    if (tnw == 2 && tkw == 2) TG_TN(2, 2);
    else if (tnw == 4 && tkw == 4) TG_TN(4, 4);
    else if (tnw == 4 && tkw == 5) TG_TN(4, 5);
    else if (tnw == 5 && tkw == 5) TG_TN(5, 5);
    else if (tnw == 4 && tkw == 2) TG_TN(4, 2);
    else TG_REQUIRE(false, "tg_gemm_tn(split): no %d x %d tile", tnw, tkw);
#undef TG_TN
    return check_launch("tg_gemm_tn(split)");
}
