// tg_gemm_tn, the big weight gradients: dW[n][k] += sum_m dY[m][n] * A(m, k) on the bf16 matrix cores at fp32 accuracy (bf16 x 3), with MOVER
// WAVES -- the weight-gradient counterpart of gemm_mw.hip.
//
// gemm_tn_split_kernel (gemm_split.hip) runs 64 x 64 / 128 x 64 tiles on 256-thread workgroups whose waves stage, barrier, multiply,
// barrier: 89 TFLOP/s on the four weight gradients of a GRU layer, bound by re-reading its two row-major operands (~940 MB of L2 traffic
// per group launch).  Here one PERSISTENT 768-thread workgroup per CU owns a 192 x 160 tile of dW (2.3 x fewer
// operand bytes per flop) and walks work items (problem, tile, row split):
//   * mover waves 8-11 (one per SIMD) fetch the fp32 slabs of dY and of the A window (32 rows of the reduction) two slabs ahead into
//     registers, split every value exactly into three bf16 terms and store the planes row-major into a double-buffered LDS image;
//   * matrix waves 0-7 (two per SIMD, 4 (n) x 2 (k), wave tile 48 x 80) read their fragments TRANSPOSED (ds_read_b64_tr_b16: eight
//     consecutive rows of one column per lane, tr_image.hpp) and issue nothing but those reads and MFMAs -- 90 per wave and slab, the next
//     row tile's fragments in flight under the current one's MFMAs, the next slab's k fragments reloaded one by one under the last row tile's.
// One workgroup barrier per slab; the slab stream runs on across items.  Row splits are combined with float atomics straight from the
// accumulators (the tile menu keeps the group at one item per CU: 29 MB of atomics for a GRU layer's four gradients instead of 91).
// The bias gradient costs nothing: a tile whose k range has a padding column past K stages a column of ONES there, so the product's
// column K is sum_m dY[m][n] = dbias[n] (exact: 1.0 = its own hi term).
// Out-of-range pieces (rows past the item's range, columns past N / K, window padding) are bounds-checked buffer loads that return zero.
#include "common.hpp"
#include "tr_image.hpp"
#include <stdlib.h>
#include <type_traits>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace tg {

constexpr unsigned TW_RSRC3 = 0x00020000u;
constexpr unsigned TW_OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tw_rsrc(const void* base, unsigned bytes) {     // provably wave-uniform descriptor (guide T20)
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), TW_RSRC3);
}

// fp16 x 2: the piece's four COLUMNS each carry their own power-of-two scale 2^k (the reduction runs over rows, so scales are per column);
// `kb` packs the four k as signed bytes, first column in the low byte
__device__ __forceinline__ void tw_split4_h2(const f32x4 v, const int kb, u32x2 (&out)[2]) {
    const float x0 = __builtin_ldexpf(v[0], (kb << 24) >> 24), x1 = __builtin_ldexpf(v[1], (kb << 16) >> 24);
    const float x2 = __builtin_ldexpf(v[2], (kb << 8) >> 24), x3 = __builtin_ldexpf(v[3], kb >> 24);
    unsigned h0, l0, h1, l1;
    h2_split2(x0, x1, h0, l0);
    h2_split2(x2, x3, h1, l1);
    out[0] = u32x2{h0, h1};
    out[1] = u32x2{l0, l1};
}
// four column magnitudes -> their packed scale exponents k = 141 - e (in [-109, 109])
__device__ __forceinline__ int tw_pack_k(const f32x4 cm) {
    int kb = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float f = cm[q]; kb |= ((141 - h2_exp_of_bits(__float_as_uint(f))) & 0xff) << (8 * q); }
    return kb;
}

template <int NS>
__device__ __forceinline__ void tw_split4(const f32x4 v, u32x2 (&out)[NS]) {
    static_assert(NS != 2, "fp16 x 2 pieces go through tw_split4_h2");
    if constexpr (NS == 1) {                         // plain bf16 tier (math mode 1): round to nearest even
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 r;
        r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
        out[0] = __builtin_bit_cast(u32x2, r);
        return;
    }
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xf = v[i];
        split3_bits(xf, h[i], m[i], l[i]);
    }
    if constexpr (NS == 3) {
        out[0] = u32x2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
        out[1] = u32x2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
        out[2] = u32x2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
    }
}

// wave tile (16 TNT) x (16 TKT) of dW; matrix waves WNW (n) x WKW (k); workgroup tile BN x BK.  NS = 3: bf16 x 3 (fp32-accurate); NS = 1: plain
// bf16 operands (math mode 1)
// ABL (lab builds only, -DTG_LAB_ABLATE, tools/tn_mw_ablate.py; results are WRONG by construction): bit 0 drops the MFMAs, bit 1 the movers' split
// arithmetic + LDS stores, bit 2 the movers' global loads, bit 3 the matrix waves' LDS fragment reads, bit 4 the epilogue's atomics
template <int TNT, int TKT, int WNW, int WKW, int NS, int ABL = 0>
__global__ __launch_bounds__(768, 3) void gemm_tn_mw_kernel(const TnGroup g) {
    static_assert(WNW * WKW == 8, "eight matrix waves");
    constexpr int BN = 16 * TNT * WNW, BK = 16 * TKT * WKW;
    using YI = TrImage<BN>;
    using XI = TrImage<BK>;
    constexpr int LDN = YI::LD, LDK = XI::LD;
    constexpr int PY = BN / 4, PX = BK / 4;                    // 4-column pieces per slab row
    constexpr int NPY = 32 * PY / 256, NPX = 32 * PX / 256;    // pieces per mover thread and slab
    static_assert(32 * PY % 256 == 0 && 32 * PX % 256 == 0, "whole pieces per mover thread");
    constexpr int Y_PLANE = 32 * LDN, X_PLANE = 32 * LDK;      // bf16 elements
    __shared__ __attribute__((aligned(16))) __bf16 ys[2][NS][Y_PLANE];
    __shared__ __attribute__((aligned(16))) __bf16 xs[2][NS][X_PLANE];
    __shared__ int kexp[NS == 2 ? 2 : 1][NS == 2 ? NPY + NPX : 1][NS == 2 ? 256 : 1];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int total_items = g.wg_begin[TG_MAX_GROUP];
    const int G = gridDim.x;

    // work item `vb` -> problem, tile origin, row range; false for a padding id
    int lid_last = 0;                                          // item index inside its problem (the slot of its partial tile) of the last decode
    auto decode = [&](int vb, int& pi, int& n0, int& k0, int& m_begin, int& m_end) __attribute__((always_inline)) -> bool {
        pi = group_find(g, vb);
        const TnProb& pr = g.p[pi];
        const int lid = xcd_chunked_id(vb - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
        lid_last = lid;
        const int tn_n = lid % pr.n_nt, tn_k = (lid / pr.n_nt) % pr.n_kt, tn_s = lid / (pr.n_nt * pr.n_kt);
        n0 = tn_n * BN; k0 = tn_k * BK;
        m_begin = tn_s * pr.rows_per_split;
        m_end = min(pr.M, m_begin + pr.rows_per_split);
        return m_begin < pr.M;
    };
    int total = 0;                                             // slabs this workgroup walks (both roles count the same barriers)
    for (int vb = blockIdx.x; vb < total_items; vb += G) {
        int pi, n0, k0, mb, me;
        if (decode(vb, pi, n0, k0, mb, me)) total += (me - mb + 31) >> 5;
    }

    if (wave >= 8) {
        // ============================================================================================ movers (waves 8-11)
        const int mt = t - 512;
        // piece i of this thread: slab row (mt + 256 i) / PY, columns 4 ((mt + 256 i) % PY) .. + 3 (PX for the A image).  Only the LDS positions
        // stay in registers; next_item() re-derives rows and columns from a laundered copy of mt (compile-time divisors: a few instructions per
        // item) -- as loop invariants they would cost 22 registers of the 168 a wave has here
        int ypos[NPY], xpos[NPX];
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
            const int p = mt + 256 * i;
            ypos[i] = YI::at(p / PY, 4 * (p % PY));
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int p = mt + 256 * i;
            xpos[i] = XI::at(p / PX, 4 * (p % PX));
        }
        // fetch cursor: item vb_f, slab s_f of nslab_f; runs three slabs ahead of the slab the matrix waves multiply
        int vb_f = blockIdx.x - G, s_f = 0, nslab_f = 0;
        bool live = true;
        // Per-slab address work is kept to a few full-rate instructions per piece (round 5: the float-reciprocal row division, the row-range
        // compares and four 32-bit integer multiplies per A piece had made the fetch as expensive as the staging arithmetic -- ~680 vector
        // instructions per mover wave and slab, the movers then set the pace of the kernel):
        //   * dY: the piece's offset is fixed per item (TW_OOB for columns past N) and the slab advances through the instruction's scalar
        //     offset; rows past the item's range read ZERO because the buffer descriptor ends at the item's last row -- and with dY zero there
        //     the A rows (and the ones column of the bias trick) need no row check at all: 0 x finite = 0;
        //   * A: offset and row-in-batch advance by increments (one compare-and-select per slab for the batch wrap: windows have >= 32 rows
        //     per batch on this path), the padding check (r * step + tap displacement inside [0, rows_in)) only for windows that have padding.
        unsigned y_off[NPY];                                   // byte offset of (row m_begin + yrow, column n0 + ycol) in dY, or TW_OOB
        unsigned x_off[NPX];                                   // byte offset of the piece for the cursor's slab (valid columns), advanced per slab
        int x_r[NPX];                                          // its row inside the batch (before step / tap displacement)
        int x_tapd[NPX];                                       // tap row displacement
        int x_one = -1;                                        // 4 i + q: element q of X piece i is the ones column of this item (at most one piece of a thread holds column K), or -1
        // fp16 x 2: the pieces' packed column-scale exponents live in LDS, one private slot per thread, piece and item PARITY (the fetch cursor
        // is at most one item ahead of the set being staged: items have >= 8 slabs): kept in registers they and their per-set copies cost 33
        // of the 168 registers a wave has here
        int item_par = 0;
        unsigned y_slab_b = 0;                                 // bytes between slabs in dY
        const float* y_ptr = g.p[0].dY;
        const float* a_ptr = g.p[0].A.ptr;
        unsigned y_bytes = 0, a_bytes = 0, x_inc = 0, x_wrap = 0;
        int a_rows_out = 32, a_rows_in = 0, a_step = 1;
        bool a_pad = false;
        auto next_item = [&]() __attribute__((always_inline)) {
            int pi = 0, n0 = 0, k0 = 0, mb = 0, me = 0;        // (locals: a captured variable passed by reference here ends up in scratch memory)
            do {
                vb_f += G;
                if (vb_f >= total_items) { live = false; break; }
            } while (!decode(vb_f, pi, n0, k0, mb, me));
            s_f = 0;
            int mt_l = mt;
            asm volatile("" : "+v"(mt_l));                       // (keeps the row / column arithmetic below inside this function)
            if (!live) {
#pragma unroll
                for (int i = 0; i < NPY; ++i) y_off[i] = TW_OOB;
#pragma unroll
                for (int i = 0; i < NPX; ++i) x_off[i] = TW_OOB;
                x_one = -1;
                y_slab_b = 0; x_inc = 0; x_wrap = 0;
                return;
            }
            const TnProb& pr = g.p[pi];
            const Win A = pr.A;
            nslab_f = (me - mb + 31) >> 5;
            item_par ^= 1;
#pragma unroll
            for (int i = 0; i < NPY; ++i) {
                const int p = mt_l + 256 * i, yrow = p / PY, ycol = 4 * (p % PY);
                y_off[i] = n0 + ycol < pr.N ? (unsigned)(((long)(mb + yrow) * pr.ldy + n0 + ycol) * 4) : TW_OOB;
                if constexpr (NS == 2) kexp[item_par][i][mt] = tw_pack_k(*reinterpret_cast<const f32x4*>(pr.y_cmax + (n0 + ycol < pr.N ? n0 + ycol : 0)));
            }
            const bool bias_here = pr.dbias != nullptr && k0 <= A.K && A.K < k0 + BK;       // this tile holds the padding column K
            x_one = -1;
            const unsigned rs4 = (unsigned)(A.rs * 4), bs4 = (unsigned)(A.bs * 4);
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                const int p = mt_l + 256 * i, xrow = p / PX, xcol = 4 * (p % PX);
                const int ak = k0 + xcol;
                const bool cok = ak < A.K;
                const int kc = cok ? ak : 0;
                const int tap = kc / A.cw;
                x_tapd[i] = A.shift + tap * A.dil;
                const int m = mb + xrow;
                const int b = m / A.rows_out;
                x_r[i] = m - b * A.rows_out;
                // (unsigned wrap-around arithmetic: a row before the tensor's start gives an offset the padding check replaces anyway.  A
                // column past K starts at TW_OOB: the increments of an item add up to less than the tensor's 2^31 bytes, so it stays past
                // num_records for the item's life)
                x_off[i] = cok ? (unsigned)b * bs4 + (unsigned)(x_r[i] * A.step + x_tapd[i]) * rs4 + (unsigned)((kc - tap * A.cw) * 4) : TW_OOB;
                if (bias_here && ak <= A.K && A.K < ak + 4) x_one = 4 * i + (A.K - ak);
                if constexpr (NS == 2) {
                    int kb = tw_pack_k(*reinterpret_cast<const f32x4*>(pr.a_cmax + (kc - tap * A.cw)));
                    if (bias_here && ak <= A.K && A.K < ak + 4) kb &= ~(0xff << (8 * (A.K - ak)));          // the ones column keeps scale 1
                    kexp[item_par][NPY + i][mt] = kb;
                }
            }
            y_slab_b = (unsigned)(32 * pr.ldy * 4);
            // dY's descriptor ends behind the item's last row: later rows of the last slab read zero
            const long y_end = (long)me * pr.ldy * 4;
            y_ptr = pr.dY; y_bytes = y_end < (long)pr.y_bytes ? (unsigned)y_end : pr.y_bytes;
            a_ptr = A.ptr; a_bytes = pr.a_bytes;
            a_rows_out = A.rows_out; a_rows_in = A.rows_in; a_step = A.step;
            x_inc = 32u * (unsigned)A.step * rs4;
            x_wrap = bs4 - (unsigned)A.rows_out * (unsigned)A.step * rs4;                    // added when the row index wraps into the next batch
            // padding anywhere in the window?  rows r * step + shift + tap * dil over r in [0, rows_out), tap in [0, K / cw)
            const int taps = A.K / A.cw;
            const long lo = (A.step >= 0 ? 0 : (long)(A.rows_out - 1) * A.step) + A.shift + (A.dil >= 0 ? 0 : (long)(taps - 1) * A.dil);
            const long hi = (A.step >= 0 ? (long)(A.rows_out - 1) * A.step : 0) + A.shift + (A.dil >= 0 ? (long)(taps - 1) * A.dil : 0);
            a_pad = lo < 0 || hi >= A.rows_in;
        };
        // TWO register sets of loads in flight (round 5).  Round 3's single set was staged at the top of a step and refilled right behind it: its
        // loads had only the rest of that step to land and the ISA drained vmcnt down to 0 at the top of every step.  Now slab n + 3 is fetched
        // while slab n + 1 is staged: two full steps for the loads.  All addresses of a fetch are formed before its first load so that the loads
        // issue back to back (a fixed number of vector-memory operations per step on every path: hipcc's static counts then come out exact).
        u32x4 gy[2][NPY], gx[2][NPX];
        int set_par[2] = {0, 0};                               // ... parity of the item each register set was fetched for
        int one_el[2] = {-1, -1};                              // 4 i + q of the ones column for the item the set was FETCHED for (-1: none)
        auto fetch = [&](auto set_c) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            one_el[set] = x_one;
            set_par[set] = item_par;
            const unsigned ysoff = (unsigned)__builtin_amdgcn_readfirstlane(s_f * (int)y_slab_b);
            unsigned xo[NPX];
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                xo[i] = x_off[i];
                if (a_pad) {                                   // (wave-uniform)
                    const int sr = __mul24(x_r[i], a_step) + x_tapd[i];
                    xo[i] = (unsigned)sr < (unsigned)a_rows_in ? xo[i] : TW_OOB;
                }
                // advance to the next slab: 32 rows on, at most one batch wrap (rows_out >= 32: checked by the planner)
                const int r2 = x_r[i] + 32;
                const bool wrap = r2 >= a_rows_out;
                x_r[i] = wrap ? r2 - a_rows_out : r2;
                x_off[i] += x_inc + (wrap ? x_wrap : 0u);
            }
            const __amdgpu_buffer_rsrc_t yr = tw_rsrc(y_ptr, y_bytes), ar = tw_rsrc(a_ptr, a_bytes);
#pragma unroll
            for (int i = 0; i < NPY; ++i) {
                if constexpr (ABL & 4) gy[set][i] = u32x4{0x3f800000u + (unsigned)lane, 0x40000000u, y_off[i] | 0x3f000000u, ysoff | 0x3e800000u};
                else gy[set][i] = __builtin_amdgcn_raw_buffer_load_b128(yr, y_off[i], ysoff, 0);
            }
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                if constexpr (ABL & 4) gx[set][i] = u32x4{0x3f800000u + (unsigned)lane, 0x40400000u, xo[i] | 0x3f000000u, 0x3e800000u};
                else gx[set][i] = __builtin_amdgcn_raw_buffer_load_b128(ar, xo[i], 0, 0);
            }
            if (live && ++s_f >= nslab_f) next_item();
        };
        auto stage = [&](auto set_c, int buf) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            if constexpr (ABL & 2) {                              // loaded values stay live (the loads must still be waited for), nothing else
#pragma unroll
                for (int i = 0; i < NPY; ++i) asm volatile("" :: "v"(gy[set][i]));
#pragma unroll
                for (int i = 0; i < NPX; ++i) asm volatile("" :: "v"(gx[set][i]));
                return;
            }
            int kk[NS == 2 ? NPY + NPX : 1];
            if constexpr (NS == 2) {
#pragma unroll
                for (int i = 0; i < NPY + NPX; ++i) kk[i] = kexp[set_par[set]][i][mt];
            }
#pragma unroll
            for (int i = 0; i < NPY; ++i) {
                u32x2 o[NS];
                if constexpr (NS == 2) tw_split4_h2(__builtin_bit_cast(f32x4, gy[set][i]), kk[i], o);
                else tw_split4<NS>(__builtin_bit_cast(f32x4, gy[set][i]), o);
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(&ys[buf][s][0] + ypos[i]) = o[s];
            }
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                f32x4 v = __builtin_bit_cast(f32x4, gx[set][i]);
                const int e = one_el[set] - 4 * i;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = e == q ? 1.0f : v[q];
                u32x2 o[NS];
                if constexpr (NS == 2) tw_split4_h2(v, kk[NPY + i], o);
                else tw_split4<NS>(v, o);
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(&xs[buf][s][0] + xpos[i]) = o[s];
            }
        };
        using set0 = std::integral_constant<int, 0>;
        using set1 = std::integral_constant<int, 1>;
        next_item();
        fetch(set0{});                                         // slab 0
        fetch(set1{});                                         // slab 1
        stage(set0{}, 0);
        fetch(set0{});                                         // slab 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // step n: the matrix waves multiply slab n out of buffer n & 1; slab n + 1 (fetched two steps ago into set (n + 1) & 1) is staged into
        // the other buffer (read last during step n - 1) and slab n + 3 fetched into the set just emptied
        int n = 0;
        auto step = [&](auto set_c, int buf) __attribute__((always_inline)) {
            if (n + 1 < total) stage(set_c, buf);
            fetch(set_c);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ++n;
        };
        while (n + 2 <= total) {
            step(set1{}, 1);
            step(set0{}, 0);
        }
        if (n < total) step(set1{}, 1);
    } else {
        // ============================================================================================ matrix waves (0-7)
        const int wn = wave / WKW, wk = wave % WKW;
        const int r16 = lane & 15, kq = lane >> 4;
        f32x4 acc[TNT][TKT];
        bf16x8 fa[2][NS], fb[NS][TKT];
        auto load_fa = [&](bf16x8 (&f)[NS], int buf, int nt) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if constexpr (ABL & 8) { u32x4 c4 = {0x3f803f80u + (unsigned)lane + nt, 0x3f003f00u + s, 0x3f803f80u, 0x3e803e80u + buf}; f[s] = __builtin_bit_cast(bf16x8, c4); }
                else f[s] = YI::frag(&ys[buf][s][0], wn * (16 * TNT) + nt * 16, r16, kq);
            }
        };
        auto load_fb = [&](int buf, auto kt_c) {
            constexpr int kt = decltype(kt_c)::value;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if constexpr (ABL & 8) { u32x4 c4 = {0x3f803f80u + (unsigned)lane + kt, 0x3f003f00u + s, 0x3f803f80u, 0x3e803e80u + buf}; fb[s][kt] = __builtin_bit_cast(bf16x8, c4); }
                else fb[s][kt] = XI::frag(&xs[buf][s][0], wk * (16 * TKT) + kt * 16, r16, kq);
            }
        };
        auto mma = [&](const bf16x8 (&f)[NS], auto nt_c, auto kt_c) {       // the six significant partial products, smallest first
            constexpr int nt = decltype(nt_c)::value, kt = decltype(kt_c)::value;
            if constexpr (ABL & 1) {                   // fragments stay live, no matrix instruction
                asm volatile("" :: "v"(f[0]), "v"(f[NS - 1]), "v"(fb[0][kt]), "v"(fb[NS - 1][kt]));
                return;
            }
            f32x4 cc = acc[nt][kt];
            if constexpr (NS == 2) {                   // fp16 x 2: lo hi + hi lo + hi hi
                auto h = [](const bf16x8& v) { return __builtin_bit_cast(tg_f16x8, v); };
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(f[1]), h(fb[0][kt]), cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(f[0]), h(fb[1][kt]), cc, 0, 0, 0);
                acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(f[0]), h(fb[0][kt]), cc, 0, 0, 0);
                return;
            }
            if constexpr (NS == 3) {
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], fb[0][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fb[2][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], fb[1][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], fb[0][kt], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fb[1][kt], cc, 0, 0, 0);
            }
            acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fb[0][kt], cc, 0, 0, 0);
        };
        int n = 0;
        // one slab: fb holds its k fragments.  Row tile nt + 1's fragments are read during the MFMAs of row tile nt; before the last row tile
        // every read of this slab is back -> barrier -> under the last row tile's MFMAs the NEXT slab's k fragments are reloaded one by one,
        // each right after the MFMAs that read the old one (the next slab is slab n + 1 of the workgroup's sequence, also across items)
        auto slab = [&]() {
            const int buf = n & 1, nbuf = buf ^ 1;
            load_fa(fa[0], buf, 0);
            auto rowtile = [&](auto nt_c) {
                constexpr int nt = decltype(nt_c)::value;
                if constexpr (nt + 1 < TNT) {
                    load_fa(fa[(nt + 1) & 1], buf, nt + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(fa[nt & 1], nt_c, std::integral_constant<int, 0>{});
                    if constexpr (TKT > 1) mma(fa[nt & 1], nt_c, std::integral_constant<int, 1>{});
                    if constexpr (TKT > 2) mma(fa[nt & 1], nt_c, std::integral_constant<int, 2>{});
                    if constexpr (TKT > 3) mma(fa[nt & 1], nt_c, std::integral_constant<int, 3>{});
                    if constexpr (TKT > 4) mma(fa[nt & 1], nt_c, std::integral_constant<int, 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    auto one = [&](auto kt_c) {
                        mma(fa[nt & 1], nt_c, kt_c);
                        __builtin_amdgcn_sched_barrier(0);
                        load_fb(nbuf, kt_c);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    one(std::integral_constant<int, 0>{});
                    if constexpr (TKT > 1) one(std::integral_constant<int, 1>{});
                    if constexpr (TKT > 2) one(std::integral_constant<int, 2>{});
                    if constexpr (TKT > 3) one(std::integral_constant<int, 3>{});
                    if constexpr (TKT > 4) one(std::integral_constant<int, 4>{});
                    static_assert(TKT <= 5, "k tiles per wave");
                }
            };
            rowtile(std::integral_constant<int, 0>{});
            if constexpr (TNT > 1) rowtile(std::integral_constant<int, 1>{});
            if constexpr (TNT > 2) rowtile(std::integral_constant<int, 2>{});
            static_assert(TNT <= 3, "row tiles per wave");
            ++n;
        };
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_fb(0, std::integral_constant<int, 0>{});          // slab 0's k fragments (the loop reloads every later slab's)
        if constexpr (TKT > 1) load_fb(0, std::integral_constant<int, 1>{});
        if constexpr (TKT > 2) load_fb(0, std::integral_constant<int, 2>{});
        if constexpr (TKT > 3) load_fb(0, std::integral_constant<int, 3>{});
        if constexpr (TKT > 4) load_fb(0, std::integral_constant<int, 4>{});
        for (int vb = blockIdx.x; vb < total_items; vb += G) {
            int pi, n0, k0, m_begin, m_end;
            if (!decode(vb, pi, n0, k0, m_begin, m_end)) continue;
#pragma unroll
            for (int i = 0; i < TNT; ++i)
#pragma unroll
                for (int j = 0; j < TKT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int nslab = (m_end - m_begin + 31) >> 5;
            for (int s = 0; s < nslab; ++s) slab();
            // ---- combine.  With a workspace (TnProb.partial): the accumulators leave AS THEY ARE -- 16-byte stores, a wave instruction covers
            // 1 KB of contiguous memory -- into this item's slot [wave][nt][kt][lane][4] and tn_mw_reduce_kernel adds the splits of a tile in
            // split order (fixed order: run-to-run reproducible) and scatters the sum into dW / dbias.  Round 5: the float-atomic combine below
            // was 28 of the 116 us of a GRU layer's launch (tools/tn_mw_ablate.py: every item finishes at the same time and 7.4 M atomics meet
            // in the L2s with nothing left to overlap them).  Without a workspace: float atomics straight from the accumulators; column K of
            // a biased problem is dbias.
            const TnProb& pr = g.p[pi];
            float* __restrict__ partial = pr.partial;
            if (partial) {
                if constexpr (!(ABL & 16)) {
                    float* pt = partial + (long)lid_last * (BN * BK) + (long)wave * (TNT * TKT * 256) + lane * 4;
#pragma unroll
                    for (int nt = 0; nt < TNT; ++nt)
#pragma unroll
                        for (int kt = 0; kt < TKT; ++kt) *reinterpret_cast<f32x4*>(pt + (nt * TKT + kt) * 256) = acc[nt][kt];
                }
                continue;
            }
            const int N = pr.N, K = pr.A.K, cw = pr.A.cw, out_kw = pr.out_kw;
            float* __restrict__ dW = pr.dW;
            float* __restrict__ dbias = pr.dbias;
            const long ldw = pr.ldw;
#pragma unroll
            for (int nt = 0; nt < TNT; ++nt)
#pragma unroll
                for (int kt = 0; kt < TKT; ++kt) {
                    const int kcol = k0 + wk * (16 * TKT) + kt * 16 + r16;
                    const long off = out_kw > 0 ? (long)(kcol % cw) * out_kw + kcol / cw : (long)kcol;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int nrow = n0 + wn * (16 * TNT) + nt * 16 + kq * 4 + i;
                        if (nrow >= N) continue;
                        if constexpr (ABL & 16) { if (acc[nt][kt][i] == 1.2345e-30f) dW[0] = 1.f; continue; }
                        float v = acc[nt][kt][i];
                        if constexpr (NS == 2) {
                            const float iy = h2_inv_of_exp(h2_exp_of_bits(__float_as_uint(pr.y_cmax[nrow])));
                            const float ia = kcol < K ? h2_inv_of_exp(h2_exp_of_bits(__float_as_uint(pr.a_cmax[kcol % cw]))) : 1.f;
                            v = (v * iy) * ia;
                        }
                        if (kcol < K) atomicAdd(&dW[(long)nrow * ldw + off], v);
                        else if (kcol == K && dbias) atomicAdd(&dbias[nrow], v);
                    }
                }
        }
    }
}

// Second pass of the workspace combine: one thread per 16-byte slot of a TILE's register image sums that slot over the tile's row splits in
// split order (fp64, one rounding) and adds the four values into dW (rows 4 kq .. 4 kq + 3 of the MFMA tile, column r16: 64-byte segments per
// 16 lanes) -- every output element has exactly one writer, so plain read-modify-writes.  Column K of a biased problem goes to dbias.
struct TnMwSplits { int s[TG_MAX_GROUP]; int tile_begin[TG_MAX_GROUP + 1]; };
template <int TNT, int TKT, int WNW, int WKW>
__global__ __launch_bounds__(256) void tn_mw_reduce_kernel(const TnGroup g, const TnMwSplits sp) {
    constexpr int BN = 16 * TNT * WNW, BK = 16 * TKT * WKW, SLOTS = BN * BK / 4, BPT = SLOTS / 256;      // 16-byte slots / workgroups per tile
    static_assert(SLOTS % 256 == 0, "whole workgroups per tile");
    const int tile = blockIdx.x / BPT, slot = (blockIdx.x % BPT) * 256 + threadIdx.x;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < TG_MAX_GROUP; ++q) pi += (q < g.n && tile >= sp.tile_begin[q]) ? 1 : 0;
    const TnProb& pr = g.p[pi];
    const int tl = tile - sp.tile_begin[pi], n_tiles = pr.n_nt * pr.n_kt;
    const int tn_n = tl % pr.n_nt, tn_k = tl / pr.n_nt;
    const int wave = slot / (TNT * TKT * 64), rem = slot % (TNT * TKT * 64);
    const int nt = rem / (TKT * 64), kt = (rem / 64) % TKT, lane = rem % 64;
    const int wn = wave / WKW, wk = wave % WKW, r16 = lane & 15, kq = lane >> 4;
    const float* __restrict__ src = pr.partial + (long)tl * (BN * BK) + (long)slot * 4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int q = 0; q < sp.s[pi]; ++q) {                       // item (tile, split q) has index tl + n_tiles * q inside its problem
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)q * n_tiles * (BN * BK));
        a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
    }
    float r[4] = {(float)a0, (float)a1, (float)a2, (float)a3};
    const int kcol = tn_k * BK + wk * (16 * TKT) + kt * 16 + r16;
    const int K = pr.A.K, cw = pr.A.cw, out_kw = pr.out_kw;
    if (pr.y_cmax) {                                           // fp16 x 2: back through the columns' exact power-of-two scales
        const float ia = kcol < K ? h2_inv_of_exp(h2_exp_of_bits(__float_as_uint(pr.a_cmax[kcol % cw]))) : 1.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nrow = tn_n * BN + wn * (16 * TNT) + nt * 16 + kq * 4 + i;
            const float iy = h2_inv_of_exp(h2_exp_of_bits(__float_as_uint(pr.y_cmax[nrow < pr.N ? nrow : 0])));
            r[i] = (r[i] * iy) * ia;
        }
    }
    const long off = out_kw > 0 ? (long)(kcol % cw) * out_kw + kcol / cw : (long)kcol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nrow = tn_n * BN + wn * (16 * TNT) + nt * 16 + kq * 4 + i;
        if (nrow >= pr.N) continue;
        if (kcol < K) pr.dW[(long)nrow * pr.ldw + off] += r[i];
        else if (kcol == K && pr.dbias) pr.dbias[nrow] += r[i];
    }
}

}  // namespace tg

using namespace tg;

static int g_tn_wg_cap = 0;
extern "C" int tg_set_tn_workgroup_cap(int32_t n) { g_tn_wg_cap = n > 0 ? n : 0; return 0; }
extern "C" int tg_get_tn_workgroup_cap(void) { return g_tn_wg_cap; }

// Plan for the mover-wave kernel, or false when the group should stay on gemm_tn_split_kernel.  Fills n_nt / n_kt / rows_per_split / extents
// of every problem and the workgroup ranges; *grid receives the launch size.
bool tg_gemm_tn_mw_plan(TnGroup& g, int* splits_out, int* grid, const long* ws_floats) {
    static const int env_on = [] { const char* e = getenv("TG_TN_MW"); return e ? atoi(e) : 1; }();
    if (!env_on) return false;
    constexpr int BN = 192, BK = 160;
    long tiles = 0;
    const int min_rows = g_tn_wg_cap > 0 ? 1024 : 2048;          // (a capped launch is sized to run beside another kernel, not to fill the chip)
    for (int i = 0; i < g.n; ++i) {
        TnProb& p = g.p[i];
        const Win& A = p.A;
        constexpr int min_k = 100;      // K = 108: the GRU's first layer (68 % of a 160-wide tile)
        if (!p.vec_y || !p.vec_a || p.N % 4 != 0 || A.K % 4 != 0 || p.N < 150 || A.K < min_k || p.M < min_rows) return false;
        if (p.dbias && A.K % BK == 0) return false;                               // no padding column for the ones trick
        if ((p.y_cmax != nullptr) != (g.p[0].y_cmax != nullptr)) return false;      // one operand format per launch
        if (A.bs < 0 || A.rs < 0 || A.rows_out < 32 || A.step < 0 || p.ldy < p.N) return false;           // (rows_out >= 32: one batch wrap per slab at most in the movers' row walk)
        const long batches = cdiv(p.M, A.rows_out);
        const long a_el = (batches - 1) * A.bs + (long)(A.rows_in - 1) * A.rs + A.cw;
        const long y_el = (long)(p.M - 1) * p.ldy + p.N;
        if (a_el <= 0 || a_el >= (1l << 29) || y_el >= (1l << 29) || (long)A.rows_in * (A.step > 0 ? A.step : 1) >= (1l << 23)) return false;     // (rows: the movers' 24-bit multiply)
        p.a_bytes = (unsigned)(a_el * 4);
        p.y_bytes = (unsigned)(y_el * 4);
        tiles += (long)cdiv(p.N, BN) * cdiv(A.K, BK);
    }
    static const int dev_cu = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus >= 8 ? cus / 8 * 8 : 8;
    }();
    // tg_set_tn_workgroup_cap: plan for (and occupy) at most that many CUs -- a launch meant to run BESIDE another kernel that needs the rest
    const int n_cu = (g_tn_wg_cap > 0 && g_tn_wg_cap < dev_cu) ? (g_tn_wg_cap >= 8 ? g_tn_wg_cap / 8 * 8 : 8) : dev_cu;
    // one item per CU for the group as a whole: row splits = CUs / tiles (every split costs one float atomic per output element), at
    // least 256 rows each
    long items = 0;
    for (int i = 0; i < g.n; ++i) {
        const TnProb& p = g.p[i];
        int s = (int)(n_cu / tiles);
        if (s < 1) s = 1;
        const int cap = cdiv(p.M, 256);
        if (s > cap) s = cap;
        items += (long)cdiv(p.N, BN) * cdiv(p.A.K, BK) * s;
    }
    if (items < n_cu / 2) return false;                                           // would leave most of the chip idle
    int wg = 0;
    for (int i = 0; i < g.n; ++i) {
        TnProb& p = g.p[i];
        int s = (int)(n_cu / tiles);
        if (s < 1) s = 1;
        const int cap = cdiv(p.M, 256);
        if (s > cap) s = cap;
        int rows = cdiv(p.M, s);
        rows = (rows + 31) / 32 * 32;
        p.rows_per_split = rows;
        splits_out[i] = cdiv(p.M, rows);
        p.n_nt = cdiv(p.N, BN); p.n_kt = cdiv(p.A.K, BK);
        // workspace combine: every item of the problem owns one tile-sized slot
        if (p.partial && ws_floats[i] < (long)p.n_nt * p.n_kt * splits_out[i] * (BN * BK)) return false;
        g.wg_begin[i] = wg;
        wg += (p.n_nt * p.n_kt * splits_out[i] + 7) / 8 * 8;
    }
    for (int i = g.n; i <= TG_MAX_GROUP; ++i) g.wg_begin[i] = wg;
    *grid = wg < n_cu ? wg : n_cu;
    return true;
}

int tg_gemm_tn_mw_reduce_launch(const TnGroup& g, const int* splits, hipStream_t s) {
    TnMwSplits sp;
    int tiles = 0;
    for (int i = 0; i < TG_MAX_GROUP; ++i) {
        sp.s[i] = i < g.n ? splits[i] : 0;
        sp.tile_begin[i] = tiles;
        if (i < g.n) tiles += g.p[i].n_nt * g.p[i].n_kt;
    }
    sp.tile_begin[TG_MAX_GROUP] = tiles;
    constexpr int BPT = 192 * 160 / 4 / 256;
    hipLaunchKernelGGL((tn_mw_reduce_kernel<3, 5, 4, 2>), dim3(tiles * BPT), dim3(256), 0, s, g, sp);
    return check_launch("tg_gemm_tn(mover waves, workspace combine)");
}

int tg_gemm_tn_mw_launch(const TnGroup& g, int grid, int splits, hipStream_t s) {
#ifdef TG_LAB_ABLATE
    {
        const char* e = getenv("TG_TNMW_ABL");
        const int abl = e ? atoi(e) : 0;
        if (abl && splits == 3) {
#define TG_ABL(A_) case A_: if (g.p[0].y_cmax) hipLaunchKernelGGL((gemm_tn_mw_kernel<3, 5, 4, 2, 2, A_>), dim3(grid), dim3(768), 0, s, g); \
                            else hipLaunchKernelGGL((gemm_tn_mw_kernel<3, 5, 4, 2, 3, A_>), dim3(grid), dim3(768), 0, s, g); return check_launch("tg_gemm_tn(mover waves, ablated)")
            switch (abl) { TG_ABL(1); TG_ABL(2); TG_ABL(4); TG_ABL(6); TG_ABL(8); TG_ABL(9); TG_ABL(16); TG_ABL(14); TG_ABL(15); default: break; }
#undef TG_ABL
        }
    }
#endif
    if (splits == 3 && g.p[0].y_cmax) hipLaunchKernelGGL((gemm_tn_mw_kernel<3, 5, 4, 2, 2>), dim3(grid), dim3(768), 0, s, g);
    else if (splits == 3) hipLaunchKernelGGL((gemm_tn_mw_kernel<3, 5, 4, 2, 3>), dim3(grid), dim3(768), 0, s, g);
    else hipLaunchKernelGGL((gemm_tn_mw_kernel<3, 5, 4, 2, 1>), dim3(grid), dim3(768), 0, s, g);
    return check_launch("tg_gemm_tn(mover waves)");
}
