// tg_gemm_nt, big products with many rows: fp32-accurate bf16 x 3 (or plain bf16) product with MOVER WAVES.
//
// gemm_split.hip's kernels stage a slab, barrier, multiply, barrier: their own ablation (profiles/r2_nt_ablate.txt) shows the matrix
// section ADDING its time to the staging skeleton (182 = 111 + 71 us on [13056 x 1800 x 600]) -- every wave alternates between a
// load / split / LDS-store phase and a 72-MFMA phase, and the three resident workgroups of a CU fall into step.  Here the two jobs
// belong to different waves of one PERSISTENT 512-thread workgroup per CU (what the H = 64 recurrence's mover waves proved, gru_h64.hip):
//   * waves 4-7 (movers, one per SIMD) own all operand traffic.  The ACTIVATION operand arrives as fp32: they fetch its slabs three
//     slabs ahead into two register sets, split every value exactly into its three bf16 terms and store the planes of slab n + 1 into
//     a double-buffered LDS image.  The WEIGHT operand arrives PRE-SPLIT (bf16 x 3 planes in the slab-tiled layout of common.hpp,
//     written once per optimiser step by layers.WeightPrep / once per forward for the weight-normed convs): it goes global -> LDS by
//     DMA (buffer_load_dwordx4 ... lds, 1 KB = 16 rows x 64 B per instruction, no registers, no arithmetic) into a ring of three slots,
//     two slabs ahead.
//   * waves 0-3 (one per SIMD) read their fragments of slab n from LDS and issue nothing but ds_read_b128 and MFMAs: 6 TM TN
//     v_mfma_f32_16x16x32_bf16 per slab and wave (144 for the 128 x 192 tile = 2304 cycles of matrix pipe), software-pipelined by hand:
//     the fragments of column tile j + 1 are in flight during the 24 MFMAs of column j, and the next slab's first fragments during the
//     last column's -- the wave reaches the slab's barrier five sixths of the way through its MFMAs, the moment its last read is back.
// One workgroup barrier per slab.  Why the weights are pre-split: with both operands fp32 the movers ran ~300 vector instructions per
// slab on SIMDs whose issue ports the matrix waves hold half of the time, and both roles waited for each other (4 500 cycles per slab
// against 2 304 of matrix pipe, profiles/r3_f_mw_roles.txt); the weight operand is 60 % of a 128 x 192 slab.  Why not both: all-DMA
// operands need 58 GB/s per CU from L2 at the matrix pipe's pace, at the ceiling of what a CU takes in (profiles/r3_h_nt_mw_probe.txt:
// 150 us against 165).
// The matrix waves write a finished tile straight from their accumulators (product taken transposed, C^T = W . X^T, so a lane holds four
// consecutive output columns of one row: 16-byte stores) and start the next tile while those stores drain.
//
// Addressing of the activation: every piece is a bounds-checked BUFFER load (raw_buffer_load_b128) whose voffset is either the piece's
// byte offset or a value past num_records -- padding rows of a conv window, rows past M and the K tail all read as zero with no select
// and no predicated load (static vector-memory counts), so one instantiation serves plain and padded windows.
// Weight rows past N read the plane buffer's zero row.
// Precondition (host): the activation is addressable with 31-bit byte offsets, pieces never straddle a tap (cw % 4 == 0), ONE weight
// matrix [N][K] with planes (no K-concatenated segments), C on the vectorisable layout.  Everything else stays on gemm_split.hip.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace tg {

constexpr unsigned MW_RSRC3 = 0x00020000u;
constexpr unsigned MW_OOB = 0x80000000u;            // voffset of a piece that must read as zero (>= num_records: extents are < 2^31)
typedef __attribute__((address_space(3))) void mw_lds_void;

template <int NS>
__device__ __forceinline__ void mw_split4(const f32x4 v, const float scale, u32x2 (&out)[NS]) {
    if constexpr (NS == 2) {                         // fp16 x 2 (common.hpp): hi / lo of the scaled values
        const float x0 = v[0] * scale, x1 = v[1] * scale, x2 = v[2] * scale, x3 = v[3] * scale;
        unsigned h0, l0, h1, l1;
        h2_split2(x0, x1, h0, l0);
        h2_split2(x2, x3, h1, l1);
        out[0] = u32x2{h0, h1};
        out[1] = u32x2{l0, l1};
    } else if constexpr (NS == 1) {                         // plain bf16 tier: round to nearest even
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 r;
        r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
        out[0] = __builtin_bit_cast(u32x2, r);
    } else {
        static_assert(NS == 3, "1, 2 or 3 terms");
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

__device__ __forceinline__ int mw_swz(int row) { return ((row >> 3) & 1) << 4; }      // as gemm_split.hip: XOR for a bf16 column index

// a buffer descriptor whose every input is PROVABLY wave-uniform to the compiler (cdna_hip_programming.md T20: otherwise each buffer
// operation is wrapped in a readfirstlane / saveexec "waterfall" loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), MW_RSRC3);
}
// one DMA: 64 lanes x 16 bytes from the buffer (per-lane byte offset `voff`, uniform `soff`) to 1 KB of LDS at byte offset `lds_off` of smem
// (the address-space cast only exists in the device pass: the host pass of this template would otherwise drop the kernel's launch stub)
__device__ __forceinline__ void mw_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* smem, unsigned lds_off, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (mw_lds_void*)(smem + lds_off), 16, voff, soff, 0, 0);
#endif
}

#ifdef TG_LAB_ABLATE
__device__ unsigned long long mw_role_cycles[512 * 24];      // [workgroup][wave 0..11][waited, lifetime]
#endif

// Workgroup = 12 waves: matrix waves 0-7 as WMW x WNW (two per SIMD: ONE wave issuing back-to-back MFMAs gets ~19-21 cycles per
// v_mfma_f32_16x16x32_bf16, two share the pipe at its 16, profiles/r2_mfma_rate.txt), wave tile (16 TM) x (16 TN); mover waves 8-11, one per SIMD.
// Workgroup tile (16 TM WMW) x (16 TN WNW).  NS = 3: bf16 x 3; NS = 1: plain bf16 operands (the activation rounded to nearest, the weights' hi
// plane).
// ABL (lab builds only, -DTG_LAB_ABLATE, tools/mw_ablate.py; results are WRONG by construction): bit 0 drops the MFMAs, bit 1 the movers'
// split arithmetic + LDS stores, bit 2 the activation's global loads, bit 3 the weight DMAs, bit 4 the epilogue's global traffic
template <int TM, int TN, int WMW, int WNW, int NS, int ABL = 0>
__global__ __launch_bounds__(768, 3) void gemm_nt_mw_kernel(const NtGroup g) {
    static_assert(WMW * WNW == 8, "eight matrix waves");
    constexpr int BM = 16 * TM * WMW, BN = 16 * TN * WNW;
    constexpr int NPA = BM / 32;                                            // 16-byte activation pieces per mover thread and slab
    static_assert(BM % 32 == 0 && BN % 16 == 0, "tile shape");
    constexpr int A_PLANE = BM * 64, A_BUF = NS * A_PLANE;                // bytes
    constexpr int B_PLANE = BN * 64, B_BUF = NS * B_PLANE;
    constexpr int B_BASE = 2 * A_BUF;                                      // [A buffer 0 | A buffer 1 | B slot 0 | B slot 1 | B slot 2]
    constexpr int GB = BN / 16, MAXGB = (GB + 3) / 4;                      // 16-row DMA groups of the weight tile; per mover wave
    constexpr int DUMMY_OFF = 2 * A_BUF + 3 * B_BUF;                       // 1 KB behind the ring: target of the DMAs a wave issues for groups it does not own
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_BUF + 3 * B_BUF + (GB % 4 ? 1024 : 0)];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int total_tiles = g.wg_begin[TG_MAX_GROUP];         // incl. the padding ids that round every problem's range up to a multiple of 8
    const int G = gridDim.x;

    // tile `vb` of the launch -> problem, origin, slab count; false for a padding id
    auto decode = [&](int vb, int& pi, int& m0, int& n0, int& nslab) __attribute__((always_inline)) -> bool {
        pi = group_find(g, vb);
        const NtProb& pr = g.p[pi];
        const int lid = xcd_chunked_id(vb - g.wg_begin[pi], g.wg_begin[pi + 1] - g.wg_begin[pi]);
        m0 = (lid / pr.n_nt) * BM;
        n0 = (lid % pr.n_nt) * BN;
        nslab = (pr.A.K + 31) >> 5;
        return m0 < pr.M;
    };
    // slabs this workgroup walks in total (both roles count the same barriers)
    int total = 0;
    for (int vb = blockIdx.x; vb < total_tiles; vb += G) {
        int pi, m0, n0, ns;
        if (decode(vb, pi, m0, n0, ns)) total += ns;
    }

#ifdef TG_LAB_ABLATE
    // lab: shader-clock cycles each role spends INSIDE its barriers (arrival -> release) against its whole life: the role that waits is not
    // the one that sets the pace.  mw_role_cycles[workgroup][role][0 = waited, 1 = total]
    unsigned long long waited = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#define MW_TIMED_BARRIER() do { const unsigned long long tb = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_barrier(); waited += __builtin_amdgcn_s_memtime() - tb; } while (0)
#else
#define MW_TIMED_BARRIER() __builtin_amdgcn_s_barrier()
#endif
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;

    if (wave >= 8) {
        // ============================================================================================ movers (waves 8-11)
        const int mw = wave - 8;
        // ---- activation: thread mt owns the 16-byte piece (mt & 7) of slab rows (mt >> 3) + 32 q: 8 consecutive lanes cover one 128-byte
        // row piece.  Its cursor (tile vb_a, slab s_a) runs three slabs ahead of the slab the matrix waves multiply.
        const int mt = t - 512;
        const int sp = 4 * (mt & 7), sr0 = mt >> 3;
        const int sp_wb = (sp ^ mw_swz(sr0)) * 2;                              // byte offset of the piece's 8-byte slot inside its 64-byte row
        int vb_a = blockIdx.x - G, s_a = 0, nslab_a = 0;
        bool live_a = true;                                    // false once the cursor has run past this workgroup's last tile
        unsigned a_boff[NPA];
        int a_r[NPA];
        bool a_ok[NPA];
        float a_s[NPA];                                        // fp16 x 2: the row's power-of-two scale
        unsigned rs4 = 0;
        int kk = 0, c = 0, kcur = 0, K_a = 0, a_cw = 4, a_dil = 0, a_rows_in = 0;
        const float* a_ptr = g.p[0].A.ptr;
        unsigned a_bytes = 0;
        auto next_tile_a = [&]() __attribute__((always_inline)) {
            int pi = 0, m0 = 0, n0 = 0;
            do {
                vb_a += G;
                if (vb_a >= total_tiles) { live_a = false; break; }
            } while (!decode(vb_a, pi, m0, n0, nslab_a));
            s_a = 0;
            if (!live_a) {
#pragma unroll
                for (int q = 0; q < NPA; ++q) a_ok[q] = false;
                return;
            }
            const NtProb& pr = g.p[pi];
            const Win A = pr.A;
            const float* __restrict__ a_scale = pr.a_scale;
            const float* __restrict__ a_rmax = pr.a_rmax;      // fp16 x 2, second form: magnitudes of the SOURCE rows (one or two taps): the scale is derived here
            const int two_taps = A.K > A.cw;
            const int rm_div = pr.a_rmax_div;                  // source rows per a_rmax entry (1: one entry per row; T: one per clip of T rows)
            const int M_a = pr.M;
#pragma unroll
            for (int q = 0; q < NPA; ++q) {
                const int m = m0 + sr0 + 32 * q;
                a_ok[q] = m < M_a;
                const int mm = a_ok[q] ? m : 0;
                const int b = mm / A.rows_out;
                a_boff[q] = (unsigned)(((long)b * A.bs) * 4);
                a_r[q] = (mm - b * A.rows_out) * A.step + A.shift;
                if constexpr (NS == 2) {
                    // (branch-free: two loads from always-valid addresses, then selects)
                    const float* __restrict__ rm = a_rmax ? a_rmax : a_scale;
                    const int s0 = a_r[q], s1 = s0 + A.dil;
                    const bool ok0 = (unsigned)s0 < (unsigned)A.rows_in, ok1 = two_taps && (unsigned)s1 < (unsigned)A.rows_in;
                    const int base = b * A.rows_in;
                    const unsigned v0 = __float_as_uint(rm[a_rmax ? (base + (ok0 ? s0 : 0)) / rm_div : mm]);
                    const unsigned v1 = __float_as_uint(rm[a_rmax ? (base + (ok1 ? s1 : 0)) / rm_div : mm]);
                    const unsigned vm = (ok0 ? v0 : 0u) > (ok1 ? v1 : 0u) ? (ok0 ? v0 : 0u) : (ok1 ? v1 : 0u);
                    a_s[q] = a_rmax ? h2_scale_of_exp(h2_exp_of_bits(vm)) : __uint_as_float(v0);
                }
            }
            rs4 = (unsigned)(A.rs * 4);
            a_cw = A.cw; a_dil = A.dil; a_rows_in = A.rows_in; K_a = A.K;
            kk = sp / a_cw; c = sp - kk * a_cw;                // tap / channel of this thread's piece, advanced by one slab per fetch
            kcur = sp;
            a_ptr = A.ptr; a_bytes = pr.a_bytes;
        };
        u32x4 ga[2][NPA];
        float ga_s[2][NPA];                                    // fp16 x 2: the scales of the rows a register set holds (the cursor may move on to the next tile
                                                               // -- other rows, other scales -- while the set still waits to be staged)
        auto fetch = [&](auto set_c) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            if constexpr (NS == 2) {
#pragma unroll
                for (int q = 0; q < NPA; ++q) ga_s[set][q] = a_s[q];
            }
            const __amdgpu_buffer_rsrc_t a_rsrc = mw_rsrc(a_ptr, a_bytes);
            const bool inb = kcur < K_a;
            const unsigned c4 = (unsigned)(c * 4);
#pragma unroll
            for (int q = 0; q < NPA; ++q) {
                const int sr = a_r[q] + kk * a_dil;
                const bool ok = a_ok[q] & inb & ((unsigned)sr < (unsigned)a_rows_in);
                if constexpr (ABL & 4) ga[set][q] = u32x4{0x3f800000u + (unsigned)lane, 0x40000000u, 0x3fc00000u + (unsigned)kcur, ok ? 0x3e800000u : 0u};
                else ga[set][q] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, ok ? a_boff[q] + (unsigned)sr * rs4 + c4 : MW_OOB, 0, 0);
            }
            kcur += 32;
            c += 32;
            while (c >= a_cw) { c -= a_cw; ++kk; }
            if (live_a && ++s_a >= nslab_a) next_tile_a();     // (wave-uniform)
        };
        auto stage = [&](auto set_c, int buf) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            unsigned char* const lb = smem + buf * A_BUF;
            if constexpr (ABL & 2) {                              // loaded values stay live (the loads must still be waited for), nothing else
#pragma unroll
                for (int q = 0; q < NPA; ++q) asm volatile("" :: "v"(ga[set][q]));
                return;
            }
#pragma unroll
            for (int q = 0; q < NPA; ++q) {
                u32x2 o[NS];
                mw_split4<NS>(__builtin_bit_cast(f32x4, ga[set][q]), NS == 2 ? ga_s[set][q] : 1.f, o);
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(lb + s * A_PLANE + (sr0 + 32 * q) * 64 + sp_wb) = o[s];
            }
        };
        // ---- weights: this wave moves the 16-row groups mw, mw + 4, mw + 8 of the tile's BN rows, NS planes each, one DMA per (group,
        // plane) and slab.  Its cursor (tile vb_b, slab s_b) runs two slabs ahead.
        const int grow = lane >> 2;                                            // row inside a 16-row group
        const unsigned chunk_b = (unsigned)(((lane & 3) ^ (((grow >> 3) & 1) << 1)) * 16);   // SOURCE chunk of this lane's LDS slot (swizzle on the source)
        int vb_b = blockIdx.x - G, s_b = 0, nslab_b = 0;
        bool live_b = true;
        unsigned b_voff[MAXGB];
        unsigned b_slab_b = 0, b_bytes = 0;                                    // bytes per 32-column slab of a plane; per plane
        const __bf16* b_ptr = g.p[0].Bpl;
        long b_plane = 0;
        auto next_tile_b = [&]() __attribute__((always_inline)) {
            int pi = 0, m0 = 0, n0 = 0;
            do {
                vb_b += G;
                if (vb_b >= total_tiles) { live_b = false; break; }
            } while (!decode(vb_b, pi, m0, n0, nslab_b));
            s_b = 0;
            if (!live_b) {
                // past this workgroup's last tile: the DMAs of the remaining steps are still ISSUED (a fixed number of vector-memory
                // operations per step on every path, see `step`), every lane past num_records: they write zeros into a ring slot nobody reads
#pragma unroll
                for (int u = 0; u < MAXGB; ++u) b_voff[u] = MW_OOB;
                return;
            }
            const NtProb& pr = g.p[pi];
#pragma unroll
            for (int u = 0; u < MAXGB; ++u) {
                const int n = n0 + 16 * (mw + 4 * u) + grow;
                b_voff[u] = (unsigned)(n < pr.N ? pr.b_row0 + n : pr.b_slab_rows - 1) * 64u + chunk_b;      // rows past N: the buffer's zero row
            }
            b_slab_b = (unsigned)pr.b_slab_rows * 64u;
            b_bytes = b_slab_b * (unsigned)((pr.A.K + 31) >> 5);
            b_ptr = pr.Bpl; b_plane = pr.bpl_plane;
        };
        // Every mover wave issues EXACTLY NS * MAXGB DMAs + NPA loads per step on every path.  hipcc counts outstanding vector-memory operations
        // statically; round 3's form (DMAs skipped once the cursor had run out, `continue` for the groups a wave does not own, a `break` in
        // the middle of the two-step loop body) left it with a lower bound of zero DMAs between a fetch and its use, and the ISA waited
        // vmcnt(7 .. 4) at the top of every step -- i.e. for the DMAs and loads issued a few hundred cycles earlier -- plus a full
        // vmcnt(0) behind every other barrier: a memory round trip per slab on the movers' path, whatever the tile (the "per-slab floor"
        // of DESIGN.md section 9.2).  Order inside a step: DMAs FIRST (their slot was released by the barrier just passed: the longest
        // possible time to land), then stage, then fetch.
        {
            constexpr int NG = MAXGB;
            constexpr int ND = NS * NG;                                        // DMAs per step of every mover wave
            auto dma = [&](int slot) __attribute__((always_inline)) {                                         // the cursor's slab -> ring slot `slot`
                if constexpr (!(ABL & 8)) {
                    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(s_b * (int)b_slab_b);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const __amdgpu_buffer_rsrc_t rs = mw_rsrc(b_ptr + s * b_plane, b_bytes);
#pragma unroll
                        for (int u = 0; u < NG; ++u) {
                            // a group past the tile's GB (BN = 160: waves 2, 3 own two groups, not three) is still issued -- out of range, into
                            // the spare KB behind the ring -- so that every wave's step has the same static count
                            const int grp = mw + 4 * u;
                            const bool own = (GB % 4 == 0) || grp < GB;
                            mw_dma16(rs, smem, own ? (unsigned)(B_BASE + slot * B_BUF + s * B_PLANE + grp * 1024) : (unsigned)DUMMY_OFF,
                                     own ? b_voff[u] : MW_OOB, soff);
                        }
                    }
                }
                if (live_b && ++s_b >= nslab_b) next_tile_b();
            };
            next_tile_a();
            next_tile_b();
            fetch(set0{});                                         // slab 0
            fetch(set1{});                                         // slab 1 (past this workgroup's work: every piece out of range, zeros, no traffic)
            dma(0);                                                // slabs 0, 1 of the weights
            dma(1);
            stage(set0{}, 0);
            fetch(set0{});                                         // slab 2
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            MW_TIMED_BARRIER();
            asm volatile("" ::: "memory");
            // step n: the matrix waves multiply slab n (activation buffer n & 1, weight slot n % 3).  Slab n + 2 of the weights is sent to the
            // slot slab n - 1 has just left, slab n + 1 of the activation is staged into the other buffer (read last during step n - 1), slab
            // n + 3 fetched into the register set just emptied; before the step's barrier the weights of slab n + 1 (sent one step ago)
            // must have landed: everything issued in THIS step -- ND DMAs, then NPA loads -- may stay in flight.
            int n = 0, slot2 = 2;                                  // slot2 = (n + 2) % 3
            auto step = [&](auto set_c, int abuf) __attribute__((always_inline)) {
                dma(slot2);
                if (n + 1 < total) stage(set_c, abuf);
                fetch(set_c);
                if constexpr (ABL & 32) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPA + 2 * ND) : "memory");       // lab: the DMAs get one more step (results wrong)
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPA + ND) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MW_TIMED_BARRIER();
                asm volatile("" ::: "memory");
                ++n;
                slot2 = slot2 == 2 ? 0 : slot2 + 1;
            };
            while (n + 2 <= total) {
                step(set1{}, 1);
                step(set0{}, 0);
            }
            if (n < total) step(set1{}, 1);
        }
    } else {
        // ============================================================================================ matrix waves (0-7)
        const int wm = wave / WNW, wn = wave % WNW;
        const int r16 = lane & 15, kq = lane >> 4;
        const int fa_off = (wm * (16 * TM) + r16) * 64 + ((8 * kq) ^ mw_swz(r16)) * 2;                 // byte offset of the lane's first A fragment
        const int fb_off = B_BASE + (wn * (16 * TN) + r16) * 64 + ((8 * kq) ^ mw_swz(r16)) * 2;        // ... and first B fragment (slot 0)
        f32x4 acc[TM][TN];
        auto load_fb = [&](bf16x8 (&fb)[NS], int slot_, int j) {
#pragma unroll
            for (int s = 0; s < NS; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(smem + slot_ * B_BUF + s * B_PLANE + j * 1024 + fb_off);
        };
        // acc[i][j] += (W tile j) . (X tile i)^T for row tiles [I0, I1): the weight fragment is the MFMA's A operand, so the lane's four accumulator
        // values are output row (i, r16), columns (j, 4 kq .. 4 kq + 3)
        bf16x8 fa[NS][TM];
        auto load_fa_row = [&](int abuf, auto i_c) {
            constexpr int i = decltype(i_c)::value;
#pragma unroll
            for (int s = 0; s < NS; ++s) fa[s][i] = *reinterpret_cast<const bf16x8*>(smem + abuf * A_BUF + s * A_PLANE + i * 1024 + fa_off);
        };
        auto mma_rows = [&](const bf16x8 (&fb)[NS], auto j_c, auto i0_c, auto i1_c) {
            constexpr int j = decltype(j_c)::value, I0 = decltype(i0_c)::value, I1 = decltype(i1_c)::value;
            if constexpr (ABL & 1) {                   // fragments stay live, no matrix instruction
#pragma unroll
                for (int i = I0; i < I1; ++i) asm volatile("" :: "v"(fb[0]), "v"(fb[NS - 1]), "v"(fa[0][i]), "v"(fa[NS - 1][i]));
                return;
            }
            // term-major over the row tiles: the MFMAs that accumulate into one tile are I1 - I0 issues apart (smallest terms first)
            if constexpr (NS == 2) {                   // fp16 x 2: lo_w hi_x + hi_w lo_x + hi_w hi_x
                auto h = [](const bf16x8& v) { return __builtin_bit_cast(tg_f16x8, v); };
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(fb[1]), h(fa[0][i]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(fb[0]), h(fa[1][i]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(fb[0]), h(fa[0][i]), acc[i][j], 0, 0, 0);
                return;
            }
            if constexpr (NS == 3) {
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[2][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[0][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[0][i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = I0; i < I1; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0][i], acc[i][j], 0, 0, 0);
        };
        int n = 0, slot = 0;                                   // global slab counter; slot = n % 3
        bf16x8 fbr[2][NS];
        // one slab: fa holds its A fragments, fbr[0] its first B fragment.  The fragments of column j + 1 are read during the MFMAs of column
        // j; before the last column every read of this slab is back -> barrier -> the NEXT slab's first B fragment goes out at once and its A
        // fragments row tile by row tile, each into the registers the last column's MFMAs of that row tile have just read.  The next slab is
        // simply slab n + 1 of this workgroup's sequence (also across a tile boundary; after the very last slab the reads fetch bytes
        // nobody uses), so the loop body has no conditional loads.  An odd TN reloads column 0 at the slab's start instead (fbr parity).
        auto slab = [&]() {
            const int nslot = slot == 2 ? 0 : slot + 1;
            constexpr bool EVEN = (TN & 1) == 0;
            if constexpr (!EVEN) load_fb(fbr[0], slot, 0);
            auto col = [&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                constexpr int cb = j & 1;
                if constexpr (j + 1 < TN) {
                    load_fb(fbr[cb ^ 1], slot, j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_rows(fbr[cb], j_c, std::integral_constant<int, 0>{}, std::integral_constant<int, TM>{});
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    MW_TIMED_BARRIER();
                    asm volatile("" ::: "memory");
                    if constexpr (EVEN) load_fb(fbr[cb ^ 1], nslot, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int nbuf = (n + 1) & 1;
                    auto row = [&](auto i_c) {
                        constexpr int i = decltype(i_c)::value;
                        mma_rows(fbr[cb], j_c, i_c, std::integral_constant<int, i + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                        load_fa_row(nbuf, i_c);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    row(std::integral_constant<int, 0>{});
                    if constexpr (TM > 1) row(std::integral_constant<int, 1>{});
                    if constexpr (TM > 2) row(std::integral_constant<int, 2>{});
                    if constexpr (TM > 3) row(std::integral_constant<int, 3>{});
                    static_assert(TM <= 4, "row tiles per wave");
                }
            };
            col(std::integral_constant<int, 0>{});
            if constexpr (TN > 1) col(std::integral_constant<int, 1>{});
            if constexpr (TN > 2) col(std::integral_constant<int, 2>{});
            if constexpr (TN > 3) col(std::integral_constant<int, 3>{});
            if constexpr (TN > 4) col(std::integral_constant<int, 4>{});
            if constexpr (TN > 5) col(std::integral_constant<int, 5>{});
            static_assert(TN <= 6, "column tiles per wave");
            ++n;
            slot = nslot;
        };
        asm volatile("" ::: "memory");
        MW_TIMED_BARRIER();
        asm volatile("" ::: "memory");
        // slab 0's first fragments (the loop prefetches every later slab's)
        load_fa_row(0, std::integral_constant<int, 0>{});
        if constexpr (TM > 1) load_fa_row(0, std::integral_constant<int, 1>{});
        if constexpr (TM > 2) load_fa_row(0, std::integral_constant<int, 2>{});
        if constexpr (TM > 3) load_fa_row(0, std::integral_constant<int, 3>{});
        load_fb(fbr[0], 0, 0);
        for (int vb = blockIdx.x; vb < total_tiles; vb += G) {
            int pi, m0, n0, nslab;
            if (!decode(vb, pi, m0, n0, nslab)) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < nslab; ++s) slab();
            // ---- epilogue straight from the accumulators: bias, activation, dropout scale, gate, accumulate, second output; every access a
            // 16-byte piece of one output row (an instruction covers 16 rows x 64 bytes).  The stores drain while the next tile runs.
            const NtProb& pr = g.p[pi];
            const float* __restrict__ bias = pr.bias;
            const float* __restrict__ mul = pr.mul;
            const uint64_t* __restrict__ drop = pr.drop_state;
            const float* __restrict__ gate = pr.gate;
            const float* __restrict__ res = pr.res;
            float* __restrict__ C2 = pr.C2;
            float* __restrict__ C = pr.C;
            const float slope = pr.slope, slope2 = pr.res_slope;
            const int cR = pr.cR, accumulate = pr.accumulate, M = pr.M, N = pr.N;
            long ro[TM];
            bool rok[TM];
            float inv_a[TM];                                   // fp16 x 2: 1 / scale of the product row (an exact power of two)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = m0 + wm * (16 * TM) + i * 16 + r16;
                rok[i] = row < M;
                const int rr = rok[i] ? row : 0;
                const int cb = rr / cR;
                ro[i] = (long)cb * pr.cbs + (long)(rr - cb * cR) * pr.crs;
                if constexpr (NS == 2) {
                    if (pr.a_rmax) {
                        const int ab = rr / pr.A.rows_out;
                        const int s0 = (rr - ab * pr.A.rows_out) * pr.A.step + pr.A.shift, s1 = s0 + pr.A.dil;
                        const float* __restrict__ rm = pr.a_rmax;
                        const int base = ab * pr.A.rows_in, dv = pr.a_rmax_div;
                        const unsigned v0 = (unsigned)s0 < (unsigned)pr.A.rows_in ? __float_as_uint(rm[(base + s0) / dv]) : 0u;
                        const unsigned v1 = (pr.A.K > pr.A.cw && (unsigned)s1 < (unsigned)pr.A.rows_in) ? __float_as_uint(rm[(base + s1) / dv]) : 0u;
                        inv_a[i] = h2_inv_of_exp(h2_exp_of_bits(v0 > v1 ? v0 : v1));
                    } else inv_a[i] = h2_inv_of_scale(pr.a_scale[rr]);
                }
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            // optional row magnitudes of what this tile writes (the next product's a_rmax): per lane over its columns here, across the row's four
            // lanes and into memory after the column loop
            float* const c_rmax = pr.c_rmax;
            float* const c2_rmax = pr.c2_rmax;
            unsigned mx1[TM], mx2[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) mx1[i] = mx2[i] = 0u;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + 4 * kq;
                const bool cok = col < N;                      // N % 4 == 0: a piece is inside or outside as a whole
                const int cc0 = cok ? col : 0;
                const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + cc0) : z4;
                f32x4 inv_w = z4;                              // fp16 x 2: 1 / scale of the weight rows (zero past N)
                if constexpr (NS == 2) { if (cok) inv_w = *reinterpret_cast<const f32x4*>(pr.b_inv + pr.b_row0 + cc0); }
                constexpr int RB = TM > 2 ? 2 : TM;            // row tiles per batch: the batch's reads first, from always-valid addresses
#pragma unroll
                for (int i0 = 0; i0 < TM; i0 += RB) {
                    f32x4 mv[RB], gv[RB], rv[RB], cv[RB];
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const long o = ro[i0 + u] + cc0;
                        if (mul) mv[u] = *reinterpret_cast<const f32x4*>(mul + o);
                        else if (drop) mv[u] = dropout_scale4(drop, pr.drop_site, pr.drop_p, (unsigned long)(pr.drop_index0 + o) >> 2);
                        if (gate) gv[u] = *reinterpret_cast<const f32x4*>(gate + o);
                        if (res) rv[u] = *reinterpret_cast<const f32x4*>(res + o);
                        if (accumulate) cv[u] = *reinterpret_cast<const f32x4*>(C + o);
                    }
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int i = i0 + u;
                        const long o = ro[i] + cc0;
                        f32x4 v = acc[i][j];
                        if constexpr (NS == 2) v = (v * inv_a[i]) * inv_w;      // two exact power-of-two steps: the pair's product could leave fp32's range
                        v += bv;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = act_fn(v[q], slope);
                        if (mul || drop) v *= mv[u];
                        if (gate) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = gv[u][q] > 0.f ? v[q] : 0.f;
                        }
                        if (accumulate) v += cv[u];
                        const bool ok = rok[i] & cok;
                        if constexpr (ABL & 16) { if (v[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(C + o) = v; continue; }
                        if (ok) *reinterpret_cast<f32x4*>(C + o) = v;
                        if (c_rmax && cok) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { const float f = v[q]; const unsigned b = __float_as_uint(f) & 0x7fffffffu; mx1[i] = mx1[i] > b ? mx1[i] : b; }
                        }
                        if (res) {
                            f32x4 w = v + rv[u];
#pragma unroll
                            for (int q = 0; q < 4; ++q) w[q] = act_fn(w[q], slope2);
                            if (ok) *reinterpret_cast<f32x4*>(C2 + o) = w;
                            if (c2_rmax && cok) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) { const float f = w[q]; const unsigned b = __float_as_uint(f) & 0x7fffffffu; mx2[i] = mx2[i] > b ? mx2[i] : b; }
                            }
                        }
                    }
                }
            }
            if (c_rmax || c2_rmax) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = m0 + wm * (16 * TM) + i * 16 + r16;
                    if (c_rmax) {
                        unsigned m = mx1[i];
                        unsigned w = (unsigned)__shfl_xor((int)m, 16, 64); m = m > w ? m : w;
                        w = (unsigned)__shfl_xor((int)m, 32, 64); m = m > w ? m : w;
                        if (kq == 0 && rok[i]) atomicMax(reinterpret_cast<unsigned*>(c_rmax) + row, m);
                    }
                    if (c2_rmax) {
                        unsigned m = mx2[i];
                        unsigned w = (unsigned)__shfl_xor((int)m, 16, 64); m = m > w ? m : w;
                        w = (unsigned)__shfl_xor((int)m, 32, 64); m = m > w ? m : w;
                        if (kq == 0 && rok[i]) atomicMax(reinterpret_cast<unsigned*>(c2_rmax) + row, m);
                    }
                }
            }
        }
    }
#ifdef TG_LAB_ABLATE
    if (lane == 0 && blockIdx.x < 512) {
        mw_role_cycles[blockIdx.x * 24 + wave * 2 + 0] = waited;
        mw_role_cycles[blockIdx.x * 24 + wave * 2 + 1] = __builtin_amdgcn_s_memtime() - t_begin;
    }
#endif
#undef MW_TIMED_BARRIER
}

}  // namespace tg

using namespace tg;

#ifdef TG_LAB_ABLATE
extern "C" int tg_lab_mw_role_cycles(void* host_out, int64_t n_words) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mw_role_cycles), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

// byte extents of the two operands as seen from their base pointers (what the buffer descriptors bound); false when an operand is not
// addressable that way (negative strides, >= 2 GB)
static bool mw_extents(NtProb& p) {
    const Win& A = p.A;
    if (A.bs < 0 || A.rs < 0 || A.rows_out <= 0) return false;
    const long batches = cdiv(p.M, A.rows_out);
    const long a_el = (batches - 1) * A.bs + (long)(A.rows_in - 1) * A.rs + A.cw;
    if (a_el <= 0 || a_el >= (1l << 29)) return false;
    p.a_bytes = (unsigned)(a_el * 4);
    p.b_bytes = 0;                           // (the weights are read from their planes: Bw and its segments are never touched here)
    return true;
}

struct MwTile { int tm, tn; };          // workgroup tile in units of 32 rows / columns

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
    // (round 6, measured and not kept: 256 x 192 for two-plane operands -- 96 accumulator registers push the epilogue into 388 spills, 138 us against
    // 122 on the stacked projection: profiles/r6_s_tile_256x192_rejected.txt)
    const MwTile menu[3] = {{4, 6}, {4, 5}, {4, 3}};
    static const int forced = [] { const char* e = getenv("TG_MW_TILE"); return e ? atoi(e) : 0; }();     // lab: 46 / 45 / 43 forces that tile
    double best = 0.0;
    bool found = false;
    for (const MwTile& tl : menu) {
        if (forced && forced != 10 * tl.tm + tl.tn) continue;
        const int bm = 32 * tl.tm, bn = 32 * tl.tn;
        long tiles = 0;
        double work = 0.0;
        for (int i = 0; i < g.n; ++i) {
            const long ti = (long)cdiv(g.p[i].M, bm) * cdiv(g.p[i].N, bn);
            tiles += ti;
            const double per_tile = cdiv(g.p[i].A.K, 32) * (6.0 * tl.tm * tl.tn * 16.0) + 6000.0;        // cycles
            work += ti * per_tile;
        }
        if (tiles < 150 && !forced) continue;
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
        if (!p.vec_c || p.A.cw % 4 != 0 || p.A.K % 4 != 0 || p.Bpl == nullptr) return false;            // (weight segments: the planes hold their concatenation)
        if (!mw_extents(p)) return false;
        if (p.h2 != g.p[0].h2) return false;                   // one operand format per launch
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
    // persistent workgroups, one per CU (120 KB of LDS each); a multiple of 8 so that a workgroup's tiles blockIdx + r * grid keep its XCD residue
    static const int n_cu = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus >= 8 ? cus / 8 * 8 : 8;
    }();
    const dim3 grid(wg < n_cu ? wg : n_cu);
#ifdef TG_LAB_ABLATE
    {
        const char* e = getenv("TG_MW_ABL");
        const int abl = e ? atoi(e) : 0;
        if (abl && tm == 4 && tn == 6 && splits == 3) {
#define TG_ABL(A_) case A_: if (g.p[0].h2) hipLaunchKernelGGL((gemm_nt_mw_kernel<4, 3, 2, 4, 2, A_>), grid, dim3(768), 0, s, g); \
                            else hipLaunchKernelGGL((gemm_nt_mw_kernel<4, 3, 2, 4, 3, A_>), grid, dim3(768), 0, s, g); return check_launch("tg_gemm_nt(mover waves, ablated)")
            switch (abl) { TG_ABL(1); TG_ABL(2); TG_ABL(6); TG_ABL(8); TG_ABL(14); TG_ABL(15); TG_ABL(16); TG_ABL(32); TG_ABL(33); default: break; }
#undef TG_ABL
        }
    }
#endif
    // 128 x 192: matrix waves 2 x 4, wave tile 64 x 48; 128 x 160: 4 x 2, wave tile 32 x 80
#define TG_MW(TM_, TN_, WM_, WN_)                                                                                          \
    do {                                                                                                                   \
        if (g.p[0].h2) hipLaunchKernelGGL((gemm_nt_mw_kernel<TM_, TN_, WM_, WN_, 2>), grid, dim3(768), 0, s, g);          \
        else if (splits == 3) hipLaunchKernelGGL((gemm_nt_mw_kernel<TM_, TN_, WM_, WN_, 3>), grid, dim3(768), 0, s, g);    \
        else hipLaunchKernelGGL((gemm_nt_mw_kernel<TM_, TN_, WM_, WN_, 1>), grid, dim3(768), 0, s, g);                     \
    } while (0)
    if (tm == 4 && tn == 6) TG_MW(4, 3, 2, 4);
    else if (tm == 4 && tn == 5) TG_MW(2, 5, 4, 2);
    else if (tm == 4 && tn == 3) TG_MW(2, 3, 4, 2);             // 128 x 96: matrix waves 4 x 2, wave tile 32 x 48
    else TG_REQUIRE(false, "tg_gemm_nt(mover waves): no %d x %d tile", bm, bn);
#undef TG_MW
    return check_launch("tg_gemm_nt(mover waves)");
}
