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
// One workgroup barrier per slab.  The workgroups are PERSISTENT (one per CU, each walks tiles blockIdx, blockIdx + grid, ...): the
// movers' slab stream runs on across tile boundaries, and the matrix waves write a finished tile straight from their accumulators
// (product taken transposed, C^T = W . X^T, so a lane holds four consecutive output columns of one row: 16-byte stores) and start
// the next tile while those stores drain -- the first, non-persistent form of this kernel spent 10 of every 44 us per tile outside
// the slab loop (first-slab latency 2.5, accumulator tile through LDS 0.8, all 256 CUs writing their tiles in the same 5 us, 1.6
// between workgroups: profiles/r3_d_mw_timeline_0.txt).  A mover shares its SIMD's issue port with one matrix wave: an MFMA holds the port for 8 of its 16
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
__device__ unsigned long long mw_role_cycles[512 * 4];
#endif

__device__ __forceinline__ int mw_swz(int row) { return ((row >> 3) & 1) << 4; }      // as gemm_split.hip: XOR for a bf16 column index

// Workgroup tile (32 TM) x (32 TN); matrix waves 0-3 as 2 x 2, wave tile (16 TM) x (16 TN).
// ABL (lab builds only, -DTG_LAB_ABLATE, tools/mw_ablate.py; results are WRONG by construction): bit 0 drops the MFMAs, bit 1 the movers'
// split arithmetic + LDS stores, bit 2 the global operand loads, bit 4 the epilogue's global traffic
template <int TM, int TN, int NS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_mw_kernel(const NtGroup g) {
    constexpr int BM = 32 * TM, BN = 32 * TN, ROWS = BM + BN;
    constexpr int PLANE = ROWS * 32;                           // bf16 elements of one plane of one slab
    constexpr int BUF = NS * PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF * 2];
    __bf16* const lds = reinterpret_cast<__bf16*>(smem);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int total_tiles = g.wg_begin[TG_MAX_GROUP];         // incl. the padding ids that round every problem's range up to a multiple of 8
    const int G = gridDim.x;

    // tile `vb` of the launch -> problem, origin, slab count; false for a padding id
    auto decode = [&](int vb, int& pi, int& m0, int& n0, int& nslab) -> bool {
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
#define MW_BARRIER_MOVER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); MW_TIMED_BARRIER(); asm volatile("" ::: "memory"); } while (0)
#define MW_BARRIER_MATRIX() do { asm volatile("" ::: "memory"); MW_TIMED_BARRIER(); asm volatile("" ::: "memory"); } while (0)
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;

    if (wave >= 4) {
        // ============================================================================================ movers (waves 4-7)
        // thread mt owns the 16-byte piece (mt & 7) of slab rows (mt >> 3) + 32 q: 8 consecutive lanes cover one 128-byte row piece.
        // The FETCH cursor (tile vb_f, slab s_f) runs three slabs ahead of the slab the matrix waves multiply.
        const int mt = t & 255;
        const int sp = 4 * (mt & 7), sr0 = mt >> 3;
        const int sp_w = sp ^ mw_swz(sr0);
        int vb_f = blockIdx.x - G, s_f = 0, nslab_f = 0;
        bool live = true;                                      // false once the cursor has run past this workgroup's last tile
        unsigned a_boff[TM], b_boff[TN];
        int a_r[TM];
        bool a_ok[TM], b_ok[TN];
        unsigned rs4 = 0, kb = 0, kb_wrap = 0;
        int kk = 0, c = 0, bc = 0, kcur = 0, K_f = 0, a_cw = 4, a_dil = 0, a_rows_in = 0, seg_k = 4;
        __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.p[0].A.ptr), 0, 0, MW_RSRC3);
        __amdgpu_buffer_rsrc_t b_rsrc = a_rsrc;
        auto next_tile = [&]() {
            int pi = 0, m0 = 0, n0 = 0;
            do {
                vb_f += G;
                if (vb_f >= total_tiles) { live = false; break; }
            } while (!decode(vb_f, pi, m0, n0, nslab_f));
            s_f = 0;
            if (!live) {
#pragma unroll
                for (int q = 0; q < TM; ++q) a_ok[q] = false;
#pragma unroll
                for (int q = 0; q < TN; ++q) b_ok[q] = false;
                return;
            }
            const NtProb& pr = g.p[pi];
            const Win A = pr.A;
#pragma unroll
            for (int q = 0; q < TM; ++q) {
                const int m = m0 + sr0 + 32 * q;
                a_ok[q] = m < pr.M;
                const int mm = a_ok[q] ? m : 0;
                const int b = mm / A.rows_out;
                a_boff[q] = (unsigned)(((long)b * A.bs) * 4);
                a_r[q] = (mm - b * A.rows_out) * A.step + A.shift;
            }
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                const int n = n0 + sr0 + 32 * q;
                b_ok[q] = n < pr.N;
                b_boff[q] = (unsigned)(((long)(b_ok[q] ? n : 0) * pr.ldb) * 4);
            }
            rs4 = (unsigned)(A.rs * 4);
            a_cw = A.cw; a_dil = A.dil; a_rows_in = A.rows_in; K_f = A.K; seg_k = pr.b_seg_k;
            // tap / channel of this thread's piece in the A window and weight segment / column in B, advanced by one slab per fetch
            kk = sp / a_cw; c = sp - kk * a_cw;
            const int bsg = sp / seg_k;
            bc = sp - bsg * seg_k;
            kb = (unsigned)(((long)bsg * pr.b_seg_stride + bc) * 4);
            kb_wrap = (unsigned)((pr.b_seg_stride - seg_k) * 4);
            kcur = sp;
            a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.ptr), 0, pr.a_bytes, MW_RSRC3);
            b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.Bw), 0, pr.b_bytes, MW_RSRC3);
        };
        u32x4 ga[2][TM], gb[2][TN];
        auto fetch = [&](auto set_c) {
            constexpr int set = decltype(set_c)::value;
            const bool inb = kcur < K_f;
            const unsigned c4 = (unsigned)(c * 4);
#pragma unroll
            for (int q = 0; q < TM; ++q) {
                const int sr = a_r[q] + kk * a_dil;
                const bool ok = a_ok[q] & inb & ((unsigned)sr < (unsigned)a_rows_in);
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
            while (c >= a_cw) { c -= a_cw; ++kk; }
            bc += 32; kb += 128u;
            while (bc >= seg_k) { bc -= seg_k; kb += kb_wrap; }
            if (live && ++s_f >= nslab_f) next_tile();         // (wave-uniform)
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
        next_tile();                                           // first tile of this workgroup (or none)
        fetch(set0{});                                         // slab 0
        fetch(set1{});                                         // slab 1
        stage(set0{}, 0);
        fetch(set0{});                                         // slab 2
        MW_BARRIER_MOVER();
        // step n: the matrix waves multiply slab n out of buffer n & 1; slab n + 1 is staged into the other buffer (read last during
        // step n - 1) and slab n + 3 fetched into the register set that has just been emptied
        int n = 0;
        while (n < total) {
            if (n + 1 < total) stage(set1{}, 1);
            fetch(set1{});
            MW_BARRIER_MOVER();
            if (++n >= total) break;
            if (n + 1 < total) stage(set0{}, 0);
            fetch(set0{});
            MW_BARRIER_MOVER();
            ++n;
        }
    } else {
        // ============================================================================================ matrix waves (0-3)
        const int wm = (wave >> 1) & 1, wn = wave & 1;
        const int r16 = lane & 15, kq = lane >> 4;
        const int fcol = (8 * kq) ^ mw_swz(r16);
        f32x4 acc[TM][TN];
        // acc[i][j] = (W tile j) . (X tile i)^T: the weight fragment is the MFMA's A operand, so the lane's four accumulator values are
        // output row (i, r16), columns (j, 4 kq .. 4 kq + 3)
        auto multiply = [&](int buf) {
            const __bf16* const lb = lds + buf * BUF;
            bf16x8 fa[NS][TM];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(lb + s * PLANE + (wm * (16 * TM) + i * 16 + r16) * 32 + fcol);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8 fb[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(lb + s * PLANE + (BM + wn * (16 * TN) + j * 16 + r16) * 32 + fcol);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    f32x4 cc = acc[i][j];
                    if constexpr (ABL & 1) {               // fragments stay live, no matrix instruction
                        asm volatile("" :: "v"(fb[0]), "v"(fb[NS - 1]), "v"(fb[NS / 2]), "v"(fa[0][i]), "v"(fa[NS - 1][i]), "v"(fa[NS / 2][i]));
                        continue;
                    }
                    if constexpr (NS == 3) {               // smallest terms first
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[2][i], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[0][i], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[1][i], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[1][i], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[0][i], cc, 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[0][i], cc, 0, 0, 0);
                }
            }
        };
        MW_BARRIER_MATRIX();
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
                MW_BARRIER_MATRIX();                           // this slab's fragments are in registers; the movers may refill its buffer
                ++n;
            }
            // ---- epilogue straight from the accumulators: bias, activation, dropout scale, gate, accumulate, second output; every access a
            // 16-byte piece of one output row (an instruction covers 16 rows x 64 bytes).  The stores drain while the next tile runs.
            const NtProb& pr = g.p[pi];
            const float* __restrict__ bias = pr.bias;
            const float* __restrict__ mul = pr.mul;
            const float* __restrict__ gate = pr.gate;
            const float* __restrict__ res = pr.res;
            float* __restrict__ C2 = pr.C2;
            float* __restrict__ C = pr.C;
            const float slope = pr.slope, slope2 = pr.res_slope;
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
                const bool cok = col < N;                      // N % 4 == 0: a piece is inside or outside as a whole
                const int cc0 = cok ? col : 0;
                const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + cc0) : z4;
                f32x4 mv[TM], gv[TM], rv[TM], cv[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {                 // every read of the column group first, from always-valid addresses
                    const long o = ro[i] + cc0;
                    if (mul) mv[i] = *reinterpret_cast<const f32x4*>(mul + o);
                    if (gate) gv[i] = *reinterpret_cast<const f32x4*>(gate + o);
                    if (res) rv[i] = *reinterpret_cast<const f32x4*>(res + o);
                    if (accumulate) cv[i] = *reinterpret_cast<const f32x4*>(C + o);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const long o = ro[i] + cc0;
                    f32x4 v = acc[i][j] + bv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = act_fn(v[q], slope);
                    if (mul) v *= mv[i];
                    if (gate) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = gv[i][q] > 0.f ? v[q] : 0.f;
                    }
                    if (accumulate) v += cv[i];
                    const bool ok = rok[i] & cok;
                    if constexpr (ABL & 16) { if (v[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(C + o) = v; continue; }
                    if (ok) *reinterpret_cast<f32x4*>(C + o) = v;
                    if (res) {
                        f32x4 w = v + rv[i];
#pragma unroll
                        for (int q = 0; q < 4; ++q) w[q] = act_fn(w[q], slope2);
                        if (ok) *reinterpret_cast<f32x4*>(C2 + o) = w;
                    }
                }
            }
        }
    }
#ifdef TG_LAB_ABLATE
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 512) {
        mw_role_cycles[blockIdx.x * 4 + (wave >> 2) * 2 + 0] = waited;
        mw_role_cycles[blockIdx.x * 4 + (wave >> 2) * 2 + 1] = __builtin_amdgcn_s_memtime() - t_begin;
    }
#endif
#undef MW_BARRIER_MOVER
#undef MW_BARRIER_MATRIX
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
        if (!p.vec_c || p.A.cw % 4 != 0 || p.b_seg_k % 4 != 0 || p.A.K % 4 != 0) return false;
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
#define TG_ABL(A_) case A_: hipLaunchKernelGGL((gemm_nt_mw_kernel<4, 6, 3, A_>), grid, dim3(512), 0, s, g); return check_launch("tg_gemm_nt(mover waves, ablated)")
            switch (abl) { TG_ABL(1); TG_ABL(2); TG_ABL(3); TG_ABL(4); TG_ABL(6); TG_ABL(7); TG_ABL(16); TG_ABL(23); default: break; }
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
