// Persistent GRU recurrence for H = 64 (the discriminator's 4-layer bi-GRU, T = 28): one workgroup owns 16 batch rows of one
// direction and walks the whole sequence; the dependent chain of T steps is what this kernel is about.
//
// Per step the recurrent product  gh^T[3 x 64 gate units][16 rows] = W_hh[192][64] . h_{t-1}^T  runs on the bf16 matrix cores
// at fp32 accuracy (exact three-way split of both operands, six partial products, common.hpp / gemm_split.hip): wave w owns
// hidden units [16w, 16w + 16) of the three gates, its 48 rows of W_hh live in registers as pre-split bf16 A-fragments for the
// whole sequence (72 VGPRs), h_{t-1} comes from LDS as three bf16 planes.  36 MFMAs of 16 cycles per wave and step instead of the
// 48 f32 MFMAs of 32 cycles of the first version.  The product is taken TRANSPOSED (weights as the A operand, h as B): a lane's
// four accumulator values are then four CONSECUTIVE hidden units of ONE batch row, so every global access of the gate epilogue
// (gi, y, saved gates, dropout mask) is a 16-byte vector and h_t goes back to LDS as one 8-byte store per plane.
// h is double-buffered in LDS: ONE barrier per step, and it waits for LDS only (s_waitcnt lgkmcnt(0) + s_barrier) -- the
// step's global stores and the next step's operand prefetch stay in flight across it.
//
// Optional fused inter-layer dropout (nn.GRU dropout=0.3 between layers): with drop_mask != NULL the forward also writes
// y_drop = y * mask (the next layer's input) and the backward multiplies the incoming gradient by the mask while loading it --
// the separate dropout / mask-multiply passes and their launches disappear.  The mask is drawn beforehand for all layers of a
// pass by ONE tg_dropout_mask launch (or injected by the parity tests).
#include "common.hpp"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace tg {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), unrolled
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

constexpr int HS = 64;

// eight consecutive fp32 -> three bf16x8 fragments (hi / mid / lo planes)
// bf16m (wave-uniform, math mode 1 = plain bf16 operands): plane 0 holds the value rounded to nearest even and is the only plane multiplied
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8 (&out)[3], int bf16m) {
    if (bf16m) {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i] = (__bf16)a[i]; r[4 + i] = (__bf16)b[i]; }
        out[0] = out[1] = out[2] = r;
        return;
    }
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xa = a[i], xb = b[i];
        split3_bits(xa, h[i], m[i], l[i]);
        split3_bits(xb, h[4 + i], m[4 + i], l[4 + i]);
    }
    out[0] = __builtin_bit_cast(bf16x8, u32x4{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7])});
    out[1] = __builtin_bit_cast(bf16x8, u32x4{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7])});
    out[2] = __builtin_bit_cast(bf16x8, u32x4{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7])});
}

// four consecutive fp32 -> three 8-byte LDS words (4 bf16 each)
__device__ __forceinline__ void split4_store(const f32x4 v, __bf16* p0, __bf16* p1, __bf16* p2, int bf16m) {
    if (bf16m) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (__bf16)v[i];
        *reinterpret_cast<u32x2*>(p0) = __builtin_bit_cast(u32x2, r);
        return;
    }
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = v[i];
        split3_bits(x, h[i], m[i], l[i]);
    }
    *reinterpret_cast<u32x2*>(p0) = u32x2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
    *reinterpret_cast<u32x2*>(p1) = u32x2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
    *reinterpret_cast<u32x2*>(p2) = u32x2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
}

// Make a fragment opaque to the optimiser.  Without this hipcc REMATERIALISES the pre-split weight fragments inside the step loop -- it keeps
// only the loaded fp32 weights live and redoes the whole three-way split (and / sub / and / sub / perm per element: ~290 of the 570
// instructions of a forward step, ~330 of a backward step) on every one of the T steps, to save 24 of 512 registers.  ISA check:
// the step loop must hold ~8 v_and_b32 (the split of h_t), not ~104.  (Measured: the step takes the same 1.2 / 1.7 us with 304 as with
// 568 instructions -- the redone split filled issue slots beside the step's dependent chain LDS -> 36 MFMAs -> 24 quarter-rate
// transcendentals -> split -> LDS -> barrier, which is what bounds it; the pin saves the energy, not the time.)
__device__ __forceinline__ void pin_fragment(bf16x8& v) {
    u32x4 t = __builtin_bit_cast(u32x4, v);
    asm volatile("" : "+v"(t));
    v = __builtin_bit_cast(bf16x8, t);
}

// Lab build only (-DTG_LAB_STAMP, make lab, tools/h64_stamps.py): s_memtime stamps of workgroup (0, 0), wave 0 at the phase boundaries of every
// step, into a device array read back through tg_lab_h64_read_stamps.  The stamp's own s_waitcnt lgkmcnt(0) also drains LDS operations, so
// a phase that ends in LDS stores includes their completion.  The product build contains none of this.
#ifdef TG_LAB_STAMP
__device__ unsigned long long tg_h64_stamps[64][8];
#define TG_STAMP(step_, i_)                                                                   \
    do {                                                                                      \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64 && (step_) < 64) {         \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
            if (threadIdx.x == 0) tg_h64_stamps[step_][i_] = t_;                              \
        }                                                                                     \
    } while (0)
#define TG_FORCE(v_) do { float f_; asm volatile("v_mov_b32 %0, %1" : "=v"(f_) : "v"(v_)); asm volatile("" :: "v"(f_)); } while (0)
#else
#define TG_STAMP(step_, i_) do { } while (0)
#define TG_FORCE(v_) do { } while (0)
#endif

// LDS-only workgroup barrier: the step's global stores / prefetched loads are NOT drained
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- forward: recurrence waves that touch no global memory + mover waves -----------------------------------------------------------
// Round 2 (profiles/r2_h64_ablate.txt; the single-role kernel it studied was removed in round 4): six 16-byte output stores per step cost the
// recurrence 0.4-0.6 us of a 1.25 us step -- not bandwidth (24 KB per step) but ISSUE: each store instruction of a wave covers 16 rows x 64
// bytes (half cache lines), the CU's one address unit takes ~60 cycles per such instruction, four waves issue them at the same moment, and a
// wave that is issuing cannot run its dependent chain.  So the workgroup has EIGHT waves: waves 0-3 are the recurrence (wave w owns hidden
// units [16 w, 16 w + 16) of the three gates, weights resident as pre-split A fragments), whose step operands (gi, dropout mask) come out
// of LDS and its outputs (h, h * mask, r, z, n, W_hn h + b) go into LDS; waves 4-7 move data: they store finished output records from LDS
// to global memory -- each lane a 16-byte piece of a row's contiguous record, so a store instruction covers whole lines -- and stage the
// operands of step t + 1 into LDS from registers loaded two steps earlier.  Both kinds of wave meet at the one LDS-only barrier per step
// that the recurrence has anyway.  Every wave's vector-memory instruction count per step is static (clamped rows and steps, no
// predicates); the movers' first flushes write zero records to step 0's own addresses and are overwritten in order later.
//
// Round 4 (tools/h64_stamps2.py, profiles/r4_a_h64_stamps2.txt: 2 900 cycles per stamped step, of which the matrix pipe 576): the step's
// dependent chain is what the kernel costs, so everything that need not be ON it is taken off --
//   * the h fragments are requested first and the recurrence waves run at s_setprio 2, the movers at 0 and a few dozen cycles late
//     (s_sleep): right after the barrier all eight waves used to hit the LDS at once and the fragments queued behind the movers' 16-byte
//     staging stores (1 190 cycles from the barrier to the last MFMA issued);
//   * the products run gate-major (r: 12 MFMAs, z: 12, n: 12 -- the same accumulation order per gate as before) so sigmoid(r) is evaluated
//     under the z and n products and sigmoid(z) under the n products (590 cycles of gate arithmetic used to start after the last MFMA);
//   * the step's output record leaves for LDS one step LATER, in the next step's MFMA shadow, from registers (310 cycles of 16-byte LDS
//     stores used to sit between the gates and the barrier): the movers therefore store records two steps behind;
//   * the operand-plane count NS (3 = fp32-accurate, 1 = bf16 tier) is a template parameter: the wave-uniform runtime branch around every
//     MFMA group cut the loop into basic blocks the scheduler could not interleave across.
// LDS rows of the exchanged bf16 planes in the mover-wave kernels: unpadded (128 / 384 bytes), 16-byte slot index XOR (row & 7).  By the bank
// model of MI355X_MICROARCH.md (ds_read_b128: four fixed 16-lane groups over 64 banks; ds_write_b64: 16 contiguous lanes over 32 banks) the
// padded rows these kernels started with (144 / 400 bytes) serve every fragment read in 8 LDS cycles (2-way), these in 4; the 8-byte stores
// stay 2-way (hidden behind the store's register transfer).  The backward reads 72 fragments per step and workgroup: 288 LDS cycles saved.
constexpr int HX2_LD = HS, DG2_LD = 3 * HS;
__device__ __forceinline__ int swz_col(int row, int col) { return (((col >> 3) ^ (row & 7)) << 3) | (col & 7); }      // bf16 column -> swizzled column
constexpr int OB_LD = 6 * HS + 4;      // floats per LDS row of an output record [h | h*mask | r | z | n | hn] (+4: rows 16 bytes apart in the banks)
constexpr int IB_LD = 4 * HS + 4;      // floats per LDS row of an operand record [gi_r | gi_z | gi_n | mask]
constexpr int MOVER_LAG = 1;           // s_sleep argument of the movers after each barrier (64 cycles per unit)

// eight consecutive fp32 -> NS bf16x8 fragments (NS == 1: rounded to nearest even; NS == 3: exact hi / mid / lo)
// fp16 x 2: eight consecutive fp32 times their power-of-two scale -> hi / lo fp16 fragments (bit patterns in bf16x8 registers)
__device__ __forceinline__ void split8_h2(const f32x4 a, const f32x4 b, const float scale, bf16x8 (&out)[2]) {
    unsigned h[4], l[4];
    h2_split2(a[0] * scale, a[1] * scale, h[0], l[0]);
    h2_split2(a[2] * scale, a[3] * scale, h[1], l[1]);
    h2_split2(b[0] * scale, b[1] * scale, h[2], l[2]);
    h2_split2(b[2] * scale, b[3] * scale, h[3], l[3]);
    out[0] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
    out[1] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}
// ... four consecutive fp32 -> two 8-byte LDS words
__device__ __forceinline__ void split4_store_h2(const f32x4 v, const float scale, __bf16* p0, __bf16* p1) {
    unsigned h0, l0, h1, l1;
    h2_split2(v[0] * scale, v[1] * scale, h0, l0);
    h2_split2(v[2] * scale, v[3] * scale, h1, l1);
    *reinterpret_cast<u32x2*>(p0) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(p1) = u32x2{l0, l1};
}

template <int NS>
__device__ __forceinline__ void split8_ns(const f32x4 a, const f32x4 b, bf16x8 (&out)[NS]) {
    static_assert(NS != 2, "fp16 x 2 operands are split with their scale: split8_h2");
    if constexpr (NS == 1) {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i] = (__bf16)a[i]; r[4 + i] = (__bf16)b[i]; }
        out[0] = r;
    } else {
        bf16x8 t[3];
        split8(a, b, t, 0);
        out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
    }
}
// four consecutive fp32 -> NS 8-byte LDS words
template <int NS>
__device__ __forceinline__ void split4_store_ns(const f32x4 v, __bf16* p0, __bf16* p1, __bf16* p2) { split4_store(v, p0, p1, p2, NS == 1 ? 1 : 0); }
template <int NS>
__device__ __forceinline__ f32x4 mma_ns(const bf16x8 (&wa)[NS], const bf16x8 (&fb)[NS], f32x4 acc) {
    static_assert(NS != 2, "lab projection: bf16 operands only");
    if constexpr (NS == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[0], acc, 0, 0, 0);
    else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[2], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], fb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[1], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[0], acc, 0, 0, 0);
    }
}

// the chain of mma_ns over KS k-steps, one MFMA at a time: MFMA number i (k-step major, then the six terms smallest first)
template <int NS, int KS>
__device__ __forceinline__ f32x4 mma_one(int i, const bf16x8 (&wa)[KS][NS], const bf16x8 (&fb)[KS][NS], f32x4 acc) {
    constexpr int PER = NS == 3 ? 6 : NS == 2 ? 3 : 1;
    const int ks = i / PER, t = i - ks * PER;
    if constexpr (NS == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][0], fb[ks][0], acc, 0, 0, 0);
    else if constexpr (NS == 2) {
        // fp16 x 2 (forward kernel): planes 0 / 1 = hi / lo fp16 of the scaled operand; lo_w hi_x, hi_w lo_x, hi_w hi_x (small terms first)
        constexpr int WP2[3] = {1, 0, 0}, FP2[3] = {0, 1, 0};
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tg_f16x8, wa[ks][WP2[t]]), __builtin_bit_cast(tg_f16x8, fb[ks][FP2[t]]), acc, 0, 0, 0);
    } else {
        constexpr int WP[6] = {2, 0, 1, 1, 0, 0}, FP[6] = {0, 2, 1, 0, 1, 0};
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][WP[t]], fb[ks][FP[t]], acc, 0, 0, 0);
    }
}
// sigmoid of one gate pre-activation in three two-instruction stages (so that a stage fits the issue gap of one MFMA)
__device__ __forceinline__ float sigmoid_stage(int stage, float t, float a, float b, float c) {
    if (stage == 0) return a + b + c;
    if (stage == 1) return __builtin_amdgcn_exp2f(t * -1.44269504088896341f);
    return __builtin_amdgcn_rcpf(1.f + t);
}

// ABL (lab build only, tools/h64_ablate2.py; results are wrong by construction): bit 0 movers do nothing but meet the barriers, bit 1 no
// transcendental gate arithmetic, bit 2 no MFMAs, bit 3 no record stores, bit 4 no split / h store, bit 5 no mover lag, bit 6 no priority
template <bool SAVE, bool DROP, int NS, int RW = 16, int ABL = 0>
__global__ __launch_bounds__(512) void gru_h64_fwd2_kernel(
    const float* __restrict__ gi, long gi_ds, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ Y, float* __restrict__ save, long save_ds,
    const float* __restrict__ drop_mask, float* __restrict__ y_drop, int B, int T) {
    __shared__ __attribute__((aligned(16))) __bf16 hs[2][3][RW][HX2_LD];
    __shared__ __attribute__((aligned(16))) float obuf[2][RW][OB_LD];
    __shared__ __attribute__((aligned(16))) float ibuf[2][RW][IB_LD];
    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // h_{-1} = 0 (the tile step 0 reads) and zero output records for the movers' first two flushes
    for (int i = threadIdx.x; i < 3 * RW * HX2_LD / 2; i += 512) reinterpret_cast<unsigned*>(&hs[1][0][0][0])[i] = 0u;
    for (int i = threadIdx.x; i < 2 * RW * OB_LD; i += 512) (&obuf[0][0][0])[i] = 0.f;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ movers
        const int ml = threadIdx.x - 256;                                    // 0 .. 255
        constexpr int OPR = (1 + (DROP ? 1 : 0) + (SAVE ? 4 : 0)) * (HS / 4); // 16-byte pieces per output row in use
        constexpr int NOUT = (RW * OPR + 255) / 256;                         // pieces per mover lane (a last partial round repeats the final piece)
        constexpr int IPR = (3 + (DROP ? 1 : 0)) * (HS / 4);
        constexpr int NIN = (RW * IPR + 255) / 256;
        // output piece i of this lane: LDS offset (floats) and the global pointer for time index 0 (advanced by tau * stride per step)
        int o_lds[NOUT];
        float* o_ptr[NOUT];
        long o_ts[NOUT];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int pc = min(i * 256 + ml, RW * OPR - 1), r = pc / OPR, c = pc - r * OPR;     // row, piece inside the used part of the record
            const int rowg = min((int)blockIdx.x * RW + r, B - 1);
            // used pieces in record order: h [0,16), then h*mask [16,32) if DROP, then the four saved gates
            int arr = c / 16, u = 4 * (c - 16 * (c / 16));
            int rec;                                                          // array index inside the LDS record
            if (arr == 0) { rec = 0; o_ptr[i] = Y + (long)rowg * T * (2 * HS) + dir * HS + u; o_ts[i] = 2 * HS; }
            else if (DROP && arr == 1) { rec = 1; o_ptr[i] = y_drop + (long)rowg * T * (2 * HS) + dir * HS + u; o_ts[i] = 2 * HS; }
            else { const int g = arr - (DROP ? 2 : 1); rec = 2 + g; o_ptr[i] = save + dir * save_ds + (long)rowg * T * (4 * HS) + g * HS + u; o_ts[i] = 4 * HS; }
            o_lds[i] = r * OB_LD + rec * HS + u;
        }
        int i_lds[NIN];
        const float* i_ptr[NIN];
        long i_ts[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int pc = min(i * 256 + ml, RW * IPR - 1), r = pc / IPR, c = pc - r * IPR;
            const int rowg = min((int)blockIdx.x * RW + r, B - 1);
            const int arr = c / 16, u = 4 * (c - 16 * (c / 16));
            if (arr < 3) { i_ptr[i] = gi + dir * gi_ds + (long)rowg * T * (3 * HS) + arr * HS + u; i_ts[i] = 3 * HS; }
            else { i_ptr[i] = drop_mask + (long)rowg * T * (2 * HS) + dir * HS + u; i_ts[i] = 2 * HS; }
            i_lds[i] = r * IB_LD + arr * HS + u;
        }
        f32x4 in_set[2][NIN];
        auto load_step = [&](auto set_c, int step_l) {
            constexpr int sc = decltype(set_c)::value;
            const int sl = step_l < T ? step_l : T - 1;                      // past the end: a valid address, never consumed
            const int tau_l = dir ? T - 1 - sl : sl;
#pragma unroll
            for (int i = 0; i < NIN; ++i) in_set[sc][i] = *reinterpret_cast<const f32x4*>(i_ptr[i] + tau_l * i_ts[i]);
        };
        auto stage = [&](auto set_c, int buf) {
            constexpr int sc = decltype(set_c)::value;
#pragma unroll
            for (int i = 0; i < NIN; ++i) *reinterpret_cast<f32x4*>(&ibuf[buf][0][0] + i_lds[i]) = in_set[sc][i];
        };
        auto flush = [&](int buf, int step_o) {                              // outputs of step step_o, sitting in obuf[buf]
            const int tau_o = dir ? T - 1 - step_o : step_o;
#pragma unroll
            for (int i = 0; i < NOUT; ++i)
                *reinterpret_cast<f32x4*>(o_ptr[i] + tau_o * o_ts[i]) = *reinterpret_cast<const f32x4*>(&obuf[buf][0][0] + o_lds[i]);
        };
        using s0 = std::integral_constant<int, 0>;
        using s1 = std::integral_constant<int, 1>;
        // lab (ABL bit 7): what would the NEXT layer's input projection cost if the movers took it?  72 MFMAs per mover wave and step on the
        // record of step t - 2 (96 of the next layer's 384 gate rows x K = 64), 144 VGPRs of resident weight fragments, six 16-byte stores
        bf16x8 pw[6][2][NS];
        const int pr16 = lane & 15, pkq = lane >> 4;
        if constexpr (ABL & 128) {
#pragma unroll
            for (int tl = 0; tl < 6; ++tl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float* p = whh0 + (long)((((wave - 4) * 6 + tl) * 16 + pr16) % 192) * HS + 32 * ks + 8 * pkq;
                    split8_ns<NS>(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), pw[tl][ks]);
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) pin_fragment(pw[tl][ks][sp]);
                }
        }
        auto project = [&](int buf, int step_o) {
            if constexpr (ABL & 128) {
                const int tau_o = dir ? T - 1 - step_o : step_o;
                const float* rec = &obuf[buf][pr16 & (RW - 1)][(DROP ? HS : 0) + 8 * pkq];
                bf16x8 fbp[2][NS];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    split8_ns<NS>(*reinterpret_cast<const f32x4*>(rec + 32 * ks), *reinterpret_cast<const f32x4*>(rec + 32 * ks + 4), fbp[ks]);
                const int rowg = min((int)blockIdx.x * RW + (pr16 & (RW - 1)), B - 1);
                float* dst = save + dir * save_ds + ((long)rowg * T + tau_o) * (4 * HS) + 4 * pkq;
#pragma unroll
                for (int tl = 0; tl < 6; ++tl) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) acc = mma_ns<NS>(pw[tl][ks], fbp[ks], acc);
                    if (pr16 < RW) *reinterpret_cast<f32x4*>(dst + 16 * tl) = acc;
                }
            }
        };
        load_step(s0{}, 0);
        stage(s0{}, 0);                                                      // operands of step 0
        load_step(s1{}, 1);                                                  // set 1 <- step 1 (staged during step 0)
        load_step(s0{}, 2);                                                  // set 0 <- step 2 (staged during step 1)
        lds_barrier();
        // during step t: stage the operands of step t + 1 (set (t + 1) & 1, loaded two steps ago), refill that set with step t + 3,
        // store the record of step t - 2, which the recurrence waves wrote to obuf[t & 1] during step t - 1 (t < 2: a zero record to
        // step 0's addresses)
        auto mover_step = [&](auto set_c, int t) {
            if constexpr (!(ABL & 32)) __builtin_amdgcn_s_sleep(MOVER_LAG);  // the recurrence waves' fragment reads go first
            if constexpr (!(ABL & 1)) {
                stage(set_c, (t + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                load_step(set_c, t + 3);
                flush(t & 1, t > 1 ? t - 2 : 0);
                __builtin_amdgcn_sched_barrier(0);
                project(t & 1, t > 1 ? t - 2 : 0);
            }
            lds_barrier();
        };
        int t = 0;
        for (; t + 1 < T; t += 2) { mover_step(s1{}, t); mover_step(s0{}, t + 1); }
        if (t < T) mover_step(s1{}, t);
        if (T > 1) flush(T & 1, T - 2);                                      // written during the last step
        lds_barrier();                                                       // the last record is in obuf[(T - 1) & 1]
        flush((T - 1) & 1, T - 1);
        return;
    }

    // -------------------------------------------------------------------------------------------------- recurrence waves
    if constexpr (!(ABL & 64)) __builtin_amdgcn_s_setprio(2);
    const float* whh = dir ? whh1 : whh0;
    const float* bhh = dir ? bhh1 : bhh0;
    const int r16 = lane & 15, kq = lane >> 4;
    // RW = 8: the product's batch columns 8 .. 15 repeat rows 0 .. 7 (same operands, same results); each half of the lanes then stores
    // one half of the step's output record
    const int rr = r16 & (RW - 1), half = r16 >> 3;
    bf16x8 wa[3][2][NS];
    // fp16 x 2 (NS == 2): W_hh row (gate g, unit 16 wave + r16) scaled by its own power of two (largest magnitude over its 64 columns -> [2^14, 2^15):
    // the row's columns sit in this lane's two fragments and in the three other lanes of the same r16), h as hi / lo of h * 2^14 (|h| < 1); a
    // gate row's product comes back through 1 / (row scale * 2^14).  A lane's accumulators are units 16 wave + 4 kq + q: their rows' maxima are
    // fetched from the lanes that hold them (wave-local shuffles: no LDS table, no extra barrier for the movers to match).
    constexpr float H_SCALE = 16384.f;
    f32x4 winv[3] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        f32x4 wv[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* p = whh + (long)(g * HS + 16 * wave + r16) * HS + 32 * ks + 8 * kq;
            wv[ks][0] = *reinterpret_cast<const f32x4*>(p);
            wv[ks][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
        if constexpr (NS == 2) {
            unsigned mx = 0u;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const unsigned v = __float_as_uint(wv[ks][hf][q]) & 0x7fffffffu; mx = mx > v ? mx : v; }
            unsigned o = (unsigned)__shfl_xor((int)mx, 16, 64); mx = mx > o ? mx : o;
            o = (unsigned)__shfl_xor((int)mx, 32, 64); mx = mx > o ? mx : o;
            const float wsc = h2_scale_of_exp(h2_exp_of_bits(mx));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned mq = (unsigned)__shfl((int)mx, 4 * kq + q, 64);          // row 16 wave + 4 kq + q lives in lanes with r16 = 4 kq + q
                winv[g][q] = h2_inv_of_exp(h2_exp_of_bits(mq)) * (1.f / H_SCALE);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) split8_h2(wv[ks][0], wv[ks][1], wsc, wa[g][ks]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) split8_ns<NS>(wv[ks][0], wv[ks][1], wa[g][ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int s = 0; s < NS; ++s) pin_fragment(wa[g][ks][s]);
    }
    const int u0 = 16 * wave + 4 * kq;
    f32x4 bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = *reinterpret_cast<const f32x4*>(bhh + g * HS + u0);
    f32x4 hp = zero;
    f32x4 q_h = zero, q_hm = zero, q_r = zero, q_z = zero, q_n = zero, q_hn = zero;      // the record of the step before, still in registers
    // piece j of the record's stores (RW = 16: six 16-byte stores per lane; RW = 8: three, the lane halves taking one array each)
    constexpr int NREC = RW == 16 ? 6 : 3;
    auto store_piece = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        float* orow = &obuf[buf][rr][u0];
        if constexpr (RW == 16) {
            if constexpr (j == 0) *reinterpret_cast<f32x4*>(orow) = q_h;
            if constexpr (j == 1 && DROP) *reinterpret_cast<f32x4*>(orow + HS) = q_hm;
            if constexpr (SAVE) {
                if constexpr (j == 2) *reinterpret_cast<f32x4*>(orow + 2 * HS) = q_r;
                if constexpr (j == 3) *reinterpret_cast<f32x4*>(orow + 3 * HS) = q_z;
                if constexpr (j == 4) *reinterpret_cast<f32x4*>(orow + 4 * HS) = q_n;
                if constexpr (j == 5) *reinterpret_cast<f32x4*>(orow + 5 * HS) = q_hn;
            }
        } else {
            if constexpr (j == 0) { if constexpr (DROP) *reinterpret_cast<f32x4*>(orow + half * HS) = half ? q_hm : q_h; else *reinterpret_cast<f32x4*>(orow) = q_h; }
            if constexpr (SAVE) {
                if constexpr (j == 1) *reinterpret_cast<f32x4*>(orow + (2 + half) * HS) = half ? q_z : q_r;
                if constexpr (j == 2) *reinterpret_cast<f32x4*>(orow + (4 + half) * HS) = half ? q_hn : q_n;
            }
        }
    };
    auto store_record = [&](int buf) { static_for<NREC>([&](auto jc) { store_piece(jc, buf); }); };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // weights and biases are in: no vector-memory wait is left inside the loop
    lds_barrier();
    for (int step = 0; step < T; ++step) {
        const int rb = (step + 1) & 1, wb = step & 1;
        TG_STAMP(step, 0);
        // h_{t-1} fragments first: the products wait for nothing else
        bf16x8 fb[2][NS];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int s = 0; s < NS; ++s) fb[ks][s] = *reinterpret_cast<const bf16x8*>(&hs[rb][s][rr][swz_col(rr, 32 * ks + 8 * kq)]);
        f32x4 gx[3], gm = zero;
#pragma unroll
        for (int g = 0; g < 3; ++g) gx[g] = *reinterpret_cast<const f32x4*>(&ibuf[wb][rr][g * HS + u0]);
        if constexpr (DROP) gm = *reinterpret_cast<const f32x4*>(&ibuf[wb][rr][3 * HS + u0]);
        __builtin_amdgcn_sched_barrier(0);
        // Gate-major products (r: 12 MFMAs, n: 12, z: 12 -- per gate the same accumulation order as mma_ns) with the step's other work in
        // their issue gaps, pinned by sched_barriers (left to itself hipcc interleaves the three accumulation chains, so that all finish
        // together, and starts every bit of gate arithmetic after the last MFMA).  An MFMA holds the SIMD's vector issue for 8 of its 16
        // cycles: two or three INDEPENDENT vector instructions per gap are nearly free -- the gate arithmetic is therefore issued
        // operation-major over the lane's four elements (an element-major order is one dependent chain: ~12 cycles per instruction).
        //   region A (r products): the record of step - 1 (zeros at step 0) leaves for obuf[(step - 1) & 1]; bias folded into the operands
        //   region B (n products): sigmoid(r);   region C (z products): tanh(.) of the n gate and hp - n;   tail: sigmoid(z), blend
        constexpr int NM = NS == 3 ? 12 : NS == 2 ? 6 : 2;    // MFMAs per gate
        constexpr float NL2E = -1.44269504088896341f;
        // (fp16 x 2: the accumulators are the products times row scale * 2^14 -- the exact inverse rides in the multiplier the gate's first
        // operation has anyway)
        const f32x4 c_r = winv[0] * NL2E, c_z = winv[1] * NL2E;
        f32x4 acc[3] = {zero, zero, zero};
        f32x4 gb0, gb1;                                       // (gi + b_hh) * -log2(e) of the r and z gates
        f32x4 r4, z4, n4, hn4, h, tn, ta, tb;
        // sigmoid in four operations per element: t = acc * -log2e + gb; t = 2^t; t = 1 + t; t = 1 / t
        auto sig_op = [&](auto kc, auto qc, f32x4& t, const f32x4& ac, const f32x4& gb, const f32x4& cs) {
            constexpr int k = decltype(kc)::value, q = decltype(qc)::value;
            if constexpr (ABL & 2) { if constexpr (k == 0) t[q] = ac[q] + gb[q]; }
            else if constexpr (k == 0) t[q] = __builtin_fmaf(ac[q], NS == 2 ? cs[q] : NL2E, gb[q]);
            else if constexpr (k == 1) t[q] = __builtin_amdgcn_exp2f(t[q]);
            else if constexpr (k == 2) t[q] = 1.f + t[q];
            else t[q] = __builtin_amdgcn_rcpf(t[q]);
        };
        // n gate in ten operations per element (the last one prepares the blend: tb = h_prev - n)
        auto tanh_op = [&](auto kc, auto qc) {
            constexpr int k = decltype(kc)::value, q = decltype(qc)::value;
            if constexpr (ABL & 2) { if constexpr (k == 0) { hn4[q] = acc[2][q]; n4[q] = gx[2][q] + r4[q] * hn4[q]; tb[q] = hp[q] - n4[q]; } }
            else if constexpr (k == 0) hn4[q] = NS == 2 ? __builtin_fmaf(acc[2][q], winv[2][q], bh[2][q]) : acc[2][q] + bh[2][q];
            else if constexpr (k == 1) tn[q] = __builtin_fmaf(r4[q], hn4[q], gx[2][q]);
            else if constexpr (k == 2) ta[q] = fabsf(tn[q]) * (2.f * NL2E);
            else if constexpr (k == 3) ta[q] = __builtin_amdgcn_exp2f(ta[q]);
            else if constexpr (k == 4) tb[q] = 1.f - ta[q];
            else if constexpr (k == 5) ta[q] = 1.f + ta[q];
            else if constexpr (k == 6) ta[q] = __builtin_amdgcn_rcpf(ta[q]);
            else if constexpr (k == 7) ta[q] = tb[q] * ta[q];
            else if constexpr (k == 8) n4[q] = copysignf(ta[q], tn[q]);
            else tb[q] = hp[q] - n4[q];
        };
        {
            static_for<NM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (!(ABL & 4)) acc[0] = mma_one<NS, 2>(i, wa[0], fb, acc[0]);
                if constexpr (!(ABL & 8)) {                   // the record's stores, one per gap (all in the first gaps at NM = 2)
                    static_for<NREC>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr ((NM > 2 ? j : j * NM / NREC) == i) store_piece(jc, rb);
                    });
                }
                // 16 operand-folding operations, two per gap from gap NM - 8 on (all in the last gap at NM = 2)
                static_for<16>([&](auto sc) {
                    constexpr int sl = decltype(sc)::value, gap = NM >= 12 ? NM - 8 + sl / 2 : NM >= 6 ? NM - 4 + sl / 4 : NM - 1;
                    if constexpr (gap == i) {
                        constexpr int q = sl & 3, k = sl >> 2;
                        if constexpr (k == 0) gb0[q] = gx[0][q] + bh[0][q];
                        else if constexpr (k == 1) gb1[q] = gx[1][q] + bh[1][q];
                        else if constexpr (k == 2) { if constexpr (!(ABL & 2)) gb0[q] = gb0[q] * NL2E; }
                        else { if constexpr (!(ABL & 2)) gb1[q] = gb1[q] * NL2E; }
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        static_for<NM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (!(ABL & 4)) acc[2] = mma_one<NS, 2>(i, wa[2], fb, acc[2]);
            static_for<16>([&](auto sc) {                      // sigmoid(r): 16 slots over the NM gaps
                constexpr int sl = decltype(sc)::value;
                if constexpr (sl * NM / 16 == i) sig_op(std::integral_constant<int, sl / 4>{}, std::integral_constant<int, sl % 4>{}, r4, acc[0], gb0, c_r);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<NM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (!(ABL & 4)) acc[1] = mma_one<NS, 2>(i, wa[1], fb, acc[1]);
            static_for<40>([&](auto sc) {                      // the n gate: 40 slots over the NM gaps
                constexpr int sl = decltype(sc)::value;
                if constexpr (sl * NM / 40 == i) tanh_op(std::integral_constant<int, sl / 4>{}, std::integral_constant<int, sl % 4>{});
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        TG_STAMP(step, 1);                   // reads back, MFMAs issued (r and n gates in their shadow)
        TG_FORCE(acc[1][3]);
        TG_STAMP(step, 2);                   // MFMA results back
        static_for<16>([&](auto sc) {
            constexpr int sl = decltype(sc)::value;
            sig_op(std::integral_constant<int, sl / 4>{}, std::integral_constant<int, sl % 4>{}, z4, acc[1], gb1, c_z);
        });
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = __builtin_fmaf(z4[q], tb[q], n4[q]);        // (1 - z) n + z h_prev
        if constexpr (ABL & 4) {          // keep the fragment reads alive
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) { u32x4 t_ = __builtin_bit_cast(u32x4, fb[ks][sp]); asm volatile("" :: "v"(t_)); }
        }
        hp = h;
        TG_FORCE(h[0]); TG_FORCE(h[1]); TG_FORCE(h[2]); TG_FORCE(h[3]);
        TG_STAMP(step, 3);                   // gates done
        if constexpr (!(ABL & 16)) {
            const int cs = swz_col(rr, u0);
            if constexpr (NS == 2) split4_store_h2(h, H_SCALE, &hs[wb][0][rr][cs], &hs[wb][1][rr][cs]);
            else split4_store_ns<NS>(h, &hs[wb][0][rr][cs], &hs[wb][1][rr][cs], &hs[wb][2][rr][cs]);
        }
        else asm volatile("" :: "v"(h));
        TG_STAMP(step, 4);                   // h_t split and in LDS (stores complete)
        q_h = h; q_r = r4; q_z = z4; q_n = n4; q_hn = hn4;
        if constexpr (DROP) q_hm = h * gm;
        TG_STAMP(step, 5);
        lds_barrier();                       // h_t complete in LDS; the movers have staged step + 1
        TG_STAMP(step, 6);                   // past the barrier
    }
    store_record((T - 1) & 1);               // the last step's record
    lds_barrier();
}

// ---- backward through time: recurrence waves + movers (see gru_h64_fwd2_kernel) ----------------------------------------------------
// dh_t needs dgh_{t+1} @ W_hh (contraction over the 192 gate rows): taken transposed like the forward, A = W_hh^T rows [16 wave, 16 wave + 16)
// (pre-split, 6 k-steps x 3 planes in registers), B = the previous step's gate-gradient tile from LDS (three bf16 planes, double-buffered; zero
// for the first step).
// A step needs 7 loads and 6 stores of 16 bytes per lane: waves 4-7 stage the step's operand record
// [dy | mask | r | z | n | W_hn h + b | h_prev] into LDS two steps ahead and store the gate-gradient record [dr | dz | dn | dn r] of an
// earlier step as the two contiguous 768-byte rows dgi = [dr, dz, dn], dgh = [dr, dz, dn r].
// Round 4, as in the forward: fragments first and at priority, the movers late; everything of the gate arithmetic that does not need the
// product (the factors d n / d h, d z / d h, d r / d n and dy * mask + dh * z of the step before) is evaluated under the MFMAs, five
// multiplies per element remain behind them; the record leaves one step later from registers.
constexpr int BO_LD = 4 * HS + 4;
// ABL (lab build only): bit 0 movers idle, bit 1 no gate arithmetic, bit 2 no MFMAs, bit 3 no record stores, bit 4 no split / plane stores, bit 5 no mover lag
template <bool MASK, int NS, int RW = 16, int ABL = 0>
__global__ __launch_bounds__(512) void gru_h64_bwd2_kernel(
    const float* __restrict__ dY, const float* __restrict__ dy_mask, const float* __restrict__ Y, const float* __restrict__ save,
    long save_ds, const float* __restrict__ wt0, const float* __restrict__ wt1, float* __restrict__ dgi, float* __restrict__ dgh,
    long dg_ds, int B, int T) {
    constexpr int NARR = MASK ? 7 : 6;                       // operand arrays per row: dy, (mask), r, z, n, hn, h_prev
    constexpr int BI_LD = NARR * HS + 4;
    __shared__ __attribute__((aligned(16))) __bf16 dgs[2][3][RW][DG2_LD];
    __shared__ __attribute__((aligned(16))) float obuf[2][RW][BO_LD];
    __shared__ __attribute__((aligned(16))) float ibuf[2][RW][BI_LD];
    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 3 * RW * DG2_LD / 2; i += 512) reinterpret_cast<unsigned*>(&dgs[1][0][0][0])[i] = 0u;
    for (int i = threadIdx.x; i < 2 * RW * BO_LD; i += 512) (&obuf[0][0][0])[i] = 0.f;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ movers
        const int ml = threadIdx.x - 256;
        constexpr int OPR = 6 * (HS / 4), NOUT = (RW * OPR + 255) / 256;     // 96 pieces per row
        constexpr int IPR = NARR * (HS / 4), NIN = (RW * IPR + 255) / 256;   // 112 (96) pieces per row; a last partial round repeats the final piece
        int o_lds[NOUT];
        float* o_ptr[NOUT];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int pc = min(i * 256 + ml, RW * OPR - 1), r = pc / OPR, c = pc - r * OPR;
            const int rowg = min((int)blockIdx.x * RW + r, B - 1);
            const int half = c / 48, cc = c - 48 * half, arr = cc / 16, u = 4 * (cc - 16 * arr);
            o_ptr[i] = (half ? dgh : dgi) + dir * dg_ds + (long)rowg * T * (3 * HS) + arr * HS + u;
            o_lds[i] = r * BO_LD + ((half && arr == 2) ? 3 : arr) * HS + u;     // dgh's third gate is dn * r
        }
        int i_lds[NIN], i_kind[NIN];
        const float* i_ptr[NIN];
        long i_ts[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int pc = min(i * 256 + ml, RW * IPR - 1), r = pc / IPR, c = pc - r * IPR;
            const int rowg = min((int)blockIdx.x * RW + r, B - 1);
            const int arr = c / 16, u = 4 * (c - 16 * arr);
            const int a = MASK ? arr : (arr == 0 ? 0 : arr + 1);            // logical array: 0 dy, 1 mask, 2..5 r z n hn, 6 h_prev
            const long yo = (long)rowg * T * (2 * HS) + dir * HS + u;
            if (a == 0) { i_ptr[i] = dY + yo; i_ts[i] = 2 * HS; i_kind[i] = 0; }
            else if (a == 1) { i_ptr[i] = dy_mask + yo; i_ts[i] = 2 * HS; i_kind[i] = 0; }
            else if (a < 6) { i_ptr[i] = save + dir * save_ds + (long)rowg * T * (4 * HS) + (a - 2) * HS + u; i_ts[i] = 4 * HS; i_kind[i] = 0; }
            else { i_ptr[i] = Y + yo; i_ts[i] = 2 * HS; i_kind[i] = 1; }    // h_prev: the neighbouring time index
            i_lds[i] = r * BI_LD + arr * HS + u;
        }
        f32x4 in_set[2][NIN];
        auto load_step = [&](auto set_c, int step_l) {
            constexpr int sc = decltype(set_c)::value;
            const int sl = step_l < T ? step_l : T - 1;
            const int tau_l = dir ? sl : T - 1 - sl;
            const int tp = dir ? tau_l + 1 : tau_l - 1;
            const int tq = (tp >= 0 && tp < T) ? tp : tau_l;                 // no predecessor (last step): a valid address, multiplied by 0
#pragma unroll
            for (int i = 0; i < NIN; ++i) in_set[sc][i] = *reinterpret_cast<const f32x4*>(i_ptr[i] + (i_kind[i] ? tq : tau_l) * i_ts[i]);
        };
        auto stage = [&](auto set_c, int buf) {
            constexpr int sc = decltype(set_c)::value;
#pragma unroll
            for (int i = 0; i < NIN; ++i) *reinterpret_cast<f32x4*>(&ibuf[buf][0][0] + i_lds[i]) = in_set[sc][i];
        };
        auto flush = [&](int buf, int step_o) {
            const int tau_o = dir ? step_o : T - 1 - step_o;
#pragma unroll
            for (int i = 0; i < NOUT; ++i)
                *reinterpret_cast<f32x4*>(o_ptr[i] + (long)tau_o * (3 * HS)) = *reinterpret_cast<const f32x4*>(&obuf[buf][0][0] + o_lds[i]);
        };
        using s0 = std::integral_constant<int, 0>;
        using s1 = std::integral_constant<int, 1>;
        load_step(s0{}, 0);
        stage(s0{}, 0);
        load_step(s1{}, 1);
        load_step(s0{}, 2);
        lds_barrier();
        auto mover_step = [&](auto set_c, int t) {
            if constexpr (!(ABL & 32)) __builtin_amdgcn_s_sleep(MOVER_LAG);
            if constexpr (!(ABL & 1)) {
                stage(set_c, (t + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                load_step(set_c, t + 3);
                flush(t & 1, t > 1 ? t - 2 : 0);                             // the record of step t - 2, written to obuf[t & 1] during step t - 1
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier();
        };
        int t = 0;
        for (; t + 1 < T; t += 2) { mover_step(s1{}, t); mover_step(s0{}, t + 1); }
        if (t < T) mover_step(s1{}, t);
        if (T > 1) flush(T & 1, T - 2);
        lds_barrier();
        flush((T - 1) & 1, T - 1);
        return;
    }

    // -------------------------------------------------------------------------------------------------- recurrence waves
    __builtin_amdgcn_s_setprio(2);
    const float* wt = dir ? wt1 : wt0;
    const int r16 = lane & 15, kq = lane >> 4;
    const int rr = r16 & (RW - 1), half = r16 >> 3;            // RW = 8: see the forward kernel
    bf16x8 wa[6][NS];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const float* p = wt + (long)(16 * wave + r16) * (3 * HS) + 32 * ks + 8 * kq;
        split8_ns<NS>(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), wa[ks]);
#pragma unroll
        for (int s = 0; s < NS; ++s) pin_fragment(wa[ks][s]);
    }
    const int u0 = 16 * wave + 4 * kq;
    f32x4 dhz = zero;
    f32x4 q_r = zero, q_z = zero, q_n = zero, q_nr = zero;       // the record of the step before, still in registers
    constexpr int NREC = RW == 16 ? 4 : 2;
    auto store_piece = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        float* orow = &obuf[buf][rr][u0];
        if constexpr (RW == 16) {
            if constexpr (j == 0) *reinterpret_cast<f32x4*>(orow) = q_r;
            if constexpr (j == 1) *reinterpret_cast<f32x4*>(orow + HS) = q_z;
            if constexpr (j == 2) *reinterpret_cast<f32x4*>(orow + 2 * HS) = q_n;
            if constexpr (j == 3) *reinterpret_cast<f32x4*>(orow + 3 * HS) = q_nr;
        } else {
            if constexpr (j == 0) *reinterpret_cast<f32x4*>(orow + half * HS) = half ? q_z : q_r;
            if constexpr (j == 1) *reinterpret_cast<f32x4*>(orow + (2 + half) * HS) = half ? q_nr : q_n;
        }
    };
    auto store_record = [&](int buf) { static_for<NREC>([&](auto jc) { store_piece(jc, buf); }); };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int step = 0; step < T; ++step) {
        const int rb = (step + 1) & 1, wb = step & 1;
        TG_STAMP(step, 0);
        // the first two k-steps' fragments are requested before anything else
        bf16x8 fb[6][NS];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int s = 0; s < NS; ++s) fb[ks][s] = *reinterpret_cast<const bf16x8*>(&dgs[rb][s][rr][swz_col(rr, 32 * ks + 8 * kq)]);
        const float* irow = &ibuf[wb][rr][u0];
        f32x4 x_dy = *reinterpret_cast<const f32x4*>(irow), x_mk = zero;
        if constexpr (MASK) x_mk = *reinterpret_cast<const f32x4*>(irow + HS);
        constexpr int A0 = MASK ? 2 : 1;
        const f32x4 x_r = *reinterpret_cast<const f32x4*>(irow + A0 * HS), x_z = *reinterpret_cast<const f32x4*>(irow + (A0 + 1) * HS);
        const f32x4 x_n = *reinterpret_cast<const f32x4*>(irow + (A0 + 2) * HS), x_hn = *reinterpret_cast<const f32x4*>(irow + (A0 + 3) * HS);
        const f32x4 x_hp = *reinterpret_cast<const f32x4*>(irow + (A0 + 4) * HS);
#pragma unroll
        for (int ks = 2; ks < 6; ++ks)
#pragma unroll
            for (int s = 0; s < NS; ++s) fb[ks][s] = *reinterpret_cast<const bf16x8*>(&dgs[rb][s][rr][swz_col(rr, 32 * ks + 8 * kq)]);
        __builtin_amdgcn_sched_barrier(0);
        const float keep = step < T - 1 ? 1.f : 0.f;                 // the sequence's first time index has no predecessor
        // One accumulation chain of 6 k-steps (36 MFMAs at NS = 3).  In its issue gaps: the record of step - 1 leaves for obuf[(step - 1) & 1]
        // (MFMAs 0-3), then the factors that do not need the product -- d n / d h, d z / d h, d r / d n and dy * mask + dh * z of the step
        // before -- operation-major over the lane's four elements (independent neighbours), pinned by sched_barriers (see the forward kernel).
        constexpr int NM = NS == 3 ? 36 : 6;
        // THREE accumulators (k-steps {0, 1}, {2, 3}, {4, 5}) taken round-robin: an MFMA that accumulates onto its predecessor is forwarded
        // only when the two are issued back to back; with vector instructions in the gap it waits for the full write-back instead
        f32x4 acc3[3] = {zero, zero, zero};
        f32x4 pre, c_n, c_z, c_r, t0, t1;
        auto pre_op = [&](auto kc, auto qc) {
            constexpr int k = decltype(kc)::value, q = decltype(qc)::value;
            if constexpr (k == 0) { if constexpr (MASK) pre[q] = x_dy[q] * x_mk[q]; else pre[q] = x_dy[q]; }
            else if constexpr (k == 1) pre[q] = pre[q] + dhz[q];
            else if constexpr (k == 2) t0[q] = 1.f - x_z[q];
            else if constexpr (k == 3) t1[q] = __builtin_fmaf(-x_n[q], x_n[q], 1.f);
            else if constexpr (k == 4) c_n[q] = t0[q] * t1[q];
            else if constexpr (k == 5) t1[q] = __builtin_fmaf(x_hp[q], keep, -x_n[q]);
            else if constexpr (k == 6) t0[q] = x_z[q] * t0[q];
            else if constexpr (k == 7) c_z[q] = t1[q] * t0[q];
            else if constexpr (k == 8) t0[q] = 1.f - x_r[q];
            else if constexpr (k == 9) t1[q] = x_hn[q] * x_r[q];
            else c_r[q] = t1[q] * t0[q];
        };
        {
            static_for<NM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                {
                    constexpr int PER = NS == 3 ? 6 : 1;                     // MFMA i of the round-robin: accumulator i % 3, its (i / 3)-th MFMA
                    constexpr int a = i % 3, j = i / 3, ks = 2 * a + j / PER, t = j % PER;
                    if constexpr (!(ABL & 4)) acc3[a] = mma_one<NS, 6>(ks * PER + t, wa, fb, acc3[a]);
                }
                if constexpr (!(ABL & 8)) {
                    static_for<NREC>([&](auto jc) {
                        if constexpr (decltype(jc)::value == i) store_piece(jc, rb);
                    });
                }
                static_for<44>([&](auto sc) {                 // 44 slots over the gaps 4 .. NM - 1
                    constexpr int sl = decltype(sc)::value, gap = NM > 6 ? 4 + sl * (NM - 4) / 44 : 4 + sl / 22;
                    if constexpr (gap == i && !(ABL & 2)) pre_op(std::integral_constant<int, sl / 4>{}, std::integral_constant<int, sl % 4>{});
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        const f32x4 acc = (acc3[0] + acc3[1]) + acc3[2];
        if constexpr (ABL & 2) { pre = x_dy + dhz; c_n = x_n; c_z = x_z + x_hp * keep + x_mk; c_r = x_r + x_hn; }
        if constexpr (ABL & 4) {
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) { u32x4 t_ = __builtin_bit_cast(u32x4, fb[ks][sp]); asm volatile("" :: "v"(t_)); }
        }
        TG_STAMP(step, 1);
        TG_FORCE(acc[3]);
        TG_STAMP(step, 2);
        f32x4 g_r, g_z, g_n, g_nr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dh = pre[q] + acc[q];                          // step 0: zero tile, dhz = 0
            const float dn = dh * c_n[q];
            g_z[q] = dh * c_z[q];
            g_r[q] = dn * c_r[q];
            dhz[q] = dh * x_z[q];
            g_n[q] = dn; g_nr[q] = dn * x_r[q];
        }
        TG_FORCE(g_r[3]); TG_FORCE(g_z[3]); TG_FORCE(g_nr[3]);
        TG_STAMP(step, 3);
        if constexpr (ABL & 16) { asm volatile("" :: "v"(g_r), "v"(g_z), "v"(g_nr)); }
        else {
            const int c0 = swz_col(rr, u0), c1 = swz_col(rr, HS + u0), c2 = swz_col(rr, 2 * HS + u0);
            split4_store_ns<NS>(g_r, &dgs[wb][0][rr][c0], &dgs[wb][1][rr][c0], &dgs[wb][2][rr][c0]);
            split4_store_ns<NS>(g_z, &dgs[wb][0][rr][c1], &dgs[wb][1][rr][c1], &dgs[wb][2][rr][c1]);
            split4_store_ns<NS>(g_nr, &dgs[wb][0][rr][c2], &dgs[wb][1][rr][c2], &dgs[wb][2][rr][c2]);
        }
        TG_STAMP(step, 4);
        q_r = g_r; q_z = g_z; q_n = g_n; q_nr = g_nr;
        TG_STAMP(step, 5);
        lds_barrier();
        TG_STAMP(step, 6);
    }
    store_record((T - 1) & 1);
    lds_barrier();
}

}  // namespace tg

using namespace tg;

extern "C" int tg_get_math_mode(void);

// Batch rows per workgroup of the mover-wave kernels.  8 (the product's other eight batch columns repeat them) while that still leaves at most
// one workgroup per CU: what bounds these kernels next to the step's dependent chain is the traffic of ONE CU's memory pipe -- a 16-row
// workgroup moves 40 KB (forward) / 52 KB (backward) per step through it (tools/h64_ablate2.py, profiles/r4_g_h64_ablate2.txt: the backward
// step took 1.10 us with the movers against 0.74 us without) -- and most of the chip is idle anyway.  TG_H64_ROWS = 8 | 16 forces one.
static int h64_rows(int B) {
    const char* e = getenv("TG_H64_ROWS");          // read per call: tests and probes compare both
    if (e && (atoi(e) == 8 || atoi(e) == 16)) return atoi(e);
    return 2 * cdiv(B, 8) <= 256 ? 8 : 16;
}

extern "C" int tg_gru_h64_forward(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                                  const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                                  const float* drop_mask, float* y_drop, int32_t B, int32_t T, void* stream) {
    TG_REQUIRE(gi && w_hh_fwd && w_hh_rev && b_hh_fwd && b_hh_rev && y, "tg_gru_h64_forward: null pointer");
    TG_REQUIRE(B > 0 && T > 0, "tg_gru_h64_forward: bad sizes B=%d T=%d", B, T);
    TG_REQUIRE((drop_mask == nullptr) == (y_drop == nullptr), "tg_gru_h64_forward: drop_mask and y_drop go together");
    TG_REQUIRE(aligned16(gi) && aligned16(w_hh_fwd) && aligned16(w_hh_rev) && aligned16(b_hh_fwd) && aligned16(b_hh_rev) && aligned16(y) &&
               (save == nullptr || aligned16(save)) && (drop_mask == nullptr || (aligned16(drop_mask) && aligned16(y_drop))) &&
               gi_dir_stride % 4 == 0 && save_dir_stride % 4 == 0, "tg_gru_h64_forward: operands must be 16-byte aligned");
    {
#define TG_H64_FWD2(SAVE_, DROP_, NS_, RW_)                                                                                                     \
    hipLaunchKernelGGL((gru_h64_fwd2_kernel<SAVE_, DROP_, NS_, RW_>), dim3(cdiv(B, RW_), 2), dim3(512), 0, (hipStream_t)stream, gi,           \
                       (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, save, (long)save_dir_stride, drop_mask, y_drop, B, T)
#define TG_H64_FWD2_RW(NS_, RW_)                                   \
    do {                                                           \
        if (save && drop_mask) TG_H64_FWD2(true, true, NS_, RW_);  \
        else if (save) TG_H64_FWD2(true, false, NS_, RW_);         \
        else if (drop_mask) TG_H64_FWD2(false, true, NS_, RW_);    \
        else TG_H64_FWD2(false, false, NS_, RW_);                  \
    } while (0)
#define TG_H64_FWD2_NS(NS_)                                        \
    do {                                                           \
        if (h64_rows(B) == 8) TG_H64_FWD2_RW(NS_, 8);              \
        else TG_H64_FWD2_RW(NS_, 16);                              \
    } while (0)
#if defined(TG_LAB_STAMP) || defined(TG_LAB_ABL)
        {
            const char* const ae = getenv("TG_H64_ABL");
            const int abl = ae ? atoi(ae) : 0;
            if (abl && save && drop_mask) {
#define TG_H64_ABL_CASE(A_)                                                                                                                   \
    case A_:                                                                                                                                  \
        hipLaunchKernelGGL((gru_h64_fwd2_kernel<true, true, 3, 8, A_>), dim3(cdiv(B, 8), 2), dim3(512), 0, (hipStream_t)stream, gi,          \
                           (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, save, (long)save_dir_stride, drop_mask, y_drop, B, T); \
        return check_launch("tg_gru_h64_forward");
                switch (abl) {
                    TG_H64_ABL_CASE(1) TG_H64_ABL_CASE(2) TG_H64_ABL_CASE(4) TG_H64_ABL_CASE(8) TG_H64_ABL_CASE(16) TG_H64_ABL_CASE(32) TG_H64_ABL_CASE(64)
                    TG_H64_ABL_CASE(96) TG_H64_ABL_CASE(3) TG_H64_ABL_CASE(6) TG_H64_ABL_CASE(7) TG_H64_ABL_CASE(15) TG_H64_ABL_CASE(31) TG_H64_ABL_CASE(30)
                    TG_H64_ABL_CASE(9) TG_H64_ABL_CASE(22) TG_H64_ABL_CASE(128)
                    default: break;
                }
#undef TG_H64_ABL_CASE
            }
        }
#endif
        // fp32-accurate mode: fp16 x 2 operands (three MFMAs per product, round 6) unless TG_H64_H2=0 asks for bf16 x 3 (six)
        static const int h2 = [] { const char* e = getenv("TG_H64_H2"); return e ? atoi(e) : 1; }();
        if (tg_get_math_mode() == 1) TG_H64_FWD2_NS(1);
        else if (h2) TG_H64_FWD2_NS(2);
        else TG_H64_FWD2_NS(3);
#undef TG_H64_FWD2_NS
#undef TG_H64_FWD2_RW
#undef TG_H64_FWD2
        return check_launch("tg_gru_h64_forward");
    }
}

#ifdef TG_LAB_STAMP
extern "C" int tg_lab_h64_read_stamps(unsigned long long* out) {      // out: [64][8]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tg::tg_h64_stamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int tg_gru_h64_backward(const float* dy, const float* dy_mask, const float* y, const float* save, int64_t save_dir_stride,
                                   const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                                   int32_t B, int32_t T, void* stream) {
    TG_REQUIRE(dy && y && save && w_hh_t_fwd && w_hh_t_rev && dgi && dgh, "tg_gru_h64_backward: null pointer");
    TG_REQUIRE(B > 0 && T > 0, "tg_gru_h64_backward: bad sizes B=%d T=%d", B, T);
    TG_REQUIRE(aligned16(dy) && aligned16(y) && aligned16(save) && aligned16(w_hh_t_fwd) && aligned16(w_hh_t_rev) && aligned16(dgi) &&
               aligned16(dgh) && (dy_mask == nullptr || aligned16(dy_mask)) && save_dir_stride % 4 == 0 && dg_dir_stride % 4 == 0,
               "tg_gru_h64_backward: operands must be 16-byte aligned");
    {
#define TG_H64_BWD2_RW(MASK_, NS_, RW_)                                                                                                        \
    hipLaunchKernelGGL((gru_h64_bwd2_kernel<MASK_, NS_, RW_>), dim3(cdiv(B, RW_), 2), dim3(512), 0, (hipStream_t)stream, dy, dy_mask, y, save, \
                       (long)save_dir_stride, w_hh_t_fwd, w_hh_t_rev, dgi, dgh, (long)dg_dir_stride, B, T)
#define TG_H64_BWD2(MASK_, NS_)                                \
    do {                                                       \
        if (h64_rows(B) == 8) TG_H64_BWD2_RW(MASK_, NS_, 8);   \
        else TG_H64_BWD2_RW(MASK_, NS_, 16);                   \
    } while (0)
#if defined(TG_LAB_STAMP) || defined(TG_LAB_ABL)
        {
            const char* const ae = getenv("TG_H64_ABL");
            const int abl = ae ? atoi(ae) : 0;
            if (abl && dy_mask) {
#define TG_H64_ABL_CASE(A_)                                                                                                                         \
    case A_:                                                                                                                                        \
        hipLaunchKernelGGL((gru_h64_bwd2_kernel<true, 3, 8, A_>), dim3(cdiv(B, 8), 2), dim3(512), 0, (hipStream_t)stream, dy, dy_mask, y, save,    \
                           (long)save_dir_stride, w_hh_t_fwd, w_hh_t_rev, dgi, dgh, (long)dg_dir_stride, B, T);                                    \
        return check_launch("tg_gru_h64_backward");
                switch (abl) {
                    TG_H64_ABL_CASE(1) TG_H64_ABL_CASE(2) TG_H64_ABL_CASE(4) TG_H64_ABL_CASE(8) TG_H64_ABL_CASE(16) TG_H64_ABL_CASE(32)
                    TG_H64_ABL_CASE(3) TG_H64_ABL_CASE(6) TG_H64_ABL_CASE(7) TG_H64_ABL_CASE(15) TG_H64_ABL_CASE(31) TG_H64_ABL_CASE(30) TG_H64_ABL_CASE(9)
                    default: break;
                }
#undef TG_H64_ABL_CASE
            }
        }
#endif
        const bool one = tg_get_math_mode() == 1;
        if (dy_mask) { if (one) TG_H64_BWD2(true, 1); else TG_H64_BWD2(true, 3); }
        else { if (one) TG_H64_BWD2(false, 1); else TG_H64_BWD2(false, 3); }
#undef TG_H64_BWD2
#undef TG_H64_BWD2_RW
        return check_launch("tg_gru_h64_backward");
    }
}
