// Persistent, cluster-synchronised GRU recurrence for the generator (H <= 320), recurrent product on the bf16 matrix cores at fp32
// accuracy.  Same cluster structure and hand-off protocol as gru_cluster.hip (the f32-MFMA version, kept as the reference and the
// TG_GRU_X3=0 fallback): the CW = ceil(H / 32) workgroups that share a batch tile exchange h_t through write-through (sc1) stores
// + a drained flag word per member, one polling wave, sc1 loads straight into MFMA fragments; every spin is bounded.
//
// What changes is the arithmetic of the dependent chain.  v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 matrix rate, and the 120
// f32 MFMAs per wave and step (3.2 us at two waves per SIMD) were half of the ~7 us step.  Here every fp32 operand is split EXACTLY
// into three bf16 terms (common.hpp split3_bits) and the six significant partial products run on v_mfma_f32_16x16x32_bf16:
//   * W_hh: member m's 96 rows are split ONCE, when the kernel starts, and stay in registers as bf16 A-fragments (<= 108 VGPRs);
//   * h_t : split by its PRODUCER in the gate epilogue and published as three bf16 planes (6 bytes per element instead of 4), so
//           the consumers' critical path holds no conversion at all -- their 16-byte sc1 loads are the MFMA B-fragments.
// The product is taken transposed (C^T = W . h^T), so a lane's four accumulator values are four consecutive hidden units of one
// batch row.  K steps of 32 coincide with the members' 32-unit slices: k-step j of the product reads exactly the block member j
// published, [member][plane][row][32 units], 64 bytes per row and plane -- a wave's epilogue stores cover whole 128-byte lines.
// Per wave and step: <= 108 bf16 MFMAs of 16 cycles instead of 120 f32 MFMAs of 32.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace tg {

template <int N, typename Fn, int... I>
__device__ __forceinline__ void xc_static_for_impl(Fn&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename Fn>
__device__ __forceinline__ void xc_static_for(Fn&& f) { xc_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

typedef __attribute__((address_space(1))) unsigned gu32x;

constexpr int XC_UNITS = 32;          // hidden units per workgroup = one 32-deep k-step of the product
constexpr int XC_FLAG_STRIDE = 16;    // flag words per cluster (one 64-byte line)
constexpr unsigned XC_SPIN_LIMIT = 1u << 26;
constexpr int XC_GEN_WORD = 15;       // generation word of a cluster's 16-word flag line (members use words 0 .. CW - 1 <= 9)
constexpr int XC_POLL_WAVE = 7;       // of 8; epilogue threads live in waves 0 .. 3 (forward) / 0 .. 1 (backward)
constexpr unsigned XC_RSRC3 = 0x00020000u;
#ifndef XC_SAME_XCD_FAST
#define XC_SAME_XCD_FAST 1            // 0: always the write-through protocol (A/B builds: make CXXFLAGS+=-DXC_SAME_XCD_FAST=0)
#endif

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 as_f32x4(u32x4 v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ u32x4 as_u32x4(f32x4 v) { return __builtin_bit_cast(u32x4, v); }

// eight consecutive fp32 (two float4) -> three bf16x8 fragments (hi / mid / lo planes)
__device__ __forceinline__ void xc_split8(const f32x4 a, const f32x4 b, bf16x8 (&out)[3]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xa = a[i], xb = b[i];
        split3_bits(xa, h[i], m[i], l[i]);
        split3_bits(xb, h[4 + i], m[4 + i], l[4 + i]);
    }
    out[0] = as_bf16x8(u32x4{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7])});
    out[1] = as_bf16x8(u32x4{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7])});
    out[2] = as_bf16x8(u32x4{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7])});
}

// four consecutive fp32 -> three 8-byte words (4 bf16 each)
__device__ __forceinline__ void xc_split4(const f32x4 v, u32x2 (&out)[3]) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = v[i];
        split3_bits(x, h[i], m[i], l[i]);
    }
    out[0] = u32x2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
    out[1] = u32x2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
    out[2] = u32x2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
}

// fp16 x 2 (common.hpp "two-term fp16 split"): eight / four consecutive fp32, already multiplied by their power-of-two scale, -> hi and lo planes
__device__ __forceinline__ void xc_split8_h2(const f32x4 a, const f32x4 b, const float scale, bf16x8 (&out)[3]) {
    unsigned h[4], l[4];
    h2_split2(a[0] * scale, a[1] * scale, h[0], l[0]);
    h2_split2(a[2] * scale, a[3] * scale, h[1], l[1]);
    h2_split2(b[0] * scale, b[1] * scale, h[2], l[2]);
    h2_split2(b[2] * scale, b[3] * scale, h[3], l[3]);
    out[0] = as_bf16x8(u32x4{h[0], h[1], h[2], h[3]});
    out[1] = as_bf16x8(u32x4{l[0], l[1], l[2], l[3]});
    out[2] = out[1];
}
__device__ __forceinline__ void xc_split4_h2(const f32x4 v, const float scale, u32x2 (&out)[3]) {
    unsigned h0, l0, h1, l1;
    h2_split2(v[0] * scale, v[1] * scale, h0, l0);
    h2_split2(v[2] * scale, v[3] * scale, h1, l1);
    out[0] = u32x2{h0, h1};
    out[1] = u32x2{l0, l1};
    out[2] = out[1];
}

// fp16 x 2 exchange of the backward recurrence: a block's values are scaled to |v| < 2^11 (four binades below the usual target), so every lo
// value is <= 1/2 in magnitude and bit 14 -- the top bit of its exponent field -- is clear.  The eight lo values a consumer lane loads for a
// k-step carry the block's 8-bit exponent there: bit c in the low half of dword c, bit 4 + c in its high half.
constexpr unsigned XC_GX_EXP_SHIFT = 4;
__device__ __forceinline__ float xc_take_exp(bf16x8& lo) {
    u32x4 d = __builtin_bit_cast(u32x4, lo);
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) { t[i] = d[i] & 0x40004000u; d[i] ^= t[i]; }
    const unsigned x = (t[0] >> 14) | (t[1] >> 13) | (t[2] >> 12) | (t[3] >> 11);
    const unsigned e = (x & 0xFu) | ((x >> 12) & 0xF0u);
    lo = __builtin_bit_cast(bf16x8, d);
    return __uint_as_float((e - 14u) << 23);               // = h2_inv_of_exp(e)
}
// maximum over an aligned group of eight lanes (DPP: the two quad exchanges, then the half-row mirror)
__device__ __forceinline__ unsigned xc_max8(unsigned v) {
    unsigned w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = v > w ? v : w;    // quad_perm [1, 0, 3, 2]
    w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); v = v > w ? v : w;             // quad_perm [2, 3, 0, 1]
    w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = v > w ? v : w;            // row_half_mirror
    return v;
}

// plain-bf16 tier (math mode 1): one term per operand, rounded to nearest even; plane 0 of the exchange buffer carries it
__device__ __forceinline__ bf16x8 xc_rne8(const f32x4 a, const f32x4 b) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (__bf16)a[i]; r[4 + i] = (__bf16)b[i]; }
    return r;
}
__device__ __forceinline__ u32x2 xc_rne4(const f32x4 v) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (__bf16)v[i];
    return __builtin_bit_cast(u32x2, r);
}

// wa: hi and mid planes of the weight fragment (registers); w_lo: its lo plane (read back from LDS: it feeds one MFMA in six)
template <int NS>
__device__ __forceinline__ f32x4 xc_mma(const bf16x8 (&wa)[2], const bf16x8 w_lo, const bf16x8 (&fb)[NS], f32x4 acc) {
    if constexpr (NS == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[0], acc, 0, 0, 0);
    if constexpr (NS == 2) {                             // fp16 x 2: wa = hi / lo planes of the scaled weight rows, fb = hi / lo planes of h * 2^14
        auto hh = [](const bf16x8& v) { return __builtin_bit_cast(tg_f16x8, v); };
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh(wa[1]), hh(fb[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh(wa[0]), hh(fb[1]), acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(hh(wa[0]), hh(fb[0]), acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo, fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[1], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], fb[0], acc, 0, 0, 0);
}

// Flag words carry the publishing workgroup's XCD in their low four bits: value = (generation + step + 1) << 4 | xcc.  A cluster whose
// members all sit on ONE XCD -- which is how the dispatcher is observed to place them, but a fact only once it has been READ, from
// HW_REG_XCC_ID -- shares one L2: its members then publish with PLAIN stores (the line stays in that L2; acknowledged by the L2) instead of
// write-through sc1 stores (the line is dropped from L2 and the consumers' loads go to the fabric: MI355X_MICROARCH.md, "stores of each
// flavour").  Consumers load with sc1 (L1 bypass) either way.  Every member derives the decision from the same ten flag words of step 0, so
// the cluster switches together; a cluster that straddles XCDs keeps the write-through protocol for the whole launch.
__device__ __forceinline__ unsigned xc_my_xcc() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ unsigned xc_flag_value(unsigned count, unsigned xcc) { return (count << 4) | xcc; }

// bounded poll of the cluster's flag words by ONE wave (relaxed sc1 loads + s_sleep); returns false after a timeout (diagnostics
// written to the sticky timeout block).  *same_xcd: every member's flag carries this workgroup's XCD.
__device__ __forceinline__ bool xc_wait(gu32x* cl_flags, int CW, int lane, unsigned want, unsigned* tmo, int step, unsigned my_xcc, bool* same_xcd) {
    unsigned spins = 0;
    const unsigned want4 = want << 4;
    for (;;) {
        const unsigned v = lane < CW ? __hip_atomic_load(cl_flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (want4 | my_xcc);
        if (__all((int)(v - want4) >= 0)) {                  // generations wrap: compare the difference (the XCD bits only add 0 .. 15)
            *same_xcd = __all((v & 15u) == my_xcc);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        if (++spins > XC_SPIN_LIMIT) {                    // wave-uniform
            if (lane == 0) {
                __hip_atomic_store((gu32x*)tmo + 1, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((gu32x*)tmo + 2, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane < CW) __hip_atomic_store((gu32x*)tmo + 4 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) __hip_atomic_store((gu32x*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *same_xcd = false;
            return false;
        }
    }
}

// exchange buffer (bytes): [slot 2][dir 2][block][plane 3][b_pad rows][32 units] bf16; block = member (forward) or gate * CW + member
// (backward); plane_bytes = b_pad * 64
// ABL (lab build only, -DTG_LAB_ABL, tools/gru_cluster_ablate.py; wrong results by construction): bit 0 no flag wait, bit 1 no fragment loads,
// bit 2 no MFMAs, bit 3 no K-slice reduction / gate arithmetic, bit 4 no publishing stores, bit 5 no output stores / prefetch, bit 6 no drain
// (Round 4, measured and not kept: a "pair" form -- KS = 2 K slices, 256-thread workgroups of ONE batch tile, two per CU, so that one
// cluster's MFMAs run while the other sits in its hand-off.  At B = 384 it took 202 us per launch against 165: five k-steps per wave
// need 120 registers of resident weights, the kernel spilled 23 of them into the step loop, only three k-steps of fragment loads fit in
// flight, and the two workgroups of a CU fall into step rather than apart.  profiles/r4_p_pair_rejected.txt)
template <int MT, int NS, int ABL = 0>
__global__ __launch_bounds__(512) void gru_seq_fwd_cluster_x3_kernel(
    const float* __restrict__ gi, long gi_ds, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ Y, float* __restrict__ save, long save_ds,
    const float* __restrict__ drop_mask, float* __restrict__ y_drop, void* hx, unsigned* flags, unsigned* tmo, int B, int T, int H,
    int n_bt, int CW, int b_pad, int save_row0, int save_rows) {
    constexpr int KS = 4;                                 // K slices = waves along K; a workgroup has 2 KS waves (two 16-unit tiles x KS slices)
    constexpr int SPS = (10 + KS - 1) / KS;               // k-steps per K slice (CW <= 10)
    constexpr int POLL_WAVE = 2 * KS - 1;                 // a wave without epilogue threads (they live in waves 0 .. 2 MT - 1)
    static_assert(2 * MT <= POLL_WAVE, "the polling wave owns no epilogue threads");
    __shared__ __attribute__((aligned(16))) f32x4 red[KS][2][MT][3][64];
    __shared__ __attribute__((aligned(16))) bf16x8 wlo[2 * KS][3 * SPS][64];       // lo plane of every wave's weight fragments (lane-private slots)
    __shared__ int same_xcd_s;
    const unsigned my_xcc = xc_my_xcc();
    bool fast = false;                                   // plain-store publishing: set at step 1 when the whole cluster sits on this XCD
    const int n_cl = 2 * n_bt;
    int cl, m;
    if (n_cl % 8 == 0) {        // members of one cluster on block ids of one residue mod 8: same XCD as observed (speed only)
        cl = (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) / CW);
        m = (blockIdx.x / 8) % CW;
    } else {
        cl = blockIdx.x / CW;
        m = blockIdx.x % CW;
    }
    const int dir = cl / n_bt, bt = cl % n_bt;
    const float* whh = dir ? whh1 : whh0;
    const float* bhh = dir ? bhh1 : bhh0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: K-slice bounds and exchange offsets stay in SGPRs
    const int ut = wave & 1, ks = wave >> 1;                  // ks < KS
    const int r16 = lane & 15, kq = lane >> 4;
    const int b0 = bt * (16 * MT);

    // k-steps (= member blocks) of this wave's K slice: CW steps dealt as evenly as possible over the 4 slices
    const int s_base = CW / KS, s_rem = CW % KS;
    const int s_cnt = s_base + (ks < s_rem ? 1 : 0);
    const int s_beg = ks * s_base + (ks < s_rem ? ks : s_rem);

    // W_hh rows (gate g, unit 32 m + 16 ut + r16), k = 32 (s_beg + p) + 8 kq .. +7: pre-split A fragments, resident
    bf16x8 wa[3][SPS][2];
    // fp16 x 2: every W_hh row (gate g, unit) is scaled by its own power of two (largest magnitude over the row's H columns -> [2^14, 2^15)).  A row's
    // columns are spread over the KS waves of its unit tile: partial maxima meet in LDS; the epilogue threads read the same table for the inverse.
    __shared__ unsigned wmax[KS][2][3][16];
    float wsc[3] = {1.f, 1.f, 1.f};
    if constexpr (NS == 2) {
        const int j = m * XC_UNITS + ut * 16 + r16;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            unsigned mx = 0u;
#pragma unroll
            for (int p = 0; p < SPS; ++p) {
                const int k = 32 * (s_beg + p) + 8 * kq;
                const float* src = whh + (long)(g * H + j) * H + k;
                const bool ok = p < s_cnt && j < H;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned v = (ok && k + q < H) ? __float_as_uint(src[q]) & 0x7fffffffu : 0u;
                    mx = mx > v ? mx : v;
                }
            }
            unsigned w = (unsigned)__shfl_xor((int)mx, 16, 64); mx = mx > w ? mx : w;
            w = (unsigned)__shfl_xor((int)mx, 32, 64); mx = mx > w ? mx : w;
            if (kq == 0) wmax[ks][ut][g][r16] = mx;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            unsigned mx = wmax[0][ut][g][r16];
#pragma unroll
            for (int q = 1; q < KS; ++q) { const unsigned v = wmax[q][ut][g][r16]; mx = mx > v ? mx : v; }
            wsc[g] = h2_scale_of_exp(h2_exp_of_bits(mx));
        }
    }
    {
        const int j = m * XC_UNITS + ut * 16 + r16;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int p = 0; p < SPS; ++p) {
                const int k = 32 * (s_beg + p) + 8 * kq;
                const float* src = whh + (long)(g * H + j) * H + k;
                const bool ok = p < s_cnt && j < H;
                const f32x4 a = (ok && k < H) ? *reinterpret_cast<const f32x4*>(src) : z;          // H % 4 == 0
                const f32x4 b = (ok && k + 4 < H) ? *reinterpret_cast<const f32x4*>(src + 4) : z;
                bf16x8 pl[3];
                if constexpr (NS == 1) { pl[0] = xc_rne8(a, b); pl[1] = pl[2] = pl[0]; }
                else if constexpr (NS == 2) xc_split8_h2(a, b, wsc[g], pl);
                else xc_split8(a, b, pl);
                wa[g][p][0] = pl[0]; wa[g][p][1] = pl[1];
                if constexpr (NS == 3) wlo[wave][g * SPS + p][lane] = pl[2];
            }
    }
    // ---- epilogue role: thread e < 128 * MT finalises batch row (e / 8) of the tile, hidden units 4 * (e % 8) .. +3 of the slice
    const int e = threadIdx.x;
    const bool epi = e < 128 * MT;
    const int row_l = e >> 3, ug = e & 7;
    const int e_mt = (row_l >> 4) % MT, e_lane = (row_l & 15) + 16 * (ug & 3), e_ut = ug >> 2;
    const int row = b0 + row_l;
    const int unit0 = m * XC_UNITS + 4 * ug;
    const bool e_ok = epi && row < B && unit0 < H;           // H % 4 == 0: the four units are valid together
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bh[3];                                             // b_hr, b_hz, b_hn of this thread's four units
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = (epi && unit0 < H) ? *reinterpret_cast<const f32x4*>(bhh + g * H + unit0) : zero4;
    f32x4 hp = {0.f, 0.f, 0.f, 0.f};
    // fp16 x 2: h is published as hi / lo of h * 2^14 (|h| < 1 by construction: no measuring); the product of a gate row comes back through
    // 1 / (row scale * 2^14), an exact power of two
    constexpr float H_SCALE = 16384.f;
    f32x4 winv[3] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
    if constexpr (NS == 2) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = (4 * ug + q) & 15;
                unsigned mx = wmax[0][e_ut][g][rr];
#pragma unroll
                for (int k = 1; k < KS; ++k) { const unsigned v = wmax[k][e_ut][g][rr]; mx = mx > v ? mx : v; }
                winv[g][q] = h2_inv_of_exp(h2_exp_of_bits(mx)) * (1.f / H_SCALE);
            }
    }
    __syncthreads();                                                   // wlo visible (read back by the same lane: ordering only)

    const int plane_bytes = b_pad * 64;
    const int slot_bytes = CW * 3 * plane_bytes;                       // one (slot, dir)
    __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 4 * slot_bytes, XC_RSRC3);
    gu32x* my_flag = (gu32x*)(flags + cl * XC_FLAG_STRIDE + m);
    gu32x* cl_flags = (gu32x*)(flags + cl * XC_FLAG_STRIDE);
    bool aborted = false;
    // Flag words are never zeroed between launches: member m publishes gen + step + 1, where gen is the cluster's GENERATION word
    // (word 15 of its flag line), read here by every member and advanced by T + 1 by member 0 when it leaves.  Member 0 can only
    // finish step T - 1 after every member has published step T - 2, i.e. after every member has read gen (T >= 2; with T == 1
    // nothing is published or advanced -- there is no consumer).  Every value a launch leaves behind is <= gen + T < the next gen,
    // so stale words never satisfy a wait; the workspace only has to be zero ONCE, before its first use.
    const unsigned gen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cl_flags + XC_GEN_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));

    // The per-step streams (gi in; y, saved gates, dropped y out) go through BUFFER instructions: a lane without a valid (row, unit)
    // carries an offset past num_records -- its loads return 0 and its stores are dropped by the bounds check -- so every one of them is
    // issued unconditionally.  A lane-predicated load or store (s_cbranch_execz around it) makes the count of outstanding operations
    // dynamic, and hipcc then drains everything (s_waitcnt vmcnt(0)) at the next dependent use: the first version of this loop waited
    // at the top of every step for the previous step's output stores, then for the gi loads, then for the bias loads -- three exposed
    // round trips per step.  Now gi for step s + 1 is requested right after step s has published its flag and is in flight during the
    // next hand-off; nothing but the exchange itself is waited for.  (All byte sizes < 2^31: checked by the host.)
    constexpr unsigned OOB = 0x80000000u;
    const int epi_wave = wave < 2 * MT;                                // scalar: waves that own epilogue threads
    __amdgpu_buffer_rsrc_t gi_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gi + dir * gi_ds), 0, B * T * 3 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(Y, 0, B * T * 2 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t yd_rsrc = __builtin_amdgcn_make_buffer_rsrc(y_drop ? y_drop : Y, 0, B * T * 2 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t dm_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(drop_mask ? drop_mask : Y), 0, B * T * 2 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t sv_rsrc = __builtin_amdgcn_make_buffer_rsrc(save ? save + dir * save_ds : Y, 0, save ? B * T * 4 * H * 4 : 0, XC_RSRC3);
    const unsigned gi_v = e_ok ? (unsigned)((row * T * 3 * H + unit0) * 4) : OOB;          // + (tau * 3H + g * H) * 4 (scalar)
    const unsigned y_v = e_ok ? (unsigned)((row * T * 2 * H + dir * H + unit0) * 4) : OOB;  // + tau * 2H * 4
    // gates are saved only for the batch rows that will be differentiated (of the three stacked generator calls of a GAN iteration one is:
    // two thirds of the 125 MB per layer were written for nobody); the other lanes' stores fall outside the descriptor and are dropped
    const unsigned sv_v = (e_ok && row >= save_row0 && row < save_row0 + save_rows) ? (unsigned)((row * T * 4 * H + unit0) * 4) : OOB;   // + (tau * 4H + j * H) * 4
    f32x4 gn[3] = {zero4, zero4, zero4}, mkn = zero4;                  // gi (and the dropout mask) of the NEXT step
    auto prefetch = [&](int st) {
        const int sl = st < T ? st : T - 1;                            // past the end: a valid address again, never used
        const int tl = dir ? T - 1 - sl : sl;
#pragma unroll
        for (int g = 0; g < 3; ++g) gn[g] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(gi_rsrc, gi_v, (tl * 3 * H + g * H) * 4, 0));
        if (drop_mask) mkn = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(dm_rsrc, y_v, tl * 2 * H * 4, 0));
    };
    if (epi_wave) prefetch(0);
    // (Round 4, measured and not kept: the step's output stores and the next gi prefetch issued one step LATE, behind the next step's fragment
    // loads, so that they drain under the MFMAs instead of sitting in front of those loads -- 161 -> 172 us per launch at B = 384, backward
    // 3.60 -> 3.83 us per step: vector-memory traffic inside the MFMA phase costs more than behind the flag.  profiles/r4_r_deferred_outputs_rejected.txt)
    for (int step = 0; step < T; ++step) {
        const int tau = dir ? T - 1 - step : step;
        f32x4 acc[MT][3];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            // the polling wave is one WITHOUT epilogue threads: its memory queue holds no output stores / prefetch loads of the previous
            // step, so the first poll returns after one L2 round trip (vmcnt counts in order: wave 0 would see its flag loads return
            // only behind its own stores)
            if (wave == POLL_WAVE && !aborted) {
                bool same = true;
                if constexpr (!(ABL & 1)) aborted = !xc_wait(cl_flags, CW, lane, gen + (unsigned)step, tmo, step, my_xcc, &same);
                if (step == 1 && lane == 0) same_xcd_s = same && XC_SAME_XCD_FAST;
            }
            __syncthreads();                        // the other waves load only behind the polling wave's barrier
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // compiler ordering only; every load below is sc1
            if (step == 1) fast = __builtin_amdgcn_readfirstlane(same_xcd_s) != 0;
            const int off0 = (dir * 2 + ((step - 1) & 1)) * slot_bytes;
            // fragments of at most three k-steps are in flight / live at a time: k-step p + 3 is requested into the registers of k-step p once
            // that one's MFMAs are issued (KS = 2: five k-steps per wave -- all fifteen fragments live at once spilled the resident weights)
            constexpr int FW = SPS > 3 ? 3 : SPS;
            bf16x8 fb[FW][MT][NS];
            auto load_kstep = [&](auto pc) {
                constexpr int p = decltype(pc)::value;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        // lane part in the VGPR offset (one register for all loads), wave-uniform part in the scalar offset
                        const int soff = off0 + ((s_beg + p) * 3 + s) * plane_bytes + (b0 + i * 16) * 64;
                        if constexpr (ABL & 2) fb[p % FW][i][s] = as_bf16x8(u32x4{(unsigned)soff, 1u, 2u, 3u});
                        else fb[p % FW][i][s] = as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, r16 * 64 + kq * 16, soff, 16));   // aux 16 = sc1
                    }
            };
            xc_static_for<FW>([&](auto pc) { if (decltype(pc)::value < s_cnt) load_kstep(pc); });
            xc_static_for<SPS>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if (p < s_cnt) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            if constexpr (ABL & 4) { u32x4 t_ = __builtin_bit_cast(u32x4, fb[p % FW][i][0]); asm volatile("" :: "v"(t_)); }
                            else acc[i][g] = xc_mma<NS>(wa[g][p], wlo[wave][g * SPS + p][lane], fb[p % FW][i], acc[i][g]);
                        }
                }
                if constexpr (p + FW < SPS) { if (p + FW < s_cnt) load_kstep(std::integral_constant<int, p + FW>{}); }
            });
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) red[ks][ut][i][g][lane] = acc[i][g];
        __syncthreads();

        f32x4 h = {0.f, 0.f, 0.f, 0.f}, r4 = zero4, z4 = zero4, n4 = zero4, hn4 = zero4, mk = zero4;
        if (epi_wave) {
            f32x4 gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                f32x4 s = red[0][e_ut][e_mt][g][e_lane];
                if constexpr (!(ABL & 8)) {
#pragma unroll
                    for (int q = 1; q < KS; ++q) s += red[q][e_ut][e_mt][g][e_lane];
                }
                if constexpr (NS == 2) s *= winv[g];
                gh[g] = s;
            }
            mk = mkn;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (ABL & 8) { h[q] = gn[0][q] + gh[0][q] + gh[1][q] + gh[2][q]; r4[q] = z4[q] = n4[q] = hn4[q] = h[q]; continue; }
                const float hn = gh[2][q] + bh[2][q];
                const float r = gate_sigmoid(gn[0][q] + bh[0][q] + gh[0][q]);
                const float z = gate_sigmoid(gn[1][q] + bh[1][q] + gh[1][q]);
                const float n = gate_tanh(gn[2][q] + r * hn);
                h[q] = (1.f - z) * n + z * hp[q];
                r4[q] = r; z4[q] = z; n4[q] = n; hn4[q] = hn;
            }
            hp = h;
            // publish h_t as three bf16 planes: 8-byte write-through stores; the 8 threads of a row cover its 64 bytes, a wave's 8 rows
            // four whole 128-byte lines per plane
            u32x2 pl[3];
            if constexpr (NS == 1) pl[0] = xc_rne4(h);
            else if constexpr (NS == 2) xc_split4_h2(h, H_SCALE, pl);
            else xc_split4(h, pl);
            const int woff = (dir * 2 + (step & 1)) * slot_bytes + m * 3 * plane_bytes + row * 64 + ug * 8;
            if constexpr (ABL & 16) { asm volatile("" :: "v"(pl[0]), "v"(woff)); }
            else if (fast) {                                   // one XCD: the line stays in the shared L2
#pragma unroll
                for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b64(pl[s], hx_rsrc, woff + s * plane_bytes, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b64(pl[s], hx_rsrc, woff + s * plane_bytes, 0, 16);
            }
        }
        if constexpr (!(ABL & 64)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY wave drains its stores before the flag
        __syncthreads();                                       // (also: `red` is free again)
        if (threadIdx.x == 0 && T >= 2) __hip_atomic_store(my_flag, xc_flag_value(gen + (unsigned)(step + 1), my_xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (epi_wave && !(ABL & 32)) {                         // outputs for later kernels and the next step's inputs: off the critical path
            if constexpr (!(ABL & 256)) __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(h), y_rsrc, y_v, tau * 2 * H * 4, 0);     // (lab: 256 = no output stores, 128 = no prefetch)
            if (y_drop) __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(h * mk), yd_rsrc, y_v, tau * 2 * H * 4, 0);   // fused inter-layer dropout
            if (save && !(ABL & 256)) {
                const int so = tau * 4 * H * 4;
                __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(r4), sv_rsrc, sv_v, so, 0);
                __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(z4), sv_rsrc, sv_v, so + H * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(n4), sv_rsrc, sv_v, so + 2 * H * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(hn4), sv_rsrc, sv_v, so + 3 * H * 4, 0);
            }
            if constexpr (!(ABL & 128)) prefetch(step + 1);
        }
    }
    if (m == 0 && threadIdx.x == 0 && T >= 2) __hip_atomic_store(cl_flags + XC_GEN_WORD, gen + (unsigned)(T + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------------ backward
// dh_tau = dy_tau + dh_next * z_next + dgh_next @ W_hh (contraction over the 3H gate rows).  A = W_hh^T rows [32 m, 32 m + 32)
// (pre-split, resident), B = the cluster's dgh tile of the previous step: block (gate g, member j) of the exchange buffer is k-step
// g * CW + j of the product.  One 16-row batch tile per workgroup; the 3 CW k-steps are dealt over EIGHT K slices (one per wave,
// each wave both unit tiles): 4 k-steps per wave keep the fragment registers at 64 (weights hi / mid) + 48 (dgh planes).
constexpr int XC_KSB = 8;             // K slices, backward
constexpr int XC_SPB8 = 4;            // k-steps per slice, backward: ceil(30 / 8)

template <int NS>
__global__ __launch_bounds__(512) void gru_seq_bwd_cluster_x3_kernel(
    const float* __restrict__ dY, const float* __restrict__ dy_mask, const float* __restrict__ Y, const float* __restrict__ save, long save_ds,
    const float* __restrict__ wt0, const float* __restrict__ wt1, float* __restrict__ dgi, float* __restrict__ dgh, long dg_ds,
    void* gx, unsigned* flags, unsigned* tmo, int B, int T, int H, int n_bt, int CW, int b_pad,
    float* __restrict__ gi_rmax, long rm_ds, float* __restrict__ gi_cmax, float* __restrict__ gh_cmax) {
    // gi_rmax / gi_cmax / gh_cmax (all or none; zeroed by the caller): magnitudes of what this launch writes, for the fp16 x 2 products that read it --
    // gi_rmax[dir * rm_ds + row] = largest |dgi| of batch row `row` over ALL its T steps (the input-gradient product scales a clip's rows by one power
    // of two), gi_cmax / gh_cmax[dir * 3H + c] = largest |dgi| / |dgh| of column c (the weight-gradient products' column scales).  Running maxima
    // in registers, behind each step's hand-off; atomic unsigned max ONCE, when the kernel leaves (a per-step atomic stayed in the memory queue
    // for ~0.8 us and the next step's drain before its flag waited for it: +27 us per launch, profiles/r6_m_timeline.txt)
    __shared__ __attribute__((aligned(16))) f32x4 red[XC_KSB][2][64];
    __shared__ __attribute__((aligned(16))) bf16x8 wlo[NS == 3 ? 8 : 1][2 * XC_SPB8][64];
    __shared__ int same_xcd_s;
    const unsigned my_xcc = xc_my_xcc();
    bool fast = false;                                   // see the forward kernel
    const int n_cl = 2 * n_bt;
    int cl, m;
    if (n_cl % 8 == 0) {
        cl = (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) / CW);
        m = (blockIdx.x / 8) % CW;
    } else {
        cl = blockIdx.x / CW;
        m = blockIdx.x % CW;
    }
    const int dir = cl / n_bt, bt = cl % n_bt;
    const float* wt = dir ? wt1 : wt0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: K-slice bounds and exchange offsets stay in SGPRs
    const int ks = wave;
    const int r16 = lane & 15, kq = lane >> 4;
    const int b0 = bt * 16;
    const int H3 = 3 * H;

    const int n_steps = 3 * CW;                      // k-steps of 32 gate rows: (gate, member) blocks
    const int s_base = n_steps / XC_KSB, s_rem = n_steps % XC_KSB;
    const int s_cnt = s_base + (ks < s_rem ? 1 : 0);
    const int s_beg = ks * s_base + (ks < s_rem ? ks : s_rem);

    // W_hh^T rows (unit 32 m + 16 u + r16), u = 0, 1: k-step q = (gate g, member j) covers gate rows g * H + 32 j + 8 kq .. +7
    bf16x8 wa[2][XC_SPB8][2];
    // fp16 x 2 (NS == 2): every W_hh^T row (unit) scaled by its own power of two over its 3H columns -- they are spread over the eight K-slice waves,
    // partial maxima meet in LDS; the exchanged operand (this step's dgh tile, a GRADIENT: no bound known) carries one power-of-two scale per
    // (batch row, member) -- the 96 values a member's epilogue threads hold for a row.  Its exponent travels INSIDE the lo plane (xc_take_exp;
    // a separate word per block cost 24 us per launch in extra line requests, profiles/r6_bb): a k-step's product is scaled back by its
    // block's inverse before it joins the wave's accumulator.
    __shared__ unsigned wmaxb[XC_KSB][2][16];
    float wscb[2] = {1.f, 1.f};
    if constexpr (NS == 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = m * XC_UNITS + u * 16 + r16;
            unsigned mx = 0u;
#pragma unroll
            for (int p = 0; p < XC_SPB8; ++p) {
                const int q = s_beg + p;
                const int g = q / CW, jm = q - g * CW;
                const int ku = 32 * jm + 8 * kq;
                const bool ok = p < s_cnt && j < H;
                const float* src = wt + (long)(ok ? j : 0) * H3 + (ok ? g * H + ku : 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned v = (ok && ku + i < H) ? __float_as_uint(src[i]) & 0x7fffffffu : 0u;
                    mx = mx > v ? mx : v;
                }
            }
            unsigned w = (unsigned)__shfl_xor((int)mx, 16, 64); mx = mx > w ? mx : w;
            w = (unsigned)__shfl_xor((int)mx, 32, 64); mx = mx > w ? mx : w;
            if (kq == 0) wmaxb[ks][u][r16] = mx;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            unsigned mx = wmaxb[0][u][r16];
#pragma unroll
            for (int k = 1; k < XC_KSB; ++k) { const unsigned v = wmaxb[k][u][r16]; mx = mx > v ? mx : v; }
            wscb[u] = h2_scale_of_exp(h2_exp_of_bits(mx));
        }
    }
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < XC_SPB8; ++p) {
                const int j = m * XC_UNITS + u * 16 + r16;
                const int q = s_beg + p;
                const int g = q / CW, jm = q - g * CW;
                const int ku = 32 * jm + 8 * kq;                   // unit index inside the gate
                const bool ok = p < s_cnt && j < H;
                const float* src = wt + (long)(ok ? j : 0) * H3 + (ok ? g * H + ku : 0);
                const f32x4 a = (ok && ku < H) ? *reinterpret_cast<const f32x4*>(src) : z;
                const f32x4 b = (ok && ku + 4 < H) ? *reinterpret_cast<const f32x4*>(src + 4) : z;
                bf16x8 pl[3];
                if constexpr (NS == 1) { pl[0] = xc_rne8(a, b); pl[1] = pl[2] = pl[0]; }
                else if constexpr (NS == 2) xc_split8_h2(a, b, wscb[u], pl);
                else xc_split8(a, b, pl);
                wa[u][p][0] = pl[0]; wa[u][p][1] = pl[1];
                if constexpr (NS == 3) wlo[wave][u * XC_SPB8 + p][lane] = pl[2];
            }
    }
    __syncthreads();
    const int e = threadIdx.x;
    const bool epi = e < 128;
    const int row_l = e >> 3, ug = e & 7;
    const int e_lane = (row_l & 15) + 16 * (ug & 3), e_ut = ug >> 2;
    const int row = b0 + row_l;
    const int unit0 = m * XC_UNITS + 4 * ug;
    const bool e_ok = epi && row < B && unit0 < H;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 winvb = {1.f, 1.f, 1.f, 1.f};                // fp16 x 2: 1 / scale of W_hh^T's rows = this thread's four units
    if constexpr (NS == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = (4 * ug + q) & 15;
            unsigned mx = wmaxb[0][e_ut][rr];
#pragma unroll
            for (int k = 1; k < XC_KSB; ++k) { const unsigned v = wmaxb[k][e_ut][rr]; mx = mx > v ? mx : v; }
            winvb[q] = h2_inv_of_exp(h2_exp_of_bits(mx));
        }
    }
    f32x4 dhz = zero;                                  // dh * z of the step processed before

    const int plane_bytes = b_pad * 64;
    const int slot_bytes = 3 * CW * 3 * plane_bytes;
    __amdgpu_buffer_rsrc_t gx_rsrc = __builtin_amdgcn_make_buffer_rsrc(gx, 0, 4 * slot_bytes, XC_RSRC3);
    gu32x* my_flag = (gu32x*)(flags + cl * XC_FLAG_STRIDE + m);
    gu32x* cl_flags = (gu32x*)(flags + cl * XC_FLAG_STRIDE);
    bool aborted = false;
    // Flag words are never zeroed between launches: member m publishes gen + step + 1, where gen is the cluster's GENERATION word
    // (word 15 of its flag line), read here by every member and advanced by T + 1 by member 0 when it leaves.  Member 0 can only
    // finish step T - 1 after every member has published step T - 2, i.e. after every member has read gen (T >= 2; with T == 1
    // nothing is published or advanced -- there is no consumer).  Every value a launch leaves behind is <= gen + T < the next gen,
    // so stale words never satisfy a wait; the workspace only has to be zero ONCE, before its first use.
    const unsigned gen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cl_flags + XC_GEN_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));

    // per-step streams through bounds-checked buffer instructions, requested one step ahead (see the forward kernel)
    constexpr unsigned OOB = 0x80000000u;
    const int epi_wave = wave < 2;
    __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, B * T * 2 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t dm_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy_mask ? dy_mask : dY), 0, B * T * 2 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Y), 0, B * T * 2 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t sv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(save + dir * save_ds), 0, B * T * 4 * H * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t gi_rsrc = __builtin_amdgcn_make_buffer_rsrc(dgi + dir * dg_ds, 0, B * T * H3 * 4, XC_RSRC3);
    __amdgpu_buffer_rsrc_t gh_rsrc = __builtin_amdgcn_make_buffer_rsrc(dgh + dir * dg_ds, 0, B * T * H3 * 4, XC_RSRC3);
    const unsigned y_v = e_ok ? (unsigned)((row * T * 2 * H + dir * H + unit0) * 4) : OOB;  // + tau * 2H * 4
    const unsigned sv_v = e_ok ? (unsigned)((row * T * 4 * H + unit0) * 4) : OOB;           // + (tau * 4H + j * H) * 4
    const unsigned dg_v = e_ok ? (unsigned)((row * T * H3 + unit0) * 4) : OOB;              // + (tau * 3H + g * H) * 4
    f32x4 dy = zero, dm = zero, r = zero, z = zero, n = zero, hn = zero, hp = zero;         // operands of the NEXT step's cell
    unsigned rmx = 0u;                                 // running maximum of this thread's dgi values over all steps (its batch row)
    unsigned cmx[4][4];                                // running column maxima of this thread's four units: gates r, z, n (dgi) and n * r (dgh's third)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) cmx[g][q] = 0u;
    auto prefetch = [&](int st) {
        const int sl = st < T ? st : T - 1;
        const int tl = dir ? sl : T - 1 - sl;
        const int tp = dir ? tl + 1 : tl - 1;                       // producer of h_prev for this cell
        const bool has_prev = tp >= 0 && tp < T;
        dy = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(dy_rsrc, y_v, tl * 2 * H * 4, 0));
        if (dy_mask) dm = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(dm_rsrc, y_v, tl * 2 * H * 4, 0));
        const int so = tl * 4 * H * 4;
        r = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(sv_rsrc, sv_v, so, 0));
        z = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(sv_rsrc, sv_v, so + H * 4, 0));
        n = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(sv_rsrc, sv_v, so + 2 * H * 4, 0));
        hn = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(sv_rsrc, sv_v, so + 3 * H * 4, 0));
        hp = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(y_rsrc, has_prev ? y_v : OOB, (has_prev ? tp : 0) * 2 * H * 4, 0));   // h_prev = 0 at the sequence end
    };
    if (epi_wave) prefetch(0);
    for (int step = 0; step < T; ++step) {
        const int tau = dir ? step : T - 1 - step;
        f32x4 acc[2] = {zero, zero};
        if (step > 0) {
            if (wave == XC_POLL_WAVE && !aborted) {
                bool same;
                aborted = !xc_wait(cl_flags, CW, lane, gen + (unsigned)step, tmo, step, my_xcc, &same);
                if (step == 1 && lane == 0) same_xcd_s = same && XC_SAME_XCD_FAST;
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (step == 1) fast = __builtin_amdgcn_readfirstlane(same_xcd_s) != 0;
            const int off0 = (dir * 2 + ((step - 1) & 1)) * slot_bytes + b0 * 64;
            bf16x8 fb[XC_SPB8][NS];
            if constexpr (NS == 2) {
                // no branch around a k-step this wave does not own (p >= s_cnt): its loads go out of bounds (zeros) and its product is scaled by 0,
                // so all eight loads are in flight together and the exponent decode overlaps the first products
                float binv[XC_SPB8];                           // 1 / scale of (this lane's batch row, the k-step's member)
#pragma unroll
                for (int p = 0; p < XC_SPB8; ++p) {
                    const unsigned vo = p < s_cnt ? (unsigned)(r16 * 64 + kq * 16) : 0x80000000u;
                    const int so = off0 + (p < s_cnt ? (s_beg + p) * 3 : 0) * plane_bytes;
#pragma unroll
                    for (int s = 0; s < NS; ++s) fb[p][s] = as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(gx_rsrc, vo, so + s * plane_bytes, 16));
                }
#pragma unroll
                for (int p = 0; p < XC_SPB8; ++p) {
                    const float bi = xc_take_exp(fb[p][1]);
                    binv[p] = p < s_cnt ? bi : 0.f;
                }
#pragma unroll
                for (int p = 0; p < XC_SPB8; ++p)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[u] += xc_mma<NS>(wa[u][p], wa[u][p][1], fb[p], zero) * binv[p];
            } else {
#pragma unroll
                for (int p = 0; p < XC_SPB8; ++p)
                    if (p < s_cnt) {
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            fb[p][s] = as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(gx_rsrc, r16 * 64 + kq * 16, off0 + ((s_beg + p) * 3 + s) * plane_bytes, 16));
                    }
#pragma unroll
                for (int p = 0; p < XC_SPB8; ++p) {
                    if (p < s_cnt) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = xc_mma<NS>(wa[u][p], wlo[wave][u * XC_SPB8 + p][lane], fb[p], acc[u]);
                    }
                }
            }
        }
        red[ks][0][lane] = acc[0];
        red[ks][1][lane] = acc[1];
        __syncthreads();

        f32x4 g_r = zero, g_z = zero, g_n = zero, g_nr = zero;
        if (epi_wave) {
            f32x4 s = red[0][e_ut][e_lane];
#pragma unroll
            for (int q = 1; q < XC_KSB; ++q) s += red[q][e_ut][e_lane];
            if constexpr (NS == 2) s *= winvb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dyq = dy_mask ? dy[q] * dm[q] : dy[q];
                const float dh = dyq + s[q] + dhz[q];              // step 0: s = 0 (no product), dhz = 0
                const float dn = dh * (1.f - z[q]) * (1.f - n[q] * n[q]);
                const float dz = dh * (hp[q] - n[q]) * z[q] * (1.f - z[q]);
                const float dr = dn * hn[q] * r[q] * (1.f - r[q]);
                dhz[q] = dh * z[q];
                g_r[q] = dr; g_z[q] = dz; g_n[q] = dn; g_nr[q] = dn * r[q];
            }
            // publish this step's dgh tile: blocks (gate 0..2, member m), three bf16 planes each
            const int woff = (dir * 2 + (step & 1)) * slot_bytes + row * 64 + ug * 8;
            u32x2 pl[3][3];
            if constexpr (NS == 1) { pl[0][0] = xc_rne4(g_r); pl[1][0] = xc_rne4(g_z); pl[2][0] = xc_rne4(g_nr); }
            else if constexpr (NS == 2) {
                // one power of two for the row's 96 values of this member: over this thread's twelve, then over the row's eight threads
                unsigned mx = 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned b0_ = __float_as_uint(g_r[q]) & 0x7fffffffu, b1_ = __float_as_uint(g_z[q]) & 0x7fffffffu, b2_ = __float_as_uint(g_nr[q]) & 0x7fffffffu;
                    const unsigned t3 = b0_ > b1_ ? (b0_ > b2_ ? b0_ : b2_) : (b1_ > b2_ ? b1_ : b2_);
                    mx = mx > t3 ? mx : t3;
                }
                mx = xc_max8(mx);
                const unsigned eb = (unsigned)h2_exp_of_bits(mx) + XC_GX_EXP_SHIFT;
                const float sc = h2_scale_of_exp((int)eb);
                xc_split4_h2(g_r, sc, pl[0]); xc_split4_h2(g_z, sc, pl[1]); xc_split4_h2(g_nr, sc, pl[2]);
                // the exponent rides in bit 14 of the lo plane's values (clear by the choice of scale): the consumer's 16-byte chunk is two threads'
                // values = dwords c = 2 (ug & 1) + k, dword c carries bit c (low half) and bit 4 + c (high half)
                const int c0 = 2 * (ug & 1);
                const unsigned in0 = (((eb >> c0) & 1u) << 14) | (((eb >> (4 + c0)) & 1u) << 30);
                const unsigned in1 = (((eb >> (c0 + 1)) & 1u) << 14) | (((eb >> (5 + c0)) & 1u) << 30);
#pragma unroll
                for (int g = 0; g < 3; ++g) { pl[g][1][0] |= in0; pl[g][1][1] |= in1; }
            }
            else { xc_split4(g_r, pl[0]); xc_split4(g_z, pl[1]); xc_split4(g_nr, pl[2]); }
            if (fast) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b64(pl[g][s], gx_rsrc, woff + ((g * CW + m) * 3 + s) * plane_bytes, 0, 0);
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b64(pl[g][s], gx_rsrc, woff + ((g * CW + m) * 3 + s) * plane_bytes, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && T >= 2) __hip_atomic_store(my_flag, xc_flag_value(gen + (unsigned)(step + 1), my_xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (epi_wave) {
            const int go = tau * H3 * 4;
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_r), gi_rsrc, dg_v, go, 0);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_z), gi_rsrc, dg_v, go + H * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_n), gi_rsrc, dg_v, go + 2 * H * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_r), gh_rsrc, dg_v, go, 0);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_z), gh_rsrc, dg_v, go + H * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(g_nr), gh_rsrc, dg_v, go + 2 * H * 4, 0);
            prefetch(step + 1);
            if (gi_rmax && e_ok) {                     // (behind the hand-off and the next step's requests: off the dependent chain)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned br = __float_as_uint(g_r[q]) & 0x7fffffffu, bz = __float_as_uint(g_z[q]) & 0x7fffffffu;
                    const unsigned bn = __float_as_uint(g_n[q]) & 0x7fffffffu, bnr = __float_as_uint(g_nr[q]) & 0x7fffffffu;
                    cmx[0][q] = cmx[0][q] > br ? cmx[0][q] : br;
                    cmx[1][q] = cmx[1][q] > bz ? cmx[1][q] : bz;
                    cmx[2][q] = cmx[2][q] > bn ? cmx[2][q] : bn;
                    cmx[3][q] = cmx[3][q] > bnr ? cmx[3][q] : bnr;
                    const unsigned t3 = br > bz ? (br > bn ? br : bn) : (bz > bn ? bz : bn);
                    rmx = rmx > t3 ? rmx : t3;
                }
            }
        }
    }
    if (m == 0 && threadIdx.x == 0 && T >= 2) __hip_atomic_store(cl_flags + XC_GEN_WORD, gen + (unsigned)(T + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gi_rmax && epi_wave) {
        // the batch row's maximum: over the eight threads of the row (consecutive lanes), one atomic per row and member
        {
            unsigned v = rmx;
#pragma unroll
            for (int o = 1; o <= 4; o <<= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
            if ((lane & 7) == 0 && epi && row < B) atomicMax(reinterpret_cast<unsigned*>(gi_rmax) + dir * rm_ds + row, v);
        }
        // column maxima: over the wave's eight rows (lanes 8 apart hold the same units), then one atomic per column and wave
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned v = cmx[g][q];
#pragma unroll
                for (int o = 8; o <= 32; o <<= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
                cmx[g][q] = v;
            }
        if (lane < 8 && unit0 < H) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = unit0 + q;
                unsigned* ci = reinterpret_cast<unsigned*>(gi_cmax) + dir * H3;
                unsigned* ch = reinterpret_cast<unsigned*>(gh_cmax) + dir * H3;
                atomicMax(ci + c, cmx[0][q]); atomicMax(ci + H + c, cmx[1][q]); atomicMax(ci + 2 * H + c, cmx[2][q]);
                atomicMax(ch + c, cmx[0][q]); atomicMax(ch + H + c, cmx[1][q]); atomicMax(ch + 2 * H + c, cmx[3][q]);
            }
        }
    }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_get_math_mode(void);

// fp32-accurate mode: fp16 x 2 operands in both recurrences; TG_GRU_H2 is a mask for A/B timing (1 = forward, 2 = backward; 0 = bf16 x 3 in both)
static int gru_h2_on() {
    static const int h2 = [] { const char* e = getenv("TG_GRU_H2"); return e ? atoi(e) : 3; }();
    return h2;
}

// exchange-buffer bytes of the two kernels (the flag block in front of it is laid out by gru_cluster.hip)
int64_t tg_gru_x3_fwd_exchange_bytes(int b_pad, int cw) { return 4LL * cw * 3 * b_pad * 64; }
int64_t tg_gru_x3_bwd_exchange_bytes(int b_pad, int cw) { return 4LL * 3 * cw * 3 * b_pad * 64; }

int tg_gru_x3_fwd_launch(int mt, const float* gi, long gi_ds, const float* w0, const float* w1, const float* b0, const float* b1, float* y,
                         float* save, long save_ds, const float* drop_mask, float* y_drop, void* hx, unsigned* flags, unsigned* tmo, int B, int T,
                         int H, int n_bt, int cw, int b_pad, int save_row0, int save_rows, hipStream_t s) {
    dim3 grid(2 * n_bt * cw);
    const bool bf16 = tg_get_math_mode() == 1;            // plain bf16 operands: one MFMA per product, one exchange plane
#define TG_XF(MT_, NS_) hipLaunchKernelGGL((gru_seq_fwd_cluster_x3_kernel<MT_, NS_>), grid, dim3(512), 0, s, gi, gi_ds, w0, w1, b0, b1, y, save, \
                                           save_ds, drop_mask, y_drop, hx, flags, tmo, B, T, H, n_bt, cw, b_pad, save_row0, save_rows)
#ifdef TG_LAB_ABL
    {
        const char* ae = getenv("TG_XC_ABL");
        const int abl = ae ? atoi(ae) : 0;
        if (abl && mt == 2 && !bf16) {
#define TG_XF_ABL(A_) case A_: hipLaunchKernelGGL((gru_seq_fwd_cluster_x3_kernel<2, 2, A_>), grid, dim3(512), 0, s, gi, gi_ds, w0, w1, b0, b1, y, save, save_ds, \
                                           drop_mask, y_drop, hx, flags, tmo, B, T, H, n_bt, cw, b_pad, save_row0, save_rows); return check_launch("tg_gru_forward_cluster(x3, ablated)");
            switch (abl) {
                TG_XF_ABL(1) TG_XF_ABL(2) TG_XF_ABL(4) TG_XF_ABL(8) TG_XF_ABL(16) TG_XF_ABL(32) TG_XF_ABL(64) TG_XF_ABL(80) TG_XF_ABL(3) TG_XF_ABL(7) TG_XF_ABL(15) TG_XF_ABL(31) TG_XF_ABL(127) TG_XF_ABL(6) TG_XF_ABL(96) TG_XF_ABL(81) TG_XF_ABL(128) TG_XF_ABL(256)
                default: break;
            }
#undef TG_XF_ABL
        }
    }
#endif
    // fp32-accurate mode: fp16 x 2 operands (3 matrix instructions per product: h * 2^14 and per-row scaled W_hh as hi / lo fp16 planes) unless
    // TG_GRU_H2=0 asks for bf16 x 3 (6; A/B timing)
    const int h2 = gru_h2_on() & 1;
    if (mt == 1) { if (bf16) TG_XF(1, 1); else if (h2) TG_XF(1, 2); else TG_XF(1, 3); }
    else { if (bf16) TG_XF(2, 1); else if (h2) TG_XF(2, 2); else TG_XF(2, 3); }
#undef TG_XF
    return check_launch("tg_gru_forward_cluster(x3)");
}

int tg_gru_x3_bwd_launch(const float* dy, const float* dy_mask, const float* y, const float* save, long save_ds, const float* wt0, const float* wt1, float* dgi,
                         float* dgh, long dg_ds, void* gx, unsigned* flags, unsigned* tmo, int B, int T, int H, int n_bt, int cw, int b_pad,
                         float* gi_rmax, long rm_ds, float* gi_cmax, float* gh_cmax,
                         hipStream_t s) {
    if (tg_get_math_mode() == 1)
        hipLaunchKernelGGL(gru_seq_bwd_cluster_x3_kernel<1>, dim3(2 * n_bt * cw), dim3(512), 0, s, dy, dy_mask, y, save, save_ds, wt0, wt1, dgi, dgh, dg_ds, gx,
                           flags, tmo, B, T, H, n_bt, cw, b_pad, gi_rmax, rm_ds, gi_cmax, gh_cmax);
    else if (gru_h2_on() & 2)
        hipLaunchKernelGGL(gru_seq_bwd_cluster_x3_kernel<2>, dim3(2 * n_bt * cw), dim3(512), 0, s, dy, dy_mask, y, save, save_ds, wt0, wt1, dgi, dgh, dg_ds, gx,
                           flags, tmo, B, T, H, n_bt, cw, b_pad, gi_rmax, rm_ds, gi_cmax, gh_cmax);
    else
        hipLaunchKernelGGL(gru_seq_bwd_cluster_x3_kernel<3>, dim3(2 * n_bt * cw), dim3(512), 0, s, dy, dy_mask, y, save, save_ds, wt0, wt1, dgi, dgh, dg_ds, gx,
                           flags, tmo, B, T, H, n_bt, cw, b_pad, gi_rmax, rm_ds, gi_cmax, gh_cmax);
    return check_launch("tg_gru_backward_cluster(x3)");
}
