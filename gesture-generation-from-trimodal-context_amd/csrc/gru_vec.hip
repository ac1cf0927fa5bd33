// Persistent GRU recurrence for a HANDFUL of sequences (B <= 4, inference only: no saved gates, no dropout) -- the single-utterance synthesis
// window of scripts/synthesize.py:131-160 (generate_gestures feeds the generator one 34-frame window at a time), whose four H = 300 layers are
// 4 x 34 dependent steps: multimodal_context_net.py:155 (nn.GRU, bidirectional, batch_first).
//
// The cluster kernels of gru_cluster_x3.hip pay, per step, a flag round trip on top of the data round trip, a 16-row MFMA tile (two fp16 planes,
// 20 KB per direction) and an LDS reduction over eight K-slice waves: 3.4 us per step at ANY batch <= 16 (profiles/r6_z_decode_b1.txt).  With
// one to four rows none of that is needed:
//   * h_t travels as plain fp32 (1.2 KB per direction and row) and IS its own flag: every exchange word is reset to a sentinel (all ones -- a
//     NaN pattern no |h| < 1 can take) before it is written again, and a consumer polls the words it needs until none is the sentinel.  One memory
//     round trip per step instead of two, no flag words, no generation numbers.
//   * the product is fp32 FMAs on fp32 weights resident in registers (60 per thread): 2 directions x 10 workgroups ("members", 32 hidden units
//     each) x 512 threads; a thread owns (unit, K slice of 20): three 20-term dot products per row, summed over the 16 lanes of a DPP row.
//     Lane `row` of a unit's 16 then owns (unit, row): gates, h_prev, the publish.  A workgroup fetches h_{t-1} once (thread t polls word t)
//     and shares it through LDS: one barrier per step.
// Measured (tools/gru_vec_probe.py, profiles/r6_bh_gru_vec.txt): 39 us per launch at B = 1 (1.16 us per step), 50 at B = 2, 73 at B = 4, against
// 110 for the cluster kernel; single-utterance window 724 -> 445 us.
// Exchange slots rotate over THREE steps (slot s % 3 holds h_s).  A member that has read all of h_{s-1} knows every member has finished reading
// h_{s-2} (they published h_{s-1} after it), so it resets ITS OWN words of slot (s - 2) % 3 -- the slot it will write at step s + 1 -- and drains
// that store (s_waitcnt vmcnt(0)) before it publishes h_s: whoever sees h_s also sees the reset.  Two buffers alternate between LAUNCHES (a
// launch counter in the workspace, advanced by one member when it leaves): a launch resets its own words of the other buffer when it starts,
// so every launch finds its buffer all-sentinel whatever the one before left behind (the workspace is born all ones, ops._gru_vec_ws).
// Members sit on block ids that are multiples of 8 (one XCD as the dispatcher is observed to deal them; the other seven eighths of the grid
// leave at once): when the ids published with h_0 confirm it, stores become plain (the line stays in that XCD's L2) instead of write-through;
// loads are sc1 (L1 bypass) either way -- the same instruction flavours as the cluster kernels' hand-off.
// Every spin is bounded; a timeout sets the workspace's sticky word (ops.check_async_errors raises and refills the workspace).
#include "common.hpp"

namespace tg {

typedef __attribute__((address_space(1))) unsigned vgu32;
constexpr unsigned VEC_SENT = 0xFFFFFFFFu;
constexpr int VEC_ROWS = 4;           // batch rows at most
constexpr int VEC_KSL = 20;           // a lane's K slice at most (H <= 320 over 16 lanes)
constexpr int VEC_PAD = 16;           // words behind a row's H values: the members' hello words (row 0 of a direction)
constexpr int VEC_HDR = 32;           // workspace words in front of the exchange buffers: [0, 16) timeout block, [16] launch counter
constexpr unsigned VEC_SPIN_LIMIT = 1u << 22;
constexpr unsigned VEC_RSRC3 = 0x00020000u;

__device__ __forceinline__ unsigned vec_my_xcc() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
// sum over the 16 lanes of a DPP row; every lane ends with the total
__device__ __forceinline__ float vec_row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1, 0, 3, 2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2, 3, 0, 1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}

template <int RB>
__global__ __launch_bounds__(512) void gru_seq_fwd_vec_kernel(
    const float* __restrict__ gi, long gi_ds, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ Y, unsigned* ws, int B, int T, int H, int CW) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][RB][16 * VEC_KSL];
    if (blockIdx.x & 7) return;
    const int id = blockIdx.x >> 3;
    const int dir = id / CW, m = id - dir * CW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ksl = lane & 15;
    const int unit = 32 * m + wave * 4 + (lane >> 4);
    const bool u_ok = unit < H;
    const int ksz = ((H + 63) >> 6) << 2;                 // K-slice length: a multiple of 4, 16 slices cover H
    const int k0 = ksl * ksz;
    const float* whh = dir ? whh1 : whh0;
    const float* bhh = dir ? bhh1 : bhh0;

    // W_hh rows (gate g, unit), columns of this lane's K slice: fp32, resident
    float w[3][VEC_KSL];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < VEC_KSL / 4; ++j) {
            const int k = k0 + 4 * j;
            const bool ok = u_ok && 4 * j < ksz && k < H;                             // (H % 4 == 0: a group of four is in or out as one)
            const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(whh + (long)(g * H + unit) * H + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) w[g][4 * j + i] = v[i];
        }
    const int row = ksl;                                   // lanes 0 .. RB - 1 of a unit's sixteen own (unit, row)
    const bool own = row < RB && u_ok;
    const bool r_ok = own && row < B;
    float bh[3] = {0.f, 0.f, 0.f};
    if (own) {
#pragma unroll
        for (int g = 0; g < 3; ++g) bh[g] = bhh[g * H + unit];
    }

    const int LD = H + VEC_PAD;
    const int slot_w = 2 * VEC_ROWS * LD;                  // words per slot (both directions)
    const int buf_w = 3 * slot_w;
    const unsigned launch = __builtin_amdgcn_readfirstlane(__hip_atomic_load((vgu32*)ws + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int P = (int)(launch & 1u);
    __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws + VEC_HDR, 0, 2 * buf_w * 4, VEC_RSRC3);
    const int my_w = (dir * VEC_ROWS + row) * LD + unit;   // this lane's word inside a slot (own lanes)
    const int hello_w = dir * VEC_ROWS * LD + H;           // + member
    const unsigned my_xcc = vec_my_xcc();
    bool fast = false, aborted = false;
    if (T >= 2) {
        // the other buffer: this member's words back to the sentinel for the launch after this one (all four rows, whatever RB is now)
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            if (ksl < VEC_ROWS && u_ok) __builtin_amdgcn_raw_buffer_store_b32(VEC_SENT, x_rsrc, (dir * VEC_ROWS + ksl) * LD * 4 + unit * 4, ((1 - P) * buf_w + sl * slot_w) * 4, 16);
            if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(VEC_SENT, x_rsrc, (hello_w + m) * 4, ((1 - P) * buf_w + sl * slot_w) * 4, 16);
        }
    }

    // the step's gi values of (unit, row), requested one step ahead
    const float* gi_p = gi + dir * gi_ds + (long)(r_ok ? row : 0) * T * 3 * H + (r_ok ? unit : 0);
    float* y_p = Y + (long)(r_ok ? row : 0) * T * 2 * H + dir * H + (r_ok ? unit : 0);
    float gx[3] = {0.f, 0.f, 0.f};
    auto prefetch = [&](int st) {
        const int tl = dir ? T - 1 - st : st;
        if (r_ok) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gx[g] = gi_p[(long)tl * 3 * H + g * H];
        }
    };
    prefetch(0);
    float hp = 0.f;
    for (int step = 0; step < T; ++step) {
        const int tau = dir ? T - 1 - step : step;
        float acc[3][RB];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[g][r] = 0.f;
        if (step > 0) {
            const int so = (P * buf_w + ((step - 1) % 3) * slot_w) * 4;
            // the workgroup fetches a row ONCE: thread t polls word t of every row (a wave: 256 contiguous bytes) and leaves it in LDS; with every
            // thread polling its own K slice the 160 waves of the launch sent 2 800 requests per row and round to the row's ten lines (1.65 us
            // per row and step, profiles/r6_bh_gru_vec.txt)
            const int t = threadIdx.x;
            float hw[RB];
            unsigned hello = my_xcc + 1u;
            unsigned spins = 0;
            for (;;) {
#pragma unroll
                for (int r = 0; r < RB; ++r) hw[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, t * 4, so + (dir * VEC_ROWS + r) * LD * 4, 16));
                if (step == 1 && lane < CW) hello = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (hello_w + lane) * 4, so, 16);
                bool ok = hello != VEC_SENT;
#pragma unroll
                for (int r = 0; r < RB; ++r) ok = ok && !(t < H && __float_as_uint(hw[r]) == VEC_SENT);
                if (__all(ok) || aborted) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > VEC_SPIN_LIMIT) {            // wave-uniform
                    if (lane == 0) {
                        __hip_atomic_store((vgu32*)ws + 1, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((vgu32*)ws + 2, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((vgu32*)ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    aborted = true;
                    break;
                }
            }
            if (t < 16 * VEC_KSL) {
#pragma unroll
                for (int r = 0; r < RB; ++r) hbuf[step & 1][r][t] = t < H ? hw[r] : 0.f;
            }
            // (two LDS buffers: a wave that is through with this step may already be writing the next step's words while another still reads these)
            __syncthreads();
            if (step == 1) fast = __all(hello == my_xcc + 1u) && !aborted;
            // every member has published h_{step-1}, so nobody reads h_{step-2} any more: its slot -- the one step + 1 writes -- back to the sentinel
            if (step >= 2) {
                const int ro = (P * buf_w + ((step + 1) % 3) * slot_w) * 4;
                if (fast) {
                    if (own) __builtin_amdgcn_raw_buffer_store_b32(VEC_SENT, x_rsrc, my_w * 4, ro, 0);
                    if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(VEC_SENT, x_rsrc, (hello_w + m) * 4, ro, 0);
                } else {
                    if (own) __builtin_amdgcn_raw_buffer_store_b32(VEC_SENT, x_rsrc, my_w * 4, ro, 16);
                    if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(VEC_SENT, x_rsrc, (hello_w + m) * 4, ro, 16);
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < VEC_KSL / 4; ++j) {
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(&hbuf[step & 1][r][k0 + 4 * j]);      // (words >= H are zero, and so are their weights)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) acc[g][r] = __builtin_fmaf(w[g][4 * j + i], hv[i], acc[g][r]);
                }
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[g][r] = vec_row_sum(acc[g][r]);
        }
        float gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float v = acc[g][0];
#pragma unroll
            for (int r = 1; r < RB; ++r) v = row == r ? acc[g][r] : v;
            gh[g] = v;
        }
        // the cell (multimodal_context_net.py:155 -> torch.nn.GRU): same operation order as the cluster kernel's epilogue
        const float hn = gh[2] + bh[2];
        const float rg = gate_sigmoid(gx[0] + bh[0] + gh[0]);
        const float zg = gate_sigmoid(gx[1] + bh[1] + gh[1]);
        const float ng = gate_tanh(gx[2] + rg * hn);
        const float h = (1.f - zg) * ng + zg * hp;
        hp = h;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the reset above has landed before h_t can be seen
        if (step + 1 < T) {
            const int po = (P * buf_w + (step % 3) * slot_w) * 4;
            if (fast) {
                if (own) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), x_rsrc, my_w * 4, po, 0);
            } else {
                if (own) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), x_rsrc, my_w * 4, po, 16);
                if (step == 0 && threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(my_xcc + 1u, x_rsrc, (hello_w + m) * 4, po, 16);
            }
        }
        if (r_ok) y_p[(long)tau * 2 * H] = h;
        if (step + 1 < T) prefetch(step + 1);
    }
    // the next launch takes the other buffer.  Member 0 of direction 0 can only be here after every member has published step T - 2, i.e. has
    // read the counter (T >= 2; with T == 1 nothing was exchanged or reset and the counter stays)
    if (id == 0 && threadIdx.x == 0 && T >= 2) __hip_atomic_store((vgu32*)ws + 16, launch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tg

using namespace tg;

static int vec_cus() {                  // CU count of the current device (queried once per device; 0: no usable device)
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 0; }
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 0; }
        cached[dev] = n > 0 ? n : -1;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}

// B rows of an inference forward this kernel takes: 1 .. 4 sequences, 64 < H <= 320 (H % 4 == 0), and a device on which the 2 * ceil(H / 32)
// members are co-resident beside nothing else of this stream (a whole MI355X: 256 CUs)
extern "C" int32_t tg_gru_vec_supported(int32_t B, int32_t H) {
    if (B < 1 || B > VEC_ROWS || H <= 64 || H > 16 * VEC_KSL || H % 4 != 0) return 0;
    return vec_cus() >= 128;
}

extern "C" int64_t tg_gru_vec_ws_bytes(int32_t H) { return (int64_t)(VEC_HDR + 2 * 3 * 2 * VEC_ROWS * (H + VEC_PAD)) * 4; }
extern "C" int32_t tg_gru_vec_ws_header_bytes(void) { return VEC_HDR * 4; }

extern "C" int tg_gru_forward_vec(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev, const float* b_hh_fwd,
                                  const float* b_hh_rev, float* y, void* ws, int64_t ws_bytes, int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(gi && w_hh_fwd && w_hh_rev && b_hh_fwd && b_hh_rev && y && ws, "tg_gru_forward_vec: null pointer");
    TG_REQUIRE(T > 0 && tg_gru_vec_supported(B, H), "tg_gru_forward_vec: unsupported shape B=%d H=%d", B, H);
    TG_REQUIRE(ws_bytes >= tg_gru_vec_ws_bytes(H), "tg_gru_forward_vec: workspace too small");
    TG_REQUIRE(aligned16(w_hh_fwd) && aligned16(w_hh_rev) && aligned16(ws), "tg_gru_forward_vec: W_hh and the workspace must be 16-byte aligned");
    const int cw = cdiv(H, 32);
    const dim3 grid(8 * 2 * cw);
    hipStream_t s = (hipStream_t)stream;
    unsigned* w = (unsigned*)ws;
    if (B == 1) hipLaunchKernelGGL(gru_seq_fwd_vec_kernel<1>, grid, dim3(512), 0, s, gi, (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, w, B, T, H, cw);
    else if (B == 2) hipLaunchKernelGGL(gru_seq_fwd_vec_kernel<2>, grid, dim3(512), 0, s, gi, (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, w, B, T, H, cw);
    else hipLaunchKernelGGL(gru_seq_fwd_vec_kernel<4>, grid, dim3(512), 0, s, gi, (long)gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, w, B, T, H, cw);
    return check_launch("tg_gru_forward_vec");
}
