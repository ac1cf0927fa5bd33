// Bidirectional GRU recurrence (nn.GRU semantics) on the f32 matrix cores.
//
// The recurrence is a chain of T dependent skinny GEMMs h_{t-1}[B,H] x W_hh^T[H,3H].  Each time step is ONE launch
// that covers both directions, every batch tile and every 16-wide slice of hidden units; the launch boundary is the
// grid-wide dependency (a kernel boundary costs ~1.5 us on MI355X, an in-kernel grid barrier 4-7 us), so there is no
// spin-wait anywhere and nothing that can dead-lock.  A workgroup owns [32 batch rows] x [16 hidden units, all three
// gates]: 8 waves = 2 row tiles x 4 K-slices, partial sums combined through LDS, then the gate maths is fused in the
// epilogue (sigmoid/tanh, h' = (1-z) n + z h) and h_t goes straight into the layer output y, which doubles as the
// state store.  Operands stream from L2 as 16-byte fragments (k-permuted MFMA feed, see gemm.hip).
#include "common.hpp"

namespace tg {

constexpr int GRU_MT = 2;   // 16-row tiles per workgroup
constexpr int GRU_KS = 4;   // K slices per workgroup
constexpr int GRU_THREADS = 64 * GRU_MT * GRU_KS;

// One MFMA operand fragment: 4 consecutive floats of a row, or zeros.
__device__ __forceinline__ f32x4 ld4(const float* p, bool ok) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? *reinterpret_cast<const f32x4*>(p) : z;
}

__global__ __launch_bounds__(GRU_THREADS) void gru_fwd_step_kernel(
    const float* __restrict__ gi, long gi_ds, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ Y, float* __restrict__ save,
    long save_ds, int B, int T, int H, int step) {
    __shared__ float red[GRU_KS][GRU_MT][3][4][64];
    const int dir = blockIdx.z;
    const int tau = dir ? T - 1 - step : step;
    const int tau_prev = dir ? tau + 1 : tau - 1;
    const bool has_prev = step > 0;
    const float* whh = dir ? whh1 : whh0;
    const float* bhh = dir ? bhh1 : bhh0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = wave % GRU_MT, ks = wave / GRU_MT;
    const int r16 = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * (GRU_MT * 16);

    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (has_prev) {
        const int b = b0 + mt * 16 + r16;
        const bool b_ok = b < B;
        const float* hrow = Y + ((long)(b_ok ? b : 0) * T + tau_prev) * (2 * H) + dir * H;
        const int j = j0 + r16;
        const bool j_ok = j < H;
        const float* wrow[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) wrow[g] = whh + (long)(g * H + (j_ok ? j : 0)) * H;
        for (int k0 = ks * 16; k0 < H; k0 += GRU_KS * 16) {
            const int k = k0 + 4 * kq;
            const bool inb = k < H;   // H % 4 == 0 (checked on the host)
            const f32x4 a = ld4(hrow + k, b_ok && inb);
            f32x4 w[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) w[g] = ld4(wrow[g] + k, j_ok && inb);
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v], w[g][v], acc[g], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[ks][mt][g][i][lane] = acc[g][i];
    __syncthreads();
    if (ks != 0) return;

    const int j = j0 + r16;
    if (j >= H) return;
    const float bh_r = bhh[j], bh_z = bhh[H + j], bh_n = bhh[2 * H + j];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = b0 + mt * 16 + kq * 4 + i;
        if (row >= B) continue;
        float gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float s = red[0][mt][g][i][lane];
#pragma unroll
            for (int q = 1; q < GRU_KS; ++q) s += red[q][mt][g][i][lane];
            gh[g] = s;
        }
        const float* gip = gi + dir * gi_ds + ((long)row * T + tau) * (3 * H);
        const float hp = has_prev ? Y[((long)row * T + tau_prev) * (2 * H) + dir * H + j] : 0.f;
        const float hn = gh[2] + bh_n;
        const float r = sigmoidf_(gip[j] + gh[0] + bh_r);
        const float z = sigmoidf_(gip[H + j] + gh[1] + bh_z);
        const float n = tanhf(gip[2 * H + j] + r * hn);
        const float h = (1.f - z) * n + z * hp;
        Y[((long)row * T + tau) * (2 * H) + dir * H + j] = h;
        if (save) {
            float* sp = save + dir * save_ds + ((long)row * T + tau) * (4 * H);
            sp[j] = r; sp[H + j] = z; sp[2 * H + j] = n; sp[3 * H + j] = hn;
        }
    }
}

// Backward step at time tau (the reverse of the forward order).  Using the gate gradients dgh of the step that
// consumed h_tau (tau_next, written by the previous launch):
//     dh_tau = dy_tau + dh_next * z_next + dgh_next @ W_hh        (W_hh passed transposed: [H][3H])
// then this step's own gate gradients for the 16 hidden units the workgroup owns:
//     dn = dh (1-z)(1-n^2),  dz = dh (h_prev - n) z (1-z),  dr = dn * hn * r (1-r)
//     dgi = [dr, dz, dn]   dgh = [dr, dz, dn * r]
__global__ __launch_bounds__(GRU_THREADS) void gru_bwd_step_kernel(
    const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ save, long save_ds,
    const float* __restrict__ wt0, const float* __restrict__ wt1, float* __restrict__ dgi, float* __restrict__ dgh, long dg_ds,
    float* __restrict__ dhbuf, int B, int T, int H, int step) {
    __shared__ float red[GRU_KS][GRU_MT][4][64];
    const int dir = blockIdx.z;
    const int tau = dir ? step : T - 1 - step;
    const int tau_next = dir ? tau - 1 : tau + 1;   // consumer of h_tau in forward order
    const int tau_prev = dir ? tau + 1 : tau - 1;   // producer of h_prev for this cell
    const bool has_next = step > 0;
    const bool has_prev = dir ? (tau < T - 1) : (tau > 0);
    const float* wt = dir ? wt1 : wt0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = wave % GRU_MT, ks = wave / GRU_MT;
    const int r16 = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * (GRU_MT * 16);
    const int H3 = 3 * H;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (has_next) {
        const int b = b0 + mt * 16 + r16;
        const bool b_ok = b < B;
        const float* arow = dgh + dir * dg_ds + ((long)(b_ok ? b : 0) * T + tau_next) * H3;
        const int j = j0 + r16;
        const bool j_ok = j < H;
        const float* wrow = wt + (long)(j_ok ? j : 0) * H3;
        for (int k0 = ks * 16; k0 < H3; k0 += GRU_KS * 16) {
            const int k = k0 + 4 * kq;
            const bool inb = k < H3;
            const f32x4 a = ld4(arow + k, b_ok && inb);
            const f32x4 w = ld4(wrow + k, j_ok && inb);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v], w[v], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[ks][mt][i][lane] = acc[i];
    __syncthreads();
    if (ks != 0) return;

    const int j = j0 + r16;
    if (j >= H) return;
    float* dh_w = dhbuf + ((long)(step & 1) * 2 + dir) * (long)B * H;
    const float* dh_r = dhbuf + ((long)((step & 1) ^ 1) * 2 + dir) * (long)B * H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = b0 + mt * 16 + kq * 4 + i;
        if (row >= B) continue;
        float dh = dY[((long)row * T + tau) * (2 * H) + dir * H + j];
        if (has_next) {
            float s = red[0][mt][i][lane];
#pragma unroll
            for (int q = 1; q < GRU_KS; ++q) s += red[q][mt][i][lane];
            const float z_next = save[dir * save_ds + ((long)row * T + tau_next) * (4 * H) + H + j];
            dh += s + dh_r[(long)row * H + j] * z_next;
        }
        const float* sp = save + dir * save_ds + ((long)row * T + tau) * (4 * H);
        const float r = sp[j], z = sp[H + j], n = sp[2 * H + j], hn = sp[3 * H + j];
        const float hp = has_prev ? Y[((long)row * T + tau_prev) * (2 * H) + dir * H + j] : 0.f;
        const float dn = dh * (1.f - z) * (1.f - n * n);
        const float dz = dh * (hp - n) * z * (1.f - z);
        const float dr = dn * hn * r * (1.f - r);
        float* gi_o = dgi + dir * dg_ds + ((long)row * T + tau) * H3;
        float* gh_o = dgh + dir * dg_ds + ((long)row * T + tau) * H3;
        gi_o[j] = dr; gi_o[H + j] = dz; gi_o[2 * H + j] = dn;
        gh_o[j] = dr; gh_o[H + j] = dz; gh_o[2 * H + j] = dn * r;
        dh_w[(long)row * H + j] = dh;
    }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_gru_forward(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                              const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                              int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(gi && w_hh_fwd && w_hh_rev && b_hh_fwd && b_hh_rev && y, "tg_gru_forward: null pointer");
    TG_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0, "tg_gru_forward: need H %% 4 == 0 (H=%d)", H);
    TG_REQUIRE(aligned16(w_hh_fwd) && aligned16(w_hh_rev) && aligned16(y), "tg_gru_forward: w_hh / y must be 16-byte aligned");
    dim3 grid(cdiv(H, 16), cdiv(B, GRU_MT * 16), 2);
    for (int step = 0; step < T; ++step)
        hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(GRU_THREADS), 0, (hipStream_t)stream, gi, (long)gi_dir_stride, w_hh_fwd,
                           w_hh_rev, b_hh_fwd, b_hh_rev, y, save, (long)save_dir_stride, B, T, H, step);
    return check_launch("tg_gru_forward");
}

extern "C" int tg_gru_backward(const float* dy, const float* y, const float* save, int64_t save_dir_stride,
                               const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                               float* dh_scratch, int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(dy && y && save && w_hh_t_fwd && w_hh_t_rev && dgi && dgh && dh_scratch, "tg_gru_backward: null pointer");
    TG_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0, "tg_gru_backward: need H %% 4 == 0 (H=%d)", H);
    TG_REQUIRE(aligned16(w_hh_t_fwd) && aligned16(w_hh_t_rev) && aligned16(dgh) && (dg_dir_stride % 4 == 0),
               "tg_gru_backward: w_hh_t / dgh must be 16-byte aligned");
    dim3 grid(cdiv(H, 16), cdiv(B, GRU_MT * 16), 2);
    for (int step = 0; step < T; ++step)
        hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(GRU_THREADS), 0, (hipStream_t)stream, dy, y, save, (long)save_dir_stride,
                           w_hh_t_fwd, w_hh_t_rev, dgi, dgh, (long)dg_dir_stride, dh_scratch, B, T, H, step);
    return check_launch("tg_gru_backward");
}
