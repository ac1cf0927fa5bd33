// Bidirectional GRU recurrence (nn.GRU semantics) on the f32 matrix cores.
//
// Two regimes, both without any in-kernel inter-workgroup synchronisation (nothing can dead-lock):
//
//  * H = 300 (generator): the recurrent product h_{t-1}[B,H] x W_hh^T[H,3H] is too large for one CU's LDS + registers
//    (W_hh fp32 = 1.08 MB per direction), so each time step is ONE launch covering both directions, every batch tile
//    and every 16-wide slice of hidden units; the launch boundary (~1.5 us) is the grid-wide dependency, cheaper than
//    an in-kernel grid barrier (4-7 us on 256 CUs).  A workgroup owns [32 batch rows] x [16 hidden units, all three
//    gates]: 8 waves = 2 row tiles x 4 K-slices.  Every wave first issues the loads of its gate-epilogue operands and
//    of ALL its K fragments (16-byte k-permuted MFMA feed, see gemm.hip), then runs its MFMAs, partial sums meet in LDS
//    and all eight waves share the fused gate epilogue (sigmoid/tanh, h' = (1-z) n + z h), writing h_t straight into the
//    layer output y, which doubles as the state store.
//
//  * H = 64 (discriminator): persistent kernels with W_hh in registers, see gru_h64.hip.
#include "common.hpp"

namespace tg {

constexpr int GRU_MT = 2;   // 16-row tiles per workgroup
constexpr int GRU_KS = 4;   // K slices per workgroup
constexpr int GRU_PF = 5;   // K fragments in flight per wave (covers H <= 320 in one batch of loads)
constexpr int GRU_THREADS = 64 * GRU_MT * GRU_KS;

__device__ __forceinline__ f32x4 ld4(const float* p, bool ok) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? *reinterpret_cast<const f32x4*>(p) : z;
}

__global__ __launch_bounds__(GRU_THREADS) void gru_fwd_step_kernel(
    const float* __restrict__ gi, long gi_ds, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ Y, float* __restrict__ save,
    long save_ds, int B, int T, int H, int step, int n_jt, int n_bt) {
    __shared__ float red[GRU_KS][GRU_MT][3][4][64];
    // logical order: batch tile fastest, then hidden-unit slice, then direction -> an XCD's chunk holds few W_hh slices
    // (57.6 KB each) for ALL batch tiles, and one direction's h_{t-1}
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int bt = lid % n_bt, jt = (lid / n_bt) % n_jt;
    const int dir = lid / (n_bt * n_jt);
    const int tau = dir ? T - 1 - step : step;
    const int tau_prev = dir ? tau + 1 : tau - 1;
    const bool has_prev = step > 0;
    const float* whh = dir ? whh1 : whh0;
    const float* bhh = dir ? bhh1 : bhh0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = wave % GRU_MT, ks = wave / GRU_MT;
    const int r16 = lane & 15, kq = lane >> 4;
    const int j0 = jt * 16, b0 = bt * (GRU_MT * 16);

    // gate epilogue ownership: wave (mt, ks) finalises accumulator row i = ks of m-tile mt.  Its operands do not
    // depend on the product, so their loads go out first.
    const int erow = b0 + mt * 16 + kq * 4 + ks;
    const int ej = j0 + r16;
    const bool e_ok = erow < B && ej < H;
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, hp = 0.f, bh_r = 0.f, bh_z = 0.f, bh_n = 0.f;
    if (e_ok) {
        const float* gip = gi + dir * gi_ds + ((long)erow * T + tau) * (3 * H);
        gi_r = gip[ej]; gi_z = gip[H + ej]; gi_n = gip[2 * H + ej];
        bh_r = bhh[ej]; bh_z = bhh[H + ej]; bh_n = bhh[2 * H + ej];
        if (has_prev) hp = Y[((long)erow * T + tau_prev) * (2 * H) + dir * H + ej];
    }

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
        for (int kbase = ks * 16; kbase < H; kbase += GRU_KS * 16 * GRU_PF) {
            f32x4 a[GRU_PF], w[3][GRU_PF];
#pragma unroll
            for (int p = 0; p < GRU_PF; ++p) {
                const int k = kbase + p * (GRU_KS * 16) + 4 * kq;
                const bool inb = k < H;   // H % 4 == 0 (checked on the host)
                a[p] = ld4(hrow + k, b_ok && inb);
#pragma unroll
                for (int g = 0; g < 3; ++g) w[g][p] = ld4(wrow[g] + k, j_ok && inb);
            }
#pragma unroll
            for (int p = 0; p < GRU_PF; ++p) {
                if (kbase + p * (GRU_KS * 16) < H) {      // wave-uniform
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][v], w[g][p][v], acc[g], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[ks][mt][g][i][lane] = acc[g][i];
    __syncthreads();
    if (!e_ok) return;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float s = red[0][mt][g][ks][lane];
#pragma unroll
        for (int q = 1; q < GRU_KS; ++q) s += red[q][mt][g][ks][lane];
        gh[g] = s;
    }
    const float hn = gh[2] + bh_n;
    const float r = gate_sigmoid(gi_r + gh[0] + bh_r);
    const float z = gate_sigmoid(gi_z + gh[1] + bh_z);
    const float n = gate_tanh(gi_n + r * hn);
    const float h = (1.f - z) * n + z * hp;
    Y[((long)erow * T + tau) * (2 * H) + dir * H + ej] = h;
    if (save) {
        float* sp = save + dir * save_ds + ((long)erow * T + tau) * (4 * H);
        sp[ej] = r; sp[H + ej] = z; sp[2 * H + ej] = n; sp[3 * H + ej] = hn;
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
    float* __restrict__ dhbuf, int B, int T, int H, int step, int n_jt, int n_bt) {
    __shared__ float red[GRU_KS][GRU_MT][4][64];
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int bt = lid % n_bt, jt = (lid / n_bt) % n_jt;
    const int dir = lid / (n_bt * n_jt);
    const int tau = dir ? step : T - 1 - step;
    const int tau_next = dir ? tau - 1 : tau + 1;   // consumer of h_tau in forward order
    const int tau_prev = dir ? tau + 1 : tau - 1;   // producer of h_prev for this cell
    const bool has_next = step > 0;
    const bool has_prev = dir ? (tau < T - 1) : (tau > 0);
    const float* wt = dir ? wt1 : wt0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = wave % GRU_MT, ks = wave / GRU_MT;
    const int r16 = lane & 15, kq = lane >> 4;
    const int j0 = jt * 16, b0 = bt * (GRU_MT * 16);
    const int H3 = 3 * H;

    // epilogue operands of accumulator row i = ks (independent of the product): issue their loads first
    const int erow = b0 + mt * 16 + kq * 4 + ks;
    const int ej = j0 + r16;
    const bool e_ok = erow < B && ej < H;
    float dy = 0.f, r = 0.f, z = 0.f, n = 0.f, hn = 0.f, hp = 0.f, z_next = 0.f, dh_next = 0.f;
    float* dh_w = dhbuf + ((long)(step & 1) * 2 + dir) * (long)B * H;
    const float* dh_r = dhbuf + ((long)((step & 1) ^ 1) * 2 + dir) * (long)B * H;
    if (e_ok) {
        dy = dY[((long)erow * T + tau) * (2 * H) + dir * H + ej];
        const float* sp = save + dir * save_ds + ((long)erow * T + tau) * (4 * H);
        r = sp[ej]; z = sp[H + ej]; n = sp[2 * H + ej]; hn = sp[3 * H + ej];
        if (has_prev) hp = Y[((long)erow * T + tau_prev) * (2 * H) + dir * H + ej];
        if (has_next) {
            z_next = save[dir * save_ds + ((long)erow * T + tau_next) * (4 * H) + H + ej];
            dh_next = dh_r[(long)erow * H + ej];
        }
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (has_next) {
        const int b = b0 + mt * 16 + r16;
        const bool b_ok = b < B;
        const float* arow = dgh + dir * dg_ds + ((long)(b_ok ? b : 0) * T + tau_next) * H3;
        const int j = j0 + r16;
        const bool j_ok = j < H;
        const float* wrow = wt + (long)(j_ok ? j : 0) * H3;
        for (int kbase = ks * 16; kbase < H3; kbase += GRU_KS * 16 * GRU_PF) {
            f32x4 a[GRU_PF], w[GRU_PF];
#pragma unroll
            for (int p = 0; p < GRU_PF; ++p) {
                const int k = kbase + p * (GRU_KS * 16) + 4 * kq;
                const bool inb = k < H3;
                a[p] = ld4(arow + k, b_ok && inb);
                w[p] = ld4(wrow + k, j_ok && inb);
            }
#pragma unroll
            for (int p = 0; p < GRU_PF; ++p) {
                if (kbase + p * (GRU_KS * 16) < H3) {
                    // two accumulators: the dependent-accumulator latency of v_mfma_f32_16x16x4_f32 (40 cycles) exceeds
                    // its issue interval (32)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][0], w[p][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][1], w[p][1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][2], w[p][2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][3], w[p][3], acc1, 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[ks][mt][i][lane] = acc0[i] + acc1[i];
    __syncthreads();
    if (!e_ok) return;
    float dh = dy;
    if (has_next) {
        float s = red[0][mt][ks][lane];
#pragma unroll
        for (int q = 1; q < GRU_KS; ++q) s += red[q][mt][ks][lane];
        dh += s + dh_next * z_next;
    }
    const float dn = dh * (1.f - z) * (1.f - n * n);
    const float dz = dh * (hp - n) * z * (1.f - z);
    const float dr = dn * hn * r * (1.f - r);
    float* gi_o = dgi + dir * dg_ds + ((long)erow * T + tau) * H3;
    float* gh_o = dgh + dir * dg_ds + ((long)erow * T + tau) * H3;
    gi_o[ej] = dr; gi_o[H + ej] = dz; gi_o[2 * H + ej] = dn;
    gh_o[ej] = dr; gh_o[H + ej] = dz; gh_o[2 * H + ej] = dn * r;
    dh_w[(long)erow * H + ej] = dh;
}

}  // namespace tg

using namespace tg;

constexpr int HS = 64;      // H = 64 runs in the persistent kernels of gru_h64.hip

extern "C" int tg_gru_forward(const float* gi, int64_t gi_dir_stride, const float* w_hh_fwd, const float* w_hh_rev,
                              const float* b_hh_fwd, const float* b_hh_rev, float* y, float* save, int64_t save_dir_stride,
                              int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(gi && w_hh_fwd && w_hh_rev && b_hh_fwd && b_hh_rev && y, "tg_gru_forward: null pointer");
    TG_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0, "tg_gru_forward: need H %% 4 == 0 (H=%d)", H);
    TG_REQUIRE(aligned16(w_hh_fwd) && aligned16(w_hh_rev) && aligned16(y), "tg_gru_forward: w_hh / y must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (H == HS)
        return tg_gru_h64_forward(gi, gi_dir_stride, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, y, save, save_dir_stride, nullptr, nullptr, B, T,
                                  stream);
    const int n_jt = cdiv(H, 16), n_bt = cdiv(B, GRU_MT * 16);
    dim3 grid(n_jt * n_bt * 2);
    for (int step = 0; step < T; ++step)
        hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(GRU_THREADS), 0, s, gi, (long)gi_dir_stride, w_hh_fwd,
                           w_hh_rev, b_hh_fwd, b_hh_rev, y, save, (long)save_dir_stride, B, T, H, step, n_jt, n_bt);
    return check_launch("tg_gru_forward");
}

extern "C" int tg_gru_backward(const float* dy, const float* y, const float* save, int64_t save_dir_stride,
                               const float* w_hh_t_fwd, const float* w_hh_t_rev, float* dgi, float* dgh, int64_t dg_dir_stride,
                               float* dh_scratch, int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(dy && y && save && w_hh_t_fwd && w_hh_t_rev && dgi && dgh && dh_scratch, "tg_gru_backward: null pointer");
    TG_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0, "tg_gru_backward: need H %% 4 == 0 (H=%d)", H);
    TG_REQUIRE(aligned16(w_hh_t_fwd) && aligned16(w_hh_t_rev) && aligned16(dgh) && (dg_dir_stride % 4 == 0),
               "tg_gru_backward: w_hh_t / dgh must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (H == HS)
        return tg_gru_h64_backward(dy, nullptr, y, save, save_dir_stride, w_hh_t_fwd, w_hh_t_rev, dgi, dgh, dg_dir_stride, B, T, stream);
    const int n_jt = cdiv(H, 16), n_bt = cdiv(B, GRU_MT * 16);
    dim3 grid(n_jt * n_bt * 2);
    for (int step = 0; step < T; ++step)
        hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(GRU_THREADS), 0, s, dy, y, save, (long)save_dir_stride,
                           w_hh_t_fwd, w_hh_t_rev, dgi, dgh, (long)dg_dir_stride, dh_scratch, B, T, H, step, n_jt, n_bt);
    return check_launch("tg_gru_backward");
}
