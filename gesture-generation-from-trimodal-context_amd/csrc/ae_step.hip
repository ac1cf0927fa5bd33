// FGD autoencoder training step (scripts/train_feature_extractor.py:54-97 on model/embedding_net.py:42-82,165-217, 34-frame branch,
// variational_encoding = False) as EIGHTEEN launches instead of ~105.
//
// The step is latency bound: 1.05 M MAC per clip forward, 128 clips, but eight train-mode BatchNorms whose batch statistics (forward) and
// gradient sums (backward) each force a device-wide dependency.  The generic path pays 4.4-17 us per launch for window GEMMs on a few
// workgroups, three-launch BatchNorms, permutes and weight packs (0.71 ms per step).  Here the step is cut ONLY where a BatchNorm needs the
// whole batch: sixteen such cuts -> phases 1..17, one CLIP per 512-thread workgroup, everything between two cuts fused:
//     [finish the previous BatchNorm from its fp64 sums] -> normalise + LeakyReLU -> conv / transposed conv / linear layers up to the next
//     BatchNorm -> per-channel partial sums of this workgroup's clip -> fp64 atomics
// and the same backwards (BatchNorm backward from the sums of g and g * xhat, input gradient through the layers down to the previous
// BatchNorm, its g and the two sums, THEN the layer's weight gradient as a per-clip partial: nothing waits for it).  Kernel boundaries, not
// in-kernel grid barriers: a dependent boundary costs ~1.5 us, a device-wide barrier 4-7 (MI355X_MICROARCH.md, price list), and nothing has
// to be co-resident.
// Phase 18 (453 workgroups, its own partition) turns the per-clip conv partials and the saved linear-layer vectors into the gradient slab:
// linear weight gradients as (8 rows x 64 columns) tiles reduced over the batch inside one workgroup, conv partials summed over eight batch
// slices in a fixed order, BatchNorm gamma / beta gradients straight from the backward sums; every thread then applies torch.optim.Adam to the
// elements it has just finished (when the caller passes the moment slabs: the step is complete; fc_logvar has no gradient and is skipped like
// a parameter whose .grad is None); it also copies the loss and re-zeroes the sums.  Adam's step counter advances in phase 1.  No float
// atomics on gradients: two runs are
// bit-identical up to the fp64 atomics of the statistics.
//
// Arithmetic: the conv-shaped layers and their weight gradients on v_mfma_f32_16x16x4_f32 (exact fp32 products; rows = 16 positions or 16
// (tap, channel) pairs, columns = 16 channels, the bias gradient as one more row of ones), one clip's activations in LDS with rows padded by
// four floats (a 16-row operand read on a row pitch of 32 banks would serialise 16-fold), weights staged into LDS per layer in the
// [tap * Cin + ci][co] order the MFMA's column operand reads.  The linear layers of a clip are matrix-vector products: they read a TRANSPOSED copy
// of the weight that phase 1 writes once per step (16-byte loads along the output index, slices of k per thread group, no cross-lane
// reduction); their input gradients read the weight as stored.  Statistics in fp64.  Measured (B = 128): 0.712 -> 0.20 ms per step.
#include "common.hpp"

namespace tg {

constexpr int AE_NT = 512;
constexpr int AE_WLD = 16768;            // staged weights: at most 256 (+ 1 for the bias gradient) rows of 64 + 1 floats (net.2: Conv1d(64, 64, 4))
constexpr int AE_BUF = 2048;             // one activation buffer (largest tensor of a clip: 30 x 64)
constexpr int AE_NBUF = 6;

// per-clip record of the activation workspace (floats)
constexpr int A_C0 = 0;                  // pre-BatchNorm conv outputs, channel-last
constexpr int A_C1 = A_C0 + 32 * 32;
constexpr int A_C2 = A_C1 + 30 * 64;
constexpr int A_FLAT = A_C2 + 14 * 64;   // net.3 output flattened channel-major (torch flatten(1) of (B, 32, 12))
constexpr int A_F1 = A_FLAT + 384;       // out_net.0 output (pre-BatchNorm) ...
constexpr int A_Y1F = A_F1 + 256;        // ... and normalised (the next linear layer's input)
constexpr int A_F2 = A_Y1F + 256;
constexpr int A_Y2F = A_F2 + 128;
constexpr int A_F3 = A_Y2F + 128;
constexpr int A_MU = A_F3 + 32;
constexpr int A_P0 = A_MU + 32;
constexpr int A_YP = A_P0 + 64;
constexpr int A_P3 = A_YP + 64;
constexpr int A_T0 = A_P3 + 136;
constexpr int A_T1 = A_T0 + 36 * 32;
constexpr int A_G = A_T1 + 38 * 32;      // gradient handed from one backward phase to the next (g = dL/d(BatchNorm output))
constexpr int A_DP3 = A_G + 1920;        // gradients at the linear layers' outputs (for their weight gradients, phase 18)
constexpr int A_DP0 = A_DP3 + 136;
constexpr int A_DMU = A_DP0 + 64;
constexpr int A_DF3 = A_DMU + 32;
constexpr int A_DF2 = A_DF3 + 32;
constexpr int A_DF1 = A_DF2 + 128;
constexpr int AE_ACT = A_DF1 + 256;

// per-clip record of the conv weight / bias gradient partials (floats), each segment laid out like its parameter
constexpr int Q_E0W = 0;
constexpr int Q_E0B = Q_E0W + 32 * 27 * 3;
constexpr int Q_E1W = Q_E0B + 32;
constexpr int Q_E1B = Q_E1W + 64 * 32 * 3;
constexpr int Q_E2W = Q_E1B + 64;
constexpr int Q_E2B = Q_E2W + 64 * 64 * 4;
constexpr int Q_E3W = Q_E2B + 64;
constexpr int Q_E3B = Q_E3W + 32 * 64 * 3;
constexpr int Q_T0W = Q_E3B + 32;
constexpr int Q_T0B = Q_T0W + 4 * 32 * 3;
constexpr int Q_T1W = Q_T0B + 32;
constexpr int Q_T1B = Q_T1W + 32 * 32 * 3;
constexpr int Q_C6W = Q_T1B + 32;
constexpr int Q_C6B = Q_C6W + 32 * 32 * 3;
constexpr int Q_C7W = Q_C6B + 32;
constexpr int Q_C7B = Q_C7W + 27 * 32 * 3;
constexpr int AE_PART = Q_C7B + 28;
static_assert(AE_ACT % 4 == 0 && AE_PART % 4 == 0, "16-byte rows");

// parameter order of tg_ae_step_args.off
enum {
    P_E0W, P_E0B, P_BN0G, P_BN0B, P_E1W, P_E1B, P_BN1G, P_BN1B, P_E2W, P_E2B, P_BN2G, P_BN2B, P_E3W, P_E3B,
    P_F1W, P_F1B, P_BN3G, P_BN3B, P_F2W, P_F2B, P_BN4G, P_BN4B, P_F3W, P_F3B, P_MUW, P_MUB,
    P_D0W, P_D0B, P_BN5G, P_BN5B, P_D1W, P_D1B, P_T0W, P_T0B, P_BN6G, P_BN6B, P_T1W, P_T1B, P_BN7G, P_BN7B,
    P_C6W, P_C6B, P_C7W, P_C7B, P_COUNT
};
static_assert(P_COUNT == 44, "tg_ae_step_args.off");

constexpr int AE_SLOTS = 17;             // 8 forward statistics, 8 backward sums, the loss; [slot][2][256] doubles

struct AeArgs {
    const float* x;
    float *P, *G;
    int off[P_COUNT];
    float* rm[8];
    float* rv[8];
    long long* nbt[8];
    double* sums;
    float* act;
    float* part;
    float* wt;
    float *loss, *recon, *feat;
    int* step;
    int B;
    float bn_eps, momentum;
    float *M, *V;                         // Adam moments (slab images); NULL: gradients only
    float lr, b1, b2, adam_eps;
};

struct AeBn { float *mean, *rstd, *ga, *be; };      // one BatchNorm's per-channel coefficients in LDS
struct AeLds {
    float* wl;
    float* b[AE_NBUF];
    AeBn s1, s2;                          // a backward phase needs two BatchNorms at once (the one it differentiates, the one in front of it)
    float *mg, *mgx;
    float* red;                           // [2][AE_NT]
};
constexpr int AE_LDS_FLOATS = AE_WLD + AE_NBUF * AE_BUF + 10 * 256 + 2 * AE_NT;

__device__ __forceinline__ AeLds ae_lds(float* smem) {
    AeLds l;
    l.wl = smem;
    for (int i = 0; i < AE_NBUF; ++i) l.b[i] = smem + AE_WLD + i * AE_BUF;
    float* s = smem + AE_WLD + AE_NBUF * AE_BUF;
    l.s1 = AeBn{s, s + 256, s + 512, s + 768};
    l.s2 = AeBn{s + 1024, s + 1280, s + 1536, s + 1792};
    l.mg = s + 2048; l.mgx = s + 2304;
    l.red = s + 2560;
    return l;
}

__device__ __forceinline__ float ae_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// A clip's [L][C] tensor lives in LDS with rows C + 4 floats apart (C = 32 / 64): the MFMA operand reads below take 16 rows at once, and a
// row stride that is a multiple of 32 banks would serialise them 16-fold.  Vectors (L = 1) are plain arrays.
__device__ __forceinline__ constexpr int ae_rs(int C) { return C + 4; }

// global [L][C] (flat) <-> LDS [L][C + 4]
// (the _t forms run on the thread subset [tid of nt] and end WITHOUT a barrier: a phase's prologue gives its independent global reads --
// staged weights, saved tensors, BatchNorm sums -- to different waves and meets once)
template <int L, int C>
__device__ __forceinline__ void ae_load_t(float* __restrict__ dst, const float* __restrict__ src, int tid, int nt) {
    static_assert(C % 4 == 0, "16-byte pieces");
    for (int i = tid; i < L * C / 4; i += nt) {
        const int r = (4 * i) / C, c = (4 * i) % C;
        *reinterpret_cast<f32x4*>(dst + r * ae_rs(C) + c) = reinterpret_cast<const f32x4*>(src)[i];
    }
}
template <int L, int C>
__device__ __forceinline__ void ae_load(float* __restrict__ dst, const float* __restrict__ src) {
    ae_load_t<L, C>(dst, src, threadIdx.x, AE_NT);
    __syncthreads();
}
template <int L, int C>
__device__ __forceinline__ void ae_store(float* __restrict__ dst, const float* __restrict__ src) {
    for (int i = threadIdx.x; i < L * C / 4; i += AE_NT) {
        const int r = (4 * i) / C, c = (4 * i) % C;
        reinterpret_cast<f32x4*>(dst)[i] = *reinterpret_cast<const f32x4*>(src + r * ae_rs(C) + c);
    }
}
// the clip's poses [34][27] -> LDS [34][36], columns 27.. zero
__device__ __forceinline__ void ae_load_x_t(float* __restrict__ dst, const float* __restrict__ x, int tid, int nt) {
    for (int i = tid; i < 34 * 36; i += nt) {
        const int r = i / 36, c = i - r * 36;
        dst[i] = c < 27 ? x[r * 27 + c] : 0.f;
    }
}
__device__ __forceinline__ void ae_load_x(float* __restrict__ dst, const float* __restrict__ x) {
    ae_load_x_t(dst, x, threadIdx.x, AE_NT);
    __syncthreads();
}

// BatchNorm `bn` from the fp64 sums of `slot`: mean / rstd / gamma / beta of its C channels into LDS; optionally (workgroup 0, forward
// phases) the running statistics (momentum, unbiased variance, num_batches_tracked += 1)
template <int C>
__device__ __forceinline__ void ae_bn_prepare_t(const AeArgs& a, const AeBn& o, int bn, int slot, double n_el, int p_gamma, bool update_running, int tid, int nt) {
    for (int c = tid; c < C; c += nt) {
        const double* s = a.sums + (size_t)slot * 512;
        const double m = s[c] / n_el;
        double var = s[256 + c] / n_el - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)a.bn_eps));
        o.mean[c] = mf; o.rstd[c] = rs;
        o.ga[c] = a.P[a.off[p_gamma] + c];
        o.be[c] = a.P[a.off[p_gamma + 1] + c];
        if (update_running && blockIdx.x == 0 && a.rm[bn] != nullptr) {
            const double unbiased = n_el > 1.0 ? var * n_el / (n_el - 1.0) : var;
            a.rm[bn][c] = (1.f - a.momentum) * a.rm[bn][c] + a.momentum * mf;
            a.rv[bn][c] = (1.f - a.momentum) * a.rv[bn][c] + a.momentum * (float)unbiased;
            if (c == 0 && a.nbt[bn] != nullptr) *a.nbt[bn] += 1;
        }
    }
}
// the sums of the backward: mg = sum(g) / n, mgx = sum(g * xhat) / n
template <int C>
__device__ __forceinline__ void ae_bwd_means_t(const AeArgs& a, const AeLds& l, int slot, double n_el, int tid, int nt) {
    for (int c = tid; c < C; c += nt) {
        const double* s = a.sums + (size_t)slot * 512;
        l.mg[c] = (float)(s[c] / n_el);
        l.mgx[c] = (float)(s[256 + c] / n_el);
    }
}
// dst = LeakyReLU(BatchNorm(src)), both [L][C + 4]
template <int L, int C>
__device__ __forceinline__ void ae_bn_act(const AeBn& l, float* __restrict__ dst, const float* __restrict__ src, float slope) {
    for (int i = threadIdx.x; i < L * C; i += AE_NT) {
        const int c = i & (C - 1), o = (i / C) * ae_rs(C) + c;
        dst[o] = ae_lrelu((src[o] - l.mean[c]) * l.rstd[c] * l.ga[c] + l.be[c], slope);
    }
    __syncthreads();
}
// g = dy * act'(BatchNorm(raw)) in place (dy -> g)
template <int L, int C>
__device__ __forceinline__ void ae_act_bwd(const AeBn& l, float* __restrict__ dy, const float* __restrict__ raw, float slope) {
    for (int i = threadIdx.x; i < L * C; i += AE_NT) {
        const int c = i & (C - 1), o = (i / C) * ae_rs(C) + c;
        const float bnv = (raw[o] - l.mean[c]) * l.rstd[c] * l.ga[c] + l.be[c];
        dy[o] = bnv > 0.f ? dy[o] : dy[o] * slope;
    }
    __syncthreads();
}
// BatchNorm backward in place: g -> dc = gamma * rstd * (g - mg - xhat * mgx)
template <int L, int C>
__device__ __forceinline__ void ae_bn_bwd(const AeLds& ll, const AeBn& l, float* __restrict__ g, const float* __restrict__ raw) {
    for (int i = threadIdx.x; i < L * C; i += AE_NT) {
        const int c = i & (C - 1), o = (i / C) * ae_rs(C) + c;
        const float xh = (raw[o] - l.mean[c]) * l.rstd[c];
        g[o] = l.ga[c] * l.rstd[c] * (g[o] - ll.mg[c] - xh * ll.mgx[c]);
    }
    __syncthreads();
}
// per-channel sums of the clip's [L][C + 4] tensor -> fp64 atomics.  BWD: (sum g, sum g * xhat) with xhat from `raw`; else (sum v, sum v^2)
template <int L, int C, bool BWD>
__device__ __forceinline__ void ae_chan_sums(const AeLds& l, const AeBn& bnc, const float* __restrict__ v, const float* __restrict__ raw, double* __restrict__ slot) {
    static_assert(AE_NT % C == 0, "channels per workgroup");
    constexpr int PARTS = AE_NT / C;
    const int c = threadIdx.x % C, part = threadIdx.x / C;
    float s0 = 0.f, s1 = 0.f;
    for (int r = part; r < L; r += PARTS) {
        const float x = v[r * ae_rs(C) + c];
        s0 += x;
        if constexpr (BWD) s1 += x * ((raw[r * ae_rs(C) + c] - bnc.mean[c]) * bnc.rstd[c]);
        else s1 += x * x;
    }
    l.red[threadIdx.x] = s0; l.red[AE_NT + threadIdx.x] = s1;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        double t0 = 0.0, t1 = 0.0;
        for (int p = 0; p < PARTS; ++p) { t0 += (double)l.red[p * C + threadIdx.x]; t1 += (double)l.red[AE_NT + p * C + threadIdx.x]; }
        atomicAdd(slot + threadIdx.x, t0);
        atomicAdd(slot + 256 + threadIdx.x, t1);
    }
    __syncthreads();
}

// ---- the multiply routines of the conv-shaped layers: v_mfma_f32_16x16x4_f32 (exact fp32 products) -------------------------------------------
// out[p][n] = bias[n] + sum_{kk < KW, m < CINP} in[p * S + kk - PAD][m] * wl[(kk * CINP + m) * (N + 1) + n]   (rows outside [0, LIN) are zero)
// in: LDS rows RS floats apart; out: LDS [LOUT][N + 4].  A wave owns (16 positions) x (16 channels) tiles: MFMA rows = positions, columns = channels;
// lane (r16, kq) supplies in[position r16][k = 4 s + kq] and wl[k][channel r16] of reduction step s and holds out[4 kq + q][r16].
template <int LIN, int CINP, int RS, int LOUT, int N, int KW, int S, int PAD>
__device__ __forceinline__ void ae_conv(const float* __restrict__ in, const float* __restrict__ wl, const float* __restrict__ bias, int n_bias,
                                        float* __restrict__ out) {
    static_assert(N % 16 == 0 && CINP % 4 == 0, "tiles");
    constexpr int MT = (LOUT + 15) / 16, NTL = N / 16, WS = N + 1, OS = N + 4, KT = KW * CINP / 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    for (int tile = wave; tile < MT * NTL; tile += AE_NT / 64) {
        const int rt = tile / NTL, ct = tile - rt * NTL;
        const int n = ct * 16 + r16, p = rt * 16 + r16;
        const float b0 = (bias != nullptr && n < n_bias) ? bias[n] : 0.f;
        f32x4 acc = {b0, b0, b0, b0};
        int kk = 0, m = kq;
        const float* wp = wl + kq * WS + n;
        // (unrolled deep: the operand reads of many steps are in flight before the first MFMA needs them -- the chain is LDS latency otherwise)
#pragma unroll 12
        for (int s = 0; s < KT; ++s) {
            const int row = p * S + kk - PAD;
            const float av = (p < LOUT && row >= 0 && row < LIN) ? in[row * RS + m] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wp[0], acc, 0, 0, 0);
            wp += 4 * WS;
            m += 4;
            if (m >= CINP) { m -= CINP; ++kk; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int po = rt * 16 + 4 * kq + q;
            if (po < LOUT) out[po * OS + n] = acc[q];
        }
    }
    __syncthreads();
}
// the same sum on the vector ALU for a 4-channel output (decoder.net.0's input gradient): thread = (channel, position)
template <int LIN, int CINP, int RS, int LOUT, int KW>
__device__ __forceinline__ void ae_conv_n4(const float* __restrict__ in, const float* __restrict__ wl, float* __restrict__ out) {
    constexpr int WS = 5;
    const int n = threadIdx.x & 3, p = threadIdx.x >> 2;
    if (p < LOUT) {
        float acc = 0.f;
        for (int kk = 0; kk < KW; ++kk)
            for (int m = 0; m < CINP; ++m) acc = fmaf(in[(p + kk) * RS + m], wl[(kk * CINP + m) * WS + n], acc);
        out[p * 4 + n] = acc;
    }
    __syncthreads();
}
// weight gradient of one clip in the staged layout: dwl[(kk * CINP + m) * (N + 1) + n] = sum_p dout[p][n] * in[p * S + kk - PAD][m].
// MFMA rows = (kk, m), columns = n, reduction over the positions p (steps of four).  Row KW * CINP is the BIAS gradient sum_p dout[p][n]
// (an input of ones): ae_bias_out copies it out
template <int LIN, int CINP, int RS, int LOUT, int N, int KW, int S, int PAD>
__device__ __forceinline__ void ae_wgrad(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ dwl) {
    static_assert(N % 16 == 0, "tiles");
    constexpr int KM = KW * CINP, MT = (KM + 1 + 15) / 16, NTL = N / 16, WS = N + 1, DS = N + 4, PT = (LOUT + 3) / 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    for (int tile = wave; tile < MT * NTL; tile += AE_NT / 64) {
        const int rt = tile / NTL, ct = tile - rt * NTL;
        const int km = rt * 16 + r16, n = ct * 16 + r16;
        const bool km_ok = km < KM;
        const int kk = km_ok ? km / CINP : 0, m = km_ok ? km - kk * CINP : 0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < PT; ++s) {
            const int p = 4 * s + kq;
            const int row = p * S + kk - PAD;
            float av = (km_ok && p < LOUT && row >= 0 && row < LIN) ? in[row * RS + m] : 0.f;
            if (km == KM && p < LOUT) av = 1.f;
            const float bv = p < LOUT ? dout[p * DS + n] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ko = rt * 16 + 4 * kq + q;
            if (ko <= KM) dwl[ko * WS + n] = acc[q];
        }
    }
    __syncthreads();
}
// the bias gradient row of ae_wgrad's result -> the partial record (dout's padding channels hold zeros, so entries past the layer's
// channel count come out zero); call between ae_wgrad and the barrier that ends ae_unstage
template <int KM, int N>
__device__ __forceinline__ void ae_bias_out(const float* __restrict__ dwl, float* __restrict__ part, int n_store) {
    if ((int)threadIdx.x < n_store) part[threadIdx.x] = dwl[KM * (N + 1) + threadIdx.x];
}

// weight index maps: parameter element idx -> (kk, m, n) of the staged layout
// Conv1d weight (CO, CI, KW) as the forward operand: kk = k, m = ci, n = co
template <int CI, int KW> struct MapConvFwd {
    __device__ static void at(int idx, int& kk, int& m, int& n) { const int k = idx % KW, r = idx / KW; kk = k; m = r % CI; n = r / CI; }
};
// Conv1d weight (CO, CI, KW) as the input-gradient operand (taps reversed, channel roles swapped): kk = KW - 1 - k, m = co, n = ci
template <int CI, int KW> struct MapConvDgrad {
    __device__ static void at(int idx, int& kk, int& m, int& n) { const int k = idx % KW, r = idx / KW; kk = KW - 1 - k; n = r % CI; m = r / CI; }
};
// ConvTranspose1d weight (CI, CO, KW) as the forward operand: out[q] = sum_k x[q - k] W[:, :, k] -> kk = KW - 1 - k with PAD = KW - 1
template <int CO, int KW> struct MapConvTFwd {
    __device__ static void at(int idx, int& kk, int& m, int& n) { const int k = idx % KW, r = idx / KW; kk = KW - 1 - k; n = r % CO; m = r / CO; }
};
// ConvTranspose1d weight (CI, CO, KW) as the input-gradient operand: dx[p] = sum_k dy[p + k] W[:, :, k]^T -> kk = k, m = co, n = ci
template <int CO, int KW> struct MapConvTDgrad {
    __device__ static void at(int idx, int& kk, int& m, int& n) { const int k = idx % KW, r = idx / KW; kk = k; m = r % CO; n = r / CO; }
};

template <int CINP, int N, class MAP>
__device__ __forceinline__ void ae_stage_t(const float* __restrict__ W, int numel, float* __restrict__ wl, int tid, int nt) {
    constexpr int WS = N + 1;
    for (int idx = tid; idx < numel; idx += nt) {
        int kk, m, n;
        MAP::at(idx, kk, m, n);
        wl[(kk * CINP + m) * WS + n] = W[idx];
    }
}
template <int CINP, int N, class MAP>
__device__ __forceinline__ void ae_stage(const float* __restrict__ W, int numel, float* __restrict__ wl, int rows, bool zero_first) {
    constexpr int WS = N + 1;
    if (zero_first) {
        for (int i = threadIdx.x; i < rows * WS; i += AE_NT) wl[i] = 0.f;
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < numel; idx += AE_NT) {
        int kk, m, n;
        MAP::at(idx, kk, m, n);
        wl[(kk * CINP + m) * WS + n] = W[idx];
    }
    __syncthreads();
}
template <int CINP, int N, class MAP>
__device__ __forceinline__ void ae_unstage(float* __restrict__ part, int numel, const float* __restrict__ dwl) {
    constexpr int WS = N + 1;
    for (int idx = threadIdx.x; idx < numel; idx += AE_NT) {
        int kk, m, n;
        MAP::at(idx, kk, m, n);
        part[idx] = dwl[(kk * CINP + m) * WS + n];
    }
    __syncthreads();
}

// ---- linear layers of one clip --------------------------------------------------------------------------------------------------------
// out[n] = b[n] + sum_k Wt[k][n] x[k] with the TRANSPOSED weight (phase 1 writes it once per step): a thread owns four consecutive n (one
// 16-byte load per k, lanes contiguous) and a slice of k, no cross-lane reduction; the slices meet in LDS (scratch: AE_BUF floats)
template <int K, int N>
__device__ __forceinline__ void ae_fc(const float* __restrict__ Wt, const float* __restrict__ b, const float* __restrict__ x, float* __restrict__ out,
                                      float* __restrict__ scratch) {
    static_assert(N % 4 == 0, "16-byte pieces");
    constexpr int NQ = N / 4, KP0 = AE_NT / NQ, KP1 = KP0 < K ? KP0 : K, KP = KP1 * N <= AE_BUF ? KP1 : AE_BUF / N, PER = (K + KP - 1) / KP;
    static_assert(KP >= 1 && KP * N <= AE_BUF, "scratch");
    const int nq = threadIdx.x % NQ, kp = threadIdx.x / NQ;
    if (kp < KP) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int k0 = kp * PER, k1 = k0 + PER < K ? k0 + PER : K;
#pragma unroll 16
        for (int k = k0; k < k1; ++k) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(Wt + (long)k * N + 4 * nq);
            const float xv = x[k];
            acc[0] = fmaf(w[0], xv, acc[0]); acc[1] = fmaf(w[1], xv, acc[1]); acc[2] = fmaf(w[2], xv, acc[2]); acc[3] = fmaf(w[3], xv, acc[3]);
        }
        *reinterpret_cast<f32x4*>(scratch + kp * N + 4 * nq) = acc;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += AE_NT) {
        float s_ = b[n];
        for (int q = 0; q < KP; ++q) s_ += scratch[q * N + n];
        out[n] = s_;
    }
    __syncthreads();
}
// dx[k] = sum_n W[n][k] dy[n]: a thread per four k and a slice of n, slices summed through LDS (scratch: AE_BUF floats)
template <int K, int N>
__device__ __forceinline__ void ae_fc_dgrad(const float* __restrict__ W, const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ scratch) {
    constexpr int KQ = K / 4, NG0 = AE_NT / KQ, NG1 = NG0 < N ? NG0 : N, NG = NG1 * K <= AE_BUF ? NG1 : AE_BUF / K, PER = (N + NG - 1) / NG;
    static_assert(NG >= 1 && NG * K <= AE_BUF, "scratch");
    const int kq = threadIdx.x % KQ, ng = threadIdx.x / KQ;
    if (ng < NG) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int n0 = ng * PER, n1 = n0 + PER < N ? n0 + PER : N;
#pragma unroll 16
        for (int n = n0; n < n1; ++n) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)n * K + 4 * kq);
            const float d = dy[n];
            acc[0] = fmaf(w[0], d, acc[0]); acc[1] = fmaf(w[1], d, acc[1]); acc[2] = fmaf(w[2], d, acc[2]); acc[3] = fmaf(w[3], d, acc[3]);
        }
        *reinterpret_cast<f32x4*>(scratch + ng * K + 4 * kq) = acc;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += AE_NT) {
        float s = 0.f;
        for (int q = 0; q < NG; ++q) s += scratch[q * K + k];
        dx[k] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ float ae_sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// transposed linear weights in the workspace (floats): [K][N] each
constexpr int WT_F1 = 0;
constexpr int WT_F2 = WT_F1 + 384 * 256;
constexpr int WT_F3 = WT_F2 + 256 * 128;
constexpr int WT_MU = WT_F3 + 128 * 32;
constexpr int WT_D0 = WT_MU + 32 * 32;
constexpr int WT_D1 = WT_D0 + 32 * 64;
constexpr int AE_WT = WT_D1 + 64 * 136;
struct AeWtLayer { int p, n, k, begin; };
__constant__ AeWtLayer ae_wt_layers[7] = {{P_F1W, 256, 384, WT_F1}, {P_F2W, 128, 256, WT_F2}, {P_F3W, 32, 128, WT_F3}, {P_MUW, 32, 32, WT_MU},
                                          {P_D0W, 64, 32, WT_D0},   {P_D1W, 136, 64, WT_D1},  {0, 0, 0, AE_WT}};

// ---- phases 1..17 ----------------------------------------------------------------------------------------------------------------------
// Every phase opens with a PROLOGUE in which the waves split the independent global reads -- waves 0-3 stage the first layer's weights,
// waves 4-7 fetch the saved tensors and finish the BatchNorm statistics -- and meet at one barrier (three or four dependent round trips to
// L2 became one).  A backward phase runs the input gradient FIRST (its sums leave as atomics as early as possible) and the layer's weight
// gradient last: nothing in the phase waits for it.
template <int PH>
__global__ __launch_bounds__(AE_NT) void ae_phase_kernel(const AeArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[AE_LDS_FLOATS];
    const AeLds l = ae_lds(smem);
    const int clip = blockIdx.x;
    float* const act = a.act + (size_t)clip * AE_ACT;
    float* const part = a.part + (size_t)clip * AE_PART;
    const float* const x = a.x + (size_t)clip * 34 * 27;
    const double nB = (double)a.B;
    float *B0 = l.b[0], *B1 = l.b[1], *B2 = l.b[2], *B3 = l.b[3], *B4 = l.b[4], *B5 = l.b[5];
    auto W = [&](int p) { return a.P + a.off[p]; };
    auto slot = [&](int s) { return a.sums + (size_t)s * 512; };
    constexpr int H = AE_NT / 2;
    const bool lo = (int)threadIdx.x < H;                 // waves 0-3 / waves 4-7 of the prologue
    const int th = lo ? threadIdx.x : threadIdx.x - H;

    if constexpr (PH == 1) {                    // poses -> net.0 conv; the linear layers' weights transposed for this step's forward
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.step != nullptr) *a.step += 1;       // Adam's step counter (read by phase 18 only)
        for (int i = threadIdx.x; i < 3 * 28 * 33; i += AE_NT) l.wl[i] = 0.f;
        for (int e = blockIdx.x * AE_NT + threadIdx.x; e < AE_WT; e += gridDim.x * AE_NT) {
            int li = 0;
            while (e >= ae_wt_layers[li + 1].begin) ++li;
            const AeWtLayer L = ae_wt_layers[li];
            const int r = e - L.begin, k = r / L.n, n = r - k * L.n;
            a.wt[e] = a.P[a.off[L.p] + n * L.k + k];
        }
        __syncthreads();
        if (lo) ae_stage_t<28, 32, MapConvFwd<27, 3>>(W(P_E0W), 32 * 27 * 3, l.wl, th, H);
        else ae_load_x_t(B0, x, th, H);
        __syncthreads();
        ae_conv<34, 28, 36, 32, 32, 3, 1, 0>(B0, l.wl, W(P_E0B), 32, B1);
        ae_store<32, 32>(act + A_C0, B1);
        ae_chan_sums<32, 32, false>(l, l.s1, B1, nullptr, slot(0));
    } else if constexpr (PH == 2) {             // BN0 + LeakyReLU(0.2) -> net.1 conv
        if (lo) ae_stage_t<32, 64, MapConvFwd<32, 3>>(W(P_E1W), 64 * 32 * 3, l.wl, th, H);
        else { ae_load_t<32, 32>(B0, act + A_C0, th, H); ae_bn_prepare_t<32>(a, l.s1, 0, 0, nB * 32, P_BN0G, true, th, H); }
        __syncthreads();
        ae_bn_act<32, 32>(l.s1, B1, B0, 0.2f);
        ae_conv<32, 32, 36, 30, 64, 3, 1, 0>(B1, l.wl, W(P_E1B), 64, B2);
        ae_store<30, 64>(act + A_C1, B2);
        ae_chan_sums<30, 64, false>(l, l.s1, B2, nullptr, slot(1));
    } else if constexpr (PH == 3) {             // BN1 -> net.2 conv (k 4, stride 2)
        if (lo) ae_stage_t<64, 64, MapConvFwd<64, 4>>(W(P_E2W), 64 * 64 * 4, l.wl, th, H);
        else { ae_load_t<30, 64>(B0, act + A_C1, th, H); ae_bn_prepare_t<64>(a, l.s1, 1, 1, nB * 30, P_BN1G, true, th, H); }
        __syncthreads();
        ae_bn_act<30, 64>(l.s1, B1, B0, 0.2f);
        ae_conv<30, 64, 68, 14, 64, 4, 2, 0>(B1, l.wl, W(P_E2B), 64, B2);
        ae_store<14, 64>(act + A_C2, B2);
        ae_chan_sums<14, 64, false>(l, l.s1, B2, nullptr, slot(2));
    } else if constexpr (PH == 4) {             // BN2 -> net.3 conv -> flatten -> out_net.0
        if (lo) ae_stage_t<64, 32, MapConvFwd<64, 3>>(W(P_E3W), 32 * 64 * 3, l.wl, th, H);
        else { ae_load_t<14, 64>(B0, act + A_C2, th, H); ae_bn_prepare_t<64>(a, l.s1, 2, 2, nB * 14, P_BN2G, true, th, H); }
        __syncthreads();
        ae_bn_act<14, 64>(l.s1, B1, B0, 0.2f);
        ae_conv<14, 64, 68, 12, 32, 3, 1, 0>(B1, l.wl, W(P_E3B), 32, B2);
        for (int i = threadIdx.x; i < 384; i += AE_NT) B3[i] = B2[(i % 12) * 36 + i / 12];       // flat[c * 12 + l] = c4[l][c]
        __syncthreads();
        ae_store<1, 384>(act + A_FLAT, B3);
        ae_fc<384, 256>(a.wt + WT_F1, W(P_F1B), B3, B4, B5);
        ae_store<1, 256>(act + A_F1, B4);
        ae_chan_sums<1, 256, false>(l, l.s1, B4, nullptr, slot(3));
    } else if constexpr (PH == 5) {             // BN3 (LeakyReLU(True): slope 1) -> out_net.3
        if (lo) ae_load_t<1, 256>(B0, act + A_F1, th, H);
        else ae_bn_prepare_t<256>(a, l.s1, 3, 3, nB, P_BN3G, true, th, H);
        __syncthreads();
        ae_bn_act<1, 256>(l.s1, B1, B0, 1.f);
        ae_store<1, 256>(act + A_Y1F, B1);
        ae_fc<256, 128>(a.wt + WT_F2, W(P_F2B), B1, B2, B5);
        ae_store<1, 128>(act + A_F2, B2);
        ae_chan_sums<1, 128, false>(l, l.s1, B2, nullptr, slot(4));
    } else if constexpr (PH == 6) {             // BN4 -> out_net.6 -> fc_mu (z = mu) -> decoder.pre_net.0
        if (lo) ae_load_t<1, 128>(B0, act + A_F2, th, H);
        else ae_bn_prepare_t<128>(a, l.s1, 4, 4, nB, P_BN4G, true, th, H);
        __syncthreads();
        ae_bn_act<1, 128>(l.s1, B1, B0, 1.f);
        ae_store<1, 128>(act + A_Y2F, B1);
        ae_fc<128, 32>(a.wt + WT_F3, W(P_F3B), B1, B2, B5);
        ae_store<1, 32>(act + A_F3, B2);
        ae_fc<32, 32>(a.wt + WT_MU, W(P_MUB), B2, B3, B5);
        ae_store<1, 32>(act + A_MU, B3);
        if (a.feat != nullptr) ae_store<1, 32>(a.feat + (size_t)clip * 32, B3);
        ae_fc<32, 64>(a.wt + WT_D0, W(P_D0B), B3, B4, B5);
        ae_store<1, 64>(act + A_P0, B4);
        ae_chan_sums<1, 64, false>(l, l.s1, B4, nullptr, slot(5));
    } else if constexpr (PH == 7) {             // BN5 -> pre_net.3 -> view(4, 34) -> net.0 transposed conv
        if (lo) ae_stage_t<4, 32, MapConvTFwd<32, 3>>(W(P_T0W), 4 * 32 * 3, l.wl, th, H);
        else { ae_load_t<1, 64>(B0, act + A_P0, th, H); ae_bn_prepare_t<64>(a, l.s1, 5, 5, nB, P_BN5G, true, th, H); }
        __syncthreads();
        ae_bn_act<1, 64>(l.s1, B1, B0, 1.f);
        ae_store<1, 64>(act + A_YP, B1);
        ae_fc<64, 136>(a.wt + WT_D1, W(P_D1B), B1, B2, B5);
        ae_store<1, 136>(act + A_P3, B2);
        for (int i = threadIdx.x; i < 136; i += AE_NT) B3[i] = B2[(i & 3) * 34 + (i >> 2)];       // x0[l][c] = p3[c * 34 + l], rows 4 floats apart
        __syncthreads();
        ae_conv<34, 4, 4, 36, 32, 3, 1, 2>(B3, l.wl, W(P_T0B), 32, B4);
        ae_store<36, 32>(act + A_T0, B4);
        ae_chan_sums<36, 32, false>(l, l.s1, B4, nullptr, slot(6));
    } else if constexpr (PH == 8) {             // BN6 -> net.3 transposed conv
        if (lo) ae_stage_t<32, 32, MapConvTFwd<32, 3>>(W(P_T1W), 32 * 32 * 3, l.wl, th, H);
        else { ae_load_t<36, 32>(B0, act + A_T0, th, H); ae_bn_prepare_t<32>(a, l.s1, 6, 6, nB * 36, P_BN6G, true, th, H); }
        __syncthreads();
        ae_bn_act<36, 32>(l.s1, B1, B0, 0.2f);
        ae_conv<36, 32, 36, 38, 32, 3, 1, 2>(B1, l.wl, W(P_T1B), 32, B2);
        ae_store<38, 32>(act + A_T1, B2);
        ae_chan_sums<38, 32, false>(l, l.s1, B2, nullptr, slot(7));
    } else if constexpr (PH == 9) {             // BN7 -> net.6, net.7 -> loss -> back through net.7, net.6 to BN7's output gradient
        // all four staged operands of the phase at once (96 rows x 33 floats each; a fifth region takes the weight gradients): net.6 / net.7
        // forward, net.7 / net.6 input gradient; net.7 has 27 channels: its padding columns / rows are zeroed here (disjoint from the scatter)
        constexpr int R = 96 * 33;
        float* const wg = l.wl + 4 * R;
        if (lo) {
            ae_stage_t<32, 32, MapConvFwd<32, 3>>(W(P_C6W), 32 * 32 * 3, l.wl, th, H);
            ae_stage_t<32, 32, MapConvFwd<32, 3>>(W(P_C7W), 27 * 32 * 3, l.wl + R, th, H);
            for (int i = th; i < 96 * 5; i += H) l.wl[R + (i / 5) * 33 + 27 + i % 5] = 0.f;
            ae_stage_t<32, 32, MapConvDgrad<32, 3>>(W(P_C6W), 32 * 32 * 3, l.wl + 3 * R, th, H);
        } else {
            ae_load_t<38, 32>(B0, act + A_T1, th, H); ae_bn_prepare_t<32>(a, l.s1, 7, 7, nB * 38, P_BN7G, true, th, H);
            ae_load_x_t(B4, x, th, H);
            ae_stage_t<32, 32, MapConvDgrad<32, 3>>(W(P_C7W), 27 * 32 * 3, l.wl + 2 * R, th, H);
            for (int i = th; i < 15 * 33; i += H) { const int r = i / 33; l.wl[2 * R + ((r / 5) * 32 + 27 + r % 5) * 33 + i % 33] = 0.f; }
        }
        __syncthreads();
        ae_bn_act<38, 32>(l.s1, B1, B0, 0.2f);                                                    // B1 = y(T1): net.6's input
        ae_conv<38, 32, 36, 36, 32, 3, 1, 0>(B1, l.wl, W(P_C6B), 32, B2);                         // B2 = c6
        ae_conv<36, 32, 36, 34, 32, 3, 1, 0>(B2, l.wl + R, W(P_C7B), 27, B3);                     // B3 = recon [34][32 + 4], channels 27..31 zero
        if (a.recon != nullptr)
            for (int i = threadIdx.x; i < 34 * 27; i += AE_NT) a.recon[(size_t)clip * 918 + i] = B3[(i / 27) * 36 + i % 27];
        for (int i = threadIdx.x; i < 34 * 32; i += AE_NT) { const int o = (i >> 5) * 36 + (i & 31); B4[o] = B3[o] - B4[o]; }      // e = recon - target
        __syncthreads();
        {   // L1 + L1 of the frame differences, mean over (frame, joint) per clip, summed over the batch (:63-72)
            const float w1 = 1.f / (34.f * 27.f), w2 = 1.f / (33.f * 27.f);
            float s = 0.f;
            for (int i = threadIdx.x; i < 34 * 32; i += AE_NT) {
                const int t = i >> 5, o = t * 36 + (i & 31);
                const float e = B4[o];
                float g = ae_sgn(e) * w1;
                s += fabsf(e) * w1;
                if (t >= 1) { const float d = e - B4[o - 36]; s += fabsf(d) * w2; g += ae_sgn(d) * w2; }
                if (t + 1 < 34) { const float d = B4[o + 36] - e; g -= ae_sgn(d) * w2; }
                B3[o] = g;                                                                         // d(loss) / d(recon)
            }
            l.red[threadIdx.x] = s;
            __syncthreads();
            if (threadIdx.x < 64) {
                double tot = 0.0;
                for (int i = threadIdx.x; i < AE_NT; i += 64) tot += (double)l.red[i];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
                if (threadIdx.x == 0) atomicAdd(slot(16), tot);
            }
            __syncthreads();
        }
        // input gradients first: net.7 -> d c6 (B4), net.6 -> d y(T1) (B5)
        ae_conv<34, 32, 36, 36, 32, 3, 1, 2>(B3, l.wl + 2 * R, nullptr, 0, B4);
        ae_conv<36, 32, 36, 38, 32, 3, 1, 2>(B4, l.wl + 3 * R, nullptr, 0, B5);
        ae_act_bwd<38, 32>(l.s1, B5, B0, 0.2f);
        ae_store<38, 32>(act + A_G, B5);
        ae_chan_sums<38, 32, true>(l, l.s1, B5, B0, slot(8));
        // weight / bias gradients
        ae_wgrad<36, 32, 36, 34, 32, 3, 1, 0>(B2, B3, wg);
        ae_bias_out<96, 32>(wg, part + Q_C7B, 28);
        ae_unstage<32, 32, MapConvFwd<32, 3>>(part + Q_C7W, 27 * 32 * 3, wg);
        ae_wgrad<38, 32, 36, 36, 32, 3, 1, 0>(B1, B4, wg);
        ae_bias_out<96, 32>(wg, part + Q_C6B, 32);
        ae_unstage<32, 32, MapConvFwd<32, 3>>(part + Q_C6W, 32 * 32 * 3, wg);
    } else if constexpr (PH == 10) {            // BN7 backward -> net.3 (transposed conv) -> BN6's output gradient
        if (lo) ae_stage_t<32, 32, MapConvTDgrad<32, 3>>(W(P_T1W), 32 * 32 * 3, l.wl, th, H);
        else {
            ae_load_t<38, 32>(B0, act + A_G, th, H); ae_load_t<38, 32>(B1, act + A_T1, th, H); ae_load_t<36, 32>(B2, act + A_T0, th, H);
            ae_bn_prepare_t<32>(a, l.s1, 7, 7, nB * 38, P_BN7G, false, th, H); ae_bwd_means_t<32>(a, l, 8, nB * 38, th, H);
            ae_bn_prepare_t<32>(a, l.s2, 6, 6, nB * 36, P_BN6G, false, th, H);
        }
        __syncthreads();
        ae_bn_bwd<38, 32>(l, l.s1, B0, B1);                                                       // B0 = d t1
        ae_conv<38, 32, 36, 36, 32, 3, 1, 0>(B0, l.wl, nullptr, 0, B4);                           // B4 = d y(T0)
        ae_act_bwd<36, 32>(l.s2, B4, B2, 0.2f);
        ae_store<36, 32>(act + A_G, B4);
        ae_chan_sums<36, 32, true>(l, l.s2, B4, B2, slot(9));
        ae_bn_act<36, 32>(l.s2, B3, B2, 0.2f);                                                    // B3 = y(T0): the layer's input
        ae_wgrad<36, 32, 36, 38, 32, 3, 1, 2>(B3, B0, l.wl);
        ae_bias_out<96, 32>(l.wl, part + Q_T1B, 32);
        ae_unstage<32, 32, MapConvTFwd<32, 3>>(part + Q_T1W, 32 * 32 * 3, l.wl);
    } else if constexpr (PH == 11) {            // BN6 backward -> net.0 (transposed conv) -> pre_net.3 -> BN5's output gradient
        if (lo) ae_stage_t<32, 4, MapConvTDgrad<32, 3>>(W(P_T0W), 4 * 32 * 3, l.wl, th, H);
        else {
            ae_load_t<36, 32>(B0, act + A_G, th, H); ae_load_t<36, 32>(B1, act + A_T0, th, H); ae_load_t<1, 136>(B2, act + A_P3, th, H);
            ae_load_t<1, 64>(B5, act + A_P0, th, H);
            ae_bn_prepare_t<32>(a, l.s1, 6, 6, nB * 36, P_BN6G, false, th, H); ae_bwd_means_t<32>(a, l, 9, nB * 36, th, H);
            ae_bn_prepare_t<64>(a, l.s2, 5, 5, nB, P_BN5G, false, th, H);
        }
        __syncthreads();
        ae_bn_bwd<36, 32>(l, l.s1, B0, B1);                                                       // B0 = d t0
        for (int i = threadIdx.x; i < 136; i += AE_NT) B3[i] = B2[(i & 3) * 34 + (i >> 2)];       // x0[l][c] (the weight gradient's input, below)
        ae_conv_n4<36, 32, 36, 34, 3>(B0, l.wl, B4);                                              // B4 = d x0 [34][4]
        for (int i = threadIdx.x; i < 136; i += AE_NT) B1[i] = B4[(i % 34) * 4 + i / 34];         // d p3[c * 34 + l]
        __syncthreads();
        ae_store<1, 136>(act + A_DP3, B1);
        ae_fc_dgrad<64, 136>(W(P_D1W), B1, B4, l.wl);                                             // B4 = d yp = g (slope 1); scratch: the staged weights are done
        ae_store<1, 64>(act + A_G, B4);
        ae_chan_sums<1, 64, true>(l, l.s2, B4, B5, slot(10));
        ae_wgrad<34, 4, 4, 36, 32, 3, 1, 2>(B3, B0, l.wl);
        ae_bias_out<12, 32>(l.wl, part + Q_T0B, 32);
        ae_unstage<4, 32, MapConvTFwd<32, 3>>(part + Q_T0W, 4 * 32 * 3, l.wl);
    } else if constexpr (PH == 12) {            // BN5 backward -> pre_net.0 -> fc_mu -> out_net.6 -> BN4's output gradient
        if (lo) { ae_load_t<1, 64>(B0, act + A_G, th, H); ae_load_t<1, 64>(B1, act + A_P0, th, H); ae_load_t<1, 128>(B5, act + A_F2, th, H); }
        else {
            ae_bn_prepare_t<64>(a, l.s1, 5, 5, nB, P_BN5G, false, th, H); ae_bwd_means_t<64>(a, l, 10, nB, th, H);
            ae_bn_prepare_t<128>(a, l.s2, 4, 4, nB, P_BN4G, false, th, H);
        }
        __syncthreads();
        ae_bn_bwd<1, 64>(l, l.s1, B0, B1);                                                        // B0 = d p0
        ae_store<1, 64>(act + A_DP0, B0);
        ae_fc_dgrad<32, 64>(W(P_D0W), B0, B2, B4);                                                // B2 = d mu
        ae_store<1, 32>(act + A_DMU, B2);
        ae_fc_dgrad<32, 32>(W(P_MUW), B2, B3, B4);                                                // B3 = d f3
        ae_store<1, 32>(act + A_DF3, B3);
        ae_fc_dgrad<128, 32>(W(P_F3W), B3, B0, B4);                                               // B0 = d y2f = g
        ae_store<1, 128>(act + A_G, B0);
        ae_chan_sums<1, 128, true>(l, l.s2, B0, B5, slot(11));
    } else if constexpr (PH == 13) {            // BN4 backward -> out_net.3 -> BN3's output gradient
        if (lo) { ae_load_t<1, 128>(B0, act + A_G, th, H); ae_load_t<1, 128>(B1, act + A_F2, th, H); ae_load_t<1, 256>(B5, act + A_F1, th, H); }
        else {
            ae_bn_prepare_t<128>(a, l.s1, 4, 4, nB, P_BN4G, false, th, H); ae_bwd_means_t<128>(a, l, 11, nB, th, H);
            ae_bn_prepare_t<256>(a, l.s2, 3, 3, nB, P_BN3G, false, th, H);
        }
        __syncthreads();
        ae_bn_bwd<1, 128>(l, l.s1, B0, B1);                                                       // B0 = d f2
        ae_store<1, 128>(act + A_DF2, B0);
        ae_fc_dgrad<256, 128>(W(P_F2W), B0, B2, B4);                                              // B2 = d y1f = g
        ae_store<1, 256>(act + A_G, B2);
        ae_chan_sums<1, 256, true>(l, l.s2, B2, B5, slot(12));
    } else if constexpr (PH == 14) {            // BN3 backward -> out_net.0 -> un-flatten -> net.3 conv -> BN2's output gradient
        if (lo) ae_stage_t<32, 64, MapConvDgrad<64, 3>>(W(P_E3W), 32 * 64 * 3, l.wl, th, H);
        else {
            ae_load_t<1, 256>(B0, act + A_G, th, H); ae_load_t<1, 256>(B1, act + A_F1, th, H); ae_load_t<14, 64>(B5, act + A_C2, th, H);
            ae_bn_prepare_t<256>(a, l.s1, 3, 3, nB, P_BN3G, false, th, H); ae_bwd_means_t<256>(a, l, 12, nB, th, H);
            ae_bn_prepare_t<64>(a, l.s2, 2, 2, nB * 14, P_BN2G, false, th, H);
        }
        __syncthreads();
        ae_bn_bwd<1, 256>(l, l.s1, B0, B1);                                                       // B0 = d f1
        ae_store<1, 256>(act + A_DF1, B0);
        ae_fc_dgrad<384, 256>(W(P_F1W), B0, B2, B4);                                              // B2 = d flat
        for (int i = threadIdx.x; i < 384; i += AE_NT) B3[(i >> 5) * 36 + (i & 31)] = B2[(i & 31) * 12 + (i >> 5)];      // d c4[l][c] = d flat[c * 12 + l]
        __syncthreads();
        ae_conv<12, 32, 36, 14, 64, 3, 1, 2>(B3, l.wl, nullptr, 0, B4);                           // B4 = d y2
        ae_act_bwd<14, 64>(l.s2, B4, B5, 0.2f);
        ae_store<14, 64>(act + A_G, B4);
        ae_chan_sums<14, 64, true>(l, l.s2, B4, B5, slot(13));
        ae_bn_act<14, 64>(l.s2, B1, B5, 0.2f);                                                    // B1 = y2
        ae_wgrad<14, 64, 68, 12, 32, 3, 1, 0>(B1, B3, l.wl);
        ae_bias_out<192, 32>(l.wl, part + Q_E3B, 32);
        ae_unstage<64, 32, MapConvFwd<64, 3>>(part + Q_E3W, 32 * 64 * 3, l.wl);
    } else if constexpr (PH == 15) {            // BN2 backward -> net.2 conv (stride 2) -> BN1's output gradient
        if (lo) ae_stage_t<64, 64, MapConvDgrad<64, 4>>(W(P_E2W), 64 * 64 * 4, l.wl, th, H);
        else {
            ae_load_t<14, 64>(B0, act + A_G, th, H); ae_load_t<14, 64>(B1, act + A_C2, th, H); ae_load_t<30, 64>(B2, act + A_C1, th, H);
            ae_bn_prepare_t<64>(a, l.s1, 2, 2, nB * 14, P_BN2G, false, th, H); ae_bwd_means_t<64>(a, l, 13, nB * 14, th, H);
            ae_bn_prepare_t<64>(a, l.s2, 1, 1, nB * 30, P_BN1G, false, th, H);
        }
        __syncthreads();
        ae_bn_bwd<14, 64>(l, l.s1, B0, B1);                                                       // B0 = d c2
        // input gradient of the stride-2 conv as a stride-1 transposed conv over d c2 with a zero row between its rows
        for (int i = threadIdx.x; i < 27 * 64; i += AE_NT) { const int r = i >> 6, c = i & 63; B5[r * 68 + c] = (r & 1) ? 0.f : B0[(r >> 1) * 68 + c]; }
        __syncthreads();
        ae_conv<27, 64, 68, 30, 64, 4, 1, 3>(B5, l.wl, nullptr, 0, B4);                           // B4 = d y1
        ae_act_bwd<30, 64>(l.s2, B4, B2, 0.2f);
        ae_store<30, 64>(act + A_G, B4);
        ae_chan_sums<30, 64, true>(l, l.s2, B4, B2, slot(14));
        ae_bn_act<30, 64>(l.s2, B3, B2, 0.2f);                                                    // B3 = y1
        ae_wgrad<30, 64, 68, 14, 64, 4, 2, 0>(B3, B0, l.wl);
        ae_bias_out<256, 64>(l.wl, part + Q_E2B, 64);
        ae_unstage<64, 64, MapConvFwd<64, 4>>(part + Q_E2W, 64 * 64 * 4, l.wl);
    } else if constexpr (PH == 16) {            // BN1 backward -> net.1 conv -> BN0's output gradient
        if (lo) ae_stage_t<64, 32, MapConvDgrad<32, 3>>(W(P_E1W), 64 * 32 * 3, l.wl, th, H);
        else {
            ae_load_t<30, 64>(B0, act + A_G, th, H); ae_load_t<30, 64>(B1, act + A_C1, th, H); ae_load_t<32, 32>(B2, act + A_C0, th, H);
            ae_bn_prepare_t<64>(a, l.s1, 1, 1, nB * 30, P_BN1G, false, th, H); ae_bwd_means_t<64>(a, l, 14, nB * 30, th, H);
            ae_bn_prepare_t<32>(a, l.s2, 0, 0, nB * 32, P_BN0G, false, th, H);
        }
        __syncthreads();
        ae_bn_bwd<30, 64>(l, l.s1, B0, B1);                                                       // B0 = d c1
        ae_conv<30, 64, 68, 32, 32, 3, 1, 2>(B0, l.wl, nullptr, 0, B4);                           // B4 = d y0
        ae_act_bwd<32, 32>(l.s2, B4, B2, 0.2f);
        ae_store<32, 32>(act + A_G, B4);
        ae_chan_sums<32, 32, true>(l, l.s2, B4, B2, slot(15));
        ae_bn_act<32, 32>(l.s2, B3, B2, 0.2f);                                                    // B3 = y0
        ae_wgrad<32, 32, 36, 30, 64, 3, 1, 0>(B3, B0, l.wl);
        ae_bias_out<96, 64>(l.wl, part + Q_E1B, 64);
        ae_unstage<32, 64, MapConvFwd<32, 3>>(part + Q_E1W, 64 * 32 * 3, l.wl);
    } else if constexpr (PH == 17) {            // BN0 backward -> net.0 conv weight gradient
        if (lo) { ae_load_t<32, 32>(B0, act + A_G, th, H); ae_load_t<32, 32>(B1, act + A_C0, th, H); }
        else { ae_load_x_t(B2, x, th, H); ae_bn_prepare_t<32>(a, l.s1, 0, 0, nB * 32, P_BN0G, false, th, H); ae_bwd_means_t<32>(a, l, 15, nB * 32, th, H); }
        __syncthreads();
        ae_bn_bwd<32, 32>(l, l.s1, B0, B1);                                                       // B0 = d c0
        ae_wgrad<34, 28, 36, 32, 32, 3, 1, 0>(B2, B0, l.wl);
        ae_bias_out<84, 32>(l.wl, part + Q_E0B, 32);
        ae_unstage<28, 32, MapConvFwd<27, 3>>(part + Q_E0W, 32 * 27 * 3, l.wl);
    }
}

// ---- phase 18: gradients into the slab --------------------------------------------------------------------------------------------------
struct AeFcLayer { int pw, n, k, x_off, dy_off, unit0, kc; };
__constant__ AeFcLayer ae_fc_layers[6] = {
    {P_F1W, 256, 384, A_FLAT, A_DF1, 0, 6},   {P_F2W, 128, 256, A_Y1F, A_DF2, 192, 4}, {P_F3W, 32, 128, A_Y2F, A_DF3, 256, 2},
    {P_MUW, 32, 32, A_F3, A_DMU, 264, 1},     {P_D0W, 64, 32, A_MU, A_DP0, 268, 1},    {P_D1W, 136, 64, A_YP, A_DP3, 276, 1}};
constexpr int AE_FC_UNITS = 293;
constexpr int AE_RED_UNITS = (AE_PART / 4 + 63) / 64;         // 159: 64 16-byte pieces per workgroup, eight batch slices each
constexpr int AE_TAIL_UNITS = AE_FC_UNITS + AE_RED_UNITS + 1;
__constant__ int ae_seg_begin[17] = {Q_E0W, Q_E0B, Q_E1W, Q_E1B, Q_E2W, Q_E2B, Q_E3W, Q_E3B, Q_T0W, Q_T0B, Q_T1W, Q_T1B, Q_C6W, Q_C6B, Q_C7W, Q_C7B, AE_PART};
__constant__ int ae_seg_param[16] = {P_E0W, P_E0B, P_E1W, P_E1B, P_E2W, P_E2B, P_E3W, P_E3B, P_T0W, P_T0B, P_T1W, P_T1B, P_C6W, P_C6B, P_C7W, P_C7B};
__constant__ int ae_bn_gamma[8] = {P_BN0G, P_BN1G, P_BN2G, P_BN3G, P_BN4G, P_BN5G, P_BN6G, P_BN7G};
__constant__ int ae_bn_ch[8] = {32, 64, 64, 256, 128, 64, 32, 32};

constexpr int AE_TAIL_NT = 512;
// torch.optim.Adam on one element whose gradient this thread has just finished (same arithmetic as adam_kernel, elementwise.hip)
__device__ __forceinline__ void ae_adam(const AeArgs& a, long i, float g, float step_size, float bc2s) {
    a.G[i] = g;
    if (a.M == nullptr) return;
    const float m = a.M[i] + (g - a.M[i]) * (1.f - a.b1);
    const float v = a.V[i] * a.b2 + g * g * (1.f - a.b2);
    a.M[i] = m; a.V[i] = v;
    a.P[i] -= step_size * m / (sqrtf(v) / bc2s + a.adam_eps);
}

__global__ __launch_bounds__(AE_TAIL_NT) void ae_tail_kernel(const AeArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[256 * 64];
    __shared__ __attribute__((aligned(16))) float ds[256 * 8];
    __shared__ float coef[2];
    const int u = blockIdx.x, t = threadIdx.x, B = a.B;
    if (t == 0) {                           // bias corrections in fp64 like the Python scalars torch.optim.Adam uses; the counter was advanced by phase 1
        const double st = a.step != nullptr ? (double)*a.step : 1.0;
        coef[0] = (float)((double)a.lr / (1.0 - pow((double)a.b1, st)));
        coef[1] = (float)sqrt(1.0 - pow((double)a.b2, st));
    }
    __syncthreads();
    const float step_size = coef[0], bc2s = coef[1];
    if (u < AE_FC_UNITS) {
        // linear layer weight gradient: rows [n0, n0 + 8) x columns [k0, k0 + 64) of dW = dy^T x, reduced over the batch here (batch order)
        int li = 0;
        while (li < 5 && u >= ae_fc_layers[li + 1].unit0) ++li;
        const AeFcLayer L = ae_fc_layers[li];
        const int lu = u - L.unit0, n0 = (lu / L.kc) * 8, k0 = (lu % L.kc) * 64;
        for (int i = t; i < B * 16; i += AE_TAIL_NT) {               // x[b][k0 .. k0 + 64) as 16-byte pieces
            const int b = i >> 4, q = (i & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 + q < L.k) v = *reinterpret_cast<const f32x4*>(a.act + (size_t)b * AE_ACT + L.x_off + k0 + q);
            *reinterpret_cast<f32x4*>(xs + b * 64 + q) = v;
        }
        for (int i = t; i < B * 8; i += AE_TAIL_NT) {
            const int b = i >> 3, r = i & 7;
            ds[i] = n0 + r < L.n ? a.act[(size_t)b * AE_ACT + L.dy_off + n0 + r] : 0.f;
        }
        __syncthreads();
        const int k = t & 63, r = t >> 6;
        float acc = 0.f;
#pragma unroll 4
        for (int b = 0; b < B; ++b) acc = fmaf(ds[b * 8 + r], xs[b * 64 + k], acc);
        if (n0 + r < L.n && k0 + k < L.k) ae_adam(a, a.off[L.pw] + (long)(n0 + r) * L.k + k0 + k, acc, step_size, bc2s);
        if (k0 == 0 && t < 8 && n0 + t < L.n) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += ds[b * 8 + t];
            ae_adam(a, a.off[L.pw + 1] + n0 + t, s, step_size, bc2s);
        }
    } else if (u < AE_FC_UNITS + AE_RED_UNITS) {
        // conv weight / bias gradients: the per-clip partials summed over the batch -- thread = (16-byte piece, one of eight batch slices), slices
        // combined in slice order through LDS (a fixed order: no atomics)
        const int pc = t & 63, sl = t >> 6;
        const int i4 = (u - AE_FC_UNITS) * 64 + pc;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (i4 < AE_PART / 4) {
            const int per = (B + 7) / 8, b0 = sl * per, b1 = b0 + per < B ? b0 + per : B;
#pragma unroll 16
            for (int b = b0; b < b1; ++b) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(a.part + (size_t)b * AE_PART + 4 * i4);
                acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
            }
        }
        *reinterpret_cast<f32x4*>(xs + (sl * 64 + pc) * 4) = acc;
        __syncthreads();
        if (sl == 0 && i4 < AE_PART / 4) {
            for (int q = 1; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xs + (q * 64 + pc) * 4);
                acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
            }
            int sg = 0;
            while (4 * i4 >= ae_seg_begin[sg + 1]) ++sg;
            const long o = a.off[ae_seg_param[sg]] + (4 * i4 - ae_seg_begin[sg]);
#pragma unroll
            for (int q = 0; q < 4; ++q) ae_adam(a, o + q, acc[q], step_size, bc2s);
        }
    } else {
        // BatchNorm gamma / beta gradients = the backward sums (slot 15 - bn: sum g * xhat, sum g); loss; Adam step counter; sums back to zero
        for (int bn = 0; bn < 8; ++bn) {
            const double* s = a.sums + (size_t)(15 - bn) * 512;
            for (int c = t; c < ae_bn_ch[bn]; c += AE_TAIL_NT) {
                ae_adam(a, a.off[ae_bn_gamma[bn]] + c, (float)s[256 + c], step_size, bc2s);
                ae_adam(a, a.off[ae_bn_gamma[bn] + 1] + c, (float)s[c], step_size, bc2s);
            }
        }
        if (t == 0) {
            *a.loss = (float)a.sums[16 * 512];
        }
        __syncthreads();
        for (int i = t; i < AE_SLOTS * 512; i += AE_TAIL_NT) a.sums[i] = 0.0;
    }
}

}  // namespace tg

using namespace tg;

extern "C" int32_t tg_ae_step_supported(int32_t B) { return B >= 2 && B <= 256 ? 1 : 0; }

extern "C" int64_t tg_ae_step_ws_bytes(int32_t B) {
    return (int64_t)AE_SLOTS * 512 * 8 + ((int64_t)B * AE_ACT + (int64_t)B * AE_PART + AE_WT) * 4;
}

extern "C" int tg_ae_train_step(const tg_ae_step_args* q, void* stream) {
    TG_REQUIRE(q && q->x && q->params && q->grads && q->ws && q->loss, "tg_ae_train_step: null pointer");
    TG_REQUIRE(tg_ae_step_supported(q->B), "tg_ae_train_step: batch %d outside [2, 256]", q->B);
    TG_REQUIRE(q->ws_bytes >= tg_ae_step_ws_bytes(q->B), "tg_ae_train_step: workspace too small");
    TG_REQUIRE(aligned16(q->x) && aligned16(q->params) && aligned16(q->grads) && aligned16(q->ws) && (q->feat == nullptr || aligned16(q->feat)),
               "tg_ae_train_step: operands must be 16-byte aligned");
    TG_REQUIRE(q->last_phase >= 0 && q->last_phase <= 18, "tg_ae_train_step: last_phase %d", q->last_phase);
    AeArgs a;
    a.x = q->x; a.P = q->params; a.G = q->grads;
    for (int i = 0; i < P_COUNT; ++i) {
        TG_REQUIRE(q->off[i] >= 0 && q->off[i] % 4 == 0, "tg_ae_train_step: parameter offset %d (entry %d) must be a non-negative multiple of 4 floats", q->off[i], i);
        a.off[i] = q->off[i];
    }
    for (int i = 0; i < 8; ++i) {
        TG_REQUIRE((q->running_mean[i] == nullptr) == (q->running_var[i] == nullptr), "tg_ae_train_step: running_mean / running_var go together");
        a.rm[i] = q->running_mean[i]; a.rv[i] = q->running_var[i]; a.nbt[i] = (long long*)q->num_batches_tracked[i];
    }
    a.sums = (double*)q->ws;
    a.act = (float*)((char*)q->ws + (size_t)AE_SLOTS * 512 * 8);
    a.part = a.act + (size_t)q->B * AE_ACT;
    a.wt = a.part + (size_t)q->B * AE_PART;
    a.loss = q->loss; a.recon = q->recon; a.feat = q->feat; a.step = q->step;
    a.B = q->B; a.bn_eps = q->bn_eps; a.momentum = q->momentum;
    TG_REQUIRE((q->adam_m == nullptr) == (q->adam_v == nullptr), "tg_ae_train_step: adam_m / adam_v go together");
    TG_REQUIRE(q->adam_m == nullptr || (q->lr > 0.f && q->beta1 >= 0.f && q->beta1 < 1.f && q->beta2 >= 0.f && q->beta2 < 1.f && q->adam_eps > 0.f),
               "tg_ae_train_step: Adam hyper-parameters");
    a.M = q->adam_m; a.V = q->adam_v; a.lr = q->lr; a.b1 = q->beta1; a.b2 = q->beta2; a.adam_eps = q->adam_eps;
    hipStream_t s = (hipStream_t)stream;
    const int last = q->last_phase == 0 ? 18 : q->last_phase;
    const dim3 grid(q->B), block(AE_NT);
#define TG_AE_PHASE(PH_)                                                                     \
    do {                                                                                     \
        if (last >= PH_) hipLaunchKernelGGL(ae_phase_kernel<PH_>, grid, block, 0, s, a);     \
    } while (0)
    TG_AE_PHASE(1); TG_AE_PHASE(2); TG_AE_PHASE(3); TG_AE_PHASE(4); TG_AE_PHASE(5); TG_AE_PHASE(6); TG_AE_PHASE(7); TG_AE_PHASE(8); TG_AE_PHASE(9);
    TG_AE_PHASE(10); TG_AE_PHASE(11); TG_AE_PHASE(12); TG_AE_PHASE(13); TG_AE_PHASE(14); TG_AE_PHASE(15); TG_AE_PHASE(16); TG_AE_PHASE(17);
#undef TG_AE_PHASE
    if (last >= 18) hipLaunchKernelGGL(ae_tail_kernel, dim3(AE_TAIL_UNITS), dim3(AE_TAIL_NT), 0, s, a);
    return check_launch("tg_ae_train_step");
}
