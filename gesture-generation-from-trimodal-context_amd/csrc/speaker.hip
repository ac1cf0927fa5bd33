// Speaker / style path of PoseGenerator (model/multimodal_context_net.py:83-95,125-137; model/embedding_net.py:10-13 reparameterize):
//   se = Embedding(vid)   zc = Linear(16,16)(se)   mu = Linear(16,16)(zc)   logvar = Linear(16,16)(zc)   z = mu + eps * exp(0.5 * logvar)
// and z repeated over the T frames into its columns of the GRU input.  384 x 16 numbers: as separate launches (gather, three GEMMs,
// reparameterise, repeat; backward: two clones, reparameterise, three weight-gradient + three input-gradient GEMMs, scatter) the path cost
// ~19 launches of 4.6-8 us.  Here: one forward launch (one 16-lane group per batch row) and one backward launch (one workgroup: every
// operand sits in LDS, weight gradients summed over the batch in fixed order).
#include "common.hpp"

namespace tg {

constexpr int SZ = 16;                 // style vector size (fixed by the reference: nn.Embedding(n, 16), nn.Linear(16, 16))

__global__ __launch_bounds__(256) void speaker_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ vid, int n_rows,
                                                          const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ wmu,
                                                          const float* __restrict__ bmu, const float* __restrict__ wlv, const float* __restrict__ blv,
                                                          float* __restrict__ eps, float* __restrict__ se, float* __restrict__ zc,
                                                          float* __restrict__ mu, float* __restrict__ lv, float* __restrict__ z, int B,
                                                          float* __restrict__ rep, long rep_ld, int T, const uint64_t* __restrict__ st, uint32_t site) {
    const int j = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int bb = b < B ? b : B - 1;                    // whole 16-lane groups take part in the shuffles
    const int64_t id = vid[bb];
    const float s = (id >= 0 && id < n_rows) ? table[id * SZ + j] : 0.f;
    // y[j] = bias[j] + sum_k W[j][k] x[k], x spread over the 16 lanes of the group
    auto linear = [&](const float* __restrict__ W, const float* __restrict__ bias, float x) {
        float acc = bias[j];
#pragma unroll
        for (int k = 0; k < SZ; ++k) acc = __builtin_fmaf(W[j * SZ + k], __shfl(x, (threadIdx.x & 48) + k), acc);
        return acc;
    };
    const float c = linear(w1, b1, s);
    const float m = linear(wmu, bmu, c);
    const float l = linear(wlv, blv, c);
    float e;
    if (st) {                                            // element bb * 16 + j of tg_normal(eps, B * 16, st, site), bit for bit (normal_kernel)
        const long el = (long)bb * SZ + j;
        uint32_t r[4];
        philox4x32(st[0], (uint64_t)(el >> 2), site, (uint32_t)st[1], r);
        const int q = (int)(el & 3) >> 1;
        const float rad = sqrtf(-2.f * logf(u01(r[2 * q])));
        const float ang = 6.283185307179586f * u01(r[2 * q + 1]);
        e = (el & 1) ? rad * sinf(ang) : rad * cosf(ang);
        if (b < B) eps[el] = e;
    } else {
        e = eps[bb * SZ + j];
    }
    const float zz = m + e * expf(0.5f * l);
    if (b < B) {
        const long o = (long)b * SZ + j;
        se[o] = s; zc[o] = c; mu[o] = m; lv[o] = l; z[o] = zz;
        if (rep)
            for (int t = 0; t < T; ++t) rep[((long)b * T + t) * rep_ld + j] = zz;
    }
}

// dz [nb][16]: gradient w.r.t. z (already summed over the frames).  d_mu_in / d_lv_in: direct gradients (KLD term) or NULL.
// One workgroup of 1024 threads: every operand in LDS; the three 16 x 16 weight gradients are 768 entries, one thread each, summed over the
// batch in order.
__global__ __launch_bounds__(1024) void speaker_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ d_mu_in, const float* __restrict__ d_lv_in,
                                                           const float* __restrict__ lv, const float* __restrict__ eps, const float* __restrict__ zc,
                                                           const float* __restrict__ se, const int64_t* __restrict__ vid, int n_rows,
                                                           const float* __restrict__ w1, const float* __restrict__ wmu, const float* __restrict__ wlv,
                                                           float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dwmu,
                                                           float* __restrict__ dbmu, float* __restrict__ dwlv, float* __restrict__ dblv,
                                                           float* __restrict__ dtable, int nb) {
    extern __shared__ float sh[];                       // [5][nb][16]: dmu, dlv, zc, se, dzc
    float* s_dmu = sh;
    float* s_dlv = sh + (long)nb * SZ;
    float* s_zc = sh + 2L * nb * SZ;
    float* s_se = sh + 3L * nb * SZ;
    float* s_dzc = sh + 4L * nb * SZ;
    const int n = nb * SZ;
    const int nt = blockDim.x;
#pragma unroll 2
    for (int i = threadIdx.x; i < n; i += nt) {
        const float g = dz[i];
        s_dmu[i] = (d_mu_in ? d_mu_in[i] : 0.f) + g;                                         // reparam_bwd_kernel
        s_dlv[i] = (d_lv_in ? d_lv_in[i] : 0.f) + g * eps[i] * 0.5f * expf(0.5f * lv[i]);
        s_zc[i] = zc[i];
        s_se[i] = se[i];
    }
    __syncthreads();
    // dzc = dmu . Wmu + dlv . Wlv   (input gradients of the two heads, summed)
    for (int i = threadIdx.x; i < n; i += nt) {
        const int b = i >> 4, k = i & 15;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < SZ; ++j) a = __builtin_fmaf(s_dmu[b * SZ + j], wmu[j * SZ + k], a);
#pragma unroll
        for (int j = 0; j < SZ; ++j) a = __builtin_fmaf(s_dlv[b * SZ + j], wlv[j * SZ + k], a);
        s_dzc[i] = a;
    }
    __syncthreads();
    // weight gradients: thread (matrix, j, k) sums over the batch in order; bias gradients by the threads of column k == 0
    if (threadIdx.x < 3 * SZ * SZ) {
        const int mat = threadIdx.x >> 8, j = (threadIdx.x >> 4) & 15, k = threadIdx.x & 15;
        const float* dsrc = mat == 0 ? s_dmu : (mat == 1 ? s_dlv : s_dzc);
        const float* xsrc = mat == 2 ? s_se : s_zc;
        float gsum = 0.f, bsum = 0.f;
#pragma unroll 8
        for (int b = 0; b < nb; ++b) {
            const float dv = dsrc[b * SZ + j];
            gsum = __builtin_fmaf(dv, xsrc[b * SZ + k], gsum);
            bsum += dv;
        }
        float* dw = mat == 0 ? dwmu : (mat == 1 ? dwlv : dw1);
        float* db = mat == 0 ? dbmu : (mat == 1 ? dblv : db1);
        dw[j * SZ + k] += gsum;
        if (k == 0) db[j] += bsum;
    }
    // dse = dzc . W1, scattered into the embedding gradient (duplicates of a speaker id meet in float atomics, like tg_embed_scatter_add)
    for (int i = threadIdx.x; i < n; i += nt) {
        const int b = i >> 4, k = i & 15;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < SZ; ++j) a = __builtin_fmaf(s_dzc[b * SZ + j], w1[j * SZ + k], a);
        const int64_t id = vid[b];
        if (id >= 0 && id < n_rows) atomicAdd(&dtable[id * SZ + k], a);
    }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_speaker_fwd(const float* table, const int64_t* vid, int32_t n_rows, const float* w1, const float* b1, const float* wmu, const float* bmu,
                              const float* wlv, const float* blv, float* eps, float* se, float* zc, float* mu, float* logvar, float* z,
                              int32_t B, float* rep, int64_t rep_ld, int32_t T, const uint64_t* rng_state, uint32_t site, void* stream) {
    TG_REQUIRE(table && vid && w1 && b1 && wmu && bmu && wlv && blv && eps && se && zc && mu && logvar && z && B > 0 && n_rows > 0, "tg_speaker_fwd: bad arguments");
    TG_REQUIRE(rep == nullptr || (T > 0 && rep_ld >= SZ), "tg_speaker_fwd: bad repeat target");
    hipLaunchKernelGGL(speaker_fwd_kernel, dim3(cdiv(B, 16)), dim3(256), 0, (hipStream_t)stream, table, vid, n_rows, w1, b1, wmu, bmu, wlv, blv, eps, se, zc, mu,
                       logvar, z, B, rep, (long)rep_ld, T, rng_state, site);
    return check_launch("tg_speaker_fwd");
}

extern "C" int32_t tg_speaker_bwd_max_rows(void) { return 512; }          // 5 x 512 x 16 floats = 160 KB of LDS

extern "C" int tg_speaker_bwd(const float* dz, const float* d_mu_in, const float* d_logvar_in, const float* logvar, const float* eps, const float* zc,
                              const float* se, const int64_t* vid, int32_t n_rows, const float* w1, const float* wmu, const float* wlv, float* dw1,
                              float* db1, float* dwmu, float* dbmu, float* dwlv, float* dblv, float* dtable, int32_t nb, void* stream) {
    TG_REQUIRE(dz && logvar && eps && zc && se && vid && w1 && wmu && wlv && dw1 && db1 && dwmu && dbmu && dwlv && dblv && dtable && n_rows > 0,
               "tg_speaker_bwd: null pointer");
    TG_REQUIRE(nb > 0 && nb <= tg_speaker_bwd_max_rows(), "tg_speaker_bwd: nb=%d must be in [1, %d]", nb, tg_speaker_bwd_max_rows());
    const size_t lds = 5 * (size_t)nb * SZ * sizeof(float);
    if (lds > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(speaker_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                set_error("tg_speaker_bwd: cannot raise the dynamic LDS limit");
                return 1;
            }
            raised = true;
        }
    }
    hipLaunchKernelGGL(speaker_bwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, dz, d_mu_in, d_logvar_in, logvar, eps, zc, se, vid, n_rows, w1, wmu, wlv,
                       dw1, db1, dwmu, dbmu, dwlv, dblv, dtable, nb);
    return check_launch("tg_speaker_bwd");
}

// ---- output MLP of PoseGenerator: Linear(H, H/2) -> LeakyReLU(True) -> Linear(H/2, D)  (model/multimodal_context_net.py:100-104) ----------
// nn.LeakyReLU(True) sets negative_slope = True == 1.0: the activation is the identity (reference README.md:122), so the two linears are ONE
// linear map,  out = o (W2 W1)^T + (W2 b1 + b2).  The forward is then a single [M x H] x [H x D] product on the composed weight (instead of
// M x H x H/2 + M x H/2 x D with a K = 150 middle dimension that is not a multiple of 4), and the backward needs the [M]-sized operands only in
//   P = d_out^T o  [D x H],   s = colsum(d_out),   d o = d_out (W2 W1):
//   dW2 = P W1^T + s b1^T,  db2 = s,  dW1 = W2^T P,  db1 = W2^T s      (out_mlp_param_grads_kernel: a few hundred thousand MACs).
namespace tg {

// (reduction loops unrolled by 10: ten independent loads in flight per thread -- rolled, every iteration waited out an L2 round trip: 67 us)
// w21 [D][H] = W2 W1, w21t [H][D] (the operand of d o = d_out W21 as a row-major [N = H][K = D] matrix), b21 [D] = W2 b1 + b2
// dup = 2: the composed weight is written twice side by side, w21 [D][2 H] = [W2 W1 | W2 W1] and w21t [2 H][D] -- the map then acts on the
// GRU output [fwd | rev] directly (out = (y_fwd + y_rev) W21^T = y [W21 | W21]^T), and the direction sum is never materialised
__global__ __launch_bounds__(256) void out_mlp_compose_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                              const float* __restrict__ b2, int H, int Hm, int D, int dup, float* __restrict__ w21,
                                                              float* __restrict__ w21t, float* __restrict__ b21) {
    // eight lanes per entry split the Hm-term sum (strided partials, shuffle tree: fixed order)
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(gid >> 3), sub = (int)(gid & 7);
    float a = 0.f;
    if (i < D * H) {
        const int d = i / H, h = i - d * H;
#pragma unroll 5
        for (int m = sub; m < Hm; m += 8) a = __builtin_fmaf(w2[d * Hm + m], w1[m * H + h], a);
    } else if (i < D * H + D) {
        const int d = i - D * H;
        if (sub == 0) a = b2[d];
#pragma unroll 5
        for (int m = sub; m < Hm; m += 8) a = __builtin_fmaf(w2[d * Hm + m], b1[m], a);
    }
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    a += __shfl_xor(a, 4);
    if (sub == 0) {
        if (i < D * H) {
            const int d = i / H, h = i - d * H;
            for (int r = 0; r < dup; ++r) {
                w21[d * (dup * H) + r * H + h] = a;
                if (w21t) w21t[(r * H + h) * D + d] = a;
            }
        } else if (i < D * H + D) {
            b21[i - D * H] = a;
        }
    }
}

// P [D][H], s [D] (accumulated by the caller's weight-gradient GEMM); all four gradients accumulate.
// Entries: [dW1: Hm * H, D-term sums] [dW2: D * Hm, H-term sums] [db1: Hm] [db2: D]; EIGHT lanes per entry split the sum (fixed order:
// strided partials, then a shuffle tree), so the 300-term sums of dW2 are 38 loads per lane in flight instead of 300 in a row.
// dup = 2: P arrives as [D][2 H] = d_out^T [y_fwd | y_rev]; its halves are added on the fly (P = d_out^T (y_fwd + y_rev))
__global__ __launch_bounds__(256) void out_mlp_param_grads_kernel(const float* __restrict__ P, const float* __restrict__ s, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, const float* __restrict__ w2, int H, int Hm, int D, int dup,
                                                                  float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                                  float* __restrict__ db2) {
    const int ldp = dup * H;
    auto Pv = [&](int d, int h) { return dup == 2 ? P[d * ldp + h] + P[d * ldp + H + h] : P[d * ldp + h]; };
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(gid >> 3), sub = (int)(gid & 7);
    const int n1 = Hm * H, n2 = D * Hm;
    float a = 0.f;
    int kind = -1, idx = 0;
    if (i < n1) {                                        // dW1[m][h] += sum_d W2[d][m] P[d][h]
        kind = 0; idx = i;
        const int m = i / H, h = i - m * H;
#pragma unroll 4
        for (int d = sub; d < D; d += 8) a = __builtin_fmaf(w2[d * Hm + m], Pv(d, h), a);
    } else if (i < n1 + n2) {                            // dW2[d][m] += sum_h P[d][h] W1[m][h] + s[d] b1[m]
        kind = 1; idx = i - n1;
        const int d = idx / Hm, m = idx - d * Hm;
        if (sub == 0) a = s[d] * b1[m];
#pragma unroll 10
        for (int h = sub; h < H; h += 8) a = __builtin_fmaf(Pv(d, h), w1[m * H + h], a);
    } else if (i < n1 + n2 + Hm) {                       // db1[m] += sum_d W2[d][m] s[d]
        kind = 2; idx = i - n1 - n2;
#pragma unroll 4
        for (int d = sub; d < D; d += 8) a = __builtin_fmaf(w2[d * Hm + idx], s[d], a);
    } else if (i < n1 + n2 + Hm + D) {
        kind = 3; idx = i - n1 - n2 - Hm;
        if (sub == 0) a = s[idx];
    }
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    a += __shfl_xor(a, 4);
    if (sub == 0) {
        if (kind == 0) dw1[idx] += a;
        else if (kind == 1) dw2[idx] += a;
        else if (kind == 2) db1[idx] += a;
        else if (kind == 3) db2[idx] += a;
    }
}

}  // namespace tg

extern "C" int tg_out_mlp_compose(const float* w1, const float* b1, const float* w2, const float* b2, int32_t H, int32_t Hm, int32_t D, int32_t dup,
                                  float* w21, float* w21t, float* b21, void* stream) {
    TG_REQUIRE(w1 && b1 && w2 && b2 && w21 && b21 && H > 0 && Hm > 0 && D > 0 && (dup == 1 || dup == 2), "tg_out_mlp_compose: bad arguments");
    hipLaunchKernelGGL(out_mlp_compose_kernel, dim3(cdiv(((long)D * H + D) * 8, 256)), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, H, Hm, D, dup, w21, w21t, b21);
    return check_launch("tg_out_mlp_compose");
}

extern "C" int tg_out_mlp_param_grads(const float* P, const float* s, const float* w1, const float* b1, const float* w2, int32_t H, int32_t Hm, int32_t D,
                                      int32_t dup, float* dw1, float* db1, float* dw2, float* db2, void* stream) {
    TG_REQUIRE(P && s && w1 && b1 && w2 && dw1 && db1 && dw2 && db2 && H > 0 && Hm > 0 && D > 0 && (dup == 1 || dup == 2), "tg_out_mlp_param_grads: bad arguments");
    const long n = (long)Hm * H + (long)D * Hm + Hm + D;
    hipLaunchKernelGGL(out_mlp_param_grads_kernel, dim3(cdiv(n * 8, 256)), dim3(256), 0, (hipStream_t)stream, P, s, w1, b1, w2, H, Hm, D, dup, dw1, db1, dw2, db2);
    return check_launch("tg_out_mlp_param_grads");
}
