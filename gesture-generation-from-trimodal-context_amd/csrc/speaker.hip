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
                                                          const float* __restrict__ eps, float* __restrict__ se, float* __restrict__ zc,
                                                          float* __restrict__ mu, float* __restrict__ lv, float* __restrict__ z, int B,
                                                          float* __restrict__ rep, long rep_ld, int T) {
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
    const float e = eps[bb * SZ + j];
    const float zz = m + e * expf(0.5f * l);
    if (b < B) {
        const long o = (long)b * SZ + j;
        se[o] = s; zc[o] = c; mu[o] = m; lv[o] = l; z[o] = zz;
        if (rep)
            for (int t = 0; t < T; ++t) rep[((long)b * T + t) * rep_ld + j] = zz;
    }
}

// dz [nb][16]: gradient w.r.t. z (already summed over the frames).  d_mu_in / d_lv_in: direct gradients (KLD term) or NULL.
__global__ __launch_bounds__(256) void speaker_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ d_mu_in, const float* __restrict__ d_lv_in,
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
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += 256) {          // (unrolled: the seven loads of eight iterations in flight together)
        const float g = dz[i];
        s_dmu[i] = (d_mu_in ? d_mu_in[i] : 0.f) + g;                                         // reparam_bwd_kernel
        s_dlv[i] = (d_lv_in ? d_lv_in[i] : 0.f) + g * eps[i] * 0.5f * expf(0.5f * lv[i]);
        s_zc[i] = zc[i];
        s_se[i] = se[i];
    }
    __syncthreads();
    // dzc = dmu . Wmu + dlv . Wlv   (input gradients of the two heads, summed)
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = i >> 4, k = i & 15;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < SZ; ++j) a = __builtin_fmaf(s_dmu[b * SZ + j], wmu[j * SZ + k], a);
#pragma unroll
        for (int j = 0; j < SZ; ++j) a = __builtin_fmaf(s_dlv[b * SZ + j], wlv[j * SZ + k], a);
        s_dzc[i] = a;
    }
    __syncthreads();
    // weight gradients: thread (j, k) of the 16 x 16 matrix sums over the batch in order; bias gradients by the threads of row k == 0
    {
        const int j = threadIdx.x >> 4, k = threadIdx.x & 15;
        float gm = 0.f, gl = 0.f, g1 = 0.f, bm = 0.f, bl = 0.f, bb1 = 0.f;
        for (int b = 0; b < nb; ++b) {
            const float dm = s_dmu[b * SZ + j], dl = s_dlv[b * SZ + j], dc = s_dzc[b * SZ + j];
            gm = __builtin_fmaf(dm, s_zc[b * SZ + k], gm);
            gl = __builtin_fmaf(dl, s_zc[b * SZ + k], gl);
            g1 = __builtin_fmaf(dc, s_se[b * SZ + k], g1);
            bm += dm; bl += dl; bb1 += dc;
        }
        dwmu[j * SZ + k] += gm;
        dwlv[j * SZ + k] += gl;
        dw1[j * SZ + k] += g1;
        if (k == 0) { dbmu[j] += bm; dblv[j] += bl; db1[j] += bb1; }
    }
    // dse = dzc . W1, scattered into the embedding gradient (duplicates of a speaker id meet in float atomics, like tg_embed_scatter_add)
    for (int i = threadIdx.x; i < n; i += 256) {
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
                              const float* wlv, const float* blv, const float* eps, float* se, float* zc, float* mu, float* logvar, float* z,
                              int32_t B, float* rep, int64_t rep_ld, int32_t T, void* stream) {
    TG_REQUIRE(table && vid && w1 && b1 && wmu && bmu && wlv && blv && eps && se && zc && mu && logvar && z && B > 0 && n_rows > 0, "tg_speaker_fwd: bad arguments");
    TG_REQUIRE(rep == nullptr || (T > 0 && rep_ld >= SZ), "tg_speaker_fwd: bad repeat target");
    hipLaunchKernelGGL(speaker_fwd_kernel, dim3(cdiv(B, 16)), dim3(256), 0, (hipStream_t)stream, table, vid, n_rows, w1, b1, wmu, bmu, wlv, blv, eps, se, zc, mu,
                       logvar, z, B, rep, (long)rep_ld, T);
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
    hipLaunchKernelGGL(speaker_bwd_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, dz, d_mu_in, d_logvar_in, logvar, eps, zc, se, vid, n_rows, w1, wmu, wlv,
                       dw1, db1, dwmu, dbmu, dwlv, dblv, dtable, nb);
    return check_launch("tg_speaker_bwd");
}
