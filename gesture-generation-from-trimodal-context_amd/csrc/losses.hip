// Loss kernels of the GAN step (train_eval/train_gan.py:41,53-89) and of the FGD autoencoder
// (train_feature_extractor.py:64-72): forward value and analytic gradient in one pass, all on device so the
// step never synchronises with the host (the reference's five .item() calls per iteration become one deferred read).
#include "common.hpp"

namespace tg {

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// sum over a 256-thread block; result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}
__device__ __forceinline__ float sl1(float x) { const float a = fabsf(x); return a < 1.f ? 0.5f * x * x : a - 0.5f; }
__device__ __forceinline__ float dsl1(float x) { return fabsf(x) < 1.f ? x : (x > 0.f ? 1.f : -1.f); }
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// dis_error = -mean(log(s_r + 1e-8) + log(1 - s_f + 1e-8))
__global__ __launch_bounds__(256) void gan_d_loss_kernel(const float* __restrict__ lr, const float* __restrict__ lf, int B, float* __restrict__ out,
                                                         float* __restrict__ dlr, float* __restrict__ dlf) {
    __shared__ float sh[4];
    float s = 0.f;
    const float invB = 1.f / (float)B;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float sr = sigmoidf_(lr[i]), sf = sigmoidf_(lf[i]);
        s += logf(sr + 1e-8f) + logf(1.f - sf + 1e-8f);
        dlr[i] = -invB * sr * (1.f - sr) / (sr + 1e-8f);
        dlf[i] = invB * sf * (1.f - sf) / (1.f - sf + 1e-8f);
    }
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) out[0] = -s * invB;
}

// stage 1: one workgroup per clip -> ws[b] = P_b (pose smooth-L1 sum vs the shuffled-speaker output),
// ws[B+b] = mean_c |z - z_rand|, ws[2B+b] = sum smooth-L1((out - target)/0.1); and the clip's rows of d_out: the diversity term's
// coefficient depends on this clip's P_b and z distance only, so
//   d_out = w_h * sl1'((o-t)/0.1) / (B*TD) + coef_b * sl1'((o-orand)/0.05),  coef_b = d(w_d * div_reg)/dP_b (0 where the ratio is clamped)
// needs no other workgroup (it was a third launch after the scalar stage)
__global__ __launch_bounds__(256) void gan_g_stage1(const float* __restrict__ o, const float* __restrict__ t, const float* __restrict__ orand,
                                                    const float* __restrict__ z, const float* __restrict__ zr, int B, int TD, int Z,
                                                    float* __restrict__ ws, float w_h, float w_d, float* __restrict__ d_out) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    float p = 0.f, h = 0.f, zl = 0.f;
    for (int i = threadIdx.x; i < TD; i += 256) {
        const float ov = o[(long)b * TD + i];
        p += sl1((ov - orand[(long)b * TD + i]) / 0.05f) * 0.05f;
        h += sl1((ov - t[(long)b * TD + i]) / 0.1f);
    }
    for (int i = threadIdx.x; i < Z; i += 256) zl += fabsf(z[(long)b * Z + i] - zr[(long)b * Z + i]);
    p = block_sum256(p, sh);
    h = block_sum256(h, sh);
    zl = block_sum256(zl, sh);
    if (threadIdx.x == 0) { ws[b] = p; ws[B + b] = zl / (float)Z; ws[2 * B + b] = h; }
    const float den = zl / (float)Z + 1.0e-5f;
    const float coef = (-(p / den) < -1000.f) ? 0.f : (-1.f / den) * w_d * (1.f / (float)B);        // as stage 2 forms it
    const float hn = w_h / (float)((long)B * TD);
    for (int i = threadIdx.x; i < TD; i += 256) {
        const long e = (long)b * TD + i;
        const float ov = o[e];
        d_out[e] = hn * dsl1((ov - t[e]) / 0.1f) + coef * dsl1((ov - orand[e]) / 0.05f);
    }
}

// stage 2 (one workgroup): scalars, per-clip div_reg coefficient (overwrites ws[b]), d_mu, d_logvar, d_logit
__global__ __launch_bounds__(256) void gan_g_stage2(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ logit, int B,
                                                    int TD, int Z, float w_h, float w_k, float w_d, float w_g, int use_gan,
                                                    float* __restrict__ ws, float* __restrict__ sc, float* __restrict__ dmu, float* __restrict__ dlv,
                                                    float* __restrict__ dlogit) {
    __shared__ float sh[4];
    float hub = 0.f, div = 0.f, gen = 0.f, kld = 0.f;
    const float invB = 1.f / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        hub += ws[2 * B + b];
        const float den = ws[B + b] + 1.0e-5f;
        const float v = -(ws[b] / den);
        const bool clamped = v < -1000.f;
        div += clamped ? -1000.f : v;
        ws[b] = clamped ? 0.f : (-1.f / den) * w_d * invB;     // d(w_d * div_reg)/dP_b
        const float s = sigmoidf_(logit[b]);
        gen += logf(s + 1e-8f);
        dlogit[b] = use_gan ? -w_g * invB * s * (1.f - s) / (s + 1e-8f) : 0.f;
    }
    const long nz = (long)B * Z;
    const float invnz = 1.f / (float)nz;
    for (long i = threadIdx.x; i < nz; i += 256) {
        const float m = mu[i], l = lv[i], e = expf(l);
        kld += 1.f + l - m * m - e;
        dmu[i] = w_k * invnz * m;                         // d/dmu of -0.5*mean(1 + lv - mu^2 - e^lv)
        dlv[i] = w_k * (-0.5f) * invnz * (1.f - e);
    }
    hub = block_sum256(hub, sh);
    div = block_sum256(div, sh);
    gen = block_sum256(gen, sh);
    kld = block_sum256(kld, sh);
    if (threadIdx.x == 0) {
        const float huber = 0.1f * hub / ((float)B * (float)TD);
        const float kl = -0.5f * kld * invnz;
        const float dv = div * invB;
        const float ge = -gen * invB;
        sc[0] = huber; sc[1] = kl; sc[2] = dv; sc[3] = ge;
        sc[4] = w_h * huber + w_k * kl + w_d * dv + (use_gan ? w_g * ge : 0.f);
    }
}

__global__ __launch_bounds__(256) void l1_mean_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += fabsf(a[i] - b[i]);
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) atomicAdd(out, s / (float)n);
}

// loss = sum_b [ mean_{t,d} |r - x| + mean_{t,d} |(r_t - r_{t-1}) - (x_t - x_{t-1})| ]
__global__ __launch_bounds__(256) void ae_loss_kernel(const float* __restrict__ r, const float* __restrict__ x, int B, int T, int D, float* __restrict__ out,
                                                      float* __restrict__ dr) {
    __shared__ float sh[4];
    const long n = (long)B * T * D;
    const float w1 = 1.f / (float)(T * D), w2 = 1.f / (float)((T - 1) * D);
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)((i / D) % T);
        const float e = r[i] - x[i];
        float g = sgn(e) * w1;
        s += fabsf(e) * w1;
        if (t >= 1) {
            const float dlt = e - (r[i - D] - x[i - D]);
            s += fabsf(dlt) * w2;
            g += sgn(dlt) * w2;
        }
        if (t + 1 < T) {
            const float dlt = (r[i + D] - x[i + D]) - e;
            g -= sgn(dlt) * w2;
        }
        dr[i] = g;
    }
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// Cross-fade of consecutive synthesis windows (synthesize.py:145-153): the first n frames of the new window become
// prev_tail[j] * (n - j)/(n + 1) + next[j] * (j + 1)/(n + 1).  next: [B][T][D] (in place), prev_tail: [B][n][D].
__global__ void window_blend_kernel(const float* __restrict__ prev_tail, float* __restrict__ next, int B, int T, int D, int n) {
    const long total = (long)B * n * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int j = (int)((i / D) % n);
        const long b = i / ((long)D * n);
        float* p = next + (b * T + j) * D + d;
        *p = prev_tail[i] * (float)(n - j) / (float)(n + 1) + *p * (float)(j + 1) / (float)(n + 1);
    }
}

// evaluate_testset metrics (train.py:282-310): direction vectors (+ mean) -> 10 joint positions by walking the 9 bones
// (utils/data_utils.py:14-15,77-98); sums of |joint error| over frames >= n_pre, of |second-difference error| and of the plain
// L1 error.  One thread per (clip, frame); sums[0..2] in fp64.
__constant__ int c_bone_from[9] = {0, 1, 2, 1, 4, 5, 1, 7, 8};
__constant__ int c_bone_to[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
__constant__ float c_bone_len[9] = {0.26f, 0.18f, 0.14f, 0.22f, 0.36f, 0.33f, 0.22f, 0.36f, 0.33f};

__device__ __forceinline__ void joints_of(const float* __restrict__ v, const float* __restrict__ mean, float (&jp)[30]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) jp[q] = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) jp[c_bone_to[j] * 3 + q] = jp[c_bone_from[j] * 3 + q] + c_bone_len[j] * (v[j * 3 + q] + mean[j * 3 + q]);
}

__global__ __launch_bounds__(256) void pose_metrics_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                           const float* __restrict__ mean, int B, int T, int n_pre, double* __restrict__ sums) {
    __shared__ float sh[4];
    float s_mae = 0.f, s_acc = 0.f, s_l1 = 0.f;
    const long total = (long)B * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const float* o = out + i * 27;
        const float* g = tgt + i * 27;
        for (int q = 0; q < 27; ++q) s_l1 += fabsf(o[q] - g[q]);
        float jo[30], jg[30];
        joints_of(o, mean, jo);
        joints_of(g, mean, jg);
        if (t >= n_pre)
            for (int q = 0; q < 30; ++q) s_mae += fabsf(jo[q] - jg[q]);
        if (t + 2 < T) {      // np.diff(n=2): x[t+2] - 2 x[t+1] + x[t]
            float jo1[30], jg1[30], jo2[30], jg2[30];
            joints_of(o + 27, mean, jo1); joints_of(g + 27, mean, jg1);
            joints_of(o + 54, mean, jo2); joints_of(g + 54, mean, jg2);
            for (int q = 0; q < 30; ++q)
                s_acc += fabsf((jg2[q] - 2.f * jg1[q] + jg[q]) - (jo2[q] - 2.f * jo1[q] + jo[q]));
        }
    }
    s_mae = block_sum256(s_mae, sh);
    s_acc = block_sum256(s_acc, sh);
    s_l1 = block_sum256(s_l1, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], (double)s_mae);
        atomicAdd(&sums[1], (double)s_acc);
        atomicAdd(&sums[2], (double)s_l1);
    }
}

// ---- discriminator head (multimodal_context_net.py:243-252): sum of the GRU directions -> Linear(H -> 1) per frame -> view(B, T) ->
// Linear(T -> 1) -> sigmoid, as ONE launch (four before: add_halves, two 1-column GEMMs, sigmoid); one wave per clip.
__global__ __launch_bounds__(256) void d_head_fwd_kernel(const float* __restrict__ y, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ l1,
                                                         float* __restrict__ logit, float* __restrict__ prob, int B, int T, int H) {
    // one workgroup per clip: thread (t = tid / 8, jj = tid % 8) sums every 8th hidden unit of frame t (independent loads in flight
    // together), 8-lane shuffle reduce -> l1[b][t]; the T per-frame logits then meet in LDS for the second Linear.  T <= 32.
    __shared__ float s_l[32];
    const int b = blockIdx.x, t = threadIdx.x >> 3, jj = threadIdx.x & 7;
    float v = 0.f;
    if (t < T) {
        const float* yr = y + ((long)b * T + t) * (2 * H);
        for (int j = jj; j < H; j += 8) v += (yr[j] + yr[H + j]) * w1[j];
    }
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (jj == 0 && t < 32) {
        const float l = t < T ? v + b1[0] : 0.f;
        if (t < T) l1[(long)b * T + t] = l;
        s_l[t] = t < T ? l * w2[t] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = threadIdx.x < 32 ? s_l[threadIdx.x] : 0.f;
        a = wave_sum(a);
        if (threadIdx.x == 0) {
            const float lg = a + b2[0];
            logit[b] = lg;
            prob[b] = sigmoidf_(lg);
        }
    }
}

// backward of the head for d_logit (gradient w.r.t. the pre-sigmoid output): dy [B][T][2H] (both halves equal), and -- when
// param_grads -- dW1 / db1 / dW2 / db2 accumulated with one atomic per workgroup and entry.
// Deterministic form of the parameter gradients (tg_set_deterministic): ONE workgroup of 16 waves walks the clips (wave w: clips w, w + 16, ...),
// the waves' sums meet in LDS in wave order, plain += into the gradient -- no atomics.  The input gradient dy is written as below.
__global__ __launch_bounds__(1024) void d_head_bwd_det_kernel(const float* __restrict__ d_logit, const float* __restrict__ y,
                                                              const float* __restrict__ l1, const float* __restrict__ w1,
                                                              const float* __restrict__ w2, float* __restrict__ dy, float* __restrict__ dw1,
                                                              float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2, int B,
                                                              int T, int H) {
    __shared__ float s_w1[16][64], s_w2[16][64], s_b[16][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float a_w1 = 0.f, a_w2 = 0.f, a_b1 = 0.f, a_b2 = 0.f;
    for (int b = wv; b < B; b += 16) {
        const float dl = d_logit[b];
        a_b2 += dl;
        if (lane < T) a_w2 += dl * l1[(long)b * T + lane];
        for (int t = 0; t < T; ++t) {
            const float dl1 = dl * w2[t];
            a_b1 += dl1;
            if (lane < H) {
                const long o = ((long)b * T + t) * (2 * H) + lane;
                a_w1 += dl1 * (y[o] + y[o + H]);
            }
        }
    }
    s_w1[wv][lane] = a_w1; s_w2[wv][lane] = a_w2;
    if (lane == 0) { s_b[wv][0] = a_b1; s_b[wv][1] = a_b2; }
    __syncthreads();
    if (wv == 0) {
        float t1 = 0.f, t2 = 0.f, b1 = 0.f, b2 = 0.f;
        for (int q = 0; q < 16; ++q) { t1 += s_w1[q][lane]; t2 += s_w2[q][lane]; b1 += s_b[q][0]; b2 += s_b[q][1]; }
        if (lane < H) dw1[lane] += t1;
        if (lane < T) dw2[lane] += t2;
        if (lane == 0) { *db1 += b1; *db2 += b2; }
    }
}

__global__ __launch_bounds__(256) void d_head_bwd_kernel(const float* __restrict__ d_logit, const float* __restrict__ y,
                                                         const float* __restrict__ l1, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, float* __restrict__ dy, float* __restrict__ dw1,
                                                         float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2, int B,
                                                         int T, int H) {
    __shared__ float s_w1[4][64], s_w2[4][64], s_b[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x * 4 + wv;
    float a_w1 = 0.f, a_w2 = 0.f, a_b1 = 0.f, a_b2 = 0.f;      // H <= 64 on this path: lane j owns dW1[j]; lane t < T owns dW2[t]
    if (b < B) {
        const float dl = d_logit[b];
        a_b2 = dl;
        if (lane < T) a_w2 = dl * l1[(long)b * T + lane];
        for (int t = 0; t < T; ++t) {
            const float dl1 = dl * w2[t];
            a_b1 += dl1;
            if (lane < H) {
                const long o = ((long)b * T + t) * (2 * H) + lane;
                const float dv = dl1 * w1[lane];
                if (dw1) a_w1 += dl1 * (y[o] + y[o + H]);
                dy[o] = dv;
                dy[o + H] = dv;
            }
        }
    }
    if (!dw1) return;                                          // (uniform) input gradient only
    s_w1[wv][lane] = a_w1; s_w2[wv][lane] = a_w2;
    if (lane == 0) { s_b[wv][0] = a_b1; s_b[wv][1] = a_b2; }
    __syncthreads();
    if (wv == 0) {
        const float t1 = s_w1[0][lane] + s_w1[1][lane] + s_w1[2][lane] + s_w1[3][lane];
        const float t2 = s_w2[0][lane] + s_w2[1][lane] + s_w2[2][lane] + s_w2[3][lane];
        if (lane < H) atomicAdd(&dw1[lane], t1);
        if (lane < T) atomicAdd(&dw2[lane], t2);
        if (lane == 0) {
            atomicAdd(db1, s_b[0][0] + s_b[1][0] + s_b[2][0] + s_b[3][0]);
            atomicAdd(db2, s_b[0][1] + s_b[1][1] + s_b[2][1] + s_b[3][1]);
        }
    }
}

// Head forward + the GAN loss terms of the clips + head backward in ONE launch (train_gan.py:36-41 and :55-57,86-88 over
// multimodal_context_net.py:243-252): the clip's logit is all its loss term and d_logit need (both GAN losses are means of per-clip terms),
// so the three launches d_head_fwd / loss / d_head_bwd have no cross-clip dependency except the scalar loss itself, which -- when asked
// for -- the LAST workgroup to finish sums in a fixed order (terms[], the arrival counter resets itself).  Rows [0, nb) are scored as
// "should be real" (term log(s + 1e-8), d_logit = -scale_real s (1 - s) / (s + 1e-8)), rows [nb, B) as "should be fake" (log(1 - s + 1e-8),
// +scale_fake ...): the discriminator step has B = 2 nb and both scales 1 / nb, the generator step B = nb and scale_real = w_gan / B.
// dw1 == NULL: input gradient only (the generator step; deterministic mode adds the parameter gradients with d_head_bwd_det_kernel from
// the l1 / d_logit written here).
__device__ __forceinline__ float wave_allsum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(1024) void d_head_step_kernel(const float* __restrict__ y, const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ l1,
                                                            float* __restrict__ logit, float* __restrict__ prob, float* __restrict__ d_logit,
                                                            float* __restrict__ terms, float* __restrict__ out, unsigned* __restrict__ counter,
                                                            float* __restrict__ dy, float* __restrict__ dw1, float* __restrict__ db1,
                                                            float* __restrict__ dw2, float* __restrict__ db2, int B, int nb, float scale_real, float scale_fake, int T,
                                                            int H) {
    // 1024 threads = 4 clips x 4 waves; wave q of a clip owns frames q, q + 4, ... (<= 8 of them), lane j hidden unit j.  (One wave per clip
    // walking all T frames measured 19.6 us: ~4 000 dependent instructions on one wave; here a wave runs ~1/8 of that.)
    __shared__ float s_l[4][32], s_w1[16][64], s_b1[16], s_dl[4], sh[4];
    __shared__ unsigned s_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = wv >> 2, q = wv & 3, b = blockIdx.x * 4 + c;
    const bool on = b < B, jn = lane < H;
    const float w1j = jn ? w1[lane] : 0.f;
    const float w2t = lane < T ? w2[lane] : 0.f;
    const long row = (long)b * T * (2 * H) + lane;
    float ya[8], yc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                              // every load of the wave in flight before the first use
        const int t = q + 4 * i;
        const bool ld = on && jn && t < T;
        ya[i] = ld ? y[row + t * 2 * H] : 0.f;
        yc[i] = ld ? y[row + t * 2 * H + H] : 0.f;
    }
    if (lane < 8) s_l[c][q + 4 * lane] = 0.f;                  // (frames beyond T)
    const float b1v = b1[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = q + 4 * i;
        ya[i] += yc[i];                                        // the direction sum, kept for the backward half
        if (t < T) {                                           // (uniform)
            const float p = wave_allsum(ya[i] * w1j) + b1v;
            if (lane == 0) { s_l[c][t] = p; if (on) l1[(long)b * T + t] = p; }
        }
    }
    __syncthreads();
    const float l1_mine = lane < 32 ? s_l[c][lane] : 0.f;      // lane t: the frame's logit
    const float lg = wave_allsum(l1_mine * w2t) + b2[0];
    const float s = sigmoidf_(lg);
    const bool real = b < nb;
    const float term = real ? logf(s + 1e-8f) : logf(1.f - s + 1e-8f);
    const float dl = !on ? 0.f : real ? -scale_real * s * (1.f - s) / (s + 1e-8f) : scale_fake * s * (1.f - s) / (1.f - s + 1e-8f);
    if (on && q == 0 && lane == 0) {
        logit[b] = lg; prob[b] = s; d_logit[b] = dl;
        // written through to memory (agent scope) and awaited below: the last workgroup may run on another XCD, whose L2 is a different one.
        // (No __threadfence(): at agent scope it writes back every dirty L2 line of the XCD -- dy included -- and cost 10+ us per launch.)
        __hip_atomic_store(&terms[b], term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float a_w1 = 0.f, a_b1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = q + 4 * i;
        const float dl1 = dl * __shfl(w2t, t & 31, 64);        // (w2t is 0 beyond T)
        a_b1 += dl1;
        a_w1 += dl1 * ya[i];
        if (on && jn && t < T) {
            const float dv = dl1 * w1j;
            dy[row + t * 2 * H] = dv;
            dy[row + t * 2 * H + H] = dv;
        }
    }
    if (dw1) {                                                 // (uniform)
        s_w1[wv][lane] = a_w1;
        if (lane == 0) { s_b1[wv] = a_b1; if (q == 0) s_dl[c] = dl; }
    }
    __builtin_amdgcn_s_waitcnt(0);                             // terms[] of this workgroup at memory before its arrival is counted
    __syncthreads();
    if (dw1 && wv == 0) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t1 += s_w1[k][lane];
        if (lane < 32)
#pragma unroll
            for (int k = 0; k < 4; ++k) t2 += s_dl[k] * s_l[k][lane];
        if (lane < H) atomicAdd(&dw1[lane], t1);
        if (lane < T) atomicAdd(&dw2[lane], t2);
        if (lane == 0) {
            float u = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) u += s_b1[k];
            atomicAdd(db1, u);
            atomicAdd(db2, s_dl[0] + s_dl[1] + s_dl[2] + s_dl[3]);
        }
    }
    if (!out) return;                                          // (uniform) the caller sums terms[] itself
    if (threadIdx.x == 0) s_last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;                                       // (uniform)
    float sum = 0.f;                                           // the summation order of gan_d_loss_kernel (256 threads, then 4 wave sums)
    if (threadIdx.x < 256)
        for (int i = threadIdx.x; i < nb; i += 256) {
            sum += __hip_atomic_load(&terms[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (nb + i < B) sum += __hip_atomic_load(&terms[nb + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    sum = wave_sum(sum);
    if (lane == 0 && wv < 4) sh[wv] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = -(sh[0] + sh[1] + sh[2] + sh[3]) / (float)nb;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace tg

using namespace tg;
#define ST ((hipStream_t)stream)

extern "C" {

int tg_gan_d_loss(const float* logit_real, const float* logit_fake, int32_t B, float* out, float* d_logit_real, float* d_logit_fake,
                  void* stream) {
    TG_REQUIRE(logit_real && logit_fake && out && d_logit_real && d_logit_fake && B > 0, "tg_gan_d_loss: bad arguments");
    hipLaunchKernelGGL(gan_d_loss_kernel, dim3(1), dim3(256), 0, ST, logit_real, logit_fake, B, out, d_logit_real, d_logit_fake);
    return check_launch("tg_gan_d_loss");
}

int tg_gan_g_loss(const float* out_pose, const float* target, const float* out_rand, const float* z, const float* z_rand,
                  const float* mu, const float* logvar, const float* logit_out, int32_t B, int32_t TD, int32_t Z, float w_huber,
                  float w_kld, float w_div, float w_gan, int32_t use_gan, float* ws, float* scalars, float* d_out, float* d_mu,
                  float* d_logvar, float* d_logit_out, void* stream) {
    TG_REQUIRE(out_pose && target && out_rand && z && z_rand && mu && logvar && logit_out && ws && scalars && d_out && d_mu &&
                   d_logvar && d_logit_out, "tg_gan_g_loss: null pointer");
    TG_REQUIRE(B > 0 && TD > 0 && Z > 0, "tg_gan_g_loss: bad sizes");
    hipLaunchKernelGGL(gan_g_stage1, dim3(B), dim3(256), 0, ST, out_pose, target, out_rand, z, z_rand, B, TD, Z, ws, w_huber, w_div, d_out);
    hipLaunchKernelGGL(gan_g_stage2, dim3(1), dim3(256), 0, ST, mu, logvar, logit_out, B, TD, Z, w_huber, w_kld, w_div, w_gan, use_gan, ws,
                       scalars, d_mu, d_logvar, d_logit_out);
    return check_launch("tg_gan_g_loss");
}

int tg_d_head_fwd(const float* y, const float* w1, const float* b1, const float* w2, const float* b2, float* l1, float* logit, float* prob,
                  int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(y && w1 && b1 && w2 && b2 && l1 && logit && prob && B > 0 && T > 0 && T <= 32 && H > 0, "tg_d_head_fwd: bad arguments (T <= 32)");
    hipLaunchKernelGGL(d_head_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, y, w1, b1, w2, b2, l1, logit, prob, B, T, H);
    return check_launch("tg_d_head_fwd");
}
int tg_d_head_bwd(const float* d_logit, const float* y, const float* l1, const float* w1, const float* w2, float* dy, float* dw1, float* db1,
                  float* dw2, float* db2, int32_t B, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(d_logit && y && l1 && w1 && w2 && dy && B > 0 && T > 0 && T <= 64 && H > 0 && H <= 64, "tg_d_head_bwd: bad arguments (T, H <= 64)");
    TG_REQUIRE((dw1 != nullptr) == (db1 != nullptr) && (dw1 != nullptr) == (dw2 != nullptr) && (dw1 != nullptr) == (db2 != nullptr),
               "tg_d_head_bwd: parameter gradients are all given or all NULL");
    if (dw1 && deterministic()) {
        // the input gradient by the parallel kernel (no parameter gradients: nothing to combine), the four parameter gradients by ONE workgroup
        // in a fixed summation order, no atomics (round 5: that workgroup used to write dy for the whole batch as well, 216 us)
        hipLaunchKernelGGL(d_head_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, d_logit, y, l1, w1, w2, dy, (float*)nullptr,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, B, T, H);
        hipLaunchKernelGGL(d_head_bwd_det_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_logit, y, l1, w1, w2, dy, dw1, db1, dw2, db2, B, T, H);
        return check_launch("tg_d_head_bwd(deterministic)");
    }
    hipLaunchKernelGGL(d_head_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, d_logit, y, l1, w1, w2, dy, dw1, db1, dw2, db2, B,
                       T, H);
    return check_launch("tg_d_head_bwd");
}
int tg_d_head_step(const float* y, const float* w1, const float* b1, const float* w2, const float* b2, float* l1, float* logit, float* prob,
                   float* d_logit, float* terms, float* out, uint32_t* counter, float* dy, float* dw1, float* db1, float* dw2, float* db2,
                   int32_t n_rows, int32_t n_real, float scale_real, float scale_fake, int32_t T, int32_t H, void* stream) {
    TG_REQUIRE(y && w1 && b1 && w2 && b2 && l1 && logit && prob && d_logit && terms && (out == nullptr || counter) && dy && n_rows > 0 &&
                   n_real > 0 && n_real <= n_rows && T > 0 && T <= 32 && H > 0 && H <= 64, "tg_d_head_step: bad arguments (T <= 32, H <= 64, 0 < n_real <= n_rows)");
    TG_REQUIRE((dw1 != nullptr) == (db1 != nullptr) && (dw1 != nullptr) == (dw2 != nullptr) && (dw1 != nullptr) == (db2 != nullptr),
               "tg_d_head_step: parameter gradients are all given or all NULL");
    const bool det = dw1 && deterministic();
    hipLaunchKernelGGL(d_head_step_kernel, dim3(cdiv(n_rows, 4)), dim3(1024), 0, (hipStream_t)stream, y, w1, b1, w2, b2, l1, logit, prob, d_logit,
                       terms, out, counter, dy, det ? nullptr : dw1, det ? nullptr : db1, det ? nullptr : dw2, det ? nullptr : db2, n_rows, n_real,
                       scale_real, scale_fake, T, H);
    if (det)       // parameter gradients by ONE workgroup in a fixed order (as tg_d_head_bwd does in this mode)
        hipLaunchKernelGGL(d_head_bwd_det_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_logit, y, l1, w1, w2, dy, dw1, db1, dw2, db2, n_rows, T, H);
    return check_launch("tg_d_head_step");
}
int tg_l1_mean(const float* a, const float* b, int64_t n, float* out, void* stream) {
    TG_REQUIRE(a && b && out && n > 0, "tg_l1_mean: bad arguments");
    if (zero_async(out, sizeof(float), ST)) return 1;
    hipLaunchKernelGGL(l1_mean_kernel, dim3(ew_grid(n, 256, 8)), dim3(256), 0, ST, a, b, (long)n, out);
    return check_launch("tg_l1_mean");
}

int tg_window_blend(const float* prev_tail, float* next, int32_t B, int32_t T, int32_t D, int32_t n, void* stream) {
    TG_REQUIRE(prev_tail && next && B > 0 && T > 0 && D > 0 && n > 0 && n <= T, "tg_window_blend: bad arguments");
    hipLaunchKernelGGL(window_blend_kernel, dim3(ew_grid((long)B * n * D)), dim3(256), 0, ST, prev_tail, next, B, T, D, n);
    return check_launch("tg_window_blend");
}

int tg_pose_metrics(const float* out_dir_vec, const float* target_dir_vec, const float* mean_dir_vec, int32_t B, int32_t T,
                    int32_t n_pre, double* sums, void* stream) {
    TG_REQUIRE(out_dir_vec && target_dir_vec && mean_dir_vec && sums && B > 0 && T > 2 && n_pre >= 0 && n_pre < T, "tg_pose_metrics: bad arguments");
    if (zero_async(sums, 3 * sizeof(double), ST)) return 1;
    hipLaunchKernelGGL(pose_metrics_kernel, dim3(ew_grid((long)B * T, 256, 1)), dim3(256), 0, ST, out_dir_vec, target_dir_vec, mean_dir_vec, B, T,
                       n_pre, sums);
    return check_launch("tg_pose_metrics");
}

int tg_ae_loss(const float* recon, const float* target, int32_t B, int32_t T, int32_t D, float* out, float* d_recon, void* stream) {
    TG_REQUIRE(recon && target && out && d_recon && B > 0 && T > 1 && D > 0, "tg_ae_loss: bad arguments");
    if (zero_async(out, sizeof(float), ST)) return 1;
    hipLaunchKernelGGL(ae_loss_kernel, dim3(ew_grid((long)B * T * D, 256, 4)), dim3(256), 0, ST, recon, target, B, T, D, out, d_recon);
    return check_launch("tg_ae_loss");
}

}  // extern "C"
