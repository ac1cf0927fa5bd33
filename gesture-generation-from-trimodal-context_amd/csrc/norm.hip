// BatchNorm1d (train + eval), channel-last [rows][C].  HBM-bound streaming kernels: coalesced 16-byte accesses,
// per-thread fp64 partial sums, one LDS reduction per workgroup, one fp64 atomic per (workgroup, channel).
// Statistics of "groups" stacked reference forward calls are kept apart (each call normalises with its own batch).
#include "common.hpp"

namespace tg {

// ws layout: [groups][2][C] doubles = (sum, sum of squares)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int rows_per_group, int C, double* __restrict__ ws) {
    __shared__ double sh[2][256];
    const int g = blockIdx.y;
    const int rpi = 256 / C;                 // rows per iteration of this workgroup (C <= 256)
    const int c = threadIdx.x % C, rsub = threadIdx.x / C;
    const float* xg = x + (long)g * rows_per_group * C;
    double s = 0.0, ss = 0.0;
    if (rsub < rpi)
        for (long r = (long)blockIdx.x * rpi + rsub; r < rows_per_group; r += (long)gridDim.x * rpi) {
            const float v = xg[r * C + c];
            s += v;
            ss += (double)v * v;
        }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x < C) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < rpi; ++q) { a += sh[0][q * C + c]; b += sh[1][q * C + c]; }
        atomicAdd(&ws[((long)g * 2 + 0) * C + c], a);
        atomicAdd(&ws[((long)g * 2 + 1) * C + c], b);
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ ws, int rows_per_group, int C, int groups, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                   int64_t* __restrict__ nbt, float eps, float momentum, int repeats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const double n = (double)rows_per_group;
        float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
        for (int g = 0; g < groups; ++g) {
            const double m = ws[((long)g * 2 + 0) * C + c] / n;
            double var = ws[((long)g * 2 + 1) * C + c] / n - m * m;
            if (var < 0.0) var = 0.0;
            mean[g * C + c] = (float)m;
            rstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
            const double unbiased = rows_per_group > 1 ? var * n / (n - 1.0) : var;
            for (int q = 0; q < repeats; ++q) {   // the same batch normalised by `repeats` identical forward calls
                rm = (1.f - momentum) * rm + momentum * (float)m;
                rv = (1.f - momentum) * rv + momentum * (float)unbiased;
            }
        }
        if (rmean) rmean[c] = rm;
        if (rvar) rvar[c] = rv;
    }
    if (c == 0 && nbt) *nbt += (int64_t)groups * repeats;
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mean[c] = rmean[c];
        rstd[c] = 1.f / sqrtf(rvar[c] + eps);
    }
}

// y = act((x - mean) * rstd * gamma + beta); four channels per thread (C % 4 == 0 on this path), scalar otherwise
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int C,
                                                       int rows_per_group, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float slope) {
    const int W = VEC ? 4 : 1;
    const long total = rows * C / W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long e = i * W;
        const long row = e / C;
        const int c = (int)(e - row * C);
        const int g = (int)(row / rows_per_group);
        if (VEC) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + e);
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xh = (xv[q] - mean[g * C + c + q]) * rstd[g * C + c + q];
                o[q] = act_fn(xh * gamma[c + q] + beta[c + q], slope);
            }
            *reinterpret_cast<f32x4*>(y + e) = o;
        } else {
            const float xh = (x[e] - mean[g * C + c]) * rstd[g * C + c];
            y[e] = act_fn(xh * gamma[c] + beta[c], slope);
        }
    }
}

// ws: [2][C] doubles = (sum dz, sum dz * xhat), dz = dy * act'(z)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x, int rows, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float slope, double* __restrict__ ws) {
    __shared__ double sh[2][256];
    const int rpi = 256 / C;
    const int c = threadIdx.x % C, rsub = threadIdx.x / C;
    double s = 0.0, sx = 0.0;
    if (rsub < rpi) {
        const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
        for (long r = (long)blockIdx.x * rpi + rsub; r < rows; r += (long)gridDim.x * rpi) {
            const float xh = (x[r * C + c] - mu) * rs;
            const float z = xh * ga + be;
            const float dz = dy[r * C + c] * (z >= 0.f ? 1.f : slope);
            s += dz;
            sx += (double)dz * xh;
        }
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = sx;
    __syncthreads();
    if (threadIdx.x < C) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < rpi; ++q) { a += sh[0][q * C + c]; b += sh[1][q * C + c]; }
        atomicAdd(&ws[c], a);
        atomicAdd(&ws[C + c], b);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                                           long rows, int C, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float slope,
                                                           const double* __restrict__ ws, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
    if (blockIdx.x == 0 && (int)threadIdx.x < C) {
        const int c = threadIdx.x;
        if (dbeta) dbeta[c] += (float)ws[c];
        if (dgamma) dgamma[c] += (float)ws[C + c];
    }
    const long total = rows * C;
    const double inv_n = 1.0 / (double)rows;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const float rs = rstd[c], ga = gamma[c];
        const float xh = (x[e] - mean[c]) * rs;
        const float z = xh * ga + beta[c];
        const float dz = dy[e] * (z >= 0.f ? 1.f : slope);
        // the two batch means are subtracted in fp64: rounding them to fp32 first would shift every element of the
        // channel by the same amount, and the next layer's weight-gradient sum over ~1e6 rows amplifies that coherently
        const double m1 = ws[c] * inv_n, m2 = ws[C + c] * inv_n;
        dx[e] = (float)((double)(ga * rs) * ((double)dz - m1 - (double)xh * m2));
    }
}

// ---- small tensors (the discriminator's and the autoencoder's BatchNorms: a few thousand rows): ONE workgroup does statistics,
// running-stat update and normalisation in one launch -- the multi-kernel path costs four launches of ~5 us for microseconds of work.
constexpr int BN_SMALL_THREADS = 1024;
constexpr long BN_SMALL_MAX = 1L << 19;          // elements (2 MB): stays in L2 between the two passes

__device__ __forceinline__ void block_sum2(double& a, double& b, double (*sh)[BN_SMALL_THREADS], int C, int rpi) {
    // a, b: per-thread partials of channel (threadIdx.x % C); on return threads < C hold the channel totals
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        double x = 0.0, y = 0.0;
        for (int q = 0; q < rpi; ++q) { x += sh[0][q * C + threadIdx.x]; y += sh[1][q * C + threadIdx.x]; }
        a = x; b = y;
    }
    __syncthreads();
}

__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_small_train_kernel(
    const float* __restrict__ x, float* __restrict__ y, int rows_per_group, int C, int groups, float* __restrict__ mean, float* __restrict__ rstd,
    float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt, const float* __restrict__ gamma,
    const float* __restrict__ beta, float slope, float eps, float momentum, int repeats) {
    __shared__ double sh[2][BN_SMALL_THREADS];
    __shared__ float s_mean[256], s_rstd[256];
    const int rpi = BN_SMALL_THREADS / C;
    const int c = threadIdx.x % C, rsub = threadIdx.x / C;
    float rm = 0.f, rv = 0.f;
    if ((int)threadIdx.x < C) { rm = rmean ? rmean[c] : 0.f; rv = rvar ? rvar[c] : 0.f; }
    for (int g = 0; g < groups; ++g) {
        const float* xg = x + (long)g * rows_per_group * C;
        double s = 0.0, ss = 0.0;
        if (rsub < rpi)
            for (int r = rsub; r < rows_per_group; r += rpi) {
                const float v = xg[(long)r * C + c];
                s += v;
                ss += (double)v * v;
            }
        block_sum2(s, ss, sh, C, rpi);
        if ((int)threadIdx.x < C) {
            const double n = (double)rows_per_group;
            const double m = s / n;
            double var = ss / n - m * m;
            if (var < 0.0) var = 0.0;
            const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
            mean[g * C + c] = mf; rstd[g * C + c] = rs;
            s_mean[c] = mf; s_rstd[c] = rs;
            const double unbiased = rows_per_group > 1 ? var * n / (n - 1.0) : var;
            for (int q = 0; q < repeats; ++q) {
                rm = (1.f - momentum) * rm + momentum * mf;
                rv = (1.f - momentum) * rv + momentum * (float)unbiased;
            }
        }
        __syncthreads();
        if (y) {
            float* yg = y + (long)g * rows_per_group * C;
            const long total = (long)rows_per_group * C;
            for (long e = threadIdx.x; e < total; e += BN_SMALL_THREADS) {
                const int cc = (int)(e % C);
                yg[e] = act_fn((xg[e] - s_mean[cc]) * s_rstd[cc] * gamma[cc] + beta[cc], slope);
            }
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < C) {
        if (rmean) rmean[c] = rm;
        if (rvar) rvar[c] = rv;
    }
    if (threadIdx.x == 0 && nbt) *nbt += (int64_t)groups * repeats;
}

__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_small_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int rows, int C, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sh[2][BN_SMALL_THREADS];
    __shared__ double s_m1[256], s_m2[256];
    const int rpi = BN_SMALL_THREADS / C;
    const int c = threadIdx.x % C, rsub = threadIdx.x / C;
    double s = 0.0, sx = 0.0;
    if (rsub < rpi) {
        const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
        for (int r = rsub; r < rows; r += rpi) {
            const float xh = (x[(long)r * C + c] - mu) * rs;
            const float z = xh * ga + be;
            const float dz = dy[(long)r * C + c] * (z >= 0.f ? 1.f : slope);
            s += dz;
            sx += (double)dz * xh;
        }
    }
    block_sum2(s, sx, sh, C, rpi);
    if ((int)threadIdx.x < C) {
        if (dbeta) dbeta[c] += (float)s;
        if (dgamma) dgamma[c] += (float)sx;
        s_m1[c] = s / (double)rows;
        s_m2[c] = sx / (double)rows;
    }
    __syncthreads();
    const long total = (long)rows * C;
    for (long e = threadIdx.x; e < total; e += BN_SMALL_THREADS) {
        const int cc = (int)(e % C);
        const float rs = rstd[cc], ga = gamma[cc];
        const float xh = (x[e] - mean[cc]) * rs;
        const float z = xh * ga + beta[cc];
        const float dz = dy[e] * (z >= 0.f ? 1.f : slope);
        dx[e] = (float)((double)(ga * rs) * ((double)dz - s_m1[cc] - (double)xh * s_m2[cc]));
    }
}

}  // namespace tg

using namespace tg;

// statistics + running-stat update + normalisation of a small tensor in ONE launch (y may be NULL: statistics only)
extern "C" int tg_bn_train_fused(const float* x, float* y, int32_t rows, int32_t C, int32_t groups, float* mean, float* rstd,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                                 const float* beta, float act_slope, float eps, float momentum, int32_t repeats, void* stream) {
    TG_REQUIRE(x && mean && rstd && repeats >= 1 && (y == nullptr || (gamma && beta)), "tg_bn_train_fused: null pointer / repeats < 1");
    TG_REQUIRE(C > 0 && C <= 256 && groups > 0 && rows > 0 && rows % groups == 0 && (long)rows * C <= BN_SMALL_MAX,
               "tg_bn_train_fused: C=%d (<=256), rows=%d, groups=%d, at most %ld elements", C, rows, groups, BN_SMALL_MAX);
    hipLaunchKernelGGL(bn_small_train_kernel, dim3(1), dim3(BN_SMALL_THREADS), 0, (hipStream_t)stream, x, y, rows / groups, C, groups, mean, rstd,
                       running_mean, running_var, num_batches_tracked, gamma, beta, act_slope, eps, momentum, repeats);
    return check_launch("tg_bn_train_fused");
}

extern "C" int tg_bn_train_stats(const float* x, int32_t rows, int32_t C, int32_t groups, double* ws, float* mean, float* rstd,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                                 float momentum, int32_t repeats, void* stream) {
    TG_REQUIRE(x && ws && mean && rstd && repeats >= 1, "tg_bn_train_stats: null pointer / repeats < 1");
    TG_REQUIRE(C > 0 && C <= 256 && groups > 0 && rows > 0 && rows % groups == 0, "tg_bn_train_stats: C=%d (<=256), rows=%d, groups=%d", C, rows, groups);
    hipStream_t s = (hipStream_t)stream;
    if (zero_async(ws, sizeof(double) * 2 * (size_t)groups * C, s)) return 1;
    const int rpg = rows / groups;
    const int rpi = 256 / C;
    int blocks = cdiv(rpg, rpi * 16);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks, groups), dim3(256), 0, s, x, rpg, C, ws);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, ws, rpg, C, groups, mean, rstd, running_mean, running_var,
                       num_batches_tracked, eps, momentum, repeats);
    return check_launch("tg_bn_train_stats");
}

extern "C" int tg_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps, float* mean,
                                float* rstd, void* stream) {
    TG_REQUIRE(running_mean && running_var && mean && rstd && C > 0, "tg_bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, running_mean, running_var, C, eps, mean, rstd);
    return check_launch("tg_bn_eval_stats");
}

extern "C" int tg_bn_apply(const float* x, float* y, int32_t rows, int32_t C, int32_t groups, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, float act_slope, void* stream) {
    TG_REQUIRE(x && y && mean && rstd && gamma && beta, "tg_bn_apply: null pointer");
    TG_REQUIRE(rows > 0 && C > 0 && groups > 0 && rows % groups == 0, "tg_bn_apply: bad sizes");
    const bool vec = (C % 4 == 0) && aligned16(x) && aligned16(y);
    const long total = (long)rows * C;
    if (vec) hipLaunchKernelGGL((bn_apply_kernel<true>), dim3(ew_grid(total / 4, 256, 2)), dim3(256), 0, (hipStream_t)stream, x, y, (long)rows, C, rows / groups, mean, rstd, gamma, beta, act_slope);
    else     hipLaunchKernelGGL((bn_apply_kernel<false>), dim3(ew_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, x, y, (long)rows, C, rows / groups, mean, rstd, gamma, beta, act_slope);
    return check_launch("tg_bn_apply");
}

extern "C" int tg_bn_backward(const float* dy, const float* x, float* dx, int32_t rows, int32_t C, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, float act_slope, double* ws,
                              float* dgamma, float* dbeta, void* stream) {
    TG_REQUIRE(dy && x && dx && mean && rstd && gamma && beta && ws, "tg_bn_backward: null pointer");
    TG_REQUIRE(rows > 0 && C > 0 && C <= 256, "tg_bn_backward: C=%d must be <= 256", C);
    hipStream_t s = (hipStream_t)stream;
    if ((long)rows * C <= BN_SMALL_MAX) {          // one workgroup: reduce, then apply (3 launches -> 1)
        hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(1), dim3(BN_SMALL_THREADS), 0, s, dy, x, dx, rows, C, mean, rstd, gamma, beta, act_slope,
                           dgamma, dbeta);
        return check_launch("tg_bn_backward(small)");
    }
    if (zero_async(ws, sizeof(double) * 2 * (size_t)C, s)) return 1;
    const int rpi = 256 / C;
    int blocks = cdiv(rows, rpi * 16);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, s, dy, x, rows, C, mean, rstd, gamma, beta, act_slope, ws);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid((long)rows * C, 256, 4)), dim3(256), 0, s, dy, x, dx, (long)rows, C, mean, rstd,
                       gamma, beta, act_slope, ws, dgamma, dbeta);
    return check_launch("tg_bn_backward");
}
