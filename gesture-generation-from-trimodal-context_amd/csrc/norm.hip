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

// 16-byte version (C % 4 == 0, C divides 1024): a thread's four channels never change (its element index advances by a multiple of C), so
// the partial sums are eight fp64 registers and four independent 16-byte loads are in flight per thread -- the scalar kernel above had one
// 4-byte load in flight per thread and ran the audio encoder's 21 MB BatchNorm at 0.75 TB/s.
__global__ __launch_bounds__(256) void bn_stats_vec_kernel(const float* __restrict__ x, long rows_per_group, int C, double* __restrict__ ws) {
    __shared__ double sh[2][256];
    const int g = blockIdx.y;
    const f32x4* xg = reinterpret_cast<const f32x4*>(x + (long)g * rows_per_group * C);
    const long total4 = rows_per_group * C / 4;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
    const int c0 = (int)((i0 * 4) % C);
    if ((int)threadIdx.x < C) { sh[0][threadIdx.x] = 0.0; sh[1][threadIdx.x] = 0.0; }
    __syncthreads();
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (long i = i0; i < total4; i += step) {
        const f32x4 v = xg[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) { s[q] += v[q]; ss[q] += (double)v[q] * v[q]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { atomicAdd(&sh[0][c0 + q], s[q]); atomicAdd(&sh[1][c0 + q], ss[q]); }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        atomicAdd(&ws[((long)g * 2 + 0) * C + threadIdx.x], sh[0][threadIdx.x]);
        atomicAdd(&ws[((long)g * 2 + 1) * C + threadIdx.x], sh[1][threadIdx.x]);
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

// 16-byte versions of the two backward kernels (same conditions as bn_stats_vec_kernel)
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x, long rows, int C,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float slope, double* __restrict__ ws) {
    __shared__ double sh[2][256];
    const long total4 = rows * C / 4;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
    const int c0 = (int)((i0 * 4) % C);
    if ((int)threadIdx.x < C) { sh[0][threadIdx.x] = 0.0; sh[1][threadIdx.x] = 0.0; }
    __syncthreads();
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mu[q] = mean[c0 + q]; rs[q] = rstd[c0 + q]; ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q]; }
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (long i = i0; i < total4; i += step) {
        const f32x4 xv = x4[i], dv = dy4[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xv[q] - mu[q]) * rs[q];
            const float z = xh * ga[q] + be[q];
            const float dz = dv[q] * (z >= 0.f ? 1.f : slope);
            s[q] += dz;
            sx[q] += (double)dz * xh;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { atomicAdd(&sh[0][c0 + q], s[q]); atomicAdd(&sh[1][c0 + q], sx[q]); }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        atomicAdd(&ws[threadIdx.x], sh[0][threadIdx.x]);
        atomicAdd(&ws[C + threadIdx.x], sh[1][threadIdx.x]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
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
    const long total4 = rows * C / 4;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
    const int c0 = (int)((i0 * 4) % C);
    const double inv_n = 1.0 / (double)rows;
    float mu[4], rs[4], ga[4], be[4];
    double m1[4], m2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mu[q] = mean[c0 + q]; rs[q] = rstd[c0 + q]; ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q];
        m1[q] = ws[c0 + q] * inv_n; m2[q] = ws[C + c0 + q] * inv_n;
    }
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* dx4 = reinterpret_cast<f32x4*>(dx);
#pragma unroll 2
    for (long i = i0; i < total4; i += step) {
        const f32x4 xv = x4[i], dv = dy4[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xv[q] - mu[q]) * rs[q];
            const float z = xh * ga[q] + be[q];
            const float dz = dv[q] * (z >= 0.f ? 1.f : slope);
            o[q] = (float)((double)(ga[q] * rs[q]) * ((double)dz - m1[q] - (double)xh * m2[q]));     // fp64 means: see bn_bwd_apply_kernel
        }
        dx4[i] = o;
    }
}

// ---- small tensors (the discriminator's and the autoencoder's BatchNorms: a few thousand rows): ONE workgroup does statistics,
// running-stat update and normalisation in one launch -- the multi-kernel path costs four launches of ~5 us for microseconds of work.
// 16-byte accesses throughout: with C a multiple of 4 that divides 4096, thread t always sees the same four channels (its element
// index advances by 4096 per iteration), so the per-thread partial sums are eight fp64 registers and the loads of successive
// iterations are independent requests in flight together.
constexpr int BN_SMALL_THREADS = 1024;
constexpr long BN_SMALL_MAX = 1L << 19;          // elements (2 MB): stays in L2 between the two passes

// per-channel totals of the four per-thread partial pairs: threads with equal (t * 4) % C hold the same channels
__device__ __forceinline__ void bn_small_reduce(const double (&a)[4], const double (&b)[4], double* sh_a, double* sh_b, int C) {
    // sh_a / sh_b: [C] accumulators in LDS, zeroed by the caller
    const int c0 = (threadIdx.x * 4) % C;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        atomicAdd(&sh_a[c0 + q], a[q]);          // LDS fp64 atomics: 1024 / (C / 4) adders per channel, once per pass
        atomicAdd(&sh_b[c0 + q], b[q]);
    }
    __syncthreads();
}

__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_small_train_kernel(
    const float* __restrict__ x, float* __restrict__ y, int rows_per_group, int C, int groups, float* __restrict__ mean, float* __restrict__ rstd,
    float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt, const float* __restrict__ gamma,
    const float* __restrict__ beta, float slope, float eps, float momentum, int repeats) {
    __shared__ double sh_a[256], sh_b[256];
    __shared__ float s_mean[256], s_rstd[256];
    const int c0 = (threadIdx.x * 4) % C;
    float rm = 0.f, rv = 0.f;
    if ((int)threadIdx.x < C) { rm = rmean ? rmean[threadIdx.x] : 0.f; rv = rvar ? rvar[threadIdx.x] : 0.f; }
    const long total4 = (long)rows_per_group * C / 4;
    for (int g = 0; g < groups; ++g) {
        const f32x4* xg = reinterpret_cast<const f32x4*>(x + (long)g * rows_per_group * C);
        if ((int)threadIdx.x < C) { sh_a[threadIdx.x] = 0.0; sh_b[threadIdx.x] = 0.0; }
        __syncthreads();
        double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (long i = threadIdx.x; i < total4; i += BN_SMALL_THREADS) {
            const f32x4 v = xg[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) { s[q] += v[q]; ss[q] += (double)v[q] * v[q]; }
        }
        bn_small_reduce(s, ss, sh_a, sh_b, C);
        if ((int)threadIdx.x < C) {
            const int c = threadIdx.x;
            const double n = (double)rows_per_group;
            const double m = sh_a[c] / n;
            double var = sh_b[c] / n - m * m;
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
            f32x4* yg = reinterpret_cast<f32x4*>(y + (long)g * rows_per_group * C);
            float mu[4], rs[4], ga[4], be[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { mu[q] = s_mean[c0 + q]; rs[q] = s_rstd[c0 + q]; ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q]; }
#pragma unroll 4
            for (long i = threadIdx.x; i < total4; i += BN_SMALL_THREADS) {
                const f32x4 v = xg[i];
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = act_fn((v[q] - mu[q]) * rs[q] * ga[q] + be[q], slope);   // association of bn_apply_kernel
                yg[i] = o;
            }
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < C) {
        if (rmean) rmean[threadIdx.x] = rm;
        if (rvar) rvar[threadIdx.x] = rv;
    }
    if (threadIdx.x == 0 && nbt) *nbt += (int64_t)groups * repeats;
}

__global__ __launch_bounds__(BN_SMALL_THREADS) void bn_small_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int rows, int C, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sh_a[256], sh_b[256];
    const int c0 = (threadIdx.x * 4) % C;
    if ((int)threadIdx.x < C) { sh_a[threadIdx.x] = 0.0; sh_b[threadIdx.x] = 0.0; }
    __syncthreads();
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mu[q] = mean[c0 + q]; rs[q] = rstd[c0 + q]; ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q]; }
    const long total4 = (long)rows * C / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (long i = threadIdx.x; i < total4; i += BN_SMALL_THREADS) {
        const f32x4 xv = x4[i], dv = dy4[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xv[q] - mu[q]) * rs[q];
            const float z = xh * ga[q] + be[q];
            const float dz = dv[q] * (z >= 0.f ? 1.f : slope);
            s[q] += dz;
            sx[q] += (double)dz * xh;
        }
    }
    bn_small_reduce(s, sx, sh_a, sh_b, C);
    if ((int)threadIdx.x < C) {
        if (dbeta) dbeta[threadIdx.x] += (float)sh_a[threadIdx.x];
        if (dgamma) dgamma[threadIdx.x] += (float)sh_b[threadIdx.x];
    }
    double m1[4], m2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { m1[q] = sh_a[c0 + q] / (double)rows; m2[q] = sh_b[c0 + q] / (double)rows; }
    f32x4* dx4 = reinterpret_cast<f32x4*>(dx);
#pragma unroll 4
    for (long i = threadIdx.x; i < total4; i += BN_SMALL_THREADS) {
        const f32x4 xv = x4[i], dv = dy4[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xv[q] - mu[q]) * rs[q];
            const float z = xh * ga[q] + be[q];
            const float dz = dv[q] * (z >= 0.f ? 1.f : slope);
            o[q] = (float)((double)(ga[q] * rs[q]) * ((double)dz - m1[q] - (double)xh * m2[q]));
        }
        dx4[i] = o;
    }
}

// ---- two-launch BatchNorm (train forward: partial sums | finalize + apply; backward: partial sums | finalize + apply) --------------------
// For everything between the single-workgroup kernels above (<= 32 K elements) and -- replacing them -- the zero / stats / finalize / apply
// chain: no zero fill, no atomics (per-workgroup partials combined in fixed order by every workgroup of the second launch), all statistics
// groups of a stacked forward in ONE launch (grid.y), and the backward of several groups likewise (the discriminator step ran one
// single-workgroup launch of 13-16 us per group and layer).
// part layout: [groups][P][2][C] doubles.
__global__ __launch_bounds__(256) void bn2_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, long rows_per_group, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                          double* __restrict__ part, int det) {
    // dy == nullptr: (sum x, sum x^2); else (sum dz, sum dz xhat) with dz = dy act'(z), statistics of group blockIdx.y at mean/rstd + g C
    // det (tg_set_deterministic): the threads' sums meet in LDS in thread order instead of through fp64 LDS atomics
    __shared__ double sh[2][256];
    __shared__ double sh_det[256][8];
    const int g = blockIdx.y, P = gridDim.x;
    const long total4 = rows_per_group * C / 4;
    const long base4 = (long)g * total4;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)P * 256;
    const int c0 = (int)((i0 * 4) % C);
    if ((int)threadIdx.x < C) { sh[0][threadIdx.x] = 0.0; sh[1][threadIdx.x] = 0.0; }
    __syncthreads();
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + base4;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    if (dy == nullptr) {
#pragma unroll 4
        for (long i = i0; i < total4; i += step) {
            const f32x4 v = x4[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) { s[q] += v[q]; ss[q] += (double)v[q] * v[q]; }
        }
    } else {
        const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy) + base4;
        float mu[4], rs[4], ga[4], be[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { mu[q] = mean[g * C + c0 + q]; rs[q] = rstd[g * C + c0 + q]; ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q]; }
#pragma unroll 4
        for (long i = i0; i < total4; i += step) {
            const f32x4 xv = x4[i], dv = dy4[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xh = (xv[q] - mu[q]) * rs[q];
                const float z = xh * ga[q] + be[q];
                const float dz = dv[q] * (z >= 0.f ? 1.f : slope);
                s[q] += dz;
                ss[q] += (double)dz * xh;
            }
        }
    }
    if (det) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sh_det[threadIdx.x][q] = s[q]; sh_det[threadIdx.x][4 + q] = ss[q]; }
        __syncthreads();
        if ((int)threadIdx.x < C) {                    // channel c lives in the threads t == (c >> 2) mod (C / 4), slot c & 3 (step % C == 0: fixed per thread)
            const int c = threadIdx.x;
            double a = 0.0, b = 0.0;
            for (int t = c >> 2; t < 256; t += C / 4) { a += sh_det[t][c & 3]; b += sh_det[t][4 + (c & 3)]; }
            sh[0][c] = a; sh[1][c] = b;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { atomicAdd(&sh[0][c0 + q], s[q]); atomicAdd(&sh[1][c0 + q], ss[q]); }
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        double* o = part + (((long)g * P + blockIdx.x) * 2) * C;
        o[threadIdx.x] = sh[0][threadIdx.x];
        o[C + threadIdx.x] = sh[1][threadIdx.x];
    }
}

// totals of group g's P partials into tot[2][C] (LDS), fixed order: thread (c, slice) sums every (256 / C)-th partial, then the slices
__device__ __forceinline__ void bn2_totals(const double* __restrict__ part, int g, int P, int C, double (*tot)[256], double (*sl)[256]) {
    const int S = 256 / C;                       // slices
    const int c = threadIdx.x % C, sidx = threadIdx.x / C;
    double a = 0.0, b = 0.0;
    if (sidx < S) {
#pragma unroll 8
        for (int p = sidx; p < P; p += S) {          // independent loads, eight in flight
            const double* o = part + (((long)g * P + p) * 2) * C;
            a += o[c];
            b += o[C + c];
        }
    }
    sl[0][threadIdx.x] = a;
    sl[1][threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        double ta = 0.0, tb = 0.0;
        for (int q = 0; q < S; ++q) { ta += sl[0][q * C + c]; tb += sl[1][q * C + c]; }
        tot[0][c] = ta;
        tot[1][c] = tb;
    }
    __syncthreads();
}

// many partials (large tensors): combined once per group by this launch instead of by every workgroup of the apply launch
__global__ __launch_bounds__(256) void bn2_reduce_kernel(const double* __restrict__ part, int P, int C, double* __restrict__ out) {
    __shared__ double tot[2][256], sl[2][256];
    bn2_totals(part, blockIdx.x, P, C, tot, sl);
    if ((int)threadIdx.x < C) {
        out[((long)blockIdx.x * 2) * C + threadIdx.x] = tot[0][threadIdx.x];
        out[((long)blockIdx.x * 2 + 1) * C + threadIdx.x] = tot[1][threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void bn2_fwd_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long rows_per_group, int C, int groups,
                                                            const double* __restrict__ part, int P, float* __restrict__ mean, float* __restrict__ rstd,
                                                            float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float slope, float eps,
                                                            float momentum, int repeats) {
    __shared__ double tot[2][256], sl[2][256];
    __shared__ float s_mean[256], s_rstd[256];
    const int g = blockIdx.y;
    const double n = (double)rows_per_group;
    if (blockIdx.x == 0 && g == 0) {
        // running statistics: the groups are successive forward calls of the same module -- updated in call order by one workgroup
        float rm = 0.f, rv = 0.f;
        if ((int)threadIdx.x < C) { rm = rmean ? rmean[threadIdx.x] : 0.f; rv = rvar ? rvar[threadIdx.x] : 0.f; }
        for (int gg = 0; gg < groups; ++gg) {
            bn2_totals(part, gg, P, C, tot, sl);
            if ((int)threadIdx.x < C) {
                const int c = threadIdx.x;
                const double m = tot[0][c] / n;
                double var = tot[1][c] / n - m * m;
                if (var < 0.0) var = 0.0;
                const double unbiased = rows_per_group > 1 ? var * n / (n - 1.0) : var;
                for (int q = 0; q < repeats; ++q) {
                    rm = (1.f - momentum) * rm + momentum * (float)m;
                    rv = (1.f - momentum) * rv + momentum * (float)unbiased;
                }
            }
            __syncthreads();
        }
        if ((int)threadIdx.x < C) {
            if (rmean) rmean[threadIdx.x] = rm;
            if (rvar) rvar[threadIdx.x] = rv;
        }
        if (threadIdx.x == 0 && nbt) *nbt += (int64_t)groups * repeats;
    }
    bn2_totals(part, g, P, C, tot, sl);
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        const double m = tot[0][c] / n;
        double var = tot[1][c] / n - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        s_mean[c] = mf; s_rstd[c] = rs;
        if (blockIdx.x == 0) { mean[g * C + c] = mf; rstd[g * C + c] = rs; }
    }
    __syncthreads();
    if (y == nullptr) return;
    const long total4 = rows_per_group * C / 4;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
    const int c0 = (int)((i0 * 4) % C);
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mu[q] = s_mean[c0 + q]; rs[q] = s_rstd[c0 + q]; ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q]; }
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + (long)g * total4;
    f32x4* y4 = reinterpret_cast<f32x4*>(y) + (long)g * total4;
#pragma unroll 4
    for (long i = i0; i < total4; i += step) {
        const f32x4 v = x4[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = act_fn((v[q] - mu[q]) * rs[q] * ga[q] + be[q], slope);          // association of bn_apply_kernel
        y4[i] = o;
    }
}

__global__ __launch_bounds__(256) void bn2_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                                            long rows_per_group, int C, int groups, const double* __restrict__ part, int P,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double tot[2][256], sl[2][256];
    const int g = blockIdx.y;
    if (blockIdx.x == 0 && g == 0 && (dgamma || dbeta)) {
        double a = 0.0, b = 0.0;                 // parameter gradients: the groups' sums in group order, one workgroup
        for (int gg = 0; gg < groups; ++gg) {
            bn2_totals(part, gg, P, C, tot, sl);
            if ((int)threadIdx.x < C) { a += tot[0][threadIdx.x]; b += tot[1][threadIdx.x]; }
            __syncthreads();
        }
        if ((int)threadIdx.x < C) {
            if (dbeta) dbeta[threadIdx.x] += (float)a;
            if (dgamma) dgamma[threadIdx.x] += (float)b;
        }
    }
    bn2_totals(part, g, P, C, tot, sl);
    const long total4 = rows_per_group * C / 4;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
    const int c0 = (int)((i0 * 4) % C);
    const double inv_n = 1.0 / (double)rows_per_group;
    float mu[4], rs[4], ga[4], be[4];
    double m1[4], m2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mu[q] = mean[g * C + c0 + q]; rs[q] = rstd[g * C + c0 + q]; ga[q] = gamma[c0 + q]; be[q] = beta[c0 + q];
        m1[q] = tot[0][c0 + q] * inv_n; m2[q] = tot[1][c0 + q] * inv_n;
    }
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + (long)g * total4;
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy) + (long)g * total4;
    f32x4* dx4 = reinterpret_cast<f32x4*>(dx) + (long)g * total4;
#pragma unroll 2
    for (long i = i0; i < total4; i += step) {
        const f32x4 xv = x4[i], dv = dy4[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xv[q] - mu[q]) * rs[q];
            const float z = xh * ga[q] + be[q];
            const float dz = dv[q] * (z >= 0.f ? 1.f : slope);
            o[q] = (float)((double)(ga[q] * rs[q]) * ((double)dz - m1[q] - (double)xh * m2[q]));     // fp64 means: see bn_bwd_apply_kernel
        }
        dx4[i] = o;
    }
}

// 16-byte kernels: a thread keeps the same four channels across its grid-stride loop
inline bool bn_vec_ok(int C, long elems) { return C >= 4 && C <= 256 && C % 4 == 0 && 1024 % C == 0 && elems % 4 == 0; }

}  // namespace tg

using namespace tg;

extern "C" int32_t tg_bn_fused_supported(int32_t rows, int32_t C, int32_t groups) {
    return C >= 4 && C <= 256 && C % 4 == 0 && 4096 % C == 0 && groups > 0 && rows > 0 && rows % groups == 0 && (long)rows * C <= BN_SMALL_MAX &&
           ((long)(rows / groups) * C) % 4 == 0;
}

// statistics + running-stat update + normalisation of a small tensor in ONE launch (y may be NULL: statistics only)
extern "C" int tg_bn_train_fused(const float* x, float* y, int32_t rows, int32_t C, int32_t groups, float* mean, float* rstd,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                                 const float* beta, float act_slope, float eps, float momentum, int32_t repeats, void* stream) {
    TG_REQUIRE(x && mean && rstd && repeats >= 1 && (y == nullptr || (gamma && beta)), "tg_bn_train_fused: null pointer / repeats < 1");
    TG_REQUIRE(tg_bn_fused_supported(rows, C, groups) && aligned16(x) && (y == nullptr || aligned16(y)),
               "tg_bn_train_fused: unsupported shape C=%d rows=%d groups=%d (see tg_bn_fused_supported) or unaligned pointers", C, rows, groups);
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
    if (bn_vec_ok(C, (long)rpg * C) && aligned16(x)) {
        const long total4 = (long)rpg * C / 4;
        int vb = (int)((total4 + 256 * 8 - 1) / (256 * 8));
        if (vb > 2048) vb = 2048;
        if (vb < 1) vb = 1;
        hipLaunchKernelGGL(bn_stats_vec_kernel, dim3(vb, groups), dim3(256), 0, s, x, (long)rpg, C, ws);
    } else
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
    if (tg_bn_fused_supported(rows, C, 1) && aligned16(dy) && aligned16(x) && aligned16(dx)) {   // one workgroup: reduce, then apply (3 launches -> 1)
        hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(1), dim3(BN_SMALL_THREADS), 0, s, dy, x, dx, rows, C, mean, rstd, gamma, beta, act_slope,
                           dgamma, dbeta);
        return check_launch("tg_bn_backward(small)");
    }
    if (zero_async(ws, sizeof(double) * 2 * (size_t)C, s)) return 1;
    const int rpi = 256 / C;
    int blocks = cdiv(rows, rpi * 16);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (bn_vec_ok(C, (long)rows * C) && aligned16(dy) && aligned16(x) && aligned16(dx)) {
        const long total4 = (long)rows * C / 4;
        int vb = (int)((total4 + 256 * 8 - 1) / (256 * 8));
        if (vb > 2048) vb = 2048;
        if (vb < 1) vb = 1;
        hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel, dim3(vb), dim3(256), 0, s, dy, x, (long)rows, C, mean, rstd, gamma, beta, act_slope, ws);
        hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(vb), dim3(256), 0, s, dy, x, dx, (long)rows, C, mean, rstd, gamma, beta, act_slope, ws,
                           dgamma, dbeta);
        return check_launch("tg_bn_backward(vec)");
    }
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, s, dy, x, rows, C, mean, rstd, gamma, beta, act_slope, ws);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid((long)rows * C, 256, 4)), dim3(256), 0, s, dy, x, dx, (long)rows, C, mean, rstd,
                       gamma, beta, act_slope, ws, dgamma, dbeta);
    return check_launch("tg_bn_backward");
}

// ---- two-launch forms (csrc/norm.hip bn2_*): all `groups` statistics groups of x [groups * rows_per_group][C] in one pair of launches.
extern "C" int32_t tg_bn2_supported(int32_t rows_per_group, int32_t C) {
    return rows_per_group > 0 && bn_vec_ok(C, (long)rows_per_group * C) && 256 % C == 0;
}
static int bn2_parts(long rows_per_group, int C) {
    long p = (rows_per_group * C / 4 + 2047) / 2048;
    return (int)(p < 1 ? 1 : (p > 256 ? 256 : p));
}
constexpr int BN2_INLINE_PARTS = 32;          // up to this many partials every apply workgroup combines them itself; beyond, one reduce launch
extern "C" int64_t tg_bn2_ws_doubles(int32_t rows_per_group, int32_t C, int32_t groups) {
    return tg_bn2_supported(rows_per_group, C) && groups > 0 ? (int64_t)groups * (bn2_parts(rows_per_group, C) + 1) * 2 * C : 0;
}
// partial sums of all groups -> (pointer, count) the apply launch reads
static const double* bn2_combine(double* ws, int P, int C, int groups, int* p_out, hipStream_t s) {
    if (P <= BN2_INLINE_PARTS) { *p_out = P; return ws; }
    double* tot = ws + (long)groups * P * 2 * C;
    hipLaunchKernelGGL(bn2_reduce_kernel, dim3(groups), dim3(256), 0, s, ws, P, C, tot);
    *p_out = 1;
    return tot;
}

extern "C" int tg_bn2_train(const float* x, float* y, int32_t rows_per_group, int32_t C, int32_t groups, double* ws, int64_t ws_doubles, float* mean,
                            float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                            const float* beta, float act_slope, float eps, float momentum, int32_t repeats, void* stream) {
    TG_REQUIRE(x && ws && mean && rstd && groups > 0 && repeats >= 1 && (y == nullptr || (gamma && beta)), "tg_bn2_train: null pointer / bad counts");
    TG_REQUIRE(tg_bn2_supported(rows_per_group, C) && aligned16(x) && (y == nullptr || aligned16(y)), "tg_bn2_train: unsupported shape C=%d rows=%d (tg_bn2_supported) or unaligned pointers", C, rows_per_group);
    TG_REQUIRE(ws_doubles >= tg_bn2_ws_doubles(rows_per_group, C, groups), "tg_bn2_train: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int P = bn2_parts(rows_per_group, C);
    hipLaunchKernelGGL(bn2_partial_kernel, dim3(P, groups), dim3(256), 0, s, x, (const float*)nullptr, (long)rows_per_group, C, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 1.f, ws, deterministic() ? 1 : 0);
    int Pa;
    const double* tot = bn2_combine(ws, P, C, groups, &Pa, s);
    hipLaunchKernelGGL(bn2_fwd_apply_kernel, dim3(P, groups), dim3(256), 0, s, x, y, (long)rows_per_group, C, groups, tot, Pa, mean, rstd, running_mean,
                       running_var, num_batches_tracked, gamma, beta, act_slope, eps, momentum, repeats);
    return check_launch("tg_bn2_train");
}

extern "C" int tg_bn2_backward(const float* dy, const float* x, float* dx, int32_t rows_per_group, int32_t C, int32_t groups, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, float act_slope, double* ws, int64_t ws_doubles,
                               float* dgamma, float* dbeta, void* stream) {
    TG_REQUIRE(dy && x && dx && mean && rstd && gamma && beta && ws && groups > 0, "tg_bn2_backward: null pointer / bad counts");
    TG_REQUIRE(tg_bn2_supported(rows_per_group, C) && aligned16(dy) && aligned16(x) && aligned16(dx), "tg_bn2_backward: unsupported shape C=%d rows=%d or unaligned pointers", C, rows_per_group);
    TG_REQUIRE(ws_doubles >= tg_bn2_ws_doubles(rows_per_group, C, groups), "tg_bn2_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int P = bn2_parts(rows_per_group, C);
    hipLaunchKernelGGL(bn2_partial_kernel, dim3(P, groups), dim3(256), 0, s, x, dy, (long)rows_per_group, C, mean, rstd, gamma, beta, act_slope, ws, deterministic() ? 1 : 0);
    int Pa;
    const double* tot = bn2_combine(ws, P, C, groups, &Pa, s);
    hipLaunchKernelGGL(bn2_bwd_apply_kernel, dim3(P, groups), dim3(256), 0, s, dy, x, dx, (long)rows_per_group, C, groups, tot, Pa, mean, rstd, gamma, beta,
                       act_slope, dgamma, dbeta);
    return check_launch("tg_bn2_backward");
}
