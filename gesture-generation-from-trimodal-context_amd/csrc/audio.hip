// WavEncoder front end (model/multimodal_context_net.py:13-15): Conv1d(1, 16, 15, stride 5, padding 1600) -> BatchNorm1d(16) -> LeakyReLU(0.3)
// on raw audio, forward and backward, WITHOUT the 16-channel pre-BatchNorm tensor ever touching HBM.
//
// At B = 128 that tensor is 128 x 7891 x 16 floats = 65 MB; the generic path (window GEMM, bn_stats, bn_apply; backward: bn_bwd_reduce,
// bn_bwd_apply, weight-gradient GEMM with a 1 M-row reduction) moved it through HBM nine times per iteration (~330 us).  Here:
//   forward   wav_stats_kernel   reads the audio (18 MB), recomputes the convolution on the f32 matrix cores and keeps only per-channel sums
//             wav_apply_kernel   recomputes it again (same instruction sequence: bit-identical values) and writes act(BN(conv)) once,
//                                plus one sign bit per element (the LeakyReLU gate for the backward pass)
//   backward  wav_bwd_kernel     reads d act once and reduces G^T [A | 1] on the matrix cores, G = d act * gate, A = the audio windows.
//             Every gradient of the block is a closed form in that 16 x 16 result and the forward's sums (wav_bwd_finalize_kernel):
//               d beta  = sum G                      d gamma = sum G xhat = rstd (b sum G + sum_k w_k (G^T A)_k - mean sum G)
//               d W     = gamma rstd (G^T A - mean(G) sum A - mean(G xhat) xhat^T A),   xhat^T A = rstd (X^T A - mean sum A)
//             (dx = gamma rstd (G - mean G - xhat mean(G xhat)) contracted with the windows; sums in fp64, deterministic order).
//
// Tile = 16 consecutive frames of one clip x 16 channels = one v_mfma_f32_16x16x4_f32 accumulator.  MFMA row m is frame 4 (m & 3) + (m >> 2)
// of the tile, so accumulator register i of lane l is frame 4 i + (l >> 4), channel l & 15: the 64 lanes of register i cover 64 CONSECUTIVE
// floats of the channel-last output -- coalesced 256-byte stores / loads, and a wave ballot of register i is the gate word of those 64 elements.
// The bias rides in the product (tap 15 of the window operand is 1, of the weight operand the bias).
#include "common.hpp"

namespace tg {

constexpr int WV_CO = 16, WV_KW = 15;
constexpr unsigned WV_RSRC3 = 0x00020000u;      // buffer descriptor word 3: raw buffer, 32-bit data format, bounds check on the byte offset
constexpr int WV_XA = 256;                 // fstat layout: [X^T A | sum X] 16 x 16, then sum A [16]
constexpr int WV_FSTAT = WV_XA + 16;
constexpr int WV_PART = WV_FSTAT + 32;     // per-workgroup partial of the statistics pass: fstat + sum x [16] + sum x^2 [16]
constexpr int WV_STATS_WGS = 256, WV_STATS_THREADS = 1024;
constexpr int WV_BWD_WGS = 256, WV_BWD_THREADS = 1024;

struct WavGeom {
    const float* audio;
    long a_stride;        // floats between clips
    int B, L, T1, TT;     // clips, samples per clip, output frames, tiles per clip = ceil(T1 / 16)
    int stride, pad;
};

// window sample `idx` of a clip (zero padding outside [0, L)): unconditional load from a clamped address, zeroed when used
__device__ __forceinline__ float wav_sample(const float* clip, int idx, int L, bool ok) {
    const int ci = idx < 0 ? 0 : (idx >= L ? L - 1 : idx);
    const float v = clip[ci];
    return (ok && idx == ci) ? v : 0.f;
}

// weight operand of the forward product: lane (n = channel, k) holds w[n][4 j + k] for j = 0..3, the bias at tap 15
__device__ __forceinline__ void wav_weight_frag(const float* __restrict__ w, const float* __restrict__ bias, float (&wf)[4]) {
    const int n = threadIdx.x & 15, k = (threadIdx.x >> 4) & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tap = 4 * j + k;
        wf[j] = tap < WV_KW ? w[n * WV_KW + tap] : bias[n];
    }
}

// A tile is INTERIOR when its 16 frames exist and every sample of their windows lies inside the clip (no padding): true for ~92 % of the
// tiles (padding 1600 of 36267 samples per side).  Interior tiles take the FAST loaders: one lane-constant offset plus immediates instead of
// ~15 VALU instructions of clamping and predicates per load -- the kernels were bound by exactly that address arithmetic (wav_stats: 39 us,
// of which 7 us matrix work).  `interior` is wave-uniform.
__device__ __forceinline__ bool wav_interior(const WavGeom& g, int t_first, int t_last) {
    return t_last < g.T1 && g.stride * t_first - g.pad >= 0 && g.stride * t_last - g.pad + WV_KW - 1 < g.L;
}

// window operand of the forward product for tile (clip, frames 16 tt ...): lane (m, k) holds taps 4 j + k of frame 4 (m & 3) + (m >> 2)
template <bool FAST>
__device__ __forceinline__ void wav_conv_load(const WavGeom& g, const float* clip, int tt, float (&a)[4]) {
    const int l = threadIdx.x & 63;
    const int m = l & 15, k = l >> 4;
    const int t = 16 * tt + 4 * (m & 3) + (m >> 2);
    const int base = g.stride * t - g.pad + k;
    if constexpr (FAST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = clip[base + 4 * j];        // tap 15 of the last frame may read one sample past the window: still inside the clip or its successor
        a[3] = k == 3 ? 1.f : a[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tap = 4 * j + k;
            a[j] = wav_sample(clip, base + 4 * j, g.L, tap < WV_KW);
            if (tap == WV_KW) a[j] = 1.f;
        }
    }
}
// conv output of the tile: register i = frame 16 tt + 4 i + (lane >> 4), channel lane & 15
__device__ __forceinline__ f32x4 wav_conv_mma(const float (&a)[4], const float (&wf)[4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wf[j], acc, 0, 0, 0);
    return acc;
}

// window operand of the reductions over frames (X^T A, G^T A): lane (n = tap, k) holds sample tap n of frame t = t0 + k; tap 15 is 1
template <bool FAST>
__device__ __forceinline__ float wav_window_at(const WavGeom& g, const float* clip, int t) {
    const int n = threadIdx.x & 15;
    if constexpr (FAST) {
        const float v = clip[g.stride * t - g.pad + n];
        return n == WV_KW ? 1.f : v;
    } else {
        const bool row_ok = t < g.T1;
        const float v = wav_sample(clip, g.stride * t - g.pad + n, g.L, row_ok && n < WV_KW);
        return (n == WV_KW && row_ok) ? 1.f : v;
    }
}
__device__ __forceinline__ float wav_window_frag(const WavGeom& g, const float* clip, int tt, int i) {
    return wav_window_at<false>(g, clip, 16 * tt + 4 * i + (int)((threadIdx.x & 63) >> 4));
}

// ---- forward pass 1: statistics ----------------------------------------------------------------------------------------------
// partial[wg][WV_PART] doubles: X^T [A | 1] (16 x 16, row = channel), sum A (16, entry 15 = frame count), sum x (16), sum x^2 (16)
__global__ __launch_bounds__(WV_STATS_THREADS) void wav_stats_kernel(WavGeom g, const float* __restrict__ w, const float* __restrict__ bias,
                                                                      double* __restrict__ partial) {
    __shared__ double red[WV_STATS_THREADS / 64][WV_PART];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int nwaves = WV_STATS_THREADS / 64;
    float wf[4];
    wav_weight_frag(w, bias, wf);
    f32x4 xa = {0.f, 0.f, 0.f, 0.f};
    double sx = 0.0, sxx = 0.0, sa = 0.0;
    const int tiles = g.B * g.TT;
    const int tstep = gridDim.x * nwaves;
    // software pipeline: the next tile's eight sample requests are in flight while this tile's eight MFMAs run (a wave walks ~15 tiles; without
    // it every tile cost a full memory round trip: 42 us for 3 us of matrix work)
    float a_cur[4], af_cur[4], a_nxt[4], af_nxt[4];
    int tile = blockIdx.x * nwaves + wave;
    auto load_tile = [&](int tl, float (&a)[4], float (&af)[4]) {
        const int tc = tl < tiles ? tl : tiles - 1;
        const int b = tc / g.TT, tt = tc - b * g.TT;
        const float* clip = g.audio + (long)b * g.a_stride;
        if (wav_interior(g, 16 * tt, 16 * tt + 15) && (b + 1 < g.B || g.stride * (16 * tt + 15) - g.pad + WV_KW < g.L)) {
            wav_conv_load<true>(g, clip, tt, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = wav_window_at<true>(g, clip, 16 * tt + 4 * i + (l >> 4));
        } else {
            wav_conv_load<false>(g, clip, tt, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = wav_window_at<false>(g, clip, 16 * tt + 4 * i + (l >> 4));
        }
    };
    load_tile(tile, a_cur, af_cur);
    for (; tile < tiles; tile += tstep) {
        load_tile(tile + tstep, a_nxt, af_nxt);
        const int b = tile / g.TT, tt = tile - b * g.TT;
        const f32x4 x = wav_conv_mma(a_cur, wf);
        // per-tile partial sums of four values in fp32, accumulated over the wave's tiles in fp64 (three fp64 operations per tile instead of twelve)
        float s4 = 0.f, q4 = 0.f, a4 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = 16 * tt + 4 * i + (l >> 4) < g.T1;
            float xv = x[i];
            xv = ok ? xv : 0.f;
            s4 += xv;
            q4 = __builtin_fmaf(xv, xv, q4);
            a4 += ok ? af_cur[i] : 0.f;
            xa = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, af_cur[i], xa, 0, 0, 0);
        }
        sx += (double)s4;
        sxx += (double)q4;
        sa += (double)a4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a_cur[i] = a_nxt[i]; af_cur[i] = af_nxt[i]; }
    }
    // lanes l, l ^ 16, l ^ 32 hold the same channel (sx, sxx) / the same tap (sa): fixed-order butterfly
    sx += __shfl_xor(sx, 16); sx += __shfl_xor(sx, 32);
    sxx += __shfl_xor(sxx, 16); sxx += __shfl_xor(sxx, 32);
    sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(4 * (l >> 4) + i) * 16 + (l & 15)] = (double)xa[i];      // row = channel, column = tap
    if (l < 16) {
        red[wave][WV_XA + l] = sa;
        red[wave][WV_FSTAT + l] = sx;
        red[wave][WV_FSTAT + 16 + l] = sxx;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < WV_PART; e += WV_STATS_THREADS) {
        double s = 0.0;
        for (int q = 0; q < nwaves; ++q) s += red[q][e];
        partial[(long)blockIdx.x * WV_PART + e] = s;
    }
}

// Combine the partials in fixed order: workgroup c < 16 owns channel c (row c of X^T [A | 1], sum x, sum x^2 -> mean / rstd / running statistics
// like bn_finalize_kernel), workgroup 16 owns sum A.  1024 threads = 32 outputs x 32 slices of the partial list, so every thread has its
// (<= 8) loads in flight together instead of one thread walking 256 dependent ones.
__global__ __launch_bounds__(1024) void wav_stats_finalize_kernel(const double* __restrict__ partial, int nparts, long rows, float* __restrict__ mean,
                                                                   float* __restrict__ rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                                                   int64_t* __restrict__ nbt, double* __restrict__ fstat, float eps, float momentum,
                                                                   int repeats) {
    __shared__ double sl[32][33];
    const int c = blockIdx.x;
    const int o = threadIdx.x & 31, slice = threadIdx.x >> 5;
    // output o of this workgroup -> element of a partial record (or -1)
    int e = -1;
    if (c < WV_CO) e = o < 16 ? c * 16 + o : (o == 16 ? WV_FSTAT + c : (o == 17 ? WV_FSTAT + 16 + c : -1));
    else if (o < 16) e = WV_XA + o;
    double s = 0.0;
    if (e >= 0)
        for (int q = slice; q < nparts; q += 32) s += partial[(long)q * WV_PART + e];
    sl[slice][o] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int q = 0; q < 32; ++q) t += sl[q][o];
        sl[0][o] = t;
        if (e >= 0 && e < WV_FSTAT && fstat) fstat[e] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0 && c < WV_CO) {
        const double n = (double)rows;
        const double m = sl[0][16] / n;
        double var = sl[0][17] / n - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
        float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
        for (int q = 0; q < repeats; ++q) {
            rm = (1.f - momentum) * rm + momentum * (float)m;
            rv = (1.f - momentum) * rv + momentum * (float)unbiased;
        }
        if (rmean) rmean[c] = rm;
        if (rvar) rvar[c] = rv;
        if (c == 0 && nbt) *nbt += (int64_t)repeats;
    }
}

// ---- forward pass 2: y = act((conv - mean) rstd gamma + beta), gate bits ---------------------------------------------------------
__global__ __launch_bounds__(256) void wav_apply_kernel(WavGeom g, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                         float* __restrict__ y, unsigned long long* __restrict__ gate) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int c = l & 15;
    float wf[4];
    wav_weight_frag(w, bias, wf);
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    const int tiles = g.B * g.TT;
    const int tstep = gridDim.x * 4;
    __amdgpu_buffer_rsrc_t gate_rsrc = __builtin_amdgcn_make_buffer_rsrc(gate ? (void*)gate : (void*)y, 0, gate ? tiles * 32 : 0, WV_RSRC3);
    float a_cur[4], a_nxt[4];
    int tile = blockIdx.x * 4 + wave;
    auto load_tile = [&](int tl, float (&a)[4]) {
        const int tc = tl < tiles ? tl : tiles - 1;
        const int b = tc / g.TT, tt = tc - b * g.TT;
        const float* clip = g.audio + (long)b * g.a_stride;
        if (wav_interior(g, 16 * tt, 16 * tt + 15) && (b + 1 < g.B || g.stride * (16 * tt + 15) - g.pad + WV_KW < g.L)) wav_conv_load<true>(g, clip, tt, a);
        else wav_conv_load<false>(g, clip, tt, a);
    };
    load_tile(tile, a_cur);
    for (; tile < tiles; tile += tstep) {
        load_tile(tile + tstep, a_nxt);                        // next tile's samples in flight behind this tile's stores
        const int b = tile / g.TT, tt = tile - b * g.TT;
        const f32x4 x = wav_conv_mma(a_cur, wf);
        // stores through buffer descriptors whose range ends with the clip (frames past T1 of the last tile) / is empty (no gate wanted):
        // the hardware drops out-of-range lanes, so no lane-predicated store makes the vector-memory count dynamic (the compiler would then
        // drain everything before the prefetched samples are used)
        __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(y + (long)b * g.T1 * WV_CO, 0, g.T1 * WV_CO * 4, WV_RSRC3);
        unsigned long long mine = 0ull;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = 16 * tt + 4 * i + (l >> 4) < g.T1;
            const float xh = (x[i] - mu) * rs;                 // association of bn_apply_kernel
            const float z = xh * ga + be;
            const bool pos = z >= 0.f;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pos ? z : z * slope), y_rsrc, (unsigned)((16 * tt * WV_CO + 64 * i + l) * 4), 0, 0);
            const unsigned long long word = __ballot(ok && pos);
            mine = l == i ? word : mine;
        }
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{(unsigned)mine, (unsigned)(mine >> 32)}, gate_rsrc, l < 4 ? (unsigned)(l * 8) : 0x80000000u, tile * 32, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a_cur[i] = a_nxt[i];
    }
}

// ---- backward: G^T [A | 1] ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WV_BWD_THREADS) void wav_bwd_kernel(WavGeom g, const float* __restrict__ dact, const unsigned long long* __restrict__ gate,
                                                                  float slope, double* __restrict__ partial) {
    __shared__ double red[WV_BWD_THREADS / 64][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int nwaves = WV_BWD_THREADS / 64;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int tiles = g.B * g.TT;
    const long last = ((long)g.B * g.T1) * WV_CO - 1;
    for (int tile = blockIdx.x * nwaves + wave; tile < tiles; tile += gridDim.x * nwaves) {
        const int b = tile / g.TT, tt = tile - b * g.TT;
        const float* clip = g.audio + (long)b * g.a_stride;
        const long e0 = ((long)b * g.T1 + 16 * tt) * WV_CO + l;
        float gv[4], af[4];
        unsigned long long word[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                          // loads first: eight requests in flight per wave and tile
            const long e = e0 + 64 * i;
            gv[i] = dact[e > last ? last : e];
            word[i] = gate[(long)tile * 4 + i];
            af[i] = wav_window_frag(g, clip, tt, i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = 16 * tt + 4 * i + (l >> 4) < g.T1;
            float v = gv[i] * (((word[i] >> l) & 1ull) ? 1.f : slope);
            v = ok ? v : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, af[i], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(4 * (l >> 4) + i) * 16 + (l & 15)] = (double)acc[i];     // row = channel, column = tap (15: sum G)
    __syncthreads();
    for (int e = threadIdx.x; e < 256; e += WV_BWD_THREADS) {
        double s = 0.0;
        for (int q = 0; q < nwaves; ++q) s += red[q][e];
        partial[(long)blockIdx.x * 256 + e] = s;
    }
}

// ---- backward, second form: the input gradient of the NEXT conv (Conv1d(16, 32, 15, stride 6)) computed on the fly ------------------------
// d act [b, t, ci] = sum_j sum_co d c2[b, q - j, co] W2[co, ci, p + 6 j] with t = 6 q + p.  A tile is 16 frames of ONE phase p (q = 16 qb ..),
// so one weight slice serves the whole tile: the tile is a [16 x 32 J] x [32 J x 16] product on the f32 matrix cores whose accumulator
// already has the lane layout of the G operand above -- d act (65 MB written by a GEMM, read back here) never exists.  A wave keeps one phase
// (its 8 J weight fragments stay in registers); operand rows come in as 16-byte loads feeding four MFMAs each (k-permuted feed: MFMA v of
// group u covers co = 16 u + 4 (l >> 4) + v in both operands).
constexpr int WV_C2 = 32, WV_S2 = 6, WV_KW2 = 15;

template <int NJ>
__device__ __forceinline__ void wav_bwd_fused_phase(const WavGeom& g, const float* __restrict__ dc2, int T2, const float* __restrict__ W2,
                                                    const unsigned long long* __restrict__ gate, float slope, int p, int wave_in_phase,
                                                    int waves_in_phase, f32x4& acc) {
    const int l = threadIdx.x & 63;
    const int n = l & 15, kq = l >> 4;
    float wf[NJ][2][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) wf[j][u][v] = W2[(16 * u + 4 * kq + v) * (WV_CO * WV_KW2) + n * WV_KW2 + p + WV_S2 * j];
    const int Qp = (g.T1 - 1 - p) / WV_S2 + 1;            // frames of this phase per clip (T1 > p)
    const int QB = (Qp + 15) / 16;
    const int tiles = g.B * QB;
    const int wpc = 4 * g.TT;
    const int fr_a = 4 * (n & 3) + (n >> 2);             // frame of MFMA row n (= l & 15) in the first product
    // software pipeline over the wave's ~15 tiles: the next tile's 14 requests (6 x 16-byte operand rows, 4 gate words, 4 samples) are in
    // flight while this tile's 28 MFMAs run
    struct Tile {
        f32x4 a[NJ][2];
        unsigned long long word[4];
        float af[4];
        bool fast;          // wave-uniform: interior tile
    };
    auto load_tile = [&](int tl, Tile& T_) {
        const int tc = tl < tiles ? tl : tiles - 1;
        const int b = tc / QB, qb = tc - b * QB;
        const float* clip = g.audio + (long)b * g.a_stride;
        const float* dcb = dc2 + (long)b * T2 * WV_C2 + 4 * kq;
        const unsigned long long* gb = gate + (long)b * wpc;
        const int t_first = WV_S2 * 16 * qb + p, t_last = t_first + WV_S2 * 15;
        if (16 * qb - (NJ - 1) >= 0 && 16 * qb + 15 < T2 && wav_interior(g, t_first, t_last) &&
            (b + 1 < g.B || g.stride * t_last - g.pad + WV_KW < g.L)) {      // interior tile (wave-uniform): no clamps, no predicates (lane n = 15 reads one sample past the window)
            const float* d0 = dcb + (16 * qb + fr_a) * WV_C2;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) T_.a[j][u] = *reinterpret_cast<const f32x4*>(d0 - j * WV_C2 + 16 * u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = WV_S2 * (16 * qb + 4 * i + kq) + p;
                T_.word[i] = gb[t >> 2];
                const float sv = clip[g.stride * t - g.pad + n];
                T_.af[i] = n == WV_KW ? 1.f : sv;
            }
            T_.fast = true;
            return;
        }
        T_.fast = false;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = 16 * qb + fr_a - j;
            const int rc = row < 0 ? 0 : (row >= T2 ? T2 - 1 : row);
#pragma unroll
            for (int u = 0; u < 2; ++u) T_.a[j][u] = *reinterpret_cast<const f32x4*>(dcb + (long)rc * WV_C2 + 16 * u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // register i: frame 16 qb + 4 i + kq of phase p
            const int t = WV_S2 * (16 * qb + 4 * i + kq) + p;
            const bool ok = t < g.T1;
            const int tc1 = ok ? t : g.T1 - 1;
            T_.word[i] = gb[tc1 >> 2];
            const float sv = wav_sample(clip, g.stride * tc1 - g.pad + n, g.L, ok && n < WV_KW);
            T_.af[i] = (n == WV_KW && ok) ? 1.f : sv;
        }
    };
    Tile cur, nxt;
    int tile = wave_in_phase;
    load_tile(tile, cur);
    for (; tile < tiles; tile += waves_in_phase) {
        load_tile(tile + waves_in_phase, nxt);
        const int b = tile / QB, qb = tile - b * QB;
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        if (cur.fast) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 av = cur.a[j][u];
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], wf[j][u][0], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], wf[j][u][1], d1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], wf[j][u][2], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], wf[j][u][3], d1, 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sh = 16 * ((WV_S2 * (4 * i + kq) + p + WV_S2 * 16 * qb) & 3) + n;
                const bool pos = (cur.word[i] >> sh) & 1ull;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32((d0[i] + d1[i]) * (pos ? 1.f : slope), cur.af[i], acc, 0, 0, 0);
            }
            cur = nxt;
            continue;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = 16 * qb + fr_a - j;
            const bool rok = row >= 0 && row < T2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 av = cur.a[j][u];
                av = rok ? av : f32x4{0.f, 0.f, 0.f, 0.f};
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], wf[j][u][0], d0, 0, 0, 0);      // two accumulators: the dependent-accumulator
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], wf[j][u][1], d1, 0, 0, 0);      // latency exceeds the issue interval
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], wf[j][u][2], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], wf[j][u][3], d1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = WV_S2 * (16 * qb + 4 * i + kq) + p;
            const bool ok = t < g.T1;
            const int tc1 = ok ? t : g.T1 - 1;
            const bool pos = (cur.word[i] >> (16 * (tc1 & 3) + n)) & 1ull;
            float v = (d0[i] + d1[i]) * (pos ? 1.f : slope);
            v = ok ? v : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, cur.af[i], acc, 0, 0, 0);
        }
        cur = nxt;
    }
}

__global__ __launch_bounds__(WV_BWD_THREADS) void wav_bwd_fused_kernel(WavGeom g, const float* __restrict__ dc2, int T2, const float* __restrict__ W2,
                                                                        const unsigned long long* __restrict__ gate, float slope,
                                                                        double* __restrict__ partial) {
    __shared__ double red[WV_BWD_THREADS / 64][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int nwaves = WV_BWD_THREADS / 64;
    const int gw = blockIdx.x * nwaves + wave, nw = gridDim.x * nwaves;
    const int p = gw % WV_S2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (p < g.T1) {
        const int wip = gw / WV_S2, wn = (nw - p + WV_S2 - 1) / WV_S2;
        if (p + 2 * WV_S2 < WV_KW2) wav_bwd_fused_phase<3>(g, dc2, T2, W2, gate, slope, p, wip, wn, acc);
        else wav_bwd_fused_phase<2>(g, dc2, T2, W2, gate, slope, p, wip, wn, acc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(4 * (l >> 4) + i) * 16 + (l & 15)] = (double)acc[i];
    __syncthreads();
    for (int e = threadIdx.x; e < 256; e += WV_BWD_THREADS) {
        double s = 0.0;
        for (int q = 0; q < nwaves; ++q) s += red[q][e];
        partial[(long)blockIdx.x * 256 + e] = s;
    }
}

// one workgroup per channel c: row c of G^T [A | 1] combined in fixed order (16 outputs x 64 slices of the partial list), then the closed forms
__global__ __launch_bounds__(1024) void wav_bwd_finalize_kernel(const double* __restrict__ partial, int nparts, long rows, const double* __restrict__ fstat,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma, float* __restrict__ dW, float* __restrict__ dbias,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sl[64][17];
    __shared__ double G[16];
    __shared__ double m12[2];
    const int c = blockIdx.x;
    const int o = threadIdx.x & 15, slice = threadIdx.x >> 4;
    double s = 0.0;
    for (int q = slice; q < nparts; q += 64) s += partial[(long)q * 256 + c * 16 + o];
    sl[slice][o] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0.0;
        for (int q = 0; q < 64; ++q) t += sl[q][o];
        G[o] = t;
    }
    __syncthreads();
    const double n = (double)rows;
    const double mu = (double)mean[c], rs = (double)rstd[c], ga = (double)gamma[c];
    if (threadIdx.x == 0) {
        const double sg = G[15];
        double gx = (double)bias[c] * sg;                                        // sum G x
        for (int k = 0; k < WV_KW; ++k) gx += (double)w[c * WV_KW + k] * G[k];
        const double gxh = rs * (gx - mu * sg);                                  // sum G xhat
        m12[0] = sg / n;
        m12[1] = gxh / n;
        if (dbeta) dbeta[c] += (float)sg;
        if (dgamma) dgamma[c] += (float)gxh;
        if (dbias) {
            const double sxh = rs * (fstat[c * 16 + 15] - n * mu);               // sum xhat (zero up to the rounding of mean)
            dbias[c] += (float)(ga * rs * (sg - n * (sg / n) - (gxh / n) * sxh));
        }
    }
    __syncthreads();
    if (threadIdx.x < WV_KW && dW) {
        const int k = threadIdx.x;
        const double sa = fstat[WV_XA + k];
        const double xha = rs * (fstat[c * 16 + k] - mu * sa);                   // sum xhat a_k
        dW[c * WV_KW + k] += (float)(ga * rs * (G[k] - m12[0] * sa - m12[1] * xha));
    }
}

// ---- weight gradient of the second conv, Conv1d(16, 32, 15, stride 6) -------------------------------------------------------------------
//   dW2[co][ci][k] = sum_{b, q} dc2[b, q, co] act[b, 6 q + k, ci]        db2[co] = sum dc2[b, q, co]
// a [32 x 240] result reduced over B * T2 = 168 k rows.  The generic weight-gradient GEMM tiles the OUTPUT (64 x 64, N = 32 fills half a
// tile) and re-reads the 65 MB activation through its tap windows per tile column: 75 us + 24 us of partial combine.  Here a wave keeps the WHOLE
// result -- 2 (co tiles) x 16 (taps; tap 15 is a ones column that yields db2) accumulators of v_mfma_f32_16x16x4_f32 = 128 registers -- and
// walks its share of the rows four at a time: 2 loads of dc2 and 15 of the activation rows feed 32 MFMAs; each activation row is read once
// from HBM (re-used by the 2-3 taps that touch it out of L1).  Per-wave results meet in LDS, per-workgroup partials in a fixed-order fp64
// combine (wav_conv2_wgrad_reduce_kernel).
constexpr int WV_W2_WGS = 256, WV_W2_THREADS = 768;        // 12 waves = 3 per SIMD (151 registers): what hides the groups' load latency
constexpr int WV_W2_OUT = WV_C2 * 16 * WV_CO;          // 32 co x 16 taps (15 + ones) x 16 ci = 8192 partial sums

__global__ __launch_bounds__(WV_W2_THREADS) void wav_conv2_wgrad_kernel(const float* __restrict__ dc2, const float* __restrict__ act, int B, int T1, int T2,
                                                                        float* __restrict__ partial) {
    __shared__ float red[WV_W2_OUT];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    const int r16 = l & 15, kq = l >> 4;
    constexpr int nwaves = WV_W2_THREADS / 64;
    for (int i = threadIdx.x; i < WV_W2_OUT; i += WV_W2_THREADS) red[i] = 0.f;
    __syncthreads();
    f32x4 acc[16][2];
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc[k][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int GQ = (T2 + 3) / 4;                          // groups of four output frames per clip
    const int groups = B * GQ;
    const int gstep = gridDim.x * nwaves;
    struct Grp { float a[2]; float x[15]; };
    auto load_group = [&](int gi, Grp& G_) {
        const int gc = gi < groups ? gi : groups - 1;
        const int b = gc / GQ, q0 = (gc - b * GQ) * 4;
        const int q = q0 + kq;
        const bool ok = gi < groups && q < T2;
        const int qc = q < T2 ? q : T2 - 1;
        const float* dp = dc2 + ((long)b * T2 + qc) * WV_C2 + r16;
        const float a0 = dp[0], a1 = dp[16];
        G_.a[0] = ok ? a0 : 0.f;                          // rows past the clip (or past the wave's range) contribute nothing
        G_.a[1] = ok ? a1 : 0.f;
        const float* xp = act + ((long)b * T1 + WV_S2 * qc) * WV_CO + r16;      // 6 qc + 14 <= T1 - 1 by the conv geometry
#pragma unroll
        for (int k = 0; k < WV_KW2; ++k) G_.x[k] = xp[k * WV_CO];
    };
    Grp cur, nxt;
    const float one = r16 == 0 ? 1.f : 0.f;               // tap 15: column 0 of a ones matrix -> sum of dc2 (the bias gradient)
    int gi = blockIdx.x * nwaves + wave;
    load_group(gi, cur);
    for (; gi < groups; gi += gstep) {
        load_group(gi + gstep, nxt);                      // next group's 17 requests in flight behind this group's 32 MFMAs
#pragma unroll
        for (int k = 0; k < WV_KW2; ++k) {
            acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[0], cur.x[k], acc[k][0], 0, 0, 0);
            acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[1], cur.x[k], acc[k][1], 0, 0, 0);
        }
        acc[15][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[0], one, acc[15][0], 0, 0, 0);
        acc[15][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[1], one, acc[15][1], 0, 0, 0);
        cur = nxt;
    }
    // (a ring of three register sets -- two groups of requests in flight -- measured SLOWER: 87 us against 51)
    // accumulator (tap k, co tile ct): lane holds rows co = 16 ct + 4 kq + i, column ci = r16 -> red[co][k][ci]
    for (int w = 0; w < nwaves; ++w) {                    // the waves add their results one after the other: fixed order, no atomics
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[((16 * ct + 4 * kq + i) * 16 + k) * WV_CO + r16] += acc[k][ct][i];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < WV_W2_OUT; i += WV_W2_THREADS) partial[(long)blockIdx.x * WV_W2_OUT + i] = red[i];
}

// fixed-order fp64 combine of the workgroup partials: four threads per entry (each a quarter of the partial list, loads in flight together)
__global__ __launch_bounds__(256) void wav_conv2_wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ dW2,
                                                                     float* __restrict__ db2) {
    const int e = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    double s = 0.0;
#pragma unroll 8
    for (int q = sub; q < nparts; q += 4) s += (double)partial[(long)q * WV_W2_OUT + e];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (sub == 0) {
        const int co = e / 256, k = (e >> 4) & 15, ci = e & 15;
        if (k < WV_KW2) { if (dW2) dW2[(co * WV_CO + ci) * WV_KW2 + k] += (float)s; }
        else if (ci == 0 && db2) db2[co] += (float)s;
    }
}

static int wav_geom(WavGeom& g, const char* who, const float* audio, int64_t audio_stride, int32_t B, int32_t L, int32_t stride, int32_t pad, int32_t T1) {
    TG_REQUIRE(audio && B > 0 && L > 0 && stride > 0 && pad >= 0 && T1 > 0 && audio_stride >= L, "%s: bad audio geometry (B=%d L=%d stride=%d pad=%d T1=%d)", who, B, L,
               stride, pad, T1);
    TG_REQUIRE((long)T1 == ((long)L + 2L * pad - WV_KW) / stride + 1 && (long)L + 2L * pad >= WV_KW, "%s: T1=%d is not the conv length of L=%d", who, T1, L);
    TG_REQUIRE((long)B * ((T1 + 15) / 16) < (1L << 25) && (long)stride * T1 + WV_KW < (1L << 24), "%s: problem too large for 32-bit tile / byte offsets", who);
    g.audio = audio; g.a_stride = audio_stride; g.B = B; g.L = L; g.T1 = T1; g.TT = (T1 + 15) / 16; g.stride = stride; g.pad = pad;
    return 0;
}

}  // namespace tg

using namespace tg;

extern "C" int64_t tg_wav_front_ws_doubles(void) { return (int64_t)WV_STATS_WGS * WV_PART; }          // >= WV_BWD_WGS * 256
extern "C" int64_t tg_wav_front_fstat_doubles(void) { return WV_FSTAT; }
extern "C" int64_t tg_wav_front_gate_words(int32_t B, int32_t T1) { return B > 0 && T1 > 0 ? (int64_t)B * ((T1 + 15) / 16) * 4 : 0; }

extern "C" int tg_wav_front_stats(const float* audio, int64_t audio_stride, int32_t B, int32_t L, const float* w, const float* bias, int32_t stride,
                                  int32_t pad, int32_t T1, double* ws, int64_t ws_doubles, float* mean, float* rstd, float* running_mean,
                                  float* running_var, int64_t* num_batches_tracked, double* fstat, float eps, float momentum, int32_t repeats,
                                  void* stream) {
    WavGeom g;
    if (int rc = wav_geom(g, "tg_wav_front_stats", audio, audio_stride, B, L, stride, pad, T1)) return rc;
    TG_REQUIRE(w && bias && ws && mean && rstd && repeats >= 1, "tg_wav_front_stats: null pointer / repeats < 1");
    TG_REQUIRE(ws_doubles >= tg_wav_front_ws_doubles(), "tg_wav_front_stats: workspace of %ld doubles, need %ld", (long)ws_doubles, (long)tg_wav_front_ws_doubles());
    hipStream_t s = (hipStream_t)stream;
    const int tiles = g.B * g.TT;
    const int wgs = tiles < WV_STATS_WGS * (WV_STATS_THREADS / 64) ? cdiv(tiles, WV_STATS_THREADS / 64) : WV_STATS_WGS;
    hipLaunchKernelGGL(wav_stats_kernel, dim3(wgs), dim3(WV_STATS_THREADS), 0, s, g, w, bias, ws);
    hipLaunchKernelGGL(wav_stats_finalize_kernel, dim3(WV_CO + 1), dim3(1024), 0, s, ws, wgs, (long)B * T1, mean, rstd, running_mean, running_var, num_batches_tracked,
                       fstat, eps, momentum, repeats);
    return check_launch("tg_wav_front_stats");
}

extern "C" int tg_wav_front_apply(const float* audio, int64_t audio_stride, int32_t B, int32_t L, const float* w, const float* bias, int32_t stride,
                                  int32_t pad, int32_t T1, const float* mean, const float* rstd, const float* gamma, const float* beta, float act_slope,
                                  float* y, uint64_t* gate, void* stream) {
    WavGeom g;
    if (int rc = wav_geom(g, "tg_wav_front_apply", audio, audio_stride, B, L, stride, pad, T1)) return rc;
    TG_REQUIRE(w && bias && mean && rstd && gamma && beta && y, "tg_wav_front_apply: null pointer");
    const int tiles = g.B * g.TT;
    int wgs = cdiv(tiles, 4 * 8);                       // ~8 tiles per wave
    if (wgs > 2048) wgs = 2048;
    hipLaunchKernelGGL(wav_apply_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, g, w, bias, mean, rstd, gamma, beta, act_slope, y,
                       reinterpret_cast<unsigned long long*>(gate));
    return check_launch("tg_wav_front_apply");
}

extern "C" int tg_wav_front_backward(const float* dact, const uint64_t* gate, const float* audio, int64_t audio_stride, int32_t B, int32_t L, const float* w,
                                     const float* bias, int32_t stride, int32_t pad, int32_t T1, const float* mean, const float* rstd, const float* gamma,
                                     const double* fstat, float act_slope, double* ws, int64_t ws_doubles, float* dW, float* dbias, float* dgamma,
                                     float* dbeta, void* stream) {
    WavGeom g;
    if (int rc = wav_geom(g, "tg_wav_front_backward", audio, audio_stride, B, L, stride, pad, T1)) return rc;
    TG_REQUIRE(dact && gate && w && bias && mean && rstd && gamma && fstat && ws, "tg_wav_front_backward: null pointer");
    TG_REQUIRE(ws_doubles >= tg_wav_front_ws_doubles(), "tg_wav_front_backward: workspace of %ld doubles, need %ld", (long)ws_doubles, (long)tg_wav_front_ws_doubles());
    hipStream_t s = (hipStream_t)stream;
    const int tiles = g.B * g.TT;
    const int wgs = tiles < WV_BWD_WGS * (WV_BWD_THREADS / 64) ? cdiv(tiles, WV_BWD_THREADS / 64) : WV_BWD_WGS;
    hipLaunchKernelGGL(wav_bwd_kernel, dim3(wgs), dim3(WV_BWD_THREADS), 0, s, g, dact, reinterpret_cast<const unsigned long long*>(gate), act_slope, ws);
    hipLaunchKernelGGL(wav_bwd_finalize_kernel, dim3(WV_CO), dim3(1024), 0, s, ws, wgs, (long)B * T1, fstat, w, bias, mean, rstd, gamma, dW, dbias, dgamma, dbeta);
    return check_launch("tg_wav_front_backward");
}

extern "C" int tg_wav_front_backward_fused(const float* dc2, int32_t T2, const float* w2, const uint64_t* gate, const float* audio, int64_t audio_stride,
                                           int32_t B, int32_t L, const float* w, const float* bias, int32_t stride, int32_t pad, int32_t T1,
                                           const float* mean, const float* rstd, const float* gamma, const double* fstat, float act_slope, double* ws,
                                           int64_t ws_doubles, float* dW, float* dbias, float* dgamma, float* dbeta, void* stream) {
    WavGeom g;
    if (int rc = wav_geom(g, "tg_wav_front_backward_fused", audio, audio_stride, B, L, stride, pad, T1)) return rc;
    TG_REQUIRE(dc2 && w2 && gate && w && bias && mean && rstd && gamma && fstat && ws, "tg_wav_front_backward_fused: null pointer");
    TG_REQUIRE(T2 == (T1 - WV_KW2) / WV_S2 + 1 && T1 >= WV_KW2 && aligned16(dc2), "tg_wav_front_backward_fused: T2=%d is not the length of Conv1d(16, 32, 15, stride 6) over %d frames, or dc2 unaligned", T2, T1);
    TG_REQUIRE(ws_doubles >= tg_wav_front_ws_doubles(), "tg_wav_front_backward_fused: workspace of %ld doubles, need %ld", (long)ws_doubles, (long)tg_wav_front_ws_doubles());
    hipStream_t s = (hipStream_t)stream;
    const long tiles = (long)B * WV_S2 * ((T1 / WV_S2 + 16) / 16);
    const int wgs = tiles < (long)WV_BWD_WGS * (WV_BWD_THREADS / 64) ? cdiv(tiles, WV_BWD_THREADS / 64) : WV_BWD_WGS;
    hipLaunchKernelGGL(wav_bwd_fused_kernel, dim3(wgs), dim3(WV_BWD_THREADS), 0, s, g, dc2, T2, w2, reinterpret_cast<const unsigned long long*>(gate), act_slope, ws);
    hipLaunchKernelGGL(wav_bwd_finalize_kernel, dim3(WV_CO), dim3(1024), 0, s, ws, wgs, (long)B * T1, fstat, w, bias, mean, rstd, gamma, dW, dbias, dgamma, dbeta);
    return check_launch("tg_wav_front_backward_fused");
}

extern "C" int64_t tg_wav_conv2_wgrad_ws_floats(void) { return (int64_t)WV_W2_WGS * WV_W2_OUT; }

extern "C" int tg_wav_conv2_wgrad(const float* dc2, const float* act, int32_t B, int32_t T1, int32_t T2, float* ws, int64_t ws_floats, float* dW2,
                                  float* db2, void* stream) {
    TG_REQUIRE(dc2 && act && ws && B > 0 && T1 >= WV_KW2 && T2 == (T1 - WV_KW2) / WV_S2 + 1, "tg_wav_conv2_wgrad: bad arguments (T2=%d is not the conv length of T1=%d)", T2, T1);
    TG_REQUIRE((long)B * T1 * WV_CO < (1L << 31) && ws_floats >= tg_wav_conv2_wgrad_ws_floats(), "tg_wav_conv2_wgrad: problem too large / workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long groups = (long)B * ((T2 + 3) / 4);
    const int wgs = groups < (long)WV_W2_WGS * (WV_W2_THREADS / 64) ? cdiv(groups, WV_W2_THREADS / 64) : WV_W2_WGS;
    hipLaunchKernelGGL(wav_conv2_wgrad_kernel, dim3(wgs), dim3(WV_W2_THREADS), 0, s, dc2, act, B, T1, T2, ws);
    hipLaunchKernelGGL(wav_conv2_wgrad_reduce_kernel, dim3(WV_W2_OUT / 64), dim3(256), 0, s, ws, wgs, dW2, db2);
    return check_launch("tg_wav_conv2_wgrad");
}
