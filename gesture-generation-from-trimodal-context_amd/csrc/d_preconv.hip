// ConvDiscriminator.pre_conv (model/multimodal_context_net.py:214-220), train-mode forward, ONE launch:
//     Conv1d(27, 16, 3) -> BatchNorm1d(16) -> LeakyReLU(True) -> Conv1d(16, 8, 3) -> BatchNorm1d(8) -> LeakyReLU(True) -> Conv1d(8, 8, 3)
// (nn.LeakyReLU(True) sets negative_slope = 1.0: the identity, reference README.md:122).  The generic path runs it as seven launches -- three
// window GEMMs and two two-launch BatchNorms -- on tensors of 0.1-0.5 MB: 48-52 us of which ~35 are launch-to-launch latency.  Here every
// workgroup owns CLIPS clips of one statistics group from the poses to the GRU input: its slices of the three conv outputs live in LDS, and the
// only thing the workgroups exchange are the BatchNorm partial sums (16 + 8 channels x (sum, sum of squares), fp64), through a device-wide
// barrier after conv1 and after conv2.
//
// The barrier follows the cluster GRU's hand-off protocol (gru_cluster_x3.hip): partial sums leave with write-through (sc1) stores and are
// drained (s_waitcnt vmcnt(0)) before the workgroup's relaxed agent-scope arrival; everybody spins (bounded, s_sleep) on the arrival counter and
// then reads the partials with sc1 loads.  No device-scope fence (on an 8-XCD part each one is an L2 write-back: the BatchNorm last-arriver
// experiment, profiles/r3_ae_bn2_last_arriver_rejected.txt).  Counters: ws[0] sticky timeout, ws[1..3] arrivals of barrier 1 / barrier 2 / exit;
// the last workgroup to arrive at barrier 2 re-zeroes counter 1 (nobody spins on it any more), the last to exit counters 2 and 3: the
// workspace is zero ONCE, before its first use, and again after every launch.  All workgroups must be co-resident: grid <= CU count, checked.
//
// Everything the generic backward needs is written exactly where the generic forward writes it: c1 / c2 (conv outputs, the BatchNorms'
// saved inputs), y1 / y2 (their outputs = the next conv's input), mean / rstd per group, the running statistics updated in call order by
// workgroup 0 (momentum 0.1, unbiased variance, num_batches_tracked += groups).  fp32 FMA in a fixed order, statistics in fp64.
#include "common.hpp"

namespace tg {

constexpr int DP_T0 = 34, DP_D = 27, DP_C1 = 16, DP_C2 = 8, DP_C3 = 8, DP_KW = 3;
constexpr int DP_T1 = DP_T0 - 2, DP_T2 = DP_T1 - 2, DP_T3 = DP_T2 - 2;        // 32, 30, 28
constexpr int DP_CLIPS = 4;                                                  // clips per workgroup
constexpr unsigned DP_SPIN_LIMIT = 1u << 24;
typedef __attribute__((address_space(1))) unsigned dp_gu32;

struct DPreconvArgs {
    const float* x;                  // [Bs][34][27]
    const float *w1, *b1, *g1, *be1; // conv1 (16, 27, 3), bias; BN1 gamma, beta
    const float *w2, *b2, *g2, *be2; // conv2 (8, 16, 3); BN2
    const float *w3, *b3;            // conv3 (8, 8, 3)
    float *c1, *y1, *c2, *y2, *c3;   // [Bs][32][16] x 2, [Bs][30][8] x 2, [Bs][28][8]
    float *mean1, *rstd1, *mean2, *rstd2;        // [groups][16], [groups][8]
    float *rm1, *rv1, *rm2, *rv2;    // running statistics (may be NULL together)
    int64_t *nbt1, *nbt2;
    double* part;                    // [2 layers][n workgroups][2][16] partial sums
    unsigned* ws;                    // [4] timeout, arrivals
    int Bs, groups, per;             // per = Bs / groups clips per statistics group
    float eps, momentum;
};

// one device-wide barrier: arrive on ws[idx] (after this workgroup's sc1 stores have drained), spin until all n have
__device__ __forceinline__ bool dp_barrier(unsigned* ws, int idx, unsigned n) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        dp_gu32* c = (dp_gu32*)(ws + idx);
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int good = 1;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > DP_SPIN_LIMIT) { __hip_atomic_store((dp_gu32*)ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// the partial sums cross workgroups (and XCDs, whose L2s are not coherent with each other) inside one launch: relaxed agent-scope atomic
// stores / loads, i.e. write-through and L2-coherent accesses, ordered against the arrival counter by the drain in dp_barrier
__device__ __forceinline__ void dp_store_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double dp_load_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void d_preconv_fwd_kernel(const DPreconvArgs a) {
    __shared__ __attribute__((aligned(16))) float sx[DP_CLIPS][DP_T0][DP_D + 1];          // poses of this workgroup's clips (rows padded to 28 floats, pad = 0)
    __shared__ __attribute__((aligned(16))) float s1[DP_CLIPS][DP_T1][DP_C1];              // conv1 output, then BN1 output
    __shared__ __attribute__((aligned(16))) float s2[DP_CLIPS][DP_T2][DP_C2];              // conv2 output, then BN2 output
    __shared__ __attribute__((aligned(16))) float sw1[DP_C1][DP_KW][DP_D + 1], sw2[DP_C2][DP_KW][DP_C1], sw3[DP_C3][DP_KW][DP_C2];
    __shared__ double red[2][256], tot[2][DP_C1];
    __shared__ float sc[4][DP_C1];                            // mean, rstd, gamma, beta of the current BatchNorm for this workgroup's group
    // Only the workgroups with block id = 0 mod 8 take part -- as observed they all land on ONE XCD, i.e. behind one L2, where a hand-off
    // (coherent store -> arrival -> poll -> coherent load) costs ~0.8 us per hop as in the cluster GRU; spread over the eight XCDs the same
    // protocol meets at the memory side and took ~5 us per hop (25 us per launch).  Speed only: any placement gives the same results.
    if (blockIdx.x & 7) return;
    const int wid = blockIdx.x >> 3;
    const int t = threadIdx.x;
    const int nwg = (gridDim.x >> 3) - 1;                     // the last workgroup only keeps the running statistics (below)
    const int wpg = a.per / DP_CLIPS;                         // workgroups per statistics group
    const int g = wid / wpg, b0 = wid * DP_CLIPS;        // first clip (groups are contiguous clip ranges)

    const bool keeper = (int)wid == nwg;
    // ---- BatchNorm statistics of this workgroup's group (every workgroup of the group computes them: same order, same bits).  The partials
    // of the group's workgroups are read by ALL 256 threads -- thread (channel, slice) takes every (256 / C)-th workgroup, then the slices are
    // summed in order: one round trip of the coherent loads instead of a dependent chain of 2 x 32 of them per thread (48 us per launch).
    auto totals = [&](int layer, int gg, int C) {             // -> red[0][c], red[1][c] for c < C
        const int S = 256 / C, c = t % C, sl = t / C;
        const double* p = a.part + ((long)layer * nwg + (long)gg * wpg) * 2 * DP_C1;
        double s = 0.0, ss = 0.0;
        for (int w = sl; w < wpg; w += S) {
            s += dp_load_sc1(p + (long)w * 2 * DP_C1 + c);
            ss += dp_load_sc1(p + (long)w * 2 * DP_C1 + DP_C1 + c);
        }
        __syncthreads();                                      // (red may still be read by the previous user)
        red[0][t] = s; red[1][t] = ss;
        __syncthreads();
        if (t < C) {
            double ta = 0.0, tb = 0.0;
            for (int q = 0; q < S; ++q) { ta += red[0][q * C + t]; tb += red[1][q * C + t]; }
            tot[0][t] = ta; tot[1][t] = tb;
        }
        __syncthreads();
    };
    auto stats = [&](int layer, int C, int rows_per_clip, float* mean, float* rstd, float* rm, float* rv, int64_t* nbt, const float* gamma,
                     const float* beta) {
        const double n = (double)a.per * rows_per_clip;
        if (!keeper) totals(layer, g, C);
        if (!keeper && t < C) {
            const double m = tot[0][t] / n;
            double var = tot[1][t] / n - m * m;
            if (var < 0.0) var = 0.0;
            const float rs = (float)(1.0 / sqrt(var + (double)a.eps));
            const float mf = (float)m;
            sc[0][t] = mf; sc[1][t] = rs; sc[2][t] = gamma[t]; sc[3][t] = beta[t];
            if (wid % wpg == 0) { mean[g * C + t] = mf; rstd[g * C + t] = rs; }
        }
        if (keeper && rm) {                                   // running statistics in call order (groups = successive forward calls)
            float m_ = 0.f, v_ = 0.f;
            if (t < C) { m_ = rm[t]; v_ = rv[t]; }
            for (int gg = 0; gg < a.groups; ++gg) {
                totals(layer, gg, C);
                if (t < C) {
                    const double m = tot[0][t] / n;
                    double var = tot[1][t] / n - m * m;
                    if (var < 0.0) var = 0.0;
                    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
                    m_ = (1.f - a.momentum) * m_ + a.momentum * (float)m;
                    v_ = (1.f - a.momentum) * v_ + a.momentum * (float)unbiased;
                }
            }
            if (t < C) { rm[t] = m_; rv[t] = v_; }
        }
        if (keeper && t == 0 && nbt) *nbt += a.groups;
        __syncthreads();
    };
    // ---- the keeper workgroup: arrives at both barriers (so that nobody re-zeroes a counter it still reads) and updates the running
    // statistics of BatchNorm 1 while the others run conv2, of BatchNorm 2 while they run conv3 -- in workgroup 0 the 2 x groups extra round
    // trips of coherent loads sat on every workgroup's path to the next barrier
    if (keeper) {
        if (!dp_barrier(a.ws, 1, nwg + 1)) return;
        stats(0, DP_C1, DP_T1, a.mean1, a.rstd1, a.rm1, a.rv1, a.nbt1, a.g1, a.be1);
        if (!dp_barrier(a.ws, 2, nwg + 1)) return;
        if (t == 0) __hip_atomic_store((dp_gu32*)(a.ws + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stats(1, DP_C2, DP_T2, a.mean2, a.rstd2, a.rm2, a.rv2, a.nbt2, a.g2, a.be2);
        if (t == 0) {
            dp_gu32* c = (dp_gu32*)(a.ws + 3);
            if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg) {
                __hip_atomic_store((dp_gu32*)(a.ws + 2), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // ---- stage poses and weights: every global load is issued before the first LDS store (rolled "load, store" loops paid one memory round
    // trip per iteration: 7.5 of the kernel's 25 us)
    {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (long)b0 * DP_T0 * DP_D);       // 4 clips x 918 floats: 16-byte aligned (b0 % 4 == 0)
        constexpr int NX4 = DP_CLIPS * DP_T0 * DP_D / 4;                                         // 918
        f32x4 xr[4];
        float w1r[6], w2r[2], w3r = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = t + 256 * q; xr[q] = x4[i < NX4 ? i : 0]; }
#pragma unroll
        for (int q = 0; q < 6; ++q) { const int i = t + 256 * q; w1r[q] = a.w1[i < DP_C1 * DP_D * DP_KW ? i : 0]; }
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int i = t + 256 * q; w2r[q] = a.w2[i < DP_C2 * DP_C1 * DP_KW ? i : 0]; }
        if (t < DP_C3 * DP_C2 * DP_KW) w3r = a.w3[t];
        if (t < DP_CLIPS * DP_T0) sx[t / DP_T0][t % DP_T0][DP_D] = 0.f;           // the pad column (k = 27 of every tap: multiplied by a zero weight)
        if (t < DP_C1 * DP_KW) sw1[t / DP_KW][t % DP_KW][DP_D] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i4 = t + 256 * q;
            if (i4 < NX4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * i4 + e, c = i % DP_D, r = i / DP_D;
                    sx[r / DP_T0][r % DP_T0][c] = xr[q][e];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int i = t + 256 * q;
            if (i < DP_C1 * DP_D * DP_KW) { const int k = i % DP_KW, ci = (i / DP_KW) % DP_D, co = i / (DP_KW * DP_D); sw1[co][k][ci] = w1r[q]; }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = t + 256 * q;
            if (i < DP_C2 * DP_C1 * DP_KW) { const int k = i % DP_KW, ci = (i / DP_KW) % DP_C1, co = i / (DP_KW * DP_C1); sw2[co][k][ci] = w2r[q]; }
        }
        if (t < DP_C3 * DP_C2 * DP_KW) { const int k = t % DP_KW, ci = (t / DP_KW) % DP_C2, co = t / (DP_KW * DP_C2); sw3[co][k][ci] = w3r; }
    }
    __syncthreads();

    // ---- conv1 on the f32 matrix cores: [128 rows = 4 clips x 32 frames] x [16 channels] x [K = 3 taps x 28 (27 + zero pad)] as 8 row tiles x 21
    // k-steps of v_mfma_f32_16x16x4_f32, two row tiles per wave.  A(row, k) = pose row (clip, frame + tap), channel ci, straight from LDS; B(k, co)
    // = the lane's 21 weights in registers.  (As 648 FMAs per thread on 16-byte LDS fragments the conv was LDS-bandwidth bound: 3.4 us.)
    {
        const int lane = t & 63, wave = t >> 6, r16 = lane & 15, kq = lane >> 4;
        float bw[21];
#pragma unroll
        for (int ks = 0; ks < 21; ++ks) { const int k = 4 * ks + kq; bw[ks] = sw1[r16][k / (DP_D + 1)][k % (DP_D + 1)]; }
        const float bias = a.b1[r16];
        f32x4 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (2 * wave + i) * 16 + r16;         // this lane's A row
            const float* xrow = &sx[row >> 5][row & 31][0];    // taps are consecutive 28-float rows
            acc[i] = f32x4{bias, bias, bias, bias};
#pragma unroll
            for (int ks = 0; ks < 21; ++ks) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xrow[4 * ks + kq], bw[ks], acc[i], 0, 0, 0);
        }
        // accumulator register q of the lane = output row 4 kq + q of the tile, channel r16
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (2 * wave + i) * 16 + 4 * kq + q;
                const float v = acc[i][q];
                s1[row >> 5][row & 31][r16] = v;                  // (c1 goes to global memory after the barrier, with y1: nothing but the
                s += v; ss += (double)v * v;                      //  partial sums is in flight when the workgroup drains its stores)
            }
        red[0][t] = s; red[1][t] = ss;                        // t = 16 (4 wave + kq) + channel
    }
    __syncthreads();
    if (t < DP_C1) {                                          // this workgroup's partial sums, fixed order
        double s = 0.0, ss = 0.0;
        for (int q = 0; q < 16; ++q) { s += red[0][q * 16 + t]; ss += red[1][q * 16 + t]; }
        double* o = a.part + ((long)wid * 2) * DP_C1;
        dp_store_sc1(o + t, s);
        dp_store_sc1(o + DP_C1 + t, ss);
    }
    if (!dp_barrier(a.ws, 1, nwg + 1)) return;

    stats(0, DP_C1, DP_T1, a.mean1, a.rstd1, a.rm1, a.rv1, a.nbt1, a.g1, a.be1);
    for (int i = t; i < DP_CLIPS * DP_T1 * DP_C1; i += 256) {  // y1 = BN1(c1) (LeakyReLU(True) = identity)
        const int c = i & 15;
        float* p = &s1[0][0][0] + i;
        const float cv = *p;
        const float y = (cv - sc[0][c]) * sc[1][c] * sc[2][c] + sc[3][c];        // association of bn_apply_kernel / bn2_fwd_apply_kernel
        *p = y;
        a.c1[(long)b0 * DP_T1 * DP_C1 + i] = cv;
        a.y1[(long)b0 * DP_T1 * DP_C1 + i] = y;
    }
    __syncthreads();

    // ---- conv2: 4 clips x 30 frames x 8 channels = 960 outputs
    {
        double s = 0.0, ss = 0.0;
        for (int i = t; i < DP_CLIPS * DP_T2 * DP_C2; i += 256) {
            const int co = i & 7, r = i >> 3, cl = r / DP_T2, tt = r - cl * DP_T2;
            float acc = a.b2[co];
#pragma unroll
            for (int k = 0; k < DP_KW; ++k)
#pragma unroll
                for (int c4 = 0; c4 < DP_C1 / 4; ++c4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(&s1[cl][tt + k][4 * c4]), w = *reinterpret_cast<const f32x4*>(&sw2[co][k][4 * c4]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_fmaf(xv[q], w[q], acc);
                }
            s2[cl][tt][co] = acc;
            s += acc; ss += (double)acc * acc;
        }
        red[0][t] = s; red[1][t] = ss;                        // thread t always owns channel t & 7 (256 % 8 == 0)
    }
    __syncthreads();
    if (t < DP_C2) {
        double s = 0.0, ss = 0.0;
        for (int q = 0; q < 32; ++q) { s += red[0][q * 8 + t]; ss += red[1][q * 8 + t]; }
        double* o = a.part + (((long)nwg + wid) * 2) * DP_C1;
        dp_store_sc1(o + t, s);
        dp_store_sc1(o + DP_C1 + t, ss);
    }
    if (!dp_barrier(a.ws, 2, nwg + 1)) return;
    if (t == 0) {                                             // everybody has left barrier 1: its counter goes back to zero (once)
        // (the workgroup that observes the full count FIRST is not unique, any one store of zero is right; nobody adds to it again)
        __hip_atomic_store((dp_gu32*)(a.ws + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stats(1, DP_C2, DP_T2, a.mean2, a.rstd2, a.rm2, a.rv2, a.nbt2, a.g2, a.be2);
    for (int i = t; i < DP_CLIPS * DP_T2 * DP_C2; i += 256) {
        const int c = i & 7;
        float* p = &s2[0][0][0] + i;
        const float cv = *p;
        const float y = (cv - sc[0][c]) * sc[1][c] * sc[2][c] + sc[3][c];
        *p = y;
        a.c2[(long)b0 * DP_T2 * DP_C2 + i] = cv;
        a.y2[(long)b0 * DP_T2 * DP_C2 + i] = y;
    }
    __syncthreads();

    // ---- conv3: 4 clips x 28 frames x 8 channels
    for (int i = t; i < DP_CLIPS * DP_T3 * DP_C3; i += 256) {
        const int co = i & 7, r = i >> 3, cl = r / DP_T3, tt = r - cl * DP_T3;
        float acc = a.b3[co];
#pragma unroll
        for (int k = 0; k < DP_KW; ++k)
#pragma unroll
            for (int c4 = 0; c4 < DP_C2 / 4; ++c4) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&s2[cl][tt + k][4 * c4]), w = *reinterpret_cast<const f32x4*>(&sw3[co][k][4 * c4]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_fmaf(xv[q], w[q], acc);
            }
        a.c3[(long)b0 * DP_T3 * DP_C3 + i] = acc;
    }
    // ---- exit count: the last workgroup to leave re-zeroes the counters of barrier 2 and of the exit
    __syncthreads();
    if (t == 0) {
        dp_gu32* c = (dp_gu32*)(a.ws + 3);
        if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg) {      // nwg + 1 leave
            __hip_atomic_store((dp_gu32*)(a.ws + 2), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------- backward
// The same block backwards, ONE launch: from the gradient at the GRU input (d c3) to the parameter gradients of the three convs and two
// BatchNorms and, optionally, the pose gradient.  Same work split (a workgroup owns DP_CLIPS clips), same two device-wide barriers -- here
// for the two sums every train-mode BatchNorm backward needs over its whole statistics group (sum dy, sum dy xhat).  The parameter-gradient
// partials of a workgroup (1 904 numbers) are added to the gradient tensors with float atomics, as the generic path does for these tiny
// conv gradients.  LeakyReLU(True) is the identity: dz = dy.
struct DPreconvBwdArgs {
    const float* dc3;                // [nb][28][8]
    const float *x, *c1, *y1, *c2, *y2;          // the forward's tape, rows of the nb clips
    const float *mean1, *rstd1, *mean2, *rstd2;  // [groups][C] of exactly these clips' groups
    const float *w1, *w2, *w3, *g1, *g2;
    float *dw1, *db1, *dg1, *dbe1, *dw2, *db2, *dg2, *dbe2, *dw3, *db3;      // all NULL: no parameter gradients
    float* dposes;                   // NULL, or [nb][34][27]
    int dposes_accumulate;
    double* part;                    // [2 layers][n workgroups][2][16]
    unsigned* ws;
    int nb, groups, per;
};

__global__ __launch_bounds__(256) void d_preconv_bwd_kernel(const DPreconvBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float sx[DP_CLIPS][DP_T0][DP_D + 1];
    __shared__ __attribute__((aligned(16))) float sc1[DP_CLIPS][DP_T1][DP_C1];      // c1 -> xhat1, later d c1
    __shared__ __attribute__((aligned(16))) float sy1[DP_CLIPS][DP_T1][DP_C1];      // y1
    __shared__ __attribute__((aligned(16))) float sd1[DP_CLIPS][DP_T1][DP_C1];      // d y1
    __shared__ __attribute__((aligned(16))) float sc2[DP_CLIPS][DP_T2][DP_C2];      // c2 -> xhat2, later d c2
    __shared__ __attribute__((aligned(16))) float sy2[DP_CLIPS][DP_T2][DP_C2];
    __shared__ __attribute__((aligned(16))) float sd2[DP_CLIPS][DP_T2][DP_C2];      // d y2
    __shared__ __attribute__((aligned(16))) float sd3[DP_CLIPS][DP_T3][DP_C3];      // d c3
    __shared__ float sw1[DP_C1][DP_D][DP_KW], sw2[DP_C2][DP_C1][DP_KW], sw3[DP_C3][DP_C2][DP_KW];      // reference layout (Co, Ci, kw)
    __shared__ double red[2][256], tot[2][DP_C1];
    __shared__ float sc[3][DP_C1];                            // m1 = sum dy / n, m2 = sum dy xhat / n, gamma rstd
    const int t = threadIdx.x, wid = blockIdx.x, nwg = gridDim.x;
    const int wpg = a.per / DP_CLIPS, g = wid / wpg, b0 = wid * DP_CLIPS;
    const bool pg = a.dw1 != nullptr;

    // ---- stage the tape (loads first, then LDS stores)
    {
        const long o3 = (long)b0 * DP_T3 * DP_C3, o2 = (long)b0 * DP_T2 * DP_C2, o1 = (long)b0 * DP_T1 * DP_C1;
        constexpr int N3 = DP_CLIPS * DP_T3 * DP_C3 / 4, N2 = DP_CLIPS * DP_T2 * DP_C2 / 4, N1 = DP_CLIPS * DP_T1 * DP_C1 / 4, NX4 = DP_CLIPS * DP_T0 * DP_D / 4;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 v3 = t < N3 ? reinterpret_cast<const f32x4*>(a.dc3 + o3)[t] : z4;
        const f32x4 vc2 = t < N2 ? reinterpret_cast<const f32x4*>(a.c2 + o2)[t] : z4, vy2 = t < N2 ? reinterpret_cast<const f32x4*>(a.y2 + o2)[t] : z4;
        f32x4 vc1[2], vy1[2], xr[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) { vc1[q] = reinterpret_cast<const f32x4*>(a.c1 + o1)[t + 256 * q]; vy1[q] = reinterpret_cast<const f32x4*>(a.y1 + o1)[t + 256 * q]; }
        static_assert(N1 == 512, "two 16-byte pieces of c1 / y1 per thread");
        const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (long)b0 * DP_T0 * DP_D);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = t + 256 * q; xr[q] = x4[i < NX4 ? i : 0]; }
        float w1r[6], w2r[2], w3r = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) { const int i = t + 256 * q; w1r[q] = a.w1[i < DP_C1 * DP_D * DP_KW ? i : 0]; }
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int i = t + 256 * q; w2r[q] = a.w2[i < DP_C2 * DP_C1 * DP_KW ? i : 0]; }
        if (t < DP_C3 * DP_C2 * DP_KW) w3r = a.w3[t];
        if (t < N3) reinterpret_cast<f32x4*>(&sd3[0][0][0])[t] = v3;
        if (t < N2) {
            // xhat2 = (c2 - mean) rstd, channel = (4 t) % 8 .. + 3
            const int c0 = (4 * t) & 7;
            f32x4 xh;
#pragma unroll
            for (int q = 0; q < 4; ++q) xh[q] = (vc2[q] - a.mean2[g * DP_C2 + c0 + q]) * a.rstd2[g * DP_C2 + c0 + q];
            reinterpret_cast<f32x4*>(&sc2[0][0][0])[t] = xh;
            reinterpret_cast<f32x4*>(&sy2[0][0][0])[t] = vy2;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c0 = (4 * (t + 256 * q)) & 15;
            f32x4 xh;
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[e] = (vc1[q][e] - a.mean1[g * DP_C1 + c0 + e]) * a.rstd1[g * DP_C1 + c0 + e];
            reinterpret_cast<f32x4*>(&sc1[0][0][0])[t + 256 * q] = xh;
            reinterpret_cast<f32x4*>(&sy1[0][0][0])[t + 256 * q] = vy1[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i4 = t + 256 * q;
            if (i4 < NX4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int i = 4 * i4 + e, c = i % DP_D, r = i / DP_D; sx[r / DP_T0][r % DP_T0][c] = xr[q][e]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) { const int i = t + 256 * q; if (i < DP_C1 * DP_D * DP_KW) (&sw1[0][0][0])[i] = w1r[q]; }
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int i = t + 256 * q; if (i < DP_C2 * DP_C1 * DP_KW) (&sw2[0][0][0])[i] = w2r[q]; }
        if (t < DP_C3 * DP_C2 * DP_KW) (&sw3[0][0][0])[t] = w3r;
    }
    __syncthreads();

    // ---- conv3: weight / bias gradient partials (straight to the gradient tensors), input gradient d y2
    if (pg) {
        if (t < DP_C3 * DP_C2 * DP_KW) {                      // dW3[co][ci][k] = sum_{clip, frame} d c3[frame][co] y2[frame + k][ci]
            const int k = t % DP_KW, ci = (t / DP_KW) % DP_C2, co = t / (DP_KW * DP_C2);
            float acc = 0.f;
            for (int cl = 0; cl < DP_CLIPS; ++cl)
                for (int f = 0; f < DP_T3; ++f) acc = __builtin_fmaf(sd3[cl][f][co], sy2[cl][f + k][ci], acc);
            atomicAdd(a.dw3 + t, acc);
        } else if (t < DP_C3 * DP_C2 * DP_KW + DP_C3) {
            const int co = t - DP_C3 * DP_C2 * DP_KW;
            float acc = 0.f;
            for (int cl = 0; cl < DP_CLIPS; ++cl)
                for (int f = 0; f < DP_T3; ++f) acc += sd3[cl][f][co];
            atomicAdd(a.db3 + co, acc);
        }
    }
    {
        double s = 0.0, ss = 0.0;                             // BatchNorm 2 backward sums of this thread's channel (t & 7)
        for (int i = t; i < DP_CLIPS * DP_T2 * DP_C2; i += 256) {
            const int ci = i & 7, r = i >> 3, cl = r / DP_T2, f = r - cl * DP_T2;
            float acc = 0.f;                                  // d y2[f][ci] = sum_{k, co} d c3[f - k][co] W3[co][ci][k]
#pragma unroll
            for (int k = 0; k < DP_KW; ++k) {
                const int fo = f - k;
                if (fo >= 0 && fo < DP_T3) {
#pragma unroll
                    for (int co = 0; co < DP_C3; ++co) acc = __builtin_fmaf(sd3[cl][fo][co], sw3[co][ci][k], acc);
                }
            }
            sd2[cl][f][ci] = acc;
            s += acc; ss += (double)acc * sc2[cl][f][ci];
        }
        red[0][t] = s; red[1][t] = ss;
    }
    __syncthreads();
    if (t < DP_C2) {
        double s = 0.0, ss = 0.0;
        for (int q = 0; q < 32; ++q) { s += red[0][q * 8 + t]; ss += red[1][q * 8 + t]; }
        double* o = a.part + (((long)nwg + wid) * 2) * DP_C1;
        dp_store_sc1(o + t, s);
        dp_store_sc1(o + DP_C1 + t, ss);
    }
    if (!dp_barrier(a.ws, 1, nwg)) return;

    auto totals = [&](int layer, int C) {                     // sums of this workgroup's group -> tot[0 / 1][c]
        const int S = 256 / C, c = t % C, sl = t / C;
        const double* p = a.part + ((long)layer * nwg + (long)g * wpg) * 2 * DP_C1;
        double s = 0.0, ss = 0.0;
        for (int w = sl; w < wpg; w += S) { s += dp_load_sc1(p + (long)w * 2 * DP_C1 + c); ss += dp_load_sc1(p + (long)w * 2 * DP_C1 + DP_C1 + c); }
        __syncthreads();
        red[0][t] = s; red[1][t] = ss;
        __syncthreads();
        if (t < C) {
            double ta = 0.0, tb = 0.0;
            for (int q = 0; q < S; ++q) { ta += red[0][q * C + t]; tb += red[1][q * C + t]; }
            tot[0][t] = ta; tot[1][t] = tb;
        }
        __syncthreads();
    };
    // ---- BatchNorm 2 backward: d c2 = gamma rstd (d y2 - mean(d y2) - xhat mean(d y2 xhat)); d gamma += sum d y2 xhat, d beta += sum d y2
    totals(1, DP_C2);
    if (t < DP_C2) {
        const double n = (double)a.per * DP_T2;
        sc[0][t] = (float)(tot[0][t] / n); sc[1][t] = (float)(tot[1][t] / n); sc[2][t] = a.g2[t] * a.rstd2[g * DP_C2 + t];
        if (pg && wid % wpg == 0) { atomicAdd(a.dg2 + t, (float)tot[1][t]); atomicAdd(a.dbe2 + t, (float)tot[0][t]); }
    }
    __syncthreads();
    for (int i = t; i < DP_CLIPS * DP_T2 * DP_C2; i += 256) {
        const int c = i & 7;
        float* xh = &sc2[0][0][0] + i;
        *xh = sc[2][c] * ((&sd2[0][0][0])[i] - sc[0][c] - *xh * sc[1][c]);                      // sc2 now holds d c2
    }
    __syncthreads();

    // ---- conv2: gradients, d y1, BatchNorm 1 sums
    if (pg) {
        for (int e = t; e < DP_C2 * DP_C1 * DP_KW + DP_C2; e += 256) {
            if (e < DP_C2 * DP_C1 * DP_KW) {
                const int k = e % DP_KW, ci = (e / DP_KW) % DP_C1, co = e / (DP_KW * DP_C1);
                float acc = 0.f;
                for (int cl = 0; cl < DP_CLIPS; ++cl)
                    for (int f = 0; f < DP_T2; ++f) acc = __builtin_fmaf(sc2[cl][f][co], sy1[cl][f + k][ci], acc);
                atomicAdd(a.dw2 + e, acc);
            } else {
                const int co = e - DP_C2 * DP_C1 * DP_KW;
                float acc = 0.f;
                for (int cl = 0; cl < DP_CLIPS; ++cl)
                    for (int f = 0; f < DP_T2; ++f) acc += sc2[cl][f][co];
                atomicAdd(a.db2 + co, acc);
            }
        }
    }
    {
        double s = 0.0, ss = 0.0;                             // channel t & 15
        for (int i = t; i < DP_CLIPS * DP_T1 * DP_C1; i += 256) {
            const int ci = i & 15, r = i >> 4, cl = r >> 5, f = r & 31;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < DP_KW; ++k) {
                const int fo = f - k;
                if (fo >= 0 && fo < DP_T2) {
#pragma unroll
                    for (int co = 0; co < DP_C2; ++co) acc = __builtin_fmaf(sc2[cl][fo][co], sw2[co][ci][k], acc);
                }
            }
            sd1[cl][f][ci] = acc;
            s += acc; ss += (double)acc * sc1[cl][f][ci];
        }
        red[0][t] = s; red[1][t] = ss;
    }
    __syncthreads();
    if (t < DP_C1) {
        double s = 0.0, ss = 0.0;
        for (int q = 0; q < 16; ++q) { s += red[0][q * 16 + t]; ss += red[1][q * 16 + t]; }
        double* o = a.part + ((long)wid * 2) * DP_C1;
        dp_store_sc1(o + t, s);
        dp_store_sc1(o + DP_C1 + t, ss);
    }
    if (!dp_barrier(a.ws, 2, nwg)) return;
    if (t == 0) __hip_atomic_store((dp_gu32*)(a.ws + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    totals(0, DP_C1);
    if (t < DP_C1) {
        const double n = (double)a.per * DP_T1;
        sc[0][t] = (float)(tot[0][t] / n); sc[1][t] = (float)(tot[1][t] / n); sc[2][t] = a.g1[t] * a.rstd1[g * DP_C1 + t];
        if (pg && wid % wpg == 0) { atomicAdd(a.dg1 + t, (float)tot[1][t]); atomicAdd(a.dbe1 + t, (float)tot[0][t]); }
    }
    __syncthreads();
    for (int i = t; i < DP_CLIPS * DP_T1 * DP_C1; i += 256) {
        const int c = i & 15;
        float* xh = &sc1[0][0][0] + i;
        *xh = sc[2][c] * ((&sd1[0][0][0])[i] - sc[0][c] - *xh * sc[1][c]);                      // sc1 now holds d c1
    }
    __syncthreads();

    // ---- conv1: gradients, pose gradient
    if (pg) {
        for (int e = t; e < DP_C1 * DP_D * DP_KW + DP_C1; e += 256) {
            if (e < DP_C1 * DP_D * DP_KW) {
                const int k = e % DP_KW, ci = (e / DP_KW) % DP_D, co = e / (DP_KW * DP_D);
                float acc = 0.f;
                for (int cl = 0; cl < DP_CLIPS; ++cl)
                    for (int f = 0; f < DP_T1; ++f) acc = __builtin_fmaf(sc1[cl][f][co], sx[cl][f + k][ci], acc);
                atomicAdd(a.dw1 + e, acc);
            } else {
                const int co = e - DP_C1 * DP_D * DP_KW;
                float acc = 0.f;
                for (int cl = 0; cl < DP_CLIPS; ++cl)
                    for (int f = 0; f < DP_T1; ++f) acc += sc1[cl][f][co];
                atomicAdd(a.db1 + co, acc);
            }
        }
    }
    if (a.dposes) {
        float* dp = a.dposes + (long)b0 * DP_T0 * DP_D;
        for (int i = t; i < DP_CLIPS * DP_T0 * DP_D; i += 256) {
            const int ci = i % DP_D, r = i / DP_D, cl = r / DP_T0, f = r - cl * DP_T0;
            float acc = a.dposes_accumulate ? dp[i] : 0.f;
#pragma unroll
            for (int k = 0; k < DP_KW; ++k) {
                const int fo = f - k;
                if (fo >= 0 && fo < DP_T1) {
#pragma unroll
                    for (int co = 0; co < DP_C1; ++co) acc = __builtin_fmaf(sc1[cl][fo][co], sw1[co][ci][k], acc);
                }
            }
            dp[i] = acc;
        }
    }
    __syncthreads();
    if (t == 0) {
        dp_gu32* c = (dp_gu32*)(a.ws + 3);
        if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1) {
            __hip_atomic_store((dp_gu32*)(a.ws + 2), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace tg

using namespace tg;

extern "C" int32_t tg_d_preconv_fwd_supported(int32_t Bs, int32_t groups) {
    if (Bs <= 0 || groups <= 0 || Bs % groups != 0 || (Bs / groups) % DP_CLIPS != 0) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return Bs / DP_CLIPS + 1 <= cus / 8 * 3 ? 1 : 0;           // every workgroup co-resident on ONE XCD (three per CU by LDS: 47 KB each)
}
extern "C" int64_t tg_d_preconv_ws_bytes(int32_t Bs) { return 16 + (int64_t)2 * (Bs / DP_CLIPS) * 2 * DP_C1 * 8; }

extern "C" int tg_d_preconv_fwd(const float* poses, const float* w1, const float* b1, const float* gamma1, const float* beta1, const float* w2,
                                const float* b2, const float* gamma2, const float* beta2, const float* w3, const float* b3, float* c1, float* y1,
                                float* c2, float* y2, float* c3, float* mean1, float* rstd1, float* mean2, float* rstd2, float* running_mean1,
                                float* running_var1, int64_t* nbt1, float* running_mean2, float* running_var2, int64_t* nbt2, void* ws,
                                int64_t ws_bytes, int32_t Bs, int32_t groups, float eps, float momentum, void* stream) {
    TG_REQUIRE(poses && w1 && b1 && gamma1 && beta1 && w2 && b2 && gamma2 && beta2 && w3 && b3 && c1 && y1 && c2 && y2 && c3 && mean1 && rstd1 &&
                   mean2 && rstd2 && ws, "tg_d_preconv_fwd: null pointer");
    TG_REQUIRE(tg_d_preconv_fwd_supported(Bs, groups), "tg_d_preconv_fwd: Bs=%d groups=%d unsupported (clips per group a multiple of %d, Bs / %d "
               "workgroups co-resident)", Bs, groups, DP_CLIPS, DP_CLIPS);
    TG_REQUIRE((running_mean1 == nullptr) == (running_var1 == nullptr) && (running_mean2 == nullptr) == (running_var2 == nullptr),
               "tg_d_preconv_fwd: running mean and variance go together");
    TG_REQUIRE(ws_bytes >= tg_d_preconv_ws_bytes(Bs) && aligned16(ws), "tg_d_preconv_fwd: workspace too small or unaligned");
    DPreconvArgs a;
    a.x = poses; a.w1 = w1; a.b1 = b1; a.g1 = gamma1; a.be1 = beta1; a.w2 = w2; a.b2 = b2; a.g2 = gamma2; a.be2 = beta2; a.w3 = w3; a.b3 = b3;
    a.c1 = c1; a.y1 = y1; a.c2 = c2; a.y2 = y2; a.c3 = c3; a.mean1 = mean1; a.rstd1 = rstd1; a.mean2 = mean2; a.rstd2 = rstd2;
    a.rm1 = running_mean1; a.rv1 = running_var1; a.rm2 = running_mean2; a.rv2 = running_var2; a.nbt1 = nbt1; a.nbt2 = nbt2;
    a.ws = reinterpret_cast<unsigned*>(ws);
    a.part = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + 16);
    a.Bs = Bs; a.groups = groups; a.per = Bs / groups; a.eps = eps; a.momentum = momentum;
    hipLaunchKernelGGL(d_preconv_fwd_kernel, dim3(8 * (Bs / DP_CLIPS + 1)), dim3(256), 0, (hipStream_t)stream, a);      // + the keeper; 1 id in 8 works
    return check_launch("tg_d_preconv_fwd");
}

extern "C" int tg_d_preconv_bwd(const float* dc3, const float* poses, const float* c1, const float* y1, const float* c2, const float* y2,
                                const float* mean1, const float* rstd1, const float* mean2, const float* rstd2, const float* w1, const float* w2,
                                const float* w3, const float* gamma1, const float* gamma2, float* dw1, float* db1, float* dgamma1, float* dbeta1,
                                float* dw2, float* db2, float* dgamma2, float* dbeta2, float* dw3, float* db3, float* dposes,
                                int32_t dposes_accumulate, void* ws, int64_t ws_bytes, int32_t nb, int32_t groups, void* stream) {
    TG_REQUIRE(dc3 && poses && c1 && y1 && c2 && y2 && mean1 && rstd1 && mean2 && rstd2 && w1 && w2 && w3 && gamma1 && gamma2 && ws,
               "tg_d_preconv_bwd: null pointer");
    const bool all = dw1 && db1 && dgamma1 && dbeta1 && dw2 && db2 && dgamma2 && dbeta2 && dw3 && db3;
    const bool none = !dw1 && !db1 && !dgamma1 && !dbeta1 && !dw2 && !db2 && !dgamma2 && !dbeta2 && !dw3 && !db3;
    TG_REQUIRE(all || none, "tg_d_preconv_bwd: the ten parameter gradients go together (all or none)");
    TG_REQUIRE(all || dposes, "tg_d_preconv_bwd: nothing to compute");
    TG_REQUIRE(tg_d_preconv_fwd_supported(nb, groups), "tg_d_preconv_bwd: nb=%d groups=%d unsupported", nb, groups);
    TG_REQUIRE(ws_bytes >= tg_d_preconv_ws_bytes(nb) && aligned16(ws) && aligned16(dc3) && aligned16(poses) && aligned16(c1) && aligned16(y1) &&
                   aligned16(c2) && aligned16(y2), "tg_d_preconv_bwd: workspace too small or unaligned operands");
    DPreconvBwdArgs a;
    a.dc3 = dc3; a.x = poses; a.c1 = c1; a.y1 = y1; a.c2 = c2; a.y2 = y2; a.mean1 = mean1; a.rstd1 = rstd1; a.mean2 = mean2; a.rstd2 = rstd2;
    a.w1 = w1; a.w2 = w2; a.w3 = w3; a.g1 = gamma1; a.g2 = gamma2;
    a.dw1 = dw1; a.db1 = db1; a.dg1 = dgamma1; a.dbe1 = dbeta1; a.dw2 = dw2; a.db2 = db2; a.dg2 = dgamma2; a.dbe2 = dbeta2; a.dw3 = dw3; a.db3 = db3;
    a.dposes = dposes; a.dposes_accumulate = dposes_accumulate;
    a.ws = reinterpret_cast<unsigned*>(ws);
    a.part = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + 16);
    a.nb = nb; a.groups = groups; a.per = nb / groups;
    hipLaunchKernelGGL(d_preconv_bwd_kernel, dim3(nb / DP_CLIPS), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("tg_d_preconv_bwd");
}
