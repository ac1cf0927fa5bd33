// Lab: fp32-accurate NT GEMM on the bf16 matrix cores by operand splitting (standalone, no torch).
//   C[M][N] = A[M][K] . B[N][K]^T, all fp32 in memory.
// Every fp32 operand x is split EXACTLY into three bf16 terms x = hi + mid + lo (8 + 8 + 8 significand bits, by truncation), while
// its 32-deep K slab is staged into LDS; the product keeps the six terms whose weight is >= 2^-16 of the leading one
//   hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a b|)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16: six bf16 MFMAs (16x the f32 MFMA rate each) per fp32 MFMA-equivalent, i.e. up to
// 2.67x the f32 matrix rate at fp32-level accuracy (products of bf16 pairs are exact in fp32; the only roundings are the fp32
// accumulations, as in the f32 MFMA path).  SPLITS = 2 keeps hi/mid only (3 MFMAs, error 2^-16) for comparison, SPLITS = 1 is plain bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/gemm_split_lab.bin tools/gemm_split_lab.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int xcd_chunked_id(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

// upper halves of two fp32 words -> one dword of two bf16 (element 0 in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned lo_elem, unsigned hi_elem) {
    return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u);
}

template <int SPLITS>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&out)[SPLITS]) {
    unsigned t[4], u[4], s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // NOTE: __builtin_bit_cast(unsigned, v[i]) on an ext_vector ELEMENT is miscompiled by hipcc 7.2 (every i reads element 0):
        // copy the element to a scalar first
        const float xf = v[i];
        const unsigned x = __float_as_uint(xf);
        t[i] = x & 0xffff0000u;                                   // hi (truncated): exact prefix of the significand
        if (SPLITS > 1) {
            const float r = xf - __uint_as_float(t[i]);           // exact
            u[i] = __float_as_uint(r) & 0xffff0000u;
            if (SPLITS > 2) s[i] = __float_as_uint(r - __uint_as_float(u[i]));   // <= 8 bits left: exact in bf16
        }
    }
    out[0] = u32x2{pack_hi16(t[0], t[1]), pack_hi16(t[2], t[3])};
    if (SPLITS > 1) out[1] = u32x2{pack_hi16(u[0], u[1]), pack_hi16(u[2], u[3])};
    if (SPLITS > 2) out[2] = u32x2{pack_hi16(s[0], s[1]), pack_hi16(s[2], s[3])};
}

constexpr int LDH = 40;      // bf16 per LDS row: 32 + 8 pad (80 bytes)

// Workgroup tile (32*TM) x (32*TN), 4 waves 2 x 2, wave tile (16*TM) x (16*TN) of 16x16x32 MFMAs; K slab 32.
template <int TM, int TN, int SPLITS, int DB>
__global__ __launch_bounds__(256) void gemm_split_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         int M, int N, int K, int n_nt) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int NPA = BM * 8 / 256, NPB = BN * 8 / 256;       // f32x4 pieces per thread per slab (8 pieces per 32-deep row)
    constexpr int NB = DB ? 2 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 as[NB][SPLITS][BM][LDH];
    __shared__ __attribute__((aligned(16))) __bf16 bs[NB][SPLITS][BN][LDH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    // staging: 8 consecutive lanes cover one 128-byte row piece (32 fp32)
    const int sp = 4 * (t & 7), sr0 = t >> 3;
    const float* ap[NPA]; const float* bp[NPB]; bool aok[NPA], bok[NPB];
#pragma unroll
    for (int q = 0; q < NPA; ++q) { const int m = m0 + sr0 + 32 * q; aok[q] = m < M; ap[q] = A + (long)(aok[q] ? m : 0) * K + sp; }
#pragma unroll
    for (int q = 0; q < NPB; ++q) { const int n = n0 + sr0 + 32 * q; bok[q] = n < N; bp[q] = B + (long)(bok[q] ? n : 0) * K + sp; }
    f32x4 ga[NPA], gb[NPB];
    auto fetch = [&](int k0) {
        const bool inb = k0 + sp < K;
        const int kc = inb ? k0 : 0;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NPA; ++q) { const f32x4 v = *reinterpret_cast<const f32x4*>(ap[q] + kc); ga[q] = (aok[q] && inb) ? v : z; }
#pragma unroll
        for (int q = 0; q < NPB; ++q) { const f32x4 v = *reinterpret_cast<const f32x4*>(bp[q] + kc); gb[q] = (bok[q] && inb) ? v : z; }
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += 32) {
        if (!DB && k0 > 0) __syncthreads();                     // single buffer: everybody is done reading the previous slab
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            u32x2 o[SPLITS];
            split4<SPLITS>(ga[q], o);
#pragma unroll
            for (int s = 0; s < SPLITS; ++s) *reinterpret_cast<u32x2*>(&as[buf][s][sr0 + 32 * q][sp]) = o[s];
        }
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            u32x2 o[SPLITS];
            split4<SPLITS>(gb[q], o);
#pragma unroll
            for (int s = 0; s < SPLITS; ++s) *reinterpret_cast<u32x2*>(&bs[buf][s][sr0 + 32 * q][sp]) = o[s];
        }
        __syncthreads();
        if (k0 + 32 < K) fetch(k0 + 32);
        bf16x8 fa[SPLITS][TM], fb[SPLITS][TN];
#pragma unroll
        for (int s = 0; s < SPLITS; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(&as[buf][s][wm * (16 * TM) + i * 16 + r16][8 * kq]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(&bs[buf][s][wn * (16 * TN) + j * 16 + r16][8 * kq]);
        }
        // smallest terms first: the fp32 accumulator then sees them before the large ones of this slab
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 c = acc[i][j];
                if (SPLITS > 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][i], fb[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[2][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[1][j], c, 0, 0, 0);
                }
                if (SPLITS > 1) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1][j], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0][j], c, 0, 0, 0);
                acc[i][j] = c;
            }
        if (DB) buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * (16 * TM) + i * 16 + kq * 4 + q;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + r16;
                if (col < N) C[(long)row * N + col] = acc[i][j][q];
            }
        }
}

// PF2: as gemm_split_kernel<.., DB = 1> but with TWO register sets: the global loads of slab s+2 are issued while slab s is
// multiplied, so a slab's load latency is covered by two iterations (matters when only one workgroup fits a CU / small grids).
template <int TM, int TN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_split_pf2_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                                    int M, int N, int K, int n_nt) {
    constexpr int BM = 32 * TM, BN = 32 * TN, NT = 64 * WAVES;
    constexpr int NPA = (BM * 8 + NT - 1) / NT, NPB = (BN * 8 + NT - 1) / NT;
    constexpr int WMW = WAVES / 2;                               // waves along M (2 along N)
    constexpr int WTM = 2 * TM / WMW;                            // 16-row MFMA tiles per wave along M
    __shared__ __attribute__((aligned(16))) __bf16 lds[2][3][BM + BN][LDH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    const int sp = 4 * (t & 7), sr0 = t >> 3;
    constexpr int RS = NT / 8;                                   // rows staged per pass
    const float* ap[NPA]; const float* bp[NPB]; bool aok[NPA], bok[NPB];
#pragma unroll
    for (int q = 0; q < NPA; ++q) { const int m = m0 + sr0 + RS * q; aok[q] = m < M && sr0 + RS * q < BM; ap[q] = A + (long)(aok[q] ? m : 0) * K + sp; }
#pragma unroll
    for (int q = 0; q < NPB; ++q) { const int n = n0 + sr0 + RS * q; bok[q] = n < N && sr0 + RS * q < BN; bp[q] = B + (long)(bok[q] ? n : 0) * K + sp; }
    f32x4 ga[2][NPA], gb[2][NPB];
    auto fetch = [&](int k0, f32x4 (&ra)[NPA], f32x4 (&rb)[NPB]) {
        const bool inb = k0 + sp < K;
        const int kc = inb ? k0 : 0;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NPA; ++q) { const f32x4 v = *reinterpret_cast<const f32x4*>(ap[q] + kc); ra[q] = (aok[q] && inb) ? v : z; }
#pragma unroll
        for (int q = 0; q < NPB; ++q) { const f32x4 v = *reinterpret_cast<const f32x4*>(bp[q] + kc); rb[q] = (bok[q] && inb) ? v : z; }
    };
    f32x4 acc[WTM][TN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage_mul = [&](int buf, int k_next, f32x4 (&ra)[NPA], f32x4 (&rb)[NPB]) {
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            u32x2 o[3];
            split4<3>(ra[q], o);
            if (sr0 + RS * q < BM)
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x2*>(&lds[buf][s][sr0 + RS * q][sp]) = o[s];
        }
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            u32x2 o[3];
            split4<3>(rb[q], o);
            if (sr0 + RS * q < BN)
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x2*>(&lds[buf][s][BM + sr0 + RS * q][sp]) = o[s];
        }
        __syncthreads();
        fetch(k_next, ra, rb);
        bf16x8 fa[3][WTM], fb[3][TN];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int i = 0; i < WTM; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][s][wm * (16 * WTM) + i * 16 + r16][8 * kq]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][s][BM + wn * (16 * TN) + j * 16 + r16][8 * kq]);
        }
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][i], fb[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[2][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1][j], c, 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0][j], c, 0, 0, 0);
            }
    };
    fetch(0, ga[0], gb[0]);
    fetch(32, ga[1], gb[1]);
    for (int k0 = 0; k0 < K; k0 += 64) {
        stage_mul(0, k0 + 64, ga[0], gb[0]);
        if (k0 + 32 < K) stage_mul(1, k0 + 96, ga[1], gb[1]);
    }
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * (16 * WTM) + i * 16 + kq * 4 + q;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + r16;
                if (col < N) C[(long)row * N + col] = acc[i][j][q];
            }
        }
}

// the f32-MFMA reference kernel of the library (gemm_nt_big_kernel<4, 3>: 128 x 96 tile, 16-deep slabs), plain matrices
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                       int M, int N, int K, int n_nt) {
    constexpr int BM = 32 * TM, BN = 32 * TN, LD = 20, RPP = 64;
    constexpr int NPA = (BM + RPP - 1) / RPP, NPB = (BN + RPP - 1) / RPP;
    __shared__ __attribute__((aligned(16))) float as[2][BM][LD];
    __shared__ __attribute__((aligned(16))) float bs[2][BN][LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, kq = lane >> 4;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    const int srow = t / 4, sk = 4 * (t % 4);
    const float* ap[NPA]; const float* bp[NPB]; bool aok[NPA], bok[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i) { const int m = m0 + srow + RPP * i; aok[i] = srow + RPP * i < BM && m < M; ap[i] = A + (long)(aok[i] ? m : 0) * K + sk; }
#pragma unroll
    for (int i = 0; i < NPB; ++i) { const int n = n0 + srow + RPP * i; bok[i] = srow + RPP * i < BN && n < N; bp[i] = B + (long)(bok[i] ? n : 0) * K + sk; }
    f32x4 ga[NPA], gb[NPB];
    auto fetch = [&](int k0) {
        const bool inb = k0 + sk < K;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NPA; ++i) ga[i] = (aok[i] && inb) ? *reinterpret_cast<const f32x4*>(ap[i] + k0) : z;
#pragma unroll
        for (int i = 0; i < NPB; ++i) gb[i] = (bok[i] && inb) ? *reinterpret_cast<const f32x4*>(bp[i] + k0) : z;
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) if (BM % RPP == 0 || srow + RPP * i < BM) *reinterpret_cast<f32x4*>(&as[buf][srow + RPP * i][sk]) = ga[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i) if (BN % RPP == 0 || srow + RPP * i < BN) *reinterpret_cast<f32x4*>(&bs[buf][srow + RPP * i][sk]) = gb[i];
        __syncthreads();
        if (k0 + 16 < K) fetch(k0 + 16);
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&as[buf][wm * (16 * TM) + i * 16 + r16][4 * kq]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&bs[buf][wn * (16 * TN) + j * 16 + r16][4 * kq]);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][v], fb[j][v], acc[i][j], 0, 0, 0);
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * (16 * TM) + i * 16 + kq * 4 + q;
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (16 * TN) + j * 16 + r16;
                if (col < N) C[(long)row * N + col] = acc[i][j][q];
            }
        }
}

static void check(const char* tag, const float* dC, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB,
                  float us) {
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0, rms = 0;
    int n = 0;
    unsigned s = 777;
    for (int it = 0; it < 4000; ++it) {
        s = s * 1664525u + 1013904223u; const int m = (s >> 8) % M;
        s = s * 1664525u + 1013904223u; const int c = (s >> 8) % N;
        double r = 0;
        for (int k = 0; k < K; ++k) r += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)c * K + k];
        const double e = fabs((double)hC[(size_t)m * N + c] - r);
        worst = fmax(worst, e); ref_max = fmax(ref_max, fabs(r)); rms += e * e; ++n;
    }
    printf("  %-34s %8.1f us %7.1f TF   max|err|/max|ref| %.2e  rms %.2e\n", tag, us, 2.0 * M * N * K / us / 1e6, worst / ref_max,
           sqrt(rms / n) / ref_max);
}

template <typename F>
static float time_us(F launch, int iters = 30) {
    for (int i = 0; i < 5; ++i) launch();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

template <int TM, int TN, int SPLITS, int DB>
static void run_split(const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
    const int n_nt = (N + 32 * TN - 1) / (32 * TN);
    dim3 grid(((M + 32 * TM - 1) / (32 * TM)) * n_nt);
    hipMemset(C, 0, (size_t)M * N * 4);
    const float us = time_us([&]() { hipLaunchKernelGGL((gemm_split_kernel<TM, TN, SPLITS, DB>), grid, dim3(256), 0, 0, A, B, C, M, N, K, n_nt); });
    char tag[96];
    snprintf(tag, sizeof tag, "bf16 x%d  tile %dx%d %s wgs %d", SPLITS, 32 * TM, 32 * TN, DB ? "2buf" : "1buf", grid.x);
    check(tag, C, M, N, K, hA, hB, us);
}

template <int TM, int TN, int WAVES>
static void run_pf2(const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
    const int n_nt = (N + 32 * TN - 1) / (32 * TN);
    dim3 grid(((M + 32 * TM - 1) / (32 * TM)) * n_nt);
    hipMemset(C, 0, (size_t)M * N * 4);
    const float us = time_us([&]() { hipLaunchKernelGGL((gemm_split_pf2_kernel<TM, TN, WAVES>), grid, dim3(64 * WAVES), 0, 0, A, B, C, M, N, K, n_nt); });
    char tag[96];
    snprintf(tag, sizeof tag, "bf16 x3  tile %dx%d pf2 %dw wgs %d", 32 * TM, 32 * TN, WAVES, grid.x);
    check(tag, C, M, N, K, hA, hB, us);
}

template <int TM, int TN>
static void run_f32(const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
    const int n_nt = (N + 32 * TN - 1) / (32 * TN);
    dim3 grid(((M + 32 * TM - 1) / (32 * TM)) * n_nt);
    hipMemset(C, 0, (size_t)M * N * 4);
    const float us = time_us([&]() { hipLaunchKernelGGL((gemm_f32_kernel<TM, TN>), grid, dim3(256), 0, 0, A, B, C, M, N, K, n_nt); });
    char tag[96];
    snprintf(tag, sizeof tag, "f32 mfma tile %dx%d wgs %d", 32 * TM, 32 * TN, grid.x);
    check(tag, C, M, N, K, hA, hB, us);
}

int main() {
    const int shapes[][3] = {{13056, 900, 600}, {13056, 300, 600}, {4352, 600, 900}, {4352, 600, 1800}, {4352, 300, 600}, {13056, 900, 108}, {7168, 192, 128}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        printf("M=%d N=%d K=%d  (%.2f GFLOP)\n", M, N, K, 2.0 * M * N * K / 1e9);
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd() * 0.1f;
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        run_f32<4, 3>(A, B, C, M, N, K, hA, hB);
        run_f32<4, 4>(A, B, C, M, N, K, hA, hB);
        run_split<4, 3, 3, 0>(A, B, C, M, N, K, hA, hB);
        run_split<4, 3, 3, 1>(A, B, C, M, N, K, hA, hB);
        run_split<4, 2, 3, 0>(A, B, C, M, N, K, hA, hB);
        run_split<4, 2, 3, 1>(A, B, C, M, N, K, hA, hB);
        run_split<2, 2, 3, 1>(A, B, C, M, N, K, hA, hB);
        run_split<2, 3, 3, 0>(A, B, C, M, N, K, hA, hB);
        run_split<2, 3, 3, 1>(A, B, C, M, N, K, hA, hB);
        run_pf2<4, 3, 4>(A, B, C, M, N, K, hA, hB);
        run_pf2<4, 3, 8>(A, B, C, M, N, K, hA, hB);
        run_pf2<4, 2, 4>(A, B, C, M, N, K, hA, hB);
        run_pf2<4, 2, 8>(A, B, C, M, N, K, hA, hB);
        run_pf2<2, 3, 4>(A, B, C, M, N, K, hA, hB);
        run_pf2<2, 2, 4>(A, B, C, M, N, K, hA, hB);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
