// GEMM design-space lab (standalone, no torch): C[M][N] = A[M][K] . B[N][K]^T, fp32 MFMA, 128x128 workgroup tile,
// 4 waves (2x2) each 64x64.  Variants: MFMA shape (16x16x4 | 32x32x2), slab depth BK, LDS row padding.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/gemm_lab.bin tools/gemm_lab.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int xcd_chunked_id(int bid, int nwg) {
    const int per = nwg / 8, rem = nwg % 8, x = bid % 8, q = bid / 8;
    return x * per + (x < rem ? x : rem) + q;
}

template <int MF, int BK, int PAD, int SWZ, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K, int n_nt) {
    constexpr int LD = BK + PAD;
    constexpr int PPR = BK / 4;                 // 16-byte pieces per row
    constexpr int NP = 128 * PPR / 256;         // pieces per thread per operand
    __shared__ __attribute__((aligned(16))) float as[2][128][LD];
    __shared__ __attribute__((aligned(16))) float bs[2][128][LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = SWZ ? xcd_chunked_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const int m0 = (lid / n_nt) * 128, n0 = (lid % n_nt) * 128;

    const float* ap[NP]; const float* bp[NP]; bool aok[NP], bok[NP]; int srow[NP], sk[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = t + 256 * i;
        srow[i] = p / PPR; sk[i] = 4 * (p % PPR);
        aok[i] = m0 + srow[i] < M; bok[i] = n0 + srow[i] < N;
        ap[i] = A + (long)(aok[i] ? m0 + srow[i] : 0) * K + sk[i];
        bp[i] = B + (long)(bok[i] ? n0 + srow[i] : 0) * K + sk[i];
    }
    f32x4 ga[NP], gb[NP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool inb = k0 + sk[i] < K;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ga[i] = (aok[i] && inb) ? *reinterpret_cast<const f32x4*>(ap[i] + k0) : z;
            gb[i] = (bok[i] && inb) ? *reinterpret_cast<const f32x4*>(bp[i] + k0) : z;
        }
    };
    constexpr int NT = MF == 0 ? 4 : 2;         // MFMA tiles per wave per dimension
    constexpr int NA = MF == 0 ? 4 : 16;        // accumulator floats per tile per lane
    typedef float accv __attribute__((ext_vector_type(NA)));
    accv acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < NA; ++e) acc[i][j][e] = 0.f;
    const int rr = MF == 0 ? (lane & 15) : (lane & 31);
    const int kq = MF == 0 ? (lane >> 4) : (lane >> 5);
    constexpr int TS = MF == 0 ? 16 : 32;       // tile size
    constexpr int KG = MF == 0 ? 16 : 8;        // k covered by one 16-byte fragment load across the k-groups

    fetch(0);
    int buf = 0;
    f32x4 fa[NT], fb[NT];
    if (ABL & 4) {
#pragma unroll
        for (int i = 0; i < NT; ++i) { fa[i] = ga[0] + (float)i; fb[i] = gb[0] + (float)i; }
    }
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (!(ABL & 8)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            *reinterpret_cast<f32x4*>(&as[buf][srow[i]][sk[i]]) = ga[i];
            *reinterpret_cast<f32x4*>(&bs[buf][srow[i]][sk[i]]) = gb[i];
        }
        }
        if (!(ABL & 2)) __syncthreads();
        if (!(ABL & 1) && k0 + BK < K) fetch(k0 + BK);
#pragma unroll
        for (int u = 0; u < BK / KG; ++u) {
            if (!(ABL & 4)) {
#pragma unroll
            for (int i = 0; i < NT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&as[buf][wm * 64 + i * TS + rr][KG * u + 4 * kq]);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&bs[buf][wn * 64 + j * TS + rr][KG * u + 4 * kq]);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (MF == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][v], fb[j][v], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][v], fb[j][v], acc[i][j], 0, 0, 0);
                    }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int row = m0 + wm * 64 + i * TS + (MF == 0 ? kq * 4 + e : 8 * (e >> 2) + 4 * kq + (e & 3));
                const int col = n0 + wn * 64 + j * TS + rr;
                if (row < M && col < N) C[(long)row * N + col] = acc[i][j][e];
            }
}

template <int MF, int BK, int PAD, int SWZ, int ABL = 0>
double run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& hA,
           const std::vector<float>& hB) {
    const int n_nt = (N + 127) / 128, grid = ((M + 127) / 128) * n_nt;
    hipMemset(C, 0, sizeof(float) * (size_t)M * N);
    hipLaunchKernelGGL((gemm_kernel<MF, BK, PAD, SWZ, ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, n_nt);
    hipDeviceSynchronize();
    // spot check
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), C, sizeof(float) * hC.size(), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 200; ++s) {
        const int r = (int)(((long)s * 7919 + 13) % M), c = (int)(((long)s * 104729 + 7) % N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
        worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
    }
    // also the last row / col
    {
        const int r = M - 1, c = N - 1; double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
        worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_kernel<MF, BK, PAD, SWZ, ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, n_nt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / us / 1e6;
    printf("  %-28s %8.1f us %6.1f TF  max|err| %.2e\n", name, us, tf, worst);
    return tf;
}

int main_old() {
    const int shapes[][3] = {{13056, 1800, 600}, {13056, 900, 600}, {4352, 600, 1800}, {13056, 1800, 108}, {13056, 300, 600}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        printf("M=%d N=%d K=%d\n", M, N, K);
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd() * 0.1f;
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        run<0, 16, 4, 1>("16x16x4 BK16 pad4 swz", A, B, C, M, N, K, hA, hB);
        run<0, 16, 4, 0>("16x16x4 BK16 pad4 noswz", A, B, C, M, N, K, hA, hB);
        run<1, 16, 4, 1>("32x32x2 BK16 pad4 swz", A, B, C, M, N, K, hA, hB);
        run<0, 32, 4, 1>("16x16x4 BK32 pad4 swz", A, B, C, M, N, K, hA, hB);
        run<1, 32, 4, 1>("32x32x2 BK32 pad4 swz", A, B, C, M, N, K, hA, hB);
        run<1, 16, 8, 1>("32x32x2 BK16 pad8 swz", A, B, C, M, N, K, hA, hB);
        run<1, 16, 0, 1>("32x32x2 BK16 pad0 swz", A, B, C, M, N, K, hA, hB);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// tile menu: 4 waves (2x2), wave tile (16*WTM) x (16*WTN) of 16x16x4 MFMAs, workgroup tile BM = 32*WTM, BN = 32*WTN
template <int WTM, int WTN, int BK, int ABL = 0>   // ABL bit0: no global fetch in the loop, bit1: no barrier, bit2: no LDS fragment reads, bit3: no LDS writes
__global__ __launch_bounds__(256) void gemm_menu(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                 int M, int N, int K, int n_nt) {
    constexpr int BM = 32 * WTM, BN = 32 * WTN, LD = BK + 4, PPR = BK / 4;
    constexpr int NPA = (BM * PPR + 255) / 256, NPB = (BN * PPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) float as[2][BM][LD];
    __shared__ __attribute__((aligned(16))) float bs[2][BN][LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int m0 = (lid / n_nt) * BM, n0 = (lid % n_nt) * BN;
    const float* ap[NPA]; const float* bp[NPB]; bool aok[NPA], bok[NPB]; int ar[NPA], br[NPB], ak[NPA], bk[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int p = t + 256 * i;
        ar[i] = p / PPR; ak[i] = 4 * (p % PPR);
        aok[i] = ar[i] < BM && m0 + ar[i] < M;
        ap[i] = A + (long)(aok[i] ? m0 + ar[i] : 0) * K + ak[i];
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int p = t + 256 * i;
        br[i] = p / PPR; bk[i] = 4 * (p % PPR);
        bok[i] = br[i] < BN && n0 + br[i] < N;
        bp[i] = B + (long)(bok[i] ? n0 + br[i] : 0) * K + bk[i];
    }
    f32x4 ga[NPA], gb[NPB];
    auto fetch = [&](int k0) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NPA; ++i) ga[i] = (aok[i] && k0 + ak[i] < K) ? *reinterpret_cast<const f32x4*>(ap[i] + k0) : z;
#pragma unroll
        for (int i = 0; i < NPB; ++i) gb[i] = (bok[i] && k0 + bk[i] < K) ? *reinterpret_cast<const f32x4*>(bp[i] + k0) : z;
    };
    f32x4 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    fetch(0);
    int buf = 0;
    f32x4 fa[WTM], fb[WTN];
    if (ABL & 4) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[i] = ga[0] + (float)i;
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[j] = gb[0] + (float)j;
    }
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (!(ABL & 8)) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) if (ar[i] < BM) *reinterpret_cast<f32x4*>(&as[buf][ar[i]][ak[i]]) = ga[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i) if (br[i] < BN) *reinterpret_cast<f32x4*>(&bs[buf][br[i]][bk[i]]) = gb[i];
        }
        if (!(ABL & 2)) __syncthreads();
        if (!(ABL & 1) && k0 + BK < K) fetch(k0 + BK);
#pragma unroll
        for (int u = 0; u < BK / 16; ++u) {
            if (!(ABL & 4)) {
#pragma unroll
            for (int i = 0; i < WTM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&as[buf][wm * 16 * WTM + i * 16 + r16][16 * u + 4 * kq]);
#pragma unroll
            for (int j = 0; j < WTN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&bs[buf][wn * 16 * WTN + j * 16 + r16][16 * u + 4 * kq]);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][v], fb[j][v], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm * 16 * WTM + i * 16 + kq * 4 + e;
                const int col = n0 + wn * 16 * WTN + j * 16 + r16;
                if (row < M && col < N) C[(long)row * N + col] = acc[i][j][e];
            }
}

template <int WTM, int WTN, int BK, int ABL = 0>
void run_menu(const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
    constexpr int BM = 32 * WTM, BN = 32 * WTN;
    const int n_nt = (N + BN - 1) / BN, grid = ((M + BM - 1) / BM) * n_nt;
    hipMemset(C, 0, sizeof(float) * (size_t)M * N);
    hipLaunchKernelGGL((gemm_menu<WTM, WTN, BK, ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, n_nt);
    hipDeviceSynchronize();
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), C, sizeof(float) * hC.size(), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 101; ++s) {
        const int r = s == 100 ? M - 1 : (int)(((long)s * 7919 + 13) % M), c = s == 100 ? N - 1 : (int)(((long)s * 104729 + 7) % N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
        worst = fmax(worst, fabs(ref - hC[(size_t)r * N + c]));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_menu<WTM, WTN, BK, ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, n_nt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    printf("  %3dx%3d BK%2d abl %2d wgs %5d  %8.1f us %6.1f TF  err %.1e\n", BM, BN, BK, ABL, grid, us, 2.0 * M * N * K / us / 1e6, worst);
}

int main_menu() {
    const int shapes[][3] = {{13056, 900, 600}, {13056, 900, 108}, {13056, 300, 600}, {4352, 600, 900}, {4352, 300, 600}, {4352, 108, 900},
                             {13056, 150, 300}, {4352, 300, 150}, {27776, 64, 480}, {168064, 32, 240}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        printf("M=%d N=%d K=%d\n", M, N, K);
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd() * 0.1f;
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        run_menu<4, 4, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 32>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 5, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 3, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 2, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<2, 5, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<2, 4, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<2, 3, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<2, 2, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<2, 2, 32>(A, B, C, M, N, K, hA, hB);
        run_menu<1, 5, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<1, 4, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<1, 2, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 1, 16>(A, B, C, M, N, K, hA, hB);
        run_menu<2, 1, 16>(A, B, C, M, N, K, hA, hB);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}

int main() {
    const int shapes[][3] = {{13056, 1800, 600}, {13056, 900, 600}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        printf("M=%d N=%d K=%d\n", M, N, K);
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd() * 0.1f;
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        run<1, 16, 4, 1, 0>("32x32x2 abl0", A, B, C, M, N, K, hA, hB);
        run<1, 16, 4, 1, 1>("32x32x2 abl1 (no fetch)", A, B, C, M, N, K, hA, hB);
        run<1, 16, 4, 1, 4>("32x32x2 abl4 (no lds read)", A, B, C, M, N, K, hA, hB);
        run<1, 16, 4, 1, 5>("32x32x2 abl5", A, B, C, M, N, K, hA, hB);
        run<1, 16, 4, 1, 15>("32x32x2 abl15 (mfma only)", A, B, C, M, N, K, hA, hB);
        run<0, 16, 4, 1, 15>("16x16x4 abl15 (mfma only)", A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 0>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 1>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 2>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 3>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 4>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 5>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 7>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 4, 16, 15>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 2, 16, 0>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 2, 16, 1>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 2, 16, 2>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 2, 16, 4>(A, B, C, M, N, K, hA, hB);
        run_menu<4, 2, 16, 15>(A, B, C, M, N, K, hA, hB);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
