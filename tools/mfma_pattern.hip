// What keeps a bf16 x 3 slab loop off the matrix pipe's 16 cycles per v_mfma_f32_16x16x32_bf16?  512-thread workgroups (two waves per SIMD,
// one workgroup per CU through a 120 KB LDS allocation), every wave the matrix-wave loop of csrc/gemm_mw.hip on RANDOM bf16 data:
//   mode 0: 72 MFMAs per "slab" (12 accumulators x 6 terms, chains as in the kernel), operands fixed in registers, no LDS, no barrier
//   mode 1: + the slab's 21 fragment reads (ds_read_b128) in the kernel's prefetch order
//   mode 2: + one s_barrier per slab
//   mode 3: mode 2 with the MFMAs term-major over the 4 row tiles (dependent MFMAs 4 issues apart)
//   mode 4: mode 1 with ONE wave per SIMD doing 144 MFMAs per slab (256-thread workgroups)
// prints shader cycles per MFMA per SIMD (s_memtime) and the clock (s_memtime / s_memrealtime)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDS_BYTES = 120 * 1024;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const unsigned* __restrict__ src, float* out, unsigned long long* clk, int slabs) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (MODE == 4 && wave >= 4) return;
    for (int i = t; i < LDS_BYTES / 4; i += (MODE == 4 ? 256 : 512)) reinterpret_cast<unsigned*>(smem)[i] = src[(i * 7 + blockIdx.x) & 0xffff];
    __syncthreads();
    constexpr int TM = 4, TN = (MODE == 4) ? 6 : 3;
    const int r16 = lane & 15, kq = lane >> 4;
    const int foff = (wave * 16 + r16) * 64 + ((kq ^ (((r16 >> 3) & 1) << 1)) * 16);
    f32x4 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 fa[3][TM], fb[2][3];
    auto lfa = [&](int s, int i, int buf) { fa[s][i] = *reinterpret_cast<const bf16x8*>(smem + buf * 49152 + s * 8192 + i * 1024 + foff); };
    auto lfb = [&](bf16x8 (&f)[3], int j, int buf) {
        for (int s = 0; s < 3; ++s) f[s] = *reinterpret_cast<const bf16x8*>(smem + 24576 + buf * 49152 + s * 8192 + j * 1024 + foff);
    };
    for (int s = 0; s < 3; ++s) for (int i = 0; i < TM; ++i) lfa(s, i, 0);
    lfb(fb[0], 0, 0); lfb(fb[1], 1, 0);
    auto mma = [&](const bf16x8 (&f)[3], int j, int i) {
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fa[2][i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], fa[0][i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], fa[1][i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fa[1][i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], fa[0][i], c, 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fa[0][i], c, 0, 0, 0);
    };
    auto mma_tm = [&](const bf16x8 (&f)[3], int j) {          // term-major over the row tiles
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fa[2][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], fa[0][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], fa[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fa[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], fa[0][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], fa[0][i], acc[i][j], 0, 0, 0);
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int n = 0; n < slabs; ++n) {
        const int buf = n & 1, nb = buf ^ 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (MODE >= 1) {
                if (j + 1 < TN) { lfb(fb[(j + 1) & 1], j + 1, buf); }
                else {
                    if (MODE == 2 || MODE == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
                    lfb(fb[(j + 1) & 1], 0, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE >= 1 && j + 1 == TN) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    mma(fb[j & 1], j, i);
                    __builtin_amdgcn_sched_barrier(0);
                    for (int s = 0; s < 3; ++s) lfa(s, i, nb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (MODE == 3) {
                mma_tm(fb[j & 1], j);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) mma(fb[j & 1], j, i);
            }
            if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sacc = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) sacc += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + t] = sacc;
    if (lane == 0) { clk[(blockIdx.x * 8 + wave) * 2] = t1 - t0; clk[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(const char* name, const unsigned* src, float* out, unsigned long long* clk) {
    const int slabs = 20000, blocks = 256;
    hipMemset(clk, 0, blocks * 16 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(MODE == 4 ? 256 : 512), 0, 0, src, out, clk, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(MODE == 4 ? 256 : 512), 0, 0, src, out, clk, slabs);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0; int cnt = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < (MODE == 4 ? 4 : 8); ++w) { cyc += (double)h[(b * 8 + w) * 2]; real += (double)h[(b * 8 + w) * 2 + 1]; ++cnt; }
    cyc /= cnt; real /= cnt;
    const double mfma_per_simd = 144.0 * slabs;
    printf("%-58s %8.3f ms  %6.2f cycles per MFMA per SIMD  clock %.2f GHz  chip %.0f TFLOP/s (bf16)\n", name, ms, cyc / mfma_per_simd, cyc / real * 0.1,
           mfma_per_simd * 1024 * 16384.0 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned* src; float* out; unsigned long long* clk;
    hipMalloc(&src, 65536 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 16 * 8);
    unsigned* h = (unsigned*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) {       // two random bf16 in [-2, 2) per word
        auto bf = [] { unsigned sign = rand() & 1, exp = 120 + rand() % 8, man = rand() & 127; return (sign << 15) | (exp << 7) | man; };
        h[i] = bf() | (bf() << 16);
    }
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    run<0>("mode 0: 2 waves/SIMD, registers only", src, out, clk);
    run<1>("mode 1: + 21 fragment reads per wave and slab", src, out, clk);
    run<2>("mode 2: + one s_barrier per slab", src, out, clk);
    run<3>("mode 3: mode 2, MFMAs term-major (chains 4 apart)", src, out, clk);
    run<4>("mode 4: ONE wave/SIMD, 144 MFMAs + 30 reads per slab", src, out, clk);
    return 0;
}
