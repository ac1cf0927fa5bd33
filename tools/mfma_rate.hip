// Back-to-back MFMA issue-rate probe (MFMAs pinned with inline asm so the compiler cannot shuffle accumulators through AGPRs): one wave per SIMD (256-thread blocks, 1 block per CU), independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
    if (MODE == 0) {          // f32 16x16x4, 8 accumulators
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if (MODE == 1) {   // f32 32x32x2, 4 accumulators
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if (MODE == 3) {   // bf16 32x32x16, 4 accumulators
        bf16x8 fa, fb;
        for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(a + j); fb[j] = (__bf16)(b - j); }
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
        float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {                  // bf16 16x16x32, 8 accumulators
        bf16x8 fa, fb;
        for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(a + j); fb[j] = (__bf16)(b - j); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
        float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
template <int MODE> void run(const char* name, int per_iter, double flop_per_mfma, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * per_iter;                 // MFMAs per wave
    double tf = n * flop_per_mfma * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-14s blocks %4d: %.3f ms, %.1f ns per MFMA per wave (= %.1f cycles at 2.4 GHz), chip %.1f TFLOP/s\n", name, blocks, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4, tf);
    hipFree(out);
}
int main() {
    for (int blocks : {256, 512, 1024}) {
        run<0>("f32 16x16x4", 8, 2.0 * 16 * 16 * 4, blocks);
        run<1>("f32 32x32x2", 4, 2.0 * 32 * 32 * 2, blocks);
        run<2>("bf16 16x16x32", 8, 2.0 * 16 * 16 * 32, blocks);
        run<3>("bf16 32x32x16", 4, 2.0 * 32 * 32 * 16, blocks);
    }
    return 0;
}
