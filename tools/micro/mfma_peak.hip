// Sustained MFMA-only rate of this GPU: back-to-back independent v_mfma_f32_32x32x2_f32 (and 16x16x32 bf16) with no
// memory traffic at all, 1/2/4 waves per SIMD.  The number a GEMM kernel's TFLOP/s should be read against besides the
// nominal peak (clock under sustained matrix load is part of it).
// Build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void f32_kernel(float* out, int iters, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // shader clock cycles and 100 MHz wall ticks of one wave
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

__global__ __launch_bounds__(256) void bf16_kernel(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.0f + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long* clk;
    hipMallocManaged(&clk, 16);
    const int order[6] = {4, 2, 1, 1, 2, 4};
    for (int oi = 0; oi < 6; ++oi) {                 // fp32 first, both orders, equal wall time per configuration
        const int wps = order[oi], blocks = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) {          // rep 0 warms the clocks
            const int iters = 400000 / wps;
            hipEventRecord(e0);
            hipLaunchKernelGGL(f32_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * iters * 32.0 * (2.0 * 32 * 32 * 2);
            if (rep) printf("fp32 32x32x2   %d waves/SIMD: %8.2f ms  %7.1f TFLOP/s   shader clock %.0f MHz\n", wps, ms, fl / ms / 1e9,
                            (double)clk[0] / ((double)clk[1] / 100.0));
        }
    }
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD: blocks of 4 waves, wps blocks per CU
        const int blocks = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 400000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(bf16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * iters * 32.0 * (2.0 * 16 * 16 * 32);
            if (rep) printf("bf16 16x16x32  %d waves/SIMD: %8.2f ms  %7.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
