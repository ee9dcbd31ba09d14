// Do matrix and vector instructions of DIFFERENT waves on one SIMD overlap?  One 512-thread workgroup per CU = two waves per
// SIMD (waves w and w + 4 share SIMD w, microarch guide "Two waves per SIMD"); waves 0-3 run role A, waves 4-7 role B, each a
// register-only loop of fixed length in cycles when alone:
//   M  = a dependent chain of v_mfma_f32_32x32x16_bf16 (the attention kernels' score / PV products)
//   M4 = four independent chains of the same
//   E  = v_exp_f32 on 16 independent registers        A = v_add_f32 (16 independent)       X = exp + add + max3 + cvt mix
//   -  = the wave exits at once
// For every pair the kernel time T(A,B) is printed next to T(A,-) and T(-,B): overlap = T(A,B) close to max, none = close to sum.
// Build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/coexec tools/micro/coexec.hip && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Role { NONE = 0, M1 = 1, M4 = 2, EXP = 3, ADD = 4, MIX = 5, SAME6 = 6, SAMEX = 7, ADDP = 8, M4P = 9, SAME6E = 10 };

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed) {
    if (ROLE == NONE) return 0.f;
    if (ROLE == ADDP || ROLE == M4P) __builtin_amdgcn_s_setprio(3);
    if (ROLE == SAME6 || ROLE == SAMEX || ROLE == SAME6E) {   // ONE wave's stream: every MFMA followed by its share of the vector work
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(1.0f + 0.25f * j); }
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = seed * 1e-3f + i * 0.01f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
                if (ROLE == SAME6) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(6 * u + i) & 15]) : "v"(seed));
                } else if (ROLE == SAME6E) {   // 2 exp + 2 add per gap: 16 + 8 = 24 cycles
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[(4 * u) & 15]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[(4 * u + 1) & 15]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(4 * u + 2) & 15]) : "v"(seed));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(4 * u + 3) & 15]) : "v"(seed));
                } else {   // per MFMA 1/9 of a block's softmax: ~2 exp, 2 add, 1 max3, 1 cvt
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[(2 * u) & 15]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[(2 * u + 1) & 15]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(2 * u + 2) & 15]) : "v"(seed));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(2 * u + 3) & 15]) : "v"(seed));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[(2 * u + 4) & 15]) : "v"(v[(2 * u + 5) & 15]), "v"(seed));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[(2 * u + 6) & 15]) : "v"(v[(2 * u + 7) & 15]));
                }
            }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
        return s;
    }
    if (ROLE == M1 || ROLE == M4 || ROLE == M4P) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(1.0f + 0.25f * j); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (ROLE == M1) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
                else acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        return s;
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed * 1e-3f + i * 0.01f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ROLE == EXP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            } else if (ROLE == ADD || ROLE == ADDP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            } else {  // the softmax mix of one 32-key block per lane: 16 exp, 16 add, 6 max3, 8 cvt_pk
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[i + 6]), "v"(seed));
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[i + 8]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void coexec_kernel(float* out, int ia, int ib) {
    const int wave = threadIdx.x >> 6;
    float s;
    if (wave < 4) s = run_role<RA>(ia, (float)threadIdx.x);
    else s = run_role<RB>(ib, (float)threadIdx.x);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int RA, int RB>
static float time_pair(float* out, int ia, int ib) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((coexec_kernel<RA, RB>), dim3(256), dim3(512), 0, 0, out, ia, ib);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

#define PAIR(NAME, RA, RB, IA, IB)                                                                         \
    {                                                                                                      \
        const float ta = time_pair<RA, NONE>(out, IA, 0), tb = time_pair<NONE, RB>(out, 0, IB);           \
        const float tab = time_pair<RA, RB>(out, IA, IB);                                                  \
        printf("%-10s A alone %7.3f ms   B alone %7.3f ms   together %7.3f ms   (max %7.3f, sum %7.3f)  overlap %.2f\n", NAME, ta, tb, \
               tab, ta > tb ? ta : tb, ta + tb, (ta + tb - tab) / (ta < tb ? ta : tb));                    \
    }

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 512 * 256);
    // iteration counts chosen so that both roles take about the same time alone: 16 MFMAs = 512 cycles per iteration,
    // EXP / MIX / ADD iterations are 4 x (16 x 8 | ~300 | 32 x 4) cycles
    const int n = 4000;
    PAIR("M1 | E", M1, EXP, n, n)
    PAIR("M4 | E", M4, EXP, n, n)
    PAIR("M1 | A", M1, ADD, n, n)
    PAIR("M4 | A", M4, ADD, n, n)
    PAIR("M1 | X", M1, MIX, n, n / 2)
    PAIR("M4 | X", M4, MIX, n, n / 2)
    PAIR("A | M4", ADD, M4, n, n)          // roles swapped: the vector waves are the older ones
    PAIR("X | M4", MIX, M4, n / 2, n)
    PAIR("E | M4", EXP, M4, n, n)
    PAIR("M4 | Ap", M4, ADDP, n, n)        // the vector waves at s_setprio 3
    PAIR("M4p | A", M4P, ADD, n, n)        // the matrix waves at s_setprio 3
    PAIR("S6 | -", SAME6, NONE, n, 0)      // one wave: MFMA + 6 v_add per gap (24 cycles of vector issue)
    PAIR("S6E | -", SAME6E, NONE, n, 0)    // one wave: MFMA + 2 exp + 2 add per gap
    PAIR("SX | -", SAMEX, NONE, n, 0)      // one wave: MFMA + 2 exp, 2 add, max3, cvt per gap (32 cycles)
    PAIR("S6 | S6", SAME6, SAME6, n, n)    // two such waves per SIMD
    PAIR("SX | SX", SAMEX, SAMEX, n, n)
    PAIR("M4 | M4", M4, M4, n, n)
    PAIR("E | E", EXP, EXP, n, n)
    PAIR("X | X", MIX, MIX, n / 2, n / 2)
    return 0;
}
