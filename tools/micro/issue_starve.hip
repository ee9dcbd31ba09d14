// How slow does a wave's ordinary code run next to waves that stream MFMAs on the same SIMD?  (round 4)
// The per-workgroup timeline of the fp32 GEMM (tools/gemm_timeline_f32.py) shows the ~560-instruction address set-up of a workgroup
// taking 2 us on an empty CU and 12-13 us when the CU's other workgroup is inside its K loop.  This micro isolates the effect:
// one workgroup of 12 waves per CU - waves 0..7 "aggressors" (two per SIMD, back-to-back independent v_mfma_f32_32x32x2_f32, no
// memory), waves 8..11 "victims" (one per SIMD) running ONE dependent chain of a single instruction class:
//   V0 v_add_f32 (VALU)   V1 s_add_u32 (SALU)   V2 s_load_dword pointer chase (scalar cache hit)   V3 v_rcp_f32 (transcendental)
//   V4 s_mul_hi_u32 / integer-division-like SALU mix   V5 global_load_dword pointer chase (L2 hit)
// (the aggressors stream for ~0.7 ms; a victim that reports about that long made no progress until they had finished)
// Each victim is timed with the 100 MHz wall clock, with the aggressors idle (spinning on s_sleep) and with them streaming.
// Build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_starve tools/micro/issue_starve.hip && /tmp/issue_starve
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VICTIM, int PRIO>
__global__ __launch_bounds__(768) void k(unsigned long long* out, const unsigned* chase, int aggress, int mfma_iters, int chain) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 8) {
        if (aggress) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i)
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const float a = (float)(threadIdx.x & 7) * 0.25f, b = 0.5f;
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
            float s = 0.f;
            for (int i = 0; i < 4; ++i)
                for (int r = 0; r < 16; ++r) s += acc[i][r];
            if (s == 12345.678f) out[4096] = 1;
        }
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_sleep(64);   // let the aggressors reach their loop
    const unsigned long long t0 = wall_clock64();
    unsigned sink = 0;
    if (VICTIM == 0) {
        float v = (float)threadIdx.x;
        for (int i = 0; i < chain; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v));
        sink = __float_as_uint(v);
    } else if (VICTIM == 1) {
        unsigned s = wave;
        for (int i = 0; i < chain; ++i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
        sink = s;
    } else if (VICTIM == 2) {
        const unsigned* p = chase;
        unsigned off = 0;
        for (int i = 0; i < chain / 8; ++i) {
            unsigned nx;
            asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(nx) : "s"(p), "s"(off) : "memory");
            off = nx;   // the table holds zeros: every load hits the same line
        }
        sink = off;
    } else if (VICTIM == 3) {
        float v = 1.5f + (float)threadIdx.x;
        for (int i = 0; i < chain / 4; ++i) asm volatile("v_rcp_f32 %0, %0\n\ts_nop 0" : "+v"(v));
        sink = __float_as_uint(v);
    } else if (VICTIM == 4) {
        unsigned s = 0x9E3779B9u + wave, m = 0xAAAAAAABu;
        for (int i = 0; i < chain; ++i) asm volatile("s_mul_hi_u32 %0, %0, %1\n\ts_add_u32 %0, %0, 0x12345" : "+s"(s) : "s"(m) : "scc");
        sink = s;
    } else {
        unsigned off = (threadIdx.x & 63) * 4;
        for (int i = 0; i < chain / 16; ++i) {
            unsigned nx;
            asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(nx) : "v"(off), "s"(chase) : "memory");
            off = nx + (threadIdx.x & 63) * 4;
        }
        sink = off;
    }
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (wave - 8)) * 2] = t1 - t0;
        out[(blockIdx.x * 4 + (wave - 8)) * 2 + 1] = sink;
    }
}

template <int VICTIM, int PRIO>
void run(const char* name, unsigned long long* out, const unsigned* chase, int chain) {
    std::vector<unsigned long long> h(256 * 4 * 2);
    double us[2];
    for (int aggress = 0; aggress < 2; ++aggress) {
        hipLaunchKernelGGL((k<VICTIM, PRIO>), dim3(256), dim3(768), 0, 0, out, chase, aggress, 400, chain);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < 256 * 4; ++i) s += (double)h[2 * i];
        us[aggress] = s / (256 * 4) / 100.0;
    }
    printf("%-58s chain %5d: alone %8.2f us   next to 2 MFMA waves/SIMD %8.2f us   x%.2f\n", name, chain, us[0], us[1], us[1] / us[0]);
    fflush(stdout);
}

int main() {
    unsigned long long* out;
    unsigned* chase;
    hipMalloc(&out, 8 * 8192);
    hipMalloc(&chase, 4096);
    hipMemset(chase, 0, 4096);
    run<0, 0>("V0 dependent v_add_f32", out, chase, 2000);
    run<1, 0>("V1 dependent s_add_u32", out, chase, 2000);
    run<2, 0>("V2 s_load_dword pointer chase (250 loads)", out, chase, 2000);
    run<3, 0>("V3 dependent v_rcp_f32 (500)", out, chase, 2000);
    run<4, 0>("V4 dependent s_mul_hi_u32 + s_add_u32", out, chase, 2000);
    run<5, 0>("V5 global_load_dword pointer chase (125 loads)", out, chase, 2000);
    run<0, 1>("V0 dependent v_add_f32, victim at s_setprio 3", out, chase, 2000);
    run<3, 1>("V3 dependent v_rcp_f32, victim at s_setprio 3", out, chase, 2000);
    run<5, 1>("V5 global_load_dword chase, victim at s_setprio 3", out, chase, 2000);
    return 0;
}
