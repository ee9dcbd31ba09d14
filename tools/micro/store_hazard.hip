// Library-free reproducer attempt (round 4): buffer_store_dwordx4 with an SGPR soffset, followed AT ONCE by a VALU write of the
// first store-data register.
// Found in the direct epilogue of gemm_f32_glds_kernel (OPT bit 1024): hipcc emitted
//     buffer_store_dwordx4 v[32:35], v112, s[44:47], s53 offen
//     v_add_f32_e32 v32, v36, v104
// with no wait state in between (LLVM's GCNHazardRecognizer exempts MUBUF stores whose soffset is a register from the
// ">8-byte VMEM store data followed by a VALU write of it" rule), and element 0 of lanes 12-15 / 28-31 / 44-47 / 60-63 of the
// stored float4 came out wrong (profiles/r04_store_data_hazard.txt).  With the row offset folded into the VGPR offset (soffset 0)
// the compiler inserts s_nop and the results are exact.
// Here: victims store i-dependent float4s through a buffer descriptor and clobber the first data register in the next
// instruction; every stored value is checked.  Variants: soffset in an SGPR / soffset 0; with and without two waves per SIMD
// streaming MFMAs (the GEMM's epilogue always runs next to a peer workgroup in its K loop) and LDS traffic.
// Build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_hazard tools/micro/store_hazard.hip && /tmp/store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool SGPR_SOFF, bool NOP>
__global__ __launch_bounds__(768) void k(float* out, int iters, int aggress, int mfma_iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
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
                if ((it & 7) == 7) __builtin_amdgcn_s_sleep(8);   // gaps, as the K loop's barriers leave them
            }
            float s = 0.f;
            for (int i = 0; i < 4; ++i)
                for (int r = 0; r < 16; ++r) s += acc[i][r];
            if (s == 12345.678f) out[0] = 1.f;
        }
        return;
    }
    // victim wave v of block b owns rows [((b * 4 + v) * iters + i) * 64 + lane] of float4
    const int v = wave - 8;
    float* base = out + 4096 + (size_t)((blockIdx.x * 4 + v) * (size_t)iters) * 64 * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    for (int i = 0; i < iters; ++i) {
        const float x = (float)(i * 64 + lane);
        const int soff = i * 1024;
        if (SGPR_SOFF) {
            asm volatile(
                "v_add_f32 v4, 0.5, %0\n\t"
                "v_add_f32 v5, 1.5, %0\n\t"
                "v_add_f32 v6, 2.5, %0\n\t"
                "v_add_f32 v7, 3.5, %0\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[4:7], %1, %2, %3 offen\n\t"
                "v_add_f32 v4, v5, v6\n\t"          // the next chunk's first value lands in the store's first data register
                "v_add_f32 v5, v6, v7\n\t"
                :
                : "v"(x), "v"(voff), "s"(rsrc), "s"(soff)
                : "v4", "v5", "v6", "v7", "memory");
        } else {
            const int vo2 = voff + soff;
            if (NOP)
                asm volatile(
                    "v_add_f32 v4, 0.5, %0\n\tv_add_f32 v5, 1.5, %0\n\tv_add_f32 v6, 2.5, %0\n\tv_add_f32 v7, 3.5, %0\n\ts_nop 4\n\t"
                    "buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n\t"
                    "s_nop 1\n\t"
                    "v_add_f32 v4, v5, v6\n\tv_add_f32 v5, v6, v7\n\t"
                    :
                    : "v"(x), "v"(vo2), "s"(rsrc)
                    : "v4", "v5", "v6", "v7", "memory");
            else
                asm volatile(
                    "v_add_f32 v4, 0.5, %0\n\tv_add_f32 v5, 1.5, %0\n\tv_add_f32 v6, 2.5, %0\n\tv_add_f32 v7, 3.5, %0\n\ts_nop 4\n\t"
                    "buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n\t"
                    "v_add_f32 v4, v5, v6\n\tv_add_f32 v5, v6, v7\n\t"
                    :
                    : "v"(x), "v"(vo2), "s"(rsrc)
                    : "v4", "v5", "v6", "v7", "memory");
        }
    }
}

template <bool SGPR_SOFF, bool NOP>
void run(const char* name, float* out, size_t n_floats, int iters) {
    std::vector<float> h(n_floats);
    for (int aggress = 0; aggress < 2; ++aggress) {
        hipMemset(out, 0, n_floats * 4);
        hipLaunchKernelGGL((k<SGPR_SOFF, NOP>), dim3(256), dim3(768), 0, 0, out, iters, aggress, 3000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, n_floats * 4, hipMemcpyDeviceToHost);
        long long bad = 0, bad_e0 = 0, total = 0;
        int lanes[64] = {0};
        for (size_t w = 0; w < 256 * 4; ++w)
            for (int i = 0; i < iters; ++i)
                for (int l = 0; l < 64; ++l) {
                    const float x = (float)(i * 64 + l);
                    const float* p = h.data() + 4096 + ((w * iters + i) * 64 + l) * 4;
                    for (int e = 0; e < 4; ++e) {
                        ++total;
                        if (p[e] != x + 0.5f + e) {
                            ++bad;
                            if (e == 0) ++bad_e0;
                            lanes[l]++;
                        }
                    }
                }
        printf("%-52s %s: %lld wrong of %lld stored values (%lld in element 0); lanes hit:", name, aggress ? "next to MFMA waves" : "alone             ", bad, total, bad_e0);
        for (int l = 0; l < 64; ++l)
            if (lanes[l]) printf(" %d", l);
        printf("\n");
        fflush(stdout);
    }
}

int main() {
    const int iters = 400;
    const size_t n = 4096 + (size_t)256 * 4 * iters * 64 * 4;
    float* out;
    hipMalloc(&out, n * 4);
    run<true, false>("SGPR soffset, VALU write right behind the store", out, n, iters);
    run<false, false>("soffset 0, VALU write right behind the store", out, n, iters);
    run<false, true>("soffset 0, s_nop 1 in between (compiler's fix)", out, n, iters);
    return 0;
}
