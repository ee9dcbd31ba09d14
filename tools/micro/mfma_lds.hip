// Inner-loop microbenchmark: LDS fragment reads + v_mfma_f32_32x32x2_f32, no global memory, no barriers.
//   MODE 0: wave tile 64x64 (4 accumulators): per K tile of 16: 8 ds_read_b128 then 32 MFMAs  (the production loop)
//   MODE 1: same, software pipelined: the reads of K tile t+1 are issued before the MFMAs of tile t
//   MODE 2: wave tile 128x64 (8 accumulators): 12 ds_read_b128, 64 MFMAs
//   MODE 3: MODE 2 software pipelined
// Build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lds tools/micro/mfma_lds.hip && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// FLAGS: 1 = raw s_barrier per K tile, 2 = LDS-DMA staging per K tile (3 x 16 B per thread, 3 stages, counted vmcnt)
template <int TM, int TN, bool PIPE, int FLAGS = 0>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* src = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2 stages][(256 + 128) rows][16 floats]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 384 * 16; i += blockDim.x) lds[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    const int row = lane & 31, h = lane >> 5, swz = (row >> 2) & 3;
    const float* abase = lds + ((wave >> 1) * 64 % 256 + row) * 16;
    const float* bbase = lds + (256 + (wave & 1) * 64 + row) * 16;
    int koff[2];
    for (int kq = 0; kq < 2; ++kq) koff[kq] = ((kq * 2 + h) ^ swz) * 4;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 af[2][TM][2], bf[2][TN][2];
    auto rd = [&](int buf, int stage) {
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[buf][i][kq] = *reinterpret_cast<const f32x4*>(abase + stage * 384 * 16 + (i * 32 % 64) * 16 + koff[kq]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[buf][j][kq] = *reinterpret_cast<const f32x4*>(bbase + stage * 384 * 16 + (j * 32 % 64) * 16 + koff[kq]);
        }
    };
    auto mm = [&](int buf) {
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][i][kq][c], bf[buf][j][kq][c], acc[i][j], 0, 0, 0);
    };
    if (PIPE) {
        rd(0, 0);
        for (int it = 0; it < iters; it += 2) {
            rd(1, 1);
            asm volatile("" ::: "memory");
            mm(0);
            rd(0, 0);
            asm volatile("" ::: "memory");
            mm(1);
        }
    } else if (FLAGS == 0) {
        for (int it = 0; it < iters; ++it) {
            rd(0, it & 1);
            mm(0);
            asm volatile("" ::: "memory");
        }
    } else {
        // staging area behind the two read stages: [3][384][16] floats, written by DMA only (the fragment reads keep
        // using the static image, so arithmetic stays finite)
        float* stage = lds + 2 * 384 * 16;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
        const float* g = src + ((size_t)blockIdx.x * 4096 + tid * 4) % (1 << 21);
        int cur = 0;
        for (int it = 0; it < iters; ++it) {
            if (FLAGS & 2) {
                if (it > 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            }
            if (FLAGS & 1) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            float* d = stage + cur * 384 * 16 + wave * 256;
            const float* gs = g + (size_t)(it & 1023) * 2048;
            if ((FLAGS & 2) && !(FLAGS & 4)) {
                if (FLAGS & 8) {  // buffer_load ... lds: SGPR descriptor + 32-bit offsets instead of 64-bit addresses
                    const int vo = (int)(((size_t)blockIdx.x * 4096 + tid * 4) % (1 << 21)) * 4;
                    const int so = (it & 1023) * 2048 * 4;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(d), 16, vo, so, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(d + 2048), 16, vo, so + 32768, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(d + 4096), 16, vo, so + 65536, 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        __builtin_amdgcn_global_load_lds((gptr_t)(gs + c * 8192), (lptr_t)(d + c * 2048), 16, 0, 0);
                }
            }
            rd(0, it & 1);
            if (FLAGS & 4) {  // DMA issued between the two halves of the MFMA cluster
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][i][0][c], bf[0][j][0][c], acc[i][j], 0, 0, 0);
                asm volatile("" ::: "memory");
                if (FLAGS & 8) {
                    const int vo = (int)(((size_t)blockIdx.x * 4096 + tid * 4) % (1 << 21)) * 4;
                    const int so = (it & 1023) * 2048 * 4;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(d), 16, vo, so, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(d + 2048), 16, vo, so + 32768, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(d + 4096), 16, vo, so + 65536, 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        __builtin_amdgcn_global_load_lds((gptr_t)(gs + c * 8192), (lptr_t)(d + c * 2048), 16, 0, 0);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][i][1][c], bf[0][j][1][c], acc[i][j], 0, 0, 0);
            } else {
                mm(0);
            }
            cur = cur == 2 ? 0 : cur + 1;
            asm volatile("" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int TM, int TN, bool PIPE, int FLAGS = 0>
void run(const char* name, int wgs_per_cu, int threads, float* out, const float* src = nullptr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, PIPE, FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 40000 / (TM * TN / 4);
    const int lds_bytes = 5 * 384 * 16 * 4 + (wgs_per_cu == 1 ? 40 * 1024 : 0);  // 120 KB: 1 or 2 workgroups per CU
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TM, TN, PIPE, FLAGS>), dim3(256 * wgs_per_cu), dim3(threads), lds_bytes, 0, out, iters, src);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * wgs_per_cu * (threads / 64) * iters * (TM * TN * 8.0) * (2.0 * 32 * 32 * 2);
        if (rep) printf("%-52s %d wg/CU x %2d waves: %8.2f ms  %7.1f TFLOP/s\n", name, wgs_per_cu, threads / 64, ms, fl / ms / 1e9);
    }
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 1024 * 512);
    run<2, 2, false>("64x64 wave tile, read-then-MFMA (production)", 2, 512, out);
    run<2, 2, true>("64x64 wave tile, software pipelined reads", 2, 512, out);
    run<2, 2, false>("64x64 wave tile, read-then-MFMA", 1, 512, out);
    run<2, 2, true>("64x64 wave tile, software pipelined reads", 1, 512, out);
    run<4, 2, false>("128x64 wave tile, read-then-MFMA", 1, 512, out);
    run<4, 2, true>("128x64 wave tile, software pipelined reads", 1, 512, out);
    run<4, 2, false>("128x64 wave tile, read-then-MFMA", 1, 256, out);
    run<4, 2, true>("128x64 wave tile, software pipelined reads", 1, 256, out);
    run<2, 2, false>("64x64 wave tile, read-then-MFMA (production)", 2, 512, out);
    float* src;
    hipMalloc(&src, sizeof(float) * ((1 << 21) + 1024 * 2048 + 3 * 8192 + 4096));
    hipMemset(src, 0, sizeof(float) * ((1 << 21) + 1024 * 2048 + 3 * 8192 + 4096));
    run<2, 2, false, 1>("64x64, + s_barrier per K tile", 2, 512, out, src);
    run<2, 2, false, 2>("64x64, + LDS-DMA per K tile (counted vmcnt)", 2, 512, out, src);
    run<2, 2, false, 3>("64x64, + barrier + LDS-DMA (the production main loop)", 2, 512, out, src);
    run<2, 2, false, 7>("64x64, + barrier + LDS-DMA issued mid-cluster", 2, 512, out, src);
    run<2, 2, false, 3>("64x64, + barrier + LDS-DMA (the production main loop)", 2, 512, out, src);
    run<2, 2, false, 7>("64x64, + barrier + LDS-DMA issued mid-cluster", 2, 512, out, src);
    run<2, 2, false, 15>("64x64, + barrier + buffer_load..lds issued mid-cluster (production now)", 2, 512, out, src);
    run<2, 2, false, 10>("64x64, + buffer_load..lds per K tile", 2, 512, out, src);
    run<2, 2, false, 11>("64x64, + barrier + buffer_load..lds", 2, 512, out, src);
    run<2, 2, false, 3>("64x64, + barrier + LDS-DMA (global_load_lds)", 2, 512, out, src);
    run<2, 2, false, 11>("64x64, + barrier + buffer_load..lds", 2, 512, out, src);
    run<2, 2, false, 3>("64x64, + barrier + LDS-DMA, 1 workgroup/CU", 1, 512, out, src);
    run<4, 2, false, 3>("128x64, + barrier + LDS-DMA, 1 workgroup/CU x 8 waves", 1, 512, out, src);
    return 0;
}
