// Main-loop model of a 256x256 bf16 GEMM tile on one CU: LDS fragment reads + bf16 MFMAs (+ LDS-DMA staging + one barrier
// per K tile of 64), no epilogue, no real global traffic (the DMA source is a 64 KB buffer that stays in L2).
//   MODE 0: 8 waves, wave tile 64 x 128, v_mfma_f32_16x16x32_bf16  (the production kernels' tiling): 24 KB of LDS reads per
//           wave and K tile -> 192 KB per CU and K tile against 2048 MFMA cycles
//   MODE 1: 4 waves, wave tile 128 x 128, v_mfma_f32_32x32x16_bf16: 32 KB per wave -> 128 KB per CU and K tile
// FLAGS: 1 = s_barrier per K tile, 2 = 64 KB of LDS-DMA per K tile (2 stages, counted vmcnt)
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lds_bf16 tools/micro/mfma_lds_bf16.hip && /tmp/mfma_lds_bf16
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kStage = 512 * 128;  // bytes of one stage: (256 + 256) rows x 64 bf16

template <int MODE, int FLAGS>
__global__ __launch_bounds__(MODE == 0 ? 512 : 256) void k(float* out, int iters, const char* src) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NT = MODE == 0 ? 512 : 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * kStage / 4; i += NT) reinterpret_cast<float*>(lds)[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    // rows are 128 B (64 bf16) = 8 chunks of 16 B; chunk c of row r lives at chunk c ^ ((r >> 1) & 7)
    constexpr int TM = 4, TN = MODE == 0 ? 8 : 4;            // fragments per operand and k sub-step
    constexpr int RB = MODE == 0 ? 16 : 32;                  // rows per fragment
    constexpr int SUB = MODE == 0 ? 2 : 4;                   // k sub-steps per K tile (32 / 16 wide)
    const int row = MODE == 0 ? (lane & 15) : (lane & 31);
    const int kc = MODE == 0 ? (lane >> 4) : (lane >> 5);    // this lane's 16-B chunk within a sub-step
    const int a_row0 = MODE == 0 ? (wave >> 1) * 64 : (wave >> 1) * 128;
    const int b_row0 = 256 + (MODE == 0 ? (wave & 1) * 128 : (wave & 1) * 128);
    auto addr = [&](int r, int chunk) { return r * 128 + ((chunk ^ ((r >> 1) & 7)) * 16); };
    bf16x8 af[2][TM], bfr[2][TN];
    auto rd = [&](int buf, int stage, int sub) {
        const int chunk = sub * (MODE == 0 ? 4 : 2) + kc;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[buf][i] = *reinterpret_cast<const bf16x8*>(lds + stage * kStage + addr(a_row0 + i * RB + row, chunk));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[buf][j] = *reinterpret_cast<const bf16x8*>(lds + stage * kStage + addr(b_row0 + j * RB + row, chunk));
    };
    f32x4 acc4[MODE == 0 ? TM : 1][MODE == 0 ? TN : 1];
    f32x16 acc16[MODE == 1 ? TM : 1][MODE == 1 ? TN : 1];
    for (auto& r : acc4) for (auto& v : r) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (auto& r : acc16) for (auto& v : r) for (int e = 0; e < 16; ++e) v[e] = 0.f;
    auto mm = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (MODE == 0) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[buf][i], bfr[buf][j], acc4[i][j], 0, 0, 0);
                else acc16[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[buf][i], bfr[buf][j], acc16[i][j], 0, 0, 0);
            }
    };
    constexpr int DMA_PER_THREAD = kStage / (NT * 16);  // 8 (512 threads) or 16 (256 threads)
    auto dma = [&](int stage) {
#pragma unroll
        for (int d = 0; d < DMA_PER_THREAD; ++d)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(d * NT + tid) * 16), (lptr_t)(lds + stage * kStage + (d * NT + wave * 64) * 16), 16, 0, 0);
    };
    if (FLAGS & 2) dma(1);
    if (!(FLAGS & 4)) rd(0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const int stage = it & 1;
        if (FLAGS & 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stage about to be read has landed (issued a K tile ago)
        }
        if (FLAGS & 1) __builtin_amdgcn_s_barrier();
        if (FLAGS & 2) dma(stage ^ 1);  // refill the stage consumed in the previous K tile
        if (FLAGS & 4) rd(0, stage, 0);   // legal order: a K tile's first fragments only after its wait + barrier
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            if (sub + 1 < SUB) rd((sub + 1) & 1, stage, sub + 1);
            else if (!(FLAGS & 4)) rd((sub + 1) & 1, stage ^ 1, 0);   // first fragments of the next K tile (model: no wait for its DMA)
            asm volatile("" ::: "memory");
            if (FLAGS & 8) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_setprio(1); }
            mm(sub & 1);
            if (FLAGS & 8) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_barrier(); }
        }
    }
    float s = 0.f;
    for (auto& r : acc4) for (auto& v : r) s += v[0] + v[3];
    for (auto& r : acc16) for (auto& v : r) s += v[0] + v[15];
    if (s == 12345.678f) out[tid] = s;
}

// MODE 2: 4 waves, workgroup tile 128 x 256 (wave tile 64 x 128, 16x16x32), K tiles of 32 (64-B rows), 72 KB of LDS so that TWO
// workgroups share a CU (8 waves per CU as in MODE 0, but each workgroup has its own barrier and its own prologue / epilogue).
constexpr int kStage2 = 384 * 64;  // (128 + 256) rows x 32 bf16
template <int FLAGS>
__global__ __launch_bounds__(256) void k2(float* out, int iters, const char* src) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * kStage2 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    const int row = lane & 15, kc = lane >> 4;
    const int a_row0 = (wave >> 1) * 64, b_row0 = 128 + (wave & 1) * 128;
    auto addr = [&](int r) { return r * 64 + ((kc ^ ((r >> 2) & 3)) * 16); };
    bf16x8 af[2][4], bfr[2][8];
    auto rd = [&](int buf, int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[buf][i] = *reinterpret_cast<const bf16x8*>(lds + stage * kStage2 + addr(a_row0 + i * 16 + row));
#pragma unroll
        for (int j = 0; j < 8; ++j) bfr[buf][j] = *reinterpret_cast<const bf16x8*>(lds + stage * kStage2 + addr(b_row0 + j * 16 + row));
    };
    f32x4 acc[4][8];
    for (auto& r : acc) for (auto& v : r) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto dma = [&](int stage) {
#pragma unroll
        for (int d = 0; d < 6; ++d)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(d * 256 + tid) * 16), (lptr_t)(lds + stage * kStage2 + (d * 256 + wave * 64) * 16), 16, 0, 0);
    };
    // 3 stages: K tile it is read from stage it % 3 while the DMA of K tile it + 2 lands in stage (it + 2) % 3
    if (FLAGS & 2) { dma(0); dma(1); }
    int stage = 0;
    for (int it = 0; it < iters; ++it) {
        if (FLAGS & 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K tile it has landed, it + 1 may be in flight
        if (FLAGS & 1) __builtin_amdgcn_s_barrier();
        if (FLAGS & 2) dma(stage == 0 ? 2 : stage - 1);
        rd(0, stage);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
        stage = stage == 2 ? 0 : stage + 1;
    }
    float s = 0.f;
    for (auto& r : acc) for (auto& v : r) s += v[0] + v[3];
    if (s == 12345.678f) out[tid] = s;
}

template <int FLAGS>
void run2(const char* name, float* out, const char* src) {
    const int iters = 4000, blocks = 256 * 4;
    auto kern = k2<FLAGS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kStage2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 3 * kStage2, 0, out, iters, src);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * iters * 128.0 * 256 * 32 * 2;
    printf("%-70s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

template <int MODE, int FLAGS>
void run(const char* name, float* out, const char* src) {
    const int iters = 2000, blocks = 256 * 2;
    auto kern = k<MODE, FLAGS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStage);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(MODE == 0 ? 512 : 256), 2 * kStage, 0, out, iters, src);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * iters * 256.0 * 256 * 64 * 2;
    printf("%-70s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    char* src;
    hipMalloc(&out, 4096);
    hipMalloc(&src, kStage);
    hipMemset(src, 0, kStage);
    run<0, 0>("8 waves, 64x128 wave tile, 16x16x32: reads + MFMA", out, src);
    run<0, 1>("8 waves, 64x128: + barrier per K tile", out, src);
    run<0, 3>("8 waves, 64x128: + barrier + 64 KB LDS-DMA per K tile", out, src);
    run<0, 7>("8 waves, 64x128: barrier + DMA, LEGAL order (fragments after the barrier)", out, src);
    run<0, 15>("8 waves, 64x128: legal order + 2 more barriers and setprio per sub-step", out, src);
    run<1, 0>("4 waves, 128x128 wave tile, 32x32x16: reads + MFMA", out, src);
    run<1, 1>("4 waves, 128x128: + barrier per K tile", out, src);
    run<1, 3>("4 waves, 128x128: + barrier + 64 KB LDS-DMA per K tile", out, src);
    run2<0>("2 WGs/CU x 4 waves, 128x256 tile, 64x128 wave tile, BK 32: reads + MFMA", out, src);
    run2<1>("2 WGs/CU x 4 waves: + barrier per K tile of 32", out, src);
    run2<3>("2 WGs/CU x 4 waves: + barrier + 24 KB LDS-DMA per K tile, 3 stages", out, src);
    return 0;
}
