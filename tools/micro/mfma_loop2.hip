// Main-loop microbenchmark, round 4: which K-loop SKELETON reaches the fp32 MFMA rate?
// Same ingredients as tools/micro/mfma_lds.hip (swizzled LDS fragment reads, v_mfma_f32_32x32x2_f32, raw s_barrier, counted vmcnt,
// buffer_load ... lds staging from an L2-resident source; no prologue, no epilogue, no HBM), but the wave count, the wave tile, the
// K depth per barrier and the software pipelining of the fragment reads are template parameters:
//   TM x TN   32x32 accumulators per wave (2x2 = 64x64 wave tile, 4x2 = 128x64)
//   WM x WN   waves per workgroup; workgroup tile = (WM TM 32) x (WN TN 32)
//   BK        K depth per barrier (16 / 32 / 64); ST LDS stages
//   MODE 0    production order: vmcnt + barrier at the top of a K tile, per k-step of 8: reads then MFMAs, DMA after the first step
//   MODE 1    fragments of step s+1 read before the MFMAs of step s (inside a tile); the first step of a tile is read after the barrier
//   MODE 2    as 1, and the barrier sits in front of the LAST step of a tile: the first fragments of tile t+1 are read behind it, under
//             the last step's MFMAs - a wave waits for the barrier and for nothing else
// Build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_loop2 tools/micro/mfma_loop2.hip && /tmp/mfma_loop2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
#define FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int TM, int TN, int WM, int WN, int BK, int ST, int MODE, int MINB>
__global__ __launch_bounds__(WM* WN * 64, MINB) void k(float* out, int iters, const float* src) {
    constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, ROWS = BM + BN;
    constexpr int KC = BK / 4, NKQ = BK / 8;
    constexpr int RB = KC >= 16 ? 1 : 16 / KC;           // rows per 256-byte bank row
    constexpr int DMA = ROWS * KC / NT;                   // 16-byte DMA instructions per thread and K tile
    static_assert(ROWS * KC % NT == 0, "staging must divide");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [ST][ROWS][BK]: read by the fragments, refilled by the DMA (finite source data)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < ST * ROWS * BK; i += NT) lds[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    const int wm = wave / WN, wn = wave % WN;
    const int row = lane & 31, h = lane >> 5;
    const int swz = (row / RB) % KC;
    const float* abase = lds + (wm * TM * 32 + row) * BK;
    const float* bbase = lds + (BM + wn * TN * 32 + row) * BK;
    int koff[NKQ];
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq) koff[kq] = ((kq * 2 + h) ^ swz) % KC * 4;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 af[2][TM], bf[2][TN];
    auto rd = [&](int buf, int img, int kq) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[buf][i] = *reinterpret_cast<const f32x4*>(abase + img * ROWS * BK + i * 32 * BK + koff[kq]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[buf][j] = *reinterpret_cast<const f32x4*>(bbase + img * ROWS * BK + j * 32 * BK + koff[kq]);
    };
    auto mm = [&](int buf, int c0, int c1) {
#pragma unroll
        for (int c = c0; c < c1; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][i][c], bf[buf][j][c], acc[i][j], 0, 0, 0);
    };
    float* stage = lds;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int vo = (int)(((size_t)blockIdx.x * 4096 + tid * 4) % (1 << 21)) * 4;
    int cur = MODE == 2 ? 0 : ST - 1, rs = 0;   // stage the DMA fills next / stage the fragments are read from
    auto dma = [&](int it) {
        float* d = stage + cur * ROWS * BK + wave * 256;
        const int so = (it & 1023) * 2048 * 4;
#pragma unroll
        for (int c = 0; c < DMA; ++c) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(d + c * NT * 4), 16, vo, so + c * 32768, 0, 0);
        cur = cur + 1 == ST ? 0 : cur + 1;
    };
    auto sync = [&](int it) {
        if (it > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA * (ST - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
    };
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            sync(it);
#pragma unroll
            for (int kq = 0; kq < NKQ; ++kq) {
                rd(0, rs, kq);
                if (kq == 1 || NKQ == 1) {
                    FENCE();
                    dma(it);
                }
                mm(0, 0, 4);
                FENCE();
            }
            rs = rs + 1 == ST ? 0 : rs + 1;
        }
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
            sync(it);
            rd(0, rs, 0);
#pragma unroll
            for (int kq = 0; kq < NKQ; ++kq) {
                if (kq + 1 < NKQ) rd((kq + 1) & 1, rs, kq + 1);
                FENCE();
                mm(kq & 1, 0, 1);
                if (kq == 0) {
                    FENCE();
                    dma(it);
                    FENCE();
                }
                mm(kq & 1, 1, 4);
                FENCE();
            }
            rs = rs + 1 == ST ? 0 : rs + 1;
        }
    } else {
        rd(0, 0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kq = 0; kq < NKQ; ++kq) {
                // fragments of step kq are in buffer (kq & 1); NKQ is even, so a tile's first step is always in buffer 0
                if (kq + 1 < NKQ) {
                    rd((kq + 1) & 1, rs, kq + 1);
                    FENCE();
                    mm(kq & 1, 0, 4);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave is done reading tile `it`
                    sync(it);
                    rs = rs + 1 == ST ? 0 : rs + 1;
                    rd(0, rs, 0);
                    FENCE();
                    mm(kq & 1, 0, 1);
                    FENCE();
                    dma(it);
                    FENCE();
                    mm(kq & 1, 1, 4);
                }
                FENCE();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * NT + tid] = s;
}

template <int TM, int TN, int WM, int WN, int BK, int ST, int MODE, int MINB>
void run(const char* name, int wgs_per_cu, float* out, const float* src) {
    constexpr int NT = WM * WN * 64, ROWS = (WM * TM + WN * TN) * 32;
    auto fn = k<TM, TN, WM, WN, BK, ST, MODE, MINB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 160000 / (TM * TN * (BK / 16));
    int lds_bytes = ST * ROWS * BK * 4;
    const int budget = (160 * 1024 / wgs_per_cu) & ~255;
    if (lds_bytes > budget) {
        printf("%-64s does not fit %d wg/CU (%d KB)\n", name, wgs_per_cu, lds_bytes >> 10);
        return;
    }
    if (wgs_per_cu < 8 && lds_bytes <= (160 * 1024 / (wgs_per_cu + 1))) lds_bytes = ((160 * 1024 / (wgs_per_cu + 1)) & ~255) + 256;  // pin the residency
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn));
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(256 * wgs_per_cu), dim3(NT), lds_bytes, 0, out, iters, src);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * wgs_per_cu * (NT / 64) * (double)iters * (TM * TN * (BK / 2)) * (2.0 * 32 * 32 * 2);
        if (rep)
            printf("%-64s %d wg/CU x %2d waves, %3d VGPR %2d KB: %8.2f ms  %7.1f TFLOP/s\n", name, wgs_per_cu, NT / 64, fa.numRegs, lds_bytes >> 10, ms,
                   fl / ms / 1e9);
    }
    fflush(stdout);
}

int main() {
    float *out, *src;
    hipMalloc(&out, sizeof(float) * 2048 * 1024);
    const size_t n = (size_t)(1 << 21) + 1024 * 2048 + 16 * 8192 + 4096;
    hipMalloc(&src, sizeof(float) * n);
    hipMemset(src, 0x3c, sizeof(float) * n);   // 0x3c3c3c3c = 0.0115f
    // production skeleton and its pipelined forms: 256 x 128 tile, 8 waves of 64 x 64, BK 16, 3 stages, 2 workgroups / CU
    run<2, 2, 4, 2, 16, 3, 0, 4>("256x128 8w 64x64 BK16 mode0 (production skeleton)", 2, out, src);
    run<2, 2, 4, 2, 16, 3, 1, 4>("256x128 8w 64x64 BK16 mode1 (reads pipelined in the tile)", 2, out, src);
    run<2, 2, 4, 2, 16, 3, 2, 4>("256x128 8w 64x64 BK16 mode2 (+ barrier before the last step)", 2, out, src);
    // the same tile with 4 waves of 128 x 64 (256 VGPRs, 2 waves / SIMD)
    run<4, 2, 2, 2, 16, 3, 0, 2>("256x128 4w 128x64 BK16 mode0", 2, out, src);
    run<4, 2, 2, 2, 16, 3, 1, 2>("256x128 4w 128x64 BK16 mode1", 2, out, src);
    run<4, 2, 2, 2, 16, 3, 2, 2>("256x128 4w 128x64 BK16 mode2", 2, out, src);
    run<2, 4, 2, 2, 16, 3, 2, 2>("128x256 4w 64x128 BK16 mode2", 2, out, src);
    // 128 x 128 tile, 4 waves of 64 x 64: BK 16 / 32 / 64 (the vendor's macro tile and wave count)
    run<2, 2, 2, 2, 16, 3, 0, 2>("128x128 4w 64x64 BK16 mode0 3 wg/CU", 3, out, src);
    run<2, 2, 2, 2, 16, 3, 2, 2>("128x128 4w 64x64 BK16 mode2 3 wg/CU", 3, out, src);
    run<2, 2, 2, 2, 32, 2, 0, 2>("128x128 4w 64x64 BK32 2-stage mode0 2 wg/CU", 2, out, src);
    run<2, 2, 2, 2, 32, 2, 2, 2>("128x128 4w 64x64 BK32 2-stage mode2 2 wg/CU", 2, out, src);
    run<2, 2, 2, 2, 32, 3, 2, 2>("128x128 4w 64x64 BK32 3-stage mode2 2 wg/CU", 2, out, src);
    run<2, 2, 2, 2, 64, 2, 2, 2>("128x128 4w 64x64 BK64 2-stage mode2 1 wg/CU", 1, out, src);
    // 256 x 128, 8 waves, deeper K per barrier
    run<2, 2, 4, 2, 32, 2, 0, 4>("256x128 8w 64x64 BK32 2-stage mode0 1 wg/CU", 1, out, src);
    run<2, 2, 4, 2, 32, 2, 2, 4>("256x128 8w 64x64 BK32 2-stage mode2 1 wg/CU", 1, out, src);
    run<4, 2, 2, 2, 32, 2, 2, 2>("256x128 4w 128x64 BK32 2-stage mode2 1 wg/CU", 1, out, src);
    // 256 x 256 with 8 waves of 128 x 64 (one workgroup per CU): half the DMA bytes per MFMA
    run<4, 2, 2, 4, 16, 3, 0, 2>("256x256 8w 128x64 BK16 mode0 1 wg/CU", 1, out, src);
    run<4, 2, 2, 4, 16, 3, 2, 2>("256x256 8w 128x64 BK16 mode2 1 wg/CU", 1, out, src);
    // repeat the reference line (clock drift check)
    run<2, 2, 4, 2, 16, 3, 0, 4>("256x128 8w 64x64 BK16 mode0 (production skeleton)", 2, out, src);
    return 0;
}
