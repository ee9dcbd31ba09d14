// bf16 GEMM, 256 x 256 tile: the schedule of gemm_bf16_8phase.hip.h with TWO long phases per K tile and three B buffers.
//
// Why (profiles/r03_gemm_bf16_loop_probes.txt): with every load removed the 8-phase loop - 16-MFMA clusters, a barrier before
// and after each - tops out at 1.68 PFLOP/s on K = 4096 (67 % of the matrix peak): a cluster keeps the pipe busy for 256 cycles
// and the hand-over to the other wave row costs ~100.  Here a K tile (BK = 64) is two clusters of 32 MFMAs:
//     phase 1: read A rows 0..127 of the wave row's half (16 x ds_read_b128) + B columns 0..31 (4)    -> 8 x 2 tiles x K = 64
//     phase 2: read B columns 32..63 (4), stage B of tile t+2, wait "tile t+1 has landed"              -> 8 x 2 tiles x K = 64
//              (behind the phase's first barrier: stage A of tile t+2)
// i.e. 4 barriers per K tile instead of 8, the same wave tile (128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16), the
// same fragments in registers (A of the whole K tile was resident already), the same ping-pong of the two wave rows.
// LDS: A0 A1 (32 KB each) | B0 B1 B2 = 160 KB.  B needs the third buffer because it is read in both phases: the buffer of tile
// t+2 is the one tile t-1 used.  Every wave stages its share of A and B; waits are counted in issue order:
//     ... B(t+1) [phase 2 of t-1, before its barrier], A(t+1) [behind that barrier], B(t+2) [phase 2 of t], s_waitcnt vmcnt(4)
// Hazards (rows run one barrier apart; "a" / "b" = the barrier before / after a cluster):
//   A buffer t & 1: read in phase 1 of tile t; re-staged behind 2a(t) = the other row's 1b(t) (row 0) / 2b(t) (row 1): both after
//     the other row's phase-1 reads were waited for (lgkmcnt(0) precedes its MFMAs).
//   B buffer (t+2) % 3: last read in phase 2 of tile t-1; re-staged in phase 2 of tile t, behind 1b(t).
//   tile t+1 is read from phase 1 of tile t+1 on, behind 2b(t); every wave's wait precedes its 2a(t) = the other row's 1b(t) or 2b(t).
// MFMA order per accumulator (kh = 0, 1 of each K tile, K tiles ascending) is that of the 8-phase kernel: bit-identical results.
// Requirements: N % 256 == 0, K % 128 == 0.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_bf16_8phase.hip.h"

namespace nomad {

// ABL: 1 = no epilogue stores, 4 = no LDS-DMA, 5 = neither DMA nor LDS reads (timing probes).  NTS: non-temporal output stores.
template <int ABL = 0, int NTS = 1>
__global__ __launch_bounds__(512) void gemm_bf16_p2_kernel(const GemmParams p) {
    using Cfg = P8Cfg;
    constexpr int A_BUF = 2 * Cfg::HALF_BYTES, B_BASE = 2 * A_BUF, B_BUF = 2 * Cfg::HALF_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_coords(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
    const int m0 = tile_m * Cfg::BM, n0 = tile_n * Cfg::BN;
    const int grp = blockIdx.y;
    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(p.A) + grp * p.a_goff;
    const bf16_t* Wg = reinterpret_cast<const bf16_t*>(p.W) + grp * p.w_goff;

    // DMA sources: (wave-uniform 64-bit base) + (per-lane 32-bit byte offset), as in the 8-phase kernel
    const long long tile_row0 = row_addr(p.amap, m0 < p.M ? m0 : p.M - 1);
    unsigned a_off[2][2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 512, row = id >> 3, pc = id & 7;
        const int sw = (pc ^ ((row >> 1) & 7)) * 8;
        b_off[i] = (unsigned)(((long long)row * p.ldw + sw) * 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + h * 128 + row;
            m = m < p.M ? m : p.M - 1;
            a_off[h][i] = (unsigned)((row_addr(p.amap, m) - tile_row0 + sw) * 2);
        }
    }
    const char* const a_base = reinterpret_cast<const char*>(Ag + tile_row0);
    const char* const b_base[2] = {reinterpret_cast<const char*>(Wg + (long long)n0 * p.ldw),
                                   reinterpret_cast<const char*>(Wg + (long long)(n0 + 128) * p.ldw)};
    char* const dma_dst = smem8 + wave * 1024;  // + lane * 16 implicit (lane-linear LDS-DMA destination)

#define NOMAD_P2_DMA_A(KT)                                                                                          \
    if (ABL != 4 && ABL != 5) {                                                                                     \
        const int k0_ = (KT)*64;                                                                                    \
        const int kq_ = k0_ / p.kchunk;                                                                             \
        const unsigned ko_ = (unsigned)((kq_ * p.kstride + (k0_ - kq_ * p.kchunk)) * 2);                            \
        char* d_ = dma_dst + ((KT)&1) * A_BUF;                                                                      \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_base + (a_off[h_][0] + ko_)), (lptr_t)(d_ + h_ * Cfg::HALF_BYTES), 16, 0, 0);         \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_base + (a_off[h_][1] + ko_)), (lptr_t)(d_ + h_ * Cfg::HALF_BYTES + 8192), 16, 0, 0);  \
        }                                                                                                           \
    }
#define NOMAD_P2_DMA_B(KT, BOFF)                                                                                    \
    if (ABL != 4 && ABL != 5) {                                                                                     \
        const unsigned ko_ = (unsigned)((KT)*128);                                                                  \
        char* d_ = dma_dst + B_BASE + (BOFF);                                                                       \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_base[h_] + (b_off[0] + ko_)), (lptr_t)(d_ + h_ * Cfg::HALF_BYTES), 16, 0, 0);         \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_base[h_] + (b_off[1] + ko_)), (lptr_t)(d_ + h_ * Cfg::HALF_BYTES + 8192), 16, 0, 0);  \
        }                                                                                                           \
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;  // even, >= 2
    // prologue: tiles 0 and 1 on their way (issue order B0 A0 B1 A1), tile 0 complete before the first barrier
    NOMAD_P2_DMA_B(0, 0)
    NOMAD_P2_DMA_A(0)
    NOMAD_P2_DMA_B(1, B_BUF)
    NOMAD_P2_DMA_A(1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();  // second wave row runs one barrier behind (ping-pong)

    // fragment addresses: this lane's row fr of a 16-row tile, chunk (4 kh + fq) ^ swizzle
    const int sw = (fr >> 1) & 7;
    const int koff0 = ((0 + fq) ^ sw) * 16, koff1 = ((4 + fq) ^ sw) * 16;
    const int a_frag = wr * Cfg::HALF_BYTES + fr * 128;                                         // + i * 2048
    const int b_frag = B_BASE + (wc >> 1) * Cfg::HALF_BYTES + ((wc & 1) * 64 + fr) * 128;       // + j * 2048

    bf16x8 af[8][2], bf[2][2];
    if (ABL == 5) {  // timing probe without fragment loads: keep the registers defined
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i][0] = af[i][1] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j][0] = bf[j][1] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1};
    }
    // 32 MFMAs: the 8 row tiles x 2 column tiles J0, J0 + 1 x K = 64; per accumulator kh = 0 then 1, as in the 8-phase kernel
#define NOMAD_P2_MMA(J0)                                                                                   \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                      \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
                acc[i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kh], bf[j][kh], acc[i][(J0) + j], 0, 0, 0);

    // one K tile; BUF = KT & 1 is a constant per call site, the B buffers rotate (b_cur = offset of tile t's)
#define NOMAD_P2_KTILE(KT, BUF)                                                                            \
    {                                                                                                      \
        const char* la_ = smem8 + (BUF)*A_BUF + a_frag;                                                    \
        const char* lb_ = smem8 + b_cur + b_frag;                                                          \
        const int b_nxt2_ = b_cur >= B_BUF ? b_cur - B_BUF : b_cur + 2 * B_BUF; /* buffer of tile t+2 */   \
        /* phase 1: all of A, B columns 0..31 */                                                           \
        if (ABL != 5) {                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                \
                bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff0);                       \
                bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff1);                       \
            }                                                                                              \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                \
                af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                       \
                af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                       \
            }                                                                                              \
        }                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                     \
        NOMAD_P2_MMA(0)                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                     \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                     \
        /* phase 2: B columns 32..63; B of tile t+2; "tile t+1 has landed" */                              \
        if (ABL != 5) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                      \
            bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff0);                     \
            bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff1);                     \
        }                                                                                                  \
        if ((KT) + 2 < nk) {                                                                               \
            NOMAD_P2_DMA_B((KT) + 2, b_nxt2_)                                                              \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                               \
        } else {                                                                                           \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
        }                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                     \
        if ((KT) + 2 < nk) NOMAD_P2_DMA_A((KT) + 2) /* behind 2a: the other row's phase-1 reads of this buffer are done */ \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                     \
        NOMAD_P2_MMA(2)                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                     \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                     \
        b_cur = b_cur >= 2 * B_BUF ? 0 : b_cur + B_BUF;                                                    \
    }

    int b_cur = 0;
    for (int kt = 0; kt < nk; kt += 2) {
        NOMAD_P2_KTILE(kt, 0)
        NOMAD_P2_KTILE(kt + 1, 1)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
#undef NOMAD_P2_KTILE
#undef NOMAD_P2_MMA
#undef NOMAD_P2_DMA_A
#undef NOMAD_P2_DMA_B

    p8_epilogue<ABL == 1, 0, NTS>(p, acc, smem8, grp, m0, n0, wave, wr, wc, lane, fr, fq);
}

template <int ABL = 0, int NTS = 1>
inline hipError_t launch_gemm_bf16_p2(GemmParams p, int groups, hipStream_t s) {
    p.tiles_m = (p.M + P8Cfg::BM - 1) / P8Cfg::BM;
    p.tiles_n = p.N / P8Cfg::BN;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p2_kernel<ABL, NTS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_p2_kernel<ABL, NTS>), dim3(p.tiles_m * p.tiles_n, groups), dim3(P8Cfg::THREADS), 160 * 1024, s, p);
    return hipGetLastError();
}

}  // namespace nomad
