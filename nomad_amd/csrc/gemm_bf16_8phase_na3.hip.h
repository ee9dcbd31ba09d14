// bf16 GEMM, 256 x 256 tile, the 8-phase schedule of gemm_bf16_8phase.hip.h with THREE A buffers: A is staged two K tiles
// ahead instead of one.  For the shapes whose A panel is shared by only 2-3 column tiles (fc2: N = 768, K = 3072; the conv
// layers: N = 512) every A line is a miss of its XCD's L2 for the first workgroup that asks for it, and with one K tile of
// lead (~1.5 us) the loop waits for HBM: a probe build in which every workgroup stages A tile 0 runs those shapes 12-13 %
// faster (profiles/r03_gemm_bf16_na3.txt).
//
// Why not simply a third buffer in the old kernel: s_waitcnt vmcnt counts a wave's memory operations IN ISSUE ORDER, and
// there every wave issues A and B pieces alternately - waiting for B of tile t+1 (issued during tile t) also waits for the
// A of tile t+2 issued before it, whatever buffer it went to.  Here the two wave rows split the roles, so that each wave's
// counter sees one operand only:
//   * waves 0-3 (wave row 0) stage A: both halves of tile t+3 in phase 4 of tile t (8 DMA instructions per lane), then
//     s_waitcnt vmcnt(16) = "everything but tiles t+2 and t+3 has landed";
//   * waves 4-7 (wave row 1) stage B: tile t+1 in phases 1 and 2 of tile t (4 + 4), s_waitcnt vmcnt(0) in phase 4.
// The barrier that follows each wait publishes the landed tile to the other row, exactly as before.
// LDS = 3 x A (256 x 64) + 2 x B (256 x 64) bf16 = 160 KB, all of it; the A buffer of a K tile rotates (t mod 3, a scalar
// offset), the B buffer alternates (compile-time constant per call site).  Phases, MFMA order, ping-pong of the two wave
// rows, swizzle and epilogue are those of gemm_bf16_8phase.hip.h, so the results are bit-identical to it.
//   Hazards: A buffer t mod 3 is last read in phase 2 of tile t (both rows; row 1 one barrier later) and re-staged by row 0
//   in its phase 4 of tile t, i.e. after its barrier 3b = row 1's 3a, behind row 1's phase-2 reads.  B buffer (t+1) & 1 is
//   last read in phase 3 of tile t-1 and re-staged by row 1 in its phase 1 of tile t, after its barrier 4b(t-1) = row 0's
//   1a(t), behind row 0's phase-3 reads of tile t-1.
// Requirements: N % 256 == 0, K % 128 == 0.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_bf16_8phase.hip.h"

namespace nomad {

struct P8N3Cfg {
    static constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
    static constexpr int HALF_BYTES = 128 * 128;        // 128 rows x 64 bf16
    static constexpr int A_BUF = 2 * HALF_BYTES;        // one K tile of A
    static constexpr int B_BASE = 3 * A_BUF;
    static constexpr int B_BUF = 2 * HALF_BYTES;
    static constexpr int LDS_BYTES = B_BASE + 2 * B_BUF;  // 160 KB
};

// ABL: 1 = no epilogue stores (timing only).  NTS: non-temporal output stores (p8_epilogue).
template <int ABL = 0, int NTS = 1>
__global__ __launch_bounds__(512) void gemm_bf16_8phase_na3_kernel(const GemmParams p) {
    using Cfg = P8N3Cfg;
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
    const bool a_loader = wr == 0;  // wave-uniform role

    // DMA sources as (wave-uniform 64-bit base) + (per-lane 32-bit byte offset), as in the two-buffer kernel.  A role's 256
    // lanes cover 32 rows x 8 chunks per instruction: instruction i of half h = rows 32 i + lid / 8, physical chunk lid % 8.
    const int lid = tid & 255;
    const long long tile_row0 = row_addr(p.amap, m0 < p.M ? m0 : p.M - 1);
    unsigned off[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 32 * i + (lid >> 3), pc = lid & 7;
            const int sw = (pc ^ ((row >> 1) & 7)) * 8;
            if (a_loader) {
                int m = m0 + h * 128 + row;
                m = m < p.M ? m : p.M - 1;
                off[h][i] = (unsigned)((row_addr(p.amap, m) - tile_row0 + sw) * 2);
            } else {
                off[h][i] = (unsigned)(((long long)(h * 128 + row) * p.ldw + sw) * 2);
            }
        }
    const char* const src_base = a_loader ? reinterpret_cast<const char*>(Ag + tile_row0)
                                          : reinterpret_cast<const char*>(Wg + (long long)n0 * p.ldw);
    char* const dma_dst = smem8 + (wave & 3) * 1024;  // + lane * 16 implicit (lane-linear LDS-DMA destination)

    // one half-tile (128 rows): 4 instructions per lane of the role that owns the operand
#define NOMAD_N3_DMA_HALF(KO, DST, H)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                               \
        __builtin_amdgcn_global_load_lds((gptr_t)(src_base + (off[H][i_] + (KO))), (lptr_t)((DST) + i_ * 4096), 16, 0, 0);
#define NOMAD_N3_DMA_A(KT, ABUF)                                                                                   \
    {                                                                                                              \
        const int k0_ = (KT)*64;                                                                                   \
        const int kq_ = k0_ / p.kchunk;                                                                            \
        const unsigned ko_ = (unsigned)((kq_ * p.kstride + (k0_ - kq_ * p.kchunk)) * 2);                           \
        char* d_ = dma_dst + (ABUF);                                                                               \
        NOMAD_N3_DMA_HALF(ko_, d_, 0)                                                                              \
        NOMAD_N3_DMA_HALF(ko_, d_ + Cfg::HALF_BYTES, 1)                                                            \
    }
#define NOMAD_N3_DMA_B(KT, H)                                                                                      \
    {                                                                                                              \
        const unsigned ko_ = (unsigned)((KT)*128);                                                                 \
        char* d_ = dma_dst + Cfg::B_BASE + ((KT)&1) * Cfg::B_BUF + (H)*Cfg::HALF_BYTES;                            \
        NOMAD_N3_DMA_HALF(ko_, d_, H)                                                                              \
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;  // even, >= 2
    // prologue: A of tiles 0, 1, 2 and B of tile 0 on their way; tile 0 complete before the first barrier
    if (a_loader) {
        NOMAD_N3_DMA_A(0, 0)
        NOMAD_N3_DMA_A(1, Cfg::A_BUF)
        if (nk > 2) {
            NOMAD_N3_DMA_A(2, 2 * Cfg::A_BUF)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    } else {
        NOMAD_N3_DMA_B(0, 0)
        NOMAD_N3_DMA_B(0, 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();  // second wave row runs one barrier behind (ping-pong)

    // fragment addresses: this lane's row fr of a 16-row tile, chunk (4 kh + fq) ^ swizzle
    const int sw = (fr >> 1) & 7;
    const int koff0 = ((0 + fq) ^ sw) * 16, koff1 = ((4 + fq) ^ sw) * 16;
    const int a_frag = wr * Cfg::HALF_BYTES + fr * 128;                                              // + i * 2048
    const int b_frag = Cfg::B_BASE + (wc >> 1) * Cfg::HALF_BYTES + ((wc & 1) * 64 + fr) * 128;       // + j * 2048

    bf16x8 af[8][2], bf[2][2];
#define NOMAD_N3_MMA(I0, J0)                                                                               \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
                acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[(I0) + i][kh], bf[j][kh], acc[(I0) + i][(J0) + j], 0, 0, 0);
#define NOMAD_N3_SYNC_COMPUTE(I0, J0)                   \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_setprio(1);                      \
    NOMAD_N3_MMA(I0, J0)                                \
    __builtin_amdgcn_s_setprio(0);                      \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");

    // one K tile: A buffer at the scalar offset abuf (rotating), B buffer BUF = KT & 1 (a constant per call site)
#define NOMAD_N3_KTILE(KT, BUF)                                                                            \
    {                                                                                                      \
        const char* la_ = smem8 + abuf + a_frag;                                                           \
        const char* lb_ = smem8 + (BUF)*Cfg::B_BUF + b_frag;                                               \
        /* phase 1: B columns 0..31, A rows 0..63 */                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
            bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff0);                           \
            bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff1);                           \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
            af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                           \
            af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                           \
        }                                                                                                  \
        if (!a_loader && (KT) + 1 < nk) NOMAD_N3_DMA_B((KT) + 1, 0)                                        \
        NOMAD_N3_SYNC_COMPUTE(0, 0)                                                                        \
        /* phase 2: A rows 64..127 */                                                                      \
        _Pragma("unroll") for (int i = 4; i < 8; ++i) {                                                    \
            af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                           \
            af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                           \
        }                                                                                                  \
        if (!a_loader && (KT) + 1 < nk) NOMAD_N3_DMA_B((KT) + 1, 1)                                        \
        NOMAD_N3_SYNC_COMPUTE(4, 0)                                                                        \
        /* phase 3: B columns 32..63 */                                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
            bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff0);                     \
            bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff1);                     \
        }                                                                                                  \
        NOMAD_N3_SYNC_COMPUTE(0, 2)                                                                        \
        /* phase 4: row 0 stages A of tile t+3 into this tile's buffer; both rows wait for "tile t+1 has landed" */ \
        if (a_loader) {                                                                                    \
            if ((KT) + 3 < nk) {                                                                           \
                NOMAD_N3_DMA_A((KT) + 3, abuf)                                                             \
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                          \
            } else if ((KT) + 2 < nk) {                                                                    \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                           \
            } else {                                                                                       \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
            }                                                                                              \
        } else {                                                                                           \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
        }                                                                                                  \
        NOMAD_N3_SYNC_COMPUTE(4, 2)                                                                        \
        abuf = abuf == 2 * Cfg::A_BUF ? 0 : abuf + Cfg::A_BUF;                                             \
    }

    int abuf = 0;
    for (int kt = 0; kt < nk; kt += 2) {
        NOMAD_N3_KTILE(kt, 0)
        NOMAD_N3_KTILE(kt + 1, 1)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
#undef NOMAD_N3_KTILE
#undef NOMAD_N3_SYNC_COMPUTE
#undef NOMAD_N3_MMA
#undef NOMAD_N3_DMA_A
#undef NOMAD_N3_DMA_B
#undef NOMAD_N3_DMA_HALF

    p8_epilogue<ABL == 1, 0, NTS>(p, acc, smem8, grp, m0, n0, wave, wr, wc, lane, fr, fq);
}

template <int ABL = 0, int NTS = 1>
inline hipError_t launch_gemm_bf16_8phase_na3(GemmParams p, int groups, hipStream_t s) {
    p.tiles_m = (p.M + P8N3Cfg::BM - 1) / P8N3Cfg::BM;
    p.tiles_n = p.N / P8N3Cfg::BN;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_8phase_na3_kernel<ABL, NTS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, P8N3Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_8phase_na3_kernel<ABL, NTS>), dim3(p.tiles_m * p.tiles_n, groups), dim3(P8N3Cfg::THREADS), P8N3Cfg::LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace nomad
