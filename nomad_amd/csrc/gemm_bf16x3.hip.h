// bf16x3 GEMM, 256 x 256 tile: fp32-class products from the bf16 matrix cores, with every operand plane staged ONCE.
//
//   C = A W^T with A = A_hi + A_lo, W = W_hi + W_lo (split buffers, dtypes.hip.h), computed as
//   A_hi W_hi^T + A_hi W_lo^T + A_lo W_hi^T in fp32 MFMA accumulators (the dropped A_lo W_lo^T is 2^-16 relative).
//
// gemm_bf16_8phase_kernel<.., X3> runs this as a plain GEMM on K-concatenated operands and therefore stages and reads
// A_hi and W_hi twice (6 operand tiles per k-range).  Its timing probes say the loop pays for exactly that: without
// the LDS-DMA it runs 1.43x faster, without the LDS reads 1.25x (profiles/r01_gemm_bf16x3_probes.json).  Here a stage
// holds the four planes of a 32-deep k-range - A_hi, A_lo, W_hi, W_lo, 16 KB each - and feeds 96 MFMAs per wave:
// 2/3 of the DMA bytes and LDS reads per MFMA.
//
//   * one workgroup per CU, 8 waves (2 x 4), wave tile 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16
//     (one MFMA covers the stage's whole k-range); LDS = 2 stages x 64 KB (NA = 2);
//   * rows are 64 bytes in LDS (4 rows per 256-byte bank row); the 16-byte chunk c of row r sits at chunk
//     c ^ (-(r >> 2) & 3).  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
//     (MI355X_MICROARCH.md, LDS): a group holds fragment rows 0-3 and 12-15 of one k-chunk and rows 4-11 of its
//     NEIGHBOUR chunk, and the four rows that share bank slots (r, r+4, r+8, r+12) land on four different chunks
//     only with the row-block keys 0, 3, 2, 1 - the plain (r >> 2) & 3 is 2-way (measured: SQ_LDS_BANK_CONFLICT at
//     47 % of the LDS-active cycles, profiles/r01_pmc_gemm_bf16x3.txt);
//   * six phases of 16 MFMAs per stage u (buffer u & 1); W fragments stay in registers for the whole stage:
//       phase 1: read A_hi rows 0..63, W_hi        A_hi W_hi
//       phase 2: read W_lo                         A_hi W_lo      DMA A_lo(u+1)
//       phase 3: read A_lo rows 0..63              A_lo W_hi      DMA W_hi(u+2)
//       phase 4: read A_hi rows 64..127            A_hi W_hi      DMA W_lo(u+2)
//       phase 5:                                   A_hi W_lo
//       phase 6: read A_lo rows 64..127            A_lo W_hi      DMA A_hi(u+2), then s_waitcnt vmcnt(6)
//     A plane is re-staged no earlier than two phases after its last read (the two wave rows run one barrier apart,
//     ping-pong, as in the 8-phase kernel) and is first read in the phase after the wait that retires its DMA:
//     after phase 6's issue the queue holds [.. A_lo(u+1) | W_hi(u+2) W_lo(u+2) A_hi(u+2)], so vmcnt(6) = "stage u+1
//     has landed".
//
// NA = 3 (K % 192 == 0): A - the streamed operand, first touched from HBM, where W stays cache-resident - gets a THIRD
//   buffer (3 x 32 KB for A + 2 x 32 KB for W = all 160 KB of LDS) and is staged two stages ahead: A_lo(u+2) in phase
//   2, A_hi(u+3) in phase 6, vmcnt(10).  With every workgroup staging the same (L2-resident) A tile the 2-buffer loop
//   runs 7-14 % faster (timing probe ABL 7, profiles/r01_gemm_bf16x3_probes.json): that is HBM latency the 1-stage
//   lead does not cover.
// Requirements: N % 256 == 0, K % 64 == 0 (NA = 2; stages are processed in pairs so that buffer addresses are
// constants) or K % 192 == 0 (NA = 3; in sixes).
// X3: 1 = split output, 2 = fp32 output (p8_epilogue).
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_bf16_8phase.hip.h"

namespace nomad {

struct X3Cfg {
    static constexpr int BM = 256, BN = 256, BK = 32, THREADS = 512;
    static constexpr int OPER_BYTES = 256 * 64;             // one plane of a stage: 256 rows x 32 bf16
    static constexpr int PAIR_BYTES = 2 * OPER_BYTES;       // hi + lo plane of one operand
};

// LDS: [NA buffers of (A_hi, A_lo)] [2 buffers of (W_hi, W_lo)]
// X3 = 3: the output format is a run-time property of the problem (p.c_plane != 0: split planes, else fp32) and PLAIN = the small
// epilogue for plain C / R matrices (gemm_bf16_8phase.hip.h): ONE instantiation for every GEMM of the bf16x3 transformer layers
// instead of two 60 KB ones alternating four times per layer.
template <int ABL, int X3, int NA, bool PLAIN = false>
__global__ __launch_bounds__(512) void gemm_bf16x3_kernel(const GemmParams p) {
    using Cfg = X3Cfg;
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

    // DMA sources: wave-uniform 64-bit base + per-thread 32-bit byte offset relative to the tile's first row
    // (see gemm_bf16_8phase.hip.h).  Instruction i of a plane covers row (tid + 512 i) / 4, physical chunk (tid + 512 i) % 4.
    const int m0_ld = ABL == 7 ? 0 : m0;  // ABL 7 (timing probe): every workgroup stages A tile 0 - always an L2 hit
    const long long tile_row0 = row_addr(p.amap, m0_ld < p.M ? m0_ld : p.M - 1);
    unsigned a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 512, row = id >> 2, pc = id & 3;
        const int sc = (pc ^ (-(row >> 2) & 3)) * 8;  // source chunk (elements)
        b_off[i] = (unsigned)(((long long)row * p.ldw + sc) * 2);
        int m = m0_ld + row;
        m = m < p.M ? m : p.M - 1;
        a_off[i] = (unsigned)((row_addr(p.amap, m) - tile_row0 + sc) * 2);
    }
    const char* const a_base[2] = {reinterpret_cast<const char*>(Ag + tile_row0),
                                   reinterpret_cast<const char*>(Ag + tile_row0) + p.a_plane * 2};
    const char* const b_base[2] = {reinterpret_cast<const char*>(Wg + (long long)n0 * p.ldw),
                                   reinterpret_cast<const char*>(Wg + (long long)n0 * p.ldw) + p.w_plane * 2};
    char* const dma_dst = smem8 + wave * 1024;  // + lane * 16 implicit

    constexpr int W_BASE = NA * Cfg::PAIR_BYTES;
    // PL: 0 = hi, 1 = lo plane; DB: destination buffer (compile-time)
#define NOMAD_X3_DMA_A(U, PL, DB)                                                                                \
    {                                                                                                           \
        const int k0_ = (U)*32;                                                                                 \
        const int kq_ = k0_ / p.kchunk;                                                                         \
        const unsigned ko_ = (unsigned)((kq_ * p.kstride + (k0_ - kq_ * p.kchunk)) * 2);                        \
        char* d_ = dma_dst + (DB)*Cfg::PAIR_BYTES + (PL)*Cfg::OPER_BYTES;                                       \
        if (ABL != 4) {                                                                                         \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_base[PL] + (a_off[0] + ko_)), (lptr_t)(d_), 16, 0, 0);        \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_base[PL] + (a_off[1] + ko_)), (lptr_t)(d_ + 8192), 16, 0, 0); \
        }                                                                                                       \
    }
#define NOMAD_X3_DMA_B(U, PL, DB)                                                                                \
    {                                                                                                           \
        const unsigned ko_ = (unsigned)((U)*64);                                                                \
        char* d_ = dma_dst + W_BASE + (DB)*Cfg::PAIR_BYTES + (PL)*Cfg::OPER_BYTES;                              \
        if (ABL != 4) {                                                                                         \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_base[PL] + (b_off[0] + ko_)), (lptr_t)(d_), 16, 0, 0);        \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_base[PL] + (b_off[1] + ko_)), (lptr_t)(d_ + 8192), 16, 0, 0); \
        }                                                                                                       \
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ns = p.K / 32;  // a multiple of 2 (NA = 2) or 6 (NA = 3)
    // prologue: stage 0 complete; stage 1 (and with NA = 3 A_hi of stage 2) on its way, minus what phase 2 of stage 0 issues
    NOMAD_X3_DMA_A(0, 0, 0)
    NOMAD_X3_DMA_A(0, 1, 0)
    NOMAD_X3_DMA_B(0, 0, 0)
    NOMAD_X3_DMA_B(0, 1, 0)
    if (NA == 3) {
        NOMAD_X3_DMA_A(1, 0, 1)
        NOMAD_X3_DMA_A(1, 1, 1)
        NOMAD_X3_DMA_B(1, 0, 1)
        NOMAD_X3_DMA_B(1, 1, 1)
        NOMAD_X3_DMA_A(2, 0, 2)
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
        NOMAD_X3_DMA_B(1, 0, 1)
        NOMAD_X3_DMA_B(1, 1, 1)
        NOMAD_X3_DMA_A(1, 0, 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();  // second wave row runs one barrier behind (ping-pong)

    // fragment addresses: row fr of a 16-row tile, k-chunk fq at physical chunk fq ^ (-(row >> 2) & 3)
    const int fsw = (fq ^ (-(fr >> 2) & 3)) * 16;
    const int a_frag = (wr * 128 + fr) * 64 + fsw;                               // + buffer, plane, i * 1024
    const int b_frag = W_BASE + (wc * 64 + fr) * 64 + fsw;                       // + buffer, plane, j * 1024

    bf16x8 af[4], wh[4], wl[4];
#define NOMAD_X3_READ_A(BUF, PL, I0)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        af[i] = *reinterpret_cast<const bf16x8*>(smem8 + (BUF)*Cfg::PAIR_BYTES + (PL)*Cfg::OPER_BYTES + a_frag + ((I0) + i) * 1024);
#define NOMAD_X3_READ_W(BUF, PL, DST)                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
        DST[j] = *reinterpret_cast<const bf16x8*>(smem8 + (BUF)*Cfg::PAIR_BYTES + (PL)*Cfg::OPER_BYTES + b_frag + j * 1024);
#define NOMAD_X3_SYNC_COMPUTE(I0, WF)                                                                      \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
            acc[(I0) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], WF[j], acc[(I0) + i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("" ::: "memory");

    // one stage; AB / WB = its A and W buffers (compile-time constants per call site)
#define NOMAD_X3_STAGE(U, AB, WB)                                                                          \
    {                                                                                                      \
        /* phase 1 */                                                                                      \
        NOMAD_X3_READ_W(WB, 0, wh)                                                                         \
        NOMAD_X3_READ_A(AB, 0, 0)                                                                          \
        NOMAD_X3_SYNC_COMPUTE(0, wh)                                                                       \
        /* phase 2 */                                                                                      \
        NOMAD_X3_READ_W(WB, 1, wl)                                                                         \
        if (NA == 3) {                                                                                     \
            if ((U) + 2 < ns) NOMAD_X3_DMA_A((U) + 2, 1, ((AB) + 2) % 3)                                   \
        } else {                                                                                           \
            if ((U) + 1 < ns) NOMAD_X3_DMA_A((U) + 1, 1, (AB) ^ 1)                                         \
        }                                                                                                  \
        NOMAD_X3_SYNC_COMPUTE(0, wl)                                                                       \
        /* phase 3 */                                                                                      \
        NOMAD_X3_READ_A(AB, 1, 0)                                                                          \
        if ((U) + 2 < ns) NOMAD_X3_DMA_B((U) + 2, 0, WB)                                                   \
        NOMAD_X3_SYNC_COMPUTE(0, wh)                                                                       \
        /* phase 4 */                                                                                      \
        NOMAD_X3_READ_A(AB, 0, 4)                                                                          \
        if ((U) + 2 < ns) NOMAD_X3_DMA_B((U) + 2, 1, WB)                                                   \
        NOMAD_X3_SYNC_COMPUTE(4, wh)                                                                       \
        /* phase 5 */                                                                                      \
        NOMAD_X3_SYNC_COMPUTE(4, wl)                                                                       \
        /* phase 6 */                                                                                      \
        NOMAD_X3_READ_A(AB, 1, 4)                                                                          \
        if (NA == 3) {                                                                                     \
            if ((U) + 3 < ns) {                                                                            \
                NOMAD_X3_DMA_A((U) + 3, 0, AB)                                                             \
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                          \
            } else if ((U) + 2 < ns) {                                                                     \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                           \
            } else {                                                                                       \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
            }                                                                                              \
        } else if ((U) + 2 < ns) {                                                                         \
            NOMAD_X3_DMA_A((U) + 2, 0, AB)                                                                 \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                               \
        } else {                                                                                           \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
        }                                                                                                  \
        NOMAD_X3_SYNC_COMPUTE(4, wh)                                                                       \
    }

    if (NA == 3) {
        for (int u = 0; u < ns; u += 6) {
            NOMAD_X3_STAGE(u, 0, 0)
            NOMAD_X3_STAGE(u + 1, 1, 1)
            NOMAD_X3_STAGE(u + 2, 2, 0)
            NOMAD_X3_STAGE(u + 3, 0, 1)
            NOMAD_X3_STAGE(u + 4, 1, 0)
            NOMAD_X3_STAGE(u + 5, 2, 1)
        }
    } else {
        for (int u = 0; u < ns; u += 2) {
            NOMAD_X3_STAGE(u, 0, 0)
            NOMAD_X3_STAGE(u + 1, 1, 1)
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
#undef NOMAD_X3_STAGE
#undef NOMAD_X3_SYNC_COMPUTE
#undef NOMAD_X3_READ_W
#undef NOMAD_X3_READ_A
#undef NOMAD_X3_DMA_A
#undef NOMAD_X3_DMA_B

    p8_epilogue<ABL == 1, X3, (ABL == 8 ? 1 : 0), 4, false, PLAIN>(p, acc, smem8, grp, m0, n0, wave, wr, wc, lane, fr, fq);
}

template <int ABL, int X3, int NA = 2, bool PLAIN = false>
inline hipError_t launch_gemm_bf16x3(GemmParams p, int groups, hipStream_t s) {
    p.tiles_m = (p.M + X3Cfg::BM - 1) / X3Cfg::BM;
    p.tiles_n = p.N / X3Cfg::BN;
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_bf16x3_kernel<ABL, X3, NA, PLAIN>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_bf16x3_kernel<ABL, X3, NA, PLAIN>), dim3(p.tiles_m * p.tiles_n, groups), dim3(X3Cfg::THREADS), (NA + 2) * X3Cfg::PAIR_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace nomad
