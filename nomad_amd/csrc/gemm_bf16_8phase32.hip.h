// bf16 GEMM, 256 x 256 tile, the deep-pipelined schedule of gemm_bf16_8phase.hip.h on v_mfma_f32_32x32x16_bf16.
//
// Why a second instruction shape: an MFMA holds its SIMD's vector issue port for 8 cycles whatever its length
// (MI355X_MICROARCH.md, cycle constants) - 8 of the 16 cycles of a 16x16x32, 8 of the 32 of a 32x32x16.  In the ping-pong
// schedule one wave row computes while the other issues its LDS reads and LDS-DMA; with 16x16x32 the computing wave leaves the
// loading wave only 128 issue cycles per 256-cycle phase (16 MFMAs), with 32x32x16 (8 MFMAs per phase) 192.  Same tile, same
// LDS image (the fragments of both shapes are 16-byte chunks of the same swizzled rows), same DMA, same phases:
//   phase 1: read B column tile 0 (4 x ds_read_b128) + A row tiles 0, 1 (8),   DMA B-half0 of tile t+1,  8 MFMAs (rows 0..63 x cols 0..31)
//   phase 2: read A row tiles 2, 3 (8),                                         DMA B-half1 of tile t+1,  8 MFMAs (rows 64..127 x cols 0..31)
//   phase 3: read B column tile 1 (4),                                                                     8 MFMAs (rows 0..63 x cols 32..63)
//   phase 4: DMA A-half0 / A-half1 of tile t+2, s_waitcnt vmcnt(4),                                        8 MFMAs (rows 64..127 x cols 32..63)
// Wave tile 128 x 64 = acc[4][2] of 32 x 32 (lane: column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)).
// RESULT (round 3, profiles/r03_gemm_sweep_bf16_mfma32_null.json): bit-identical to the 16x16x32 kernel on every shape, +1..4 %
// on the K = 768 shapes of config C5 (out_proj, QKV, fc1), -5..8 % on fc2 / conv4 / 4096^3, C5 end to end 1607 vs 1639
// clips/s alternating on one box: the issue port is not what holds the main loop back.  libnomad_diag.so only (tile 40 / 41).
// Requirements as gemm_bf16_8phase_kernel: N % 256 == 0, K % 128 == 0.  bf16 in / out, fp32 accumulate, bias / GELU / residual.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_bf16_8phase.hip.h"

namespace nomad {

template <bool NOSTORE>
__device__ __forceinline__ void p8_epilogue32(const GemmParams& p, const f32x16 (&acc)[4][2], char* smem8, int grp, int m0, int n0,
                                              int wave, int wr, int wc, int lane) {
    using Cfg = P8Cfg;
    bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C) + grp * p.c_goff;
    const bf16_t* Rg = p.R ? reinterpret_cast<const bf16_t*>(p.R) + grp * p.r_goff : nullptr;
    const float* biasg = p.bias ? p.bias + grp * p.bias_goff : nullptr;
    const bool c_plain = p.cmap.clip_rows >= p.M, r_plain = p.rmap.clip_rows >= p.M;
    constexpr int ELD = Cfg::ELD;
    float* slab = reinterpret_cast<float*>(smem8) + wave * (32 * ELD);
    const int col = lane & 31, rh = 4 * (lane >> 5);
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + col;
        bv[j] = (biasg && n < p.n_valid) ? biasg[n] : 0.f;
    }
    __syncthreads();  // every wave is done with the staging buffers
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {  // rows 32 s4 .. 32 s4 + 31 of the wave tile = accumulator row tile s4
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[s4][j][r] + bv[j];
                if (p.gelu) v = gelu_erf(v);
                slab[((r & 3) + 8 * (r >> 2) + rh) * ELD + j * 32 + col] = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab is private to the wave: a wave-local fence is enough
        if (!NOSTORE) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {  // 32 rows x 8 groups of 8 columns
                const int id = lane + 64 * it, row = id >> 3, cg = id & 7;
                const int m = m0 + wr * 128 + s4 * 32 + row;
                const int n = n0 + wc * 64 + cg * 8;
                if (m < p.M && n < p.n_valid) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * ELD + cg * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * ELD + cg * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (Rg) {
                        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(
                            Rg + (r_plain ? p.rmap.off + (long long)m * p.rmap.ld : row_addr(p.rmap, m)) + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                    }
                    long long c_col = n;
                    if (p.c_colblk > 0) {
                        const int blk = n / p.c_colblk;
                        c_col = (long long)blk * p.c_colblk_stride + (n - blk * p.c_colblk);
                    }
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[e];
                    *reinterpret_cast<bf16x8*>(Cg + (c_plain ? p.cmap.off + (long long)m * p.cmap.ld : row_addr(p.cmap, m)) + c_col) = ov;
                }
            }
        }
    }
}

// NOSTORE: timing ablation (no epilogue stores).
template <bool NOSTORE = false>
__global__ __launch_bounds__(512) void gemm_bf16_8phase32_kernel(const GemmParams p) {
    using Cfg = P8Cfg;
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_coords(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
    const int m0 = tile_m * Cfg::BM, n0 = tile_n * Cfg::BN;
    const int grp = blockIdx.y;
    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(p.A) + grp * p.a_goff;
    const bf16_t* Wg = reinterpret_cast<const bf16_t*>(p.W) + grp * p.w_goff;

    // DMA sources: wave-uniform 64-bit base + per-thread 32-bit byte offset (as gemm_bf16_8phase_kernel)
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
    char* const dma_dst = smem8 + wave * 1024;

#define NOMAD_P32_DMA_A(KT, H)                                                                                  \
    {                                                                                                           \
        const int k0_ = (KT)*64;                                                                                \
        const int kq_ = k0_ / p.kchunk;                                                                         \
        const unsigned ko_ = (unsigned)((kq_ * p.kstride + (k0_ - kq_ * p.kchunk)) * 2);                        \
        char* d_ = dma_dst + ((KT)&1) * Cfg::BUF_BYTES + (H)*Cfg::HALF_BYTES;                                   \
        __builtin_amdgcn_global_load_lds((gptr_t)(a_base + (a_off[H][0] + ko_)), (lptr_t)(d_), 16, 0, 0);          \
        __builtin_amdgcn_global_load_lds((gptr_t)(a_base + (a_off[H][1] + ko_)), (lptr_t)(d_ + 8192), 16, 0, 0);   \
    }
#define NOMAD_P32_DMA_B(KT, H)                                                                                  \
    {                                                                                                           \
        const unsigned ko_ = (unsigned)((KT)*128);                                                              \
        char* d_ = dma_dst + ((KT)&1) * Cfg::BUF_BYTES + (2 + (H)) * Cfg::HALF_BYTES;                           \
        __builtin_amdgcn_global_load_lds((gptr_t)(b_base[H] + (b_off[0] + ko_)), (lptr_t)(d_), 16, 0, 0);          \
        __builtin_amdgcn_global_load_lds((gptr_t)(b_base[H] + (b_off[1] + ko_)), (lptr_t)(d_ + 8192), 16, 0, 0);   \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / 64;  // even
    NOMAD_P32_DMA_A(0, 0)
    NOMAD_P32_DMA_A(0, 1)
    NOMAD_P32_DMA_B(0, 0)
    NOMAD_P32_DMA_B(0, 1)
    NOMAD_P32_DMA_A(1, 0)
    NOMAD_P32_DMA_A(1, 1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();  // second wave row runs one barrier behind (ping-pong)

    // fragment addresses: lane (row fr of a 32-row tile, k half fh): k-step ks is chunk (2 ks + fh) ^ swizzle(row)
    const int sw = (fr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((2 * ks + fh) ^ sw) * 16;
    const int a_frag = wr * Cfg::HALF_BYTES + fr * 128;                                       // + i * 4096 (32 rows)
    const int b_frag = (2 + (wc >> 1)) * Cfg::HALF_BYTES + ((wc & 1) * 64 + fr) * 128;        // + j * 4096

    bf16x8 af[4][4], bf[4];
#define NOMAD_P32_MMA(I0, J)                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
            acc[(I0) + i][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(I0) + i][ks], bf[ks], acc[(I0) + i][J], 0, 0, 0);
#define NOMAD_P32_SYNC_COMPUTE(I0, J)                   \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_setprio(1);                      \
    NOMAD_P32_MMA(I0, J)                                \
    __builtin_amdgcn_s_setprio(0);                      \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");

#define NOMAD_P32_KTILE(KT, BUF)                                                                           \
    {                                                                                                      \
        const char* la_ = smem8 + (BUF)*Cfg::BUF_BYTES + a_frag;                                           \
        const char* lb_ = smem8 + (BUF)*Cfg::BUF_BYTES + b_frag;                                           \
        /* phase 1: B column tile 0, A row tiles 0, 1 */                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const bf16x8*>(lb_ + koff[ks]);          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(la_ + i * 4096 + koff[ks]); \
        if ((KT) + 1 < nk) NOMAD_P32_DMA_B((KT) + 1, 0)                                                    \
        NOMAD_P32_SYNC_COMPUTE(0, 0)                                                                       \
        /* phase 2: A row tiles 2, 3 */                                                                    \
        _Pragma("unroll") for (int i = 2; i < 4; ++i)                                                      \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(la_ + i * 4096 + koff[ks]); \
        if ((KT) + 1 < nk) NOMAD_P32_DMA_B((KT) + 1, 1)                                                    \
        NOMAD_P32_SYNC_COMPUTE(2, 0)                                                                       \
        /* phase 3: B column tile 1 */                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const bf16x8*>(lb_ + 4096 + koff[ks]);   \
        NOMAD_P32_SYNC_COMPUTE(0, 1)                                                                       \
        /* phase 4: both A halves of tile t+2 (their last read was phase 2), then "tile t+1 has landed" */ \
        if ((KT) + 2 < nk) {                                                                               \
            NOMAD_P32_DMA_A((KT) + 2, 0)                                                                   \
            NOMAD_P32_DMA_A((KT) + 2, 1)                                                                   \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                               \
        } else {                                                                                           \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
        }                                                                                                  \
        NOMAD_P32_SYNC_COMPUTE(2, 1)                                                                       \
    }

    for (int kt = 0; kt < nk; kt += 2) {
        NOMAD_P32_KTILE(kt, 0)
        NOMAD_P32_KTILE(kt + 1, 1)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
#undef NOMAD_P32_KTILE
#undef NOMAD_P32_SYNC_COMPUTE
#undef NOMAD_P32_MMA
#undef NOMAD_P32_DMA_A
#undef NOMAD_P32_DMA_B
    p8_epilogue32<NOSTORE>(p, acc, smem8, grp, m0, n0, wave, wr, wc, lane);
}

template <bool NOSTORE = false>
inline hipError_t launch_gemm_bf16_8phase32(GemmParams p, int groups, hipStream_t s) {
    p.tiles_m = (p.M + P8Cfg::BM - 1) / P8Cfg::BM;
    p.tiles_n = p.N / P8Cfg::BN;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_8phase32_kernel<NOSTORE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_8phase32_kernel<NOSTORE>), dim3(p.tiles_m * p.tiles_n, groups), dim3(P8Cfg::THREADS), P8Cfg::LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace nomad
