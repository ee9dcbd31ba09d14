// fp32 GEMM, 256 x 256 tile, one workgroup per CU, two wave rows in ping-pong (gfx950).
//
// Why: the microbenchmark budget of the production loop (profiles/r01_mfma_lds_micro.txt) says a 64 x 64 wave tile
// with LDS-DMA staging and a barrier per K tile tops out at 140 TFLOP/s, and that the same loop with a 128 x 64
// wave tile - half the staged bytes per MFMA - models at 147.6.  A 128 x 64 wave tile needs 128 accumulator
// registers, i.e. 2 waves/SIMD, i.e. ONE 8-wave workgroup per CU; this kernel is that, with the skeleton of
// gemm_bf16_8phase.hip.h (identical byte geometry: BK = 32 floats = 128-byte LDS rows, half-tiles of 128 rows):
//   * 8 waves as 2 x 4, wave tile 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x2_f32;
//   * LDS = 2 K-tile buffers x (A 256 x 32 + B 256 x 32 floats) = 128 KB, LDS-DMA in half-tiles, XOR chunk swizzle;
//   * a K tile is 4 phases, phase q = the q-th quarter of the tile's k range: 6 ds_read_b128 (4 A + 2 B fragments, each
//     feeding 4 MFMAs through the k-permutation trick) and 32 MFMAs (2048 matrix-core cycles);
//   * phase 2 issues the whole DMA of the next K tile (8 instructions per thread) into the other buffer, phase 4
//     waits for it (two phases = ~8000 cycles later) with vmcnt(0) before its barrier;
//   * raw s_barrier twice per phase, the two wave rows one barrier apart: one row's MFMA cluster runs while the
//     other issues its LDS reads / DMA.
//   Hazards: a buffer is last read in phase 4 of tile t-1 and re-staged in phase 2 of tile t (two phases later, as
//   the one-barrier stagger requires); it is read again in phase 1 of tile t+1, after the phase-4 wait + barriers.
// Every 32x32x2 instantiation contracts k in the same order (k ascending within a lane's chunk pairs), so results
// are bit-identical to gemm_f32_glds_kernel.  Requirements: N % 256 == 0, K % 64 == 0.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_f32.hip.h"

namespace nomad {

struct PPCfg {
    static constexpr int BM = 256, BN = 256, BK = 32, THREADS = 512;
    static constexpr int HALF_BYTES = 128 * 128;       // 128 rows x 32 floats
    static constexpr int BUF_BYTES = 4 * HALF_BYTES;   // A0 A1 B0 B1
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;    // 128 KB
    static constexpr int ELD = 64 + 4;                 // epilogue slab row (floats)
};

template <bool NOEPI = false>
__global__ __launch_bounds__(512) void gemm_f32_pp_kernel(const GemmParams p) {
    using Cfg = PPCfg;
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_coords(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
    const int m0 = tile_m * Cfg::BM, n0 = tile_n * Cfg::BN;
    const int grp = blockIdx.y;
    const float* Ag = p.A + grp * p.a_goff;
    const float* Wg = p.W + grp * p.w_goff;

    // DMA sources: half h, instruction i -> row (tid + 512 i) / 8 of the half, physical 16-byte chunk (tid + 512 i) % 8.
    // A rows can lie > 4 GB apart (conv1 input: 6.7 GB), so A keeps 64-bit pointers; B uses a scalar base per half
    // plus one 32-bit offset per instruction.
    const float* a_src[2][2];
    unsigned b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 512, row = id >> 3, pc = id & 7;
        const int sw = (pc ^ ((row >> 1) & 7)) * 4;
        b_off[i] = (unsigned)(((long long)row * p.ldw + sw) * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + h * 128 + row;
            m = m < p.M ? m : p.M - 1;
            a_src[h][i] = Ag + row_addr(p.amap, m) + sw;
        }
    }
    const char* const b_base[2] = {reinterpret_cast<const char*>(Wg + (long long)n0 * p.ldw),
                                   reinterpret_cast<const char*>(Wg + (long long)(n0 + 128) * p.ldw)};
    char* const dma_dst = smem_pp + wave * 1024;

#define NOMAD_PP_DMA(KT)                                                                                        \
    {                                                                                                           \
        const int k0_ = (KT)*32;                                                                                \
        const int kq_ = k0_ / p.kchunk;                                                                         \
        const long long ako_ = (long long)kq_ * p.kstride + (k0_ - kq_ * p.kchunk);                             \
        const unsigned bko_ = (unsigned)((KT)*128);                                                             \
        char* d_ = dma_dst + ((KT)&1) * Cfg::BUF_BYTES;                                                         \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                         \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[h][0] + ako_), (lptr_t)(d_ + h * Cfg::HALF_BYTES), 16, 0, 0);        \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[h][1] + ako_), (lptr_t)(d_ + h * Cfg::HALF_BYTES + 8192), 16, 0, 0); \
        }                                                                                                       \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                         \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_base[h] + (b_off[0] + bko_)), (lptr_t)(d_ + (2 + h) * Cfg::HALF_BYTES), 16, 0, 0);        \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_base[h] + (b_off[1] + bko_)), (lptr_t)(d_ + (2 + h) * Cfg::HALF_BYTES + 8192), 16, 0, 0); \
        }                                                                                                       \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / 32;  // even
    NOMAD_PP_DMA(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();  // second wave row runs one barrier behind

    // fragment addresses (bytes): row frag_row of a 32-row tile, chunk (2 kq + h) ^ swizzle
    const int frag_row = lane & 31, h = lane >> 5;
    const int sw = (frag_row >> 1) & 7;
    const int a_frag = wr * Cfg::HALF_BYTES + frag_row * 128;                                     // + i * 4096
    const int b_frag = (2 + (wc >> 1)) * Cfg::HALF_BYTES + ((wc & 1) * 64 + frag_row) * 128;      // + j * 4096

    f32x4 af[4], bf[2];
#define NOMAD_PP_PHASE(BUF, KQ, EXTRA)                                                                     \
    {                                                                                                      \
        const int ko_ = (((KQ)*2 + h) ^ sw) * 16;                                                          \
        const char* ab_ = smem_pp + (BUF)*Cfg::BUF_BYTES + a_frag + ko_;                                   \
        const char* bb_ = smem_pp + (BUF)*Cfg::BUF_BYTES + b_frag + ko_;                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bb_ + j * 4096); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f32x4*>(ab_ + i * 4096); \
        EXTRA                                                                                              \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][c], bf[j][c], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                     \
    }
#define NOMAD_PP_KTILE(KT, BUF)                                                                            \
    NOMAD_PP_PHASE(BUF, 0, )                                                                               \
    NOMAD_PP_PHASE(BUF, 1, if ((KT) + 1 < nk) NOMAD_PP_DMA((KT) + 1))                                      \
    NOMAD_PP_PHASE(BUF, 2, )                                                                               \
    NOMAD_PP_PHASE(BUF, 3, asm volatile("s_waitcnt vmcnt(0)" ::: "memory");)

    for (int kt = 0; kt < nk; kt += 2) {
        NOMAD_PP_KTILE(kt, 0)
        NOMAD_PP_KTILE(kt + 1, 1)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
#undef NOMAD_PP_KTILE
#undef NOMAD_PP_PHASE
#undef NOMAD_PP_DMA

    // ---- epilogue: the arithmetic and its order per element are those of gemm_f32_glds_kernel ----------------
    float* Cg = p.C + grp * p.c_goff;
    const float* Rg = p.R ? p.R + grp * p.r_goff : nullptr;
    const float* biasg = p.bias ? p.bias + grp * p.bias_goff : nullptr;
    const float* DGg = p.DG ? p.DG + grp * p.dg_goff : nullptr;
    float* Ug = p.Upre ? p.Upre + grp * p.c_goff : nullptr;
    const bool c_plain = p.cmap.clip_rows >= p.M, r_plain = p.rmap.clip_rows >= p.M;
    constexpr int ELD = Cfg::ELD;
    float* slab = reinterpret_cast<float*>(smem_pp) + wave * (32 * ELD);
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + (lane & 31);
        bv[j] = (biasg && n < p.n_valid) ? biasg[n] : 0.f;
    }
    __syncthreads();  // every wave is done with the staging buffers; the slabs are wave-private from here on
    // one 32-row slab; called four times with a constant row-tile index (a loop here does not unroll - the body is too
    // large - and a runtime index would push the accumulators through scratch)
    auto slab_out = [&](const f32x16 (&a)[2], int i) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ELD + j * 32 + (lane & 31)] = a[j][r] + bv[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!NOEPI) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {  // 32 rows x 16 groups of 4 columns
                const int id = lane + 64 * it, row = id >> 4, cg = id & 15;
                const int m = m0 + wr * 128 + i * 32 + row;
                const int n = n0 + wc * 64 + cg * 4;
                if (m < p.M && n < p.n_valid) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * ELD + cg * 4);
                    long long c_col = n;
                    if (p.c_colblk > 0) {
                        const int blk = n / p.c_colblk;
                        c_col = (long long)blk * p.c_colblk_stride + (n - blk * p.c_colblk);
                    }
                    const long long c_idx = (c_plain ? p.cmap.off + (long long)m * p.cmap.ld : row_addr(p.cmap, m)) + c_col;
                    if (Ug) *reinterpret_cast<f32x4*>(Ug + c_idx) = v;
                    if (p.gelu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    }
                    if (DGg) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(DGg + row_addr(p.dgmap, m) + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= dgelu_erf_(u[e]);
                    }
                    if (Rg) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(
                            Rg + (r_plain ? p.rmap.off + (long long)m * p.rmap.ld : row_addr(p.rmap, m)) + n);
                        v += rv;
                    }
                    *reinterpret_cast<f32x4*>(Cg + c_idx) = v;
                }
            }
        }
    };
    slab_out(acc[0], 0);
    slab_out(acc[1], 1);
    slab_out(acc[2], 2);
    slab_out(acc[3], 3);
}

template <bool NOEPI = false>
inline hipError_t launch_gemm_f32_pp(GemmParams p, int groups, hipStream_t s) {
    p.tiles_m = (p.M + PPCfg::BM - 1) / PPCfg::BM;
    p.tiles_n = p.N / PPCfg::BN;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_pp_kernel<NOEPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_f32_pp_kernel<NOEPI>, dim3(p.tiles_m * p.tiles_n, groups), dim3(PPCfg::THREADS), PPCfg::LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace nomad
