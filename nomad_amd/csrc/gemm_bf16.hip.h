// bf16 MFMA GEMM / implicit-GEMM for gfx950 (config C5, long-form clips):  C = epilogue(A * W^T)
// A, W, C, R are bf16 in HBM; accumulation, bias, GELU and the residual add are fp32.
//
// Same structure as gemm_f32_glds_kernel (gemm_f32.hip.h): LDS-DMA staging (global_load_lds_dwordx4),
// unpadded [rows][128 B] LDS image with the XOR swizzle applied on the source address and again on the
// fragment read, one barrier per K tile, XCD-aware tile order, fused epilogue.  A 128-byte LDS row now
// holds 64 bf16 (BK = 64) and one 16-byte chunk is exactly one lane's operand of
// v_mfma_f32_32x32x16_bf16 (lane l: row l&31, k = 8*(l>>5) + 0..7), so a ds_read_b128 per operand tile
// feeds ONE MFMA of 32 cycles (fp32: four MFMAs of 64 cycles).  All GemmParams offsets are in elements.
#pragma once
#include <hip/hip_runtime.h>

#include "dtypes.hip.h"
#include "gemm_f32.hip.h"

namespace nomad {

template <int BM, int BN, int WM, int WN, int BK = 64, int STAGES = 2>
struct Bf16Cfg {
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int ROWB = BK * 2;                  // bytes per LDS row: 128 (BK = 64) or 64 (BK = 32)
    static constexpr int KC = ROWB / 16, RB = 256 / ROWB;  // 16-B chunks per row, rows per 256-B bank row
    static constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static constexpr int A_CHUNKS = BM * KC / THREADS, B_CHUNKS = BN * KC / THREADS;
    static constexpr int STAGE_BYTES = STAGES * (BM + BN) * ROWB;
    static constexpr int EPI_BYTES = WM * WN * 32 * (WTN + 4) * 4;  // one 32-row fp32 slab per wave
    static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
};

// ABL (timing-only ablations, wrong results): 1 = no epilogue stores, 2 = no K loop (prologue + epilogue only).
// STAGES = 2: __syncthreads() per K tile; STAGES >= 3: LDS-DMA issued STAGES-1 tiles ahead, counted vmcnt +
// raw s_barrier (as in gemm_f32_glds_kernel).
template <int BM, int BN, int WM, int WN, int ABL = 0, int BK = 64, int STAGES = 2>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_glds_kernel(const GemmParams p) {
    using Cfg = Bf16Cfg<BM, BN, WM, WN, BK, STAGES>;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, NT = Cfg::THREADS, KC = Cfg::KC, RB = Cfg::RB, ROWB = Cfg::ROWB;
    constexpr int LPT = Cfg::A_CHUNKS + Cfg::B_CHUNKS;  // DMA instructions per thread and K tile
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(BM * KC % NT == 0 && BN * KC % NT == 0 && TM >= 1 && TN >= 1, "bad tile");
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    char* As = smem_b;                          // [STAGES][BM][ROWB]
    char* Bs = smem_b + STAGES * BM * ROWB;     // [STAGES][BN][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_coords(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int grp = blockIdx.y;
    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(p.A) + grp * p.a_goff;
    const bf16_t* Wg = reinterpret_cast<const bf16_t*>(p.W) + grp * p.w_goff;

    const bf16_t* a_src[Cfg::A_CHUNKS];
    const bf16_t* b_src[Cfg::B_CHUNKS];
#pragma unroll
    for (int i = 0; i < Cfg::A_CHUNKS; ++i) {
        const int id = tid + i * NT, row = id / KC, pc = id - row * KC;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        a_src[i] = Ag + row_addr(p.amap, m) + ((pc ^ ((row / RB) % KC)) * 8);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        const int id = tid + i * NT, row = id / KC, pc = id - row * KC;
        b_src[i] = Wg + (long long)(n0 + row) * p.ldw + ((pc ^ ((row / RB) % KC)) * 8);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = ABL == 2 ? 1 : p.K / BK;
#define NOMAD_GLDS_TILE(KT, BUF)                                                                         \
    {                                                                                                    \
        const int k0_ = (KT)*BK;                                                                         \
        const int kq_ = k0_ / p.kchunk;                                                                  \
        const long long a_koff_ = (long long)kq_ * p.kstride + (k0_ - kq_ * p.kchunk);                   \
        char* as_ = As + (BUF)*BM * ROWB + wave * 1024;                                                  \
        char* bs_ = Bs + (BUF)*BN * ROWB + wave * 1024;                                                  \
        _Pragma("unroll") for (int i = 0; i < Cfg::A_CHUNKS; ++i)                                        \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + a_koff_), (lptr_t)(as_ + i * NT * 16), 16, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < Cfg::B_CHUNKS; ++i)                                        \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + k0_), (lptr_t)(bs_ + i * NT * 16), 16, 0, 0);     \
    }

    NOMAD_GLDS_TILE(0, 0)
    if (STAGES >= 3 && nk > 1) NOMAD_GLDS_TILE(1, 1)
    if (STAGES >= 4 && nk > 2) NOMAD_GLDS_TILE(2, 2)

    const int frag_row = lane & 31, h = lane >> 5;
    const int swz = (frag_row / RB) % KC;
    int koff[BK / 16];  // byte offset of this lane's chunk for k-step kq
#pragma unroll
    for (int kq = 0; kq < BK / 16; ++kq) koff[kq] = ((kq * 2 + h) ^ swz) * 16;
    const int a_row_off = (wm * Cfg::WTM + frag_row) * ROWB;
    const int b_row_off = (wn * Cfg::WTN + frag_row) * ROWB;

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (STAGES == 2) {
            __syncthreads();  // tile kt has landed (vmcnt(0)) and every wave is done with the other buffer
        } else {
            // tiles kt+1 .. kt+STAGES-2 may stay in flight
            if (STAGES == 4 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
            else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        {
            const int nxt = kt + STAGES - 1;
            int nb = cur + STAGES - 1;
            nb = nb >= STAGES ? nb - STAGES : nb;
            if (nxt < nk) NOMAD_GLDS_TILE(nxt, nb)
        }
        const char* as = As + cur * BM * ROWB + a_row_off;
        const char* bs = Bs + cur * BN * ROWB + b_row_off;
#pragma unroll
        for (int kq = 0; kq < BK / 16; ++kq) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * ROWB + koff[kq]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * ROWB + koff[kq]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
#undef NOMAD_GLDS_TILE

    // Epilogue through LDS.  An MFMA accumulator holds one column per lane, so storing it directly means 2-byte
    // scattered stores (measured: 45 % of the whole kernel).  Instead each wave parks a 32-row slab of its tile
    // in LDS as fp32 (bias and GELU applied on the way), then every lane picks up 8 consecutive columns of one
    // row, adds the residual from a 16-byte load and writes one 16-byte bf16 vector: full 128-byte segments.
    bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C) + grp * p.c_goff;
    const bf16_t* Rg = p.R ? reinterpret_cast<const bf16_t*>(p.R) + grp * p.r_goff : nullptr;
    const float* biasg = p.bias ? p.bias + grp * p.bias_goff : nullptr;
    const bool c_plain = p.cmap.clip_rows >= p.M, r_plain = p.rmap.clip_rows >= p.M;
    constexpr int ELD = Cfg::WTN + 4;            // floats per slab row (16-byte pad against bank conflicts)
    constexpr int CG = Cfg::WTN / 8;             // 8-column groups per row
    float* slab = reinterpret_cast<float*>(smem_b) + wave * (32 * ELD);
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * Cfg::WTN + j * 32 + (lane & 31);
        bv[j] = (biasg && n < p.n_valid) ? biasg[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __syncthreads();  // main loop (or the previous slab) is done with this LDS
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv[j];
                if (p.gelu) v = gelu_bf16out(v);
                slab[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ELD + j * 32 + (lane & 31)] = v;
            }
        __syncthreads();
        if (ABL != 1) {
#pragma unroll
            for (int it = 0; it < 32 * CG / 64; ++it) {
                const int id = lane + 64 * it, row = id / CG, cg = id - row * CG;
                const int m = m0 + wm * Cfg::WTM + i * 32 + row;
                const int n = n0 + wn * Cfg::WTN + cg * 8;
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
                    *reinterpret_cast<bf16x8*>(
                        Cg + (c_plain ? p.cmap.off + (long long)m * p.cmap.ld : row_addr(p.cmap, m)) + c_col) = ov;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int ABL = 0, int BK = 64, int STAGES = 2>
inline hipError_t launch_gemm_bf16(GemmParams p, int groups, hipStream_t s, int extra_lds = 0) {
    using Cfg = Bf16Cfg<BM, BN, WM, WN, BK, STAGES>;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.N / BN;
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_bf16_glds_kernel<BM, BN, WM, WN, ABL, BK, STAGES>), 160 * 1024); e != hipSuccess) return e;
    dim3 grid(p.tiles_m * p.tiles_n, groups);
    hipLaunchKernelGGL((gemm_bf16_glds_kernel<BM, BN, WM, WN, ABL, BK, STAGES>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES + extra_lds, s,
                       p);
    return hipGetLastError();
}

}  // namespace nomad
