// bf16 GEMM, 256 x 256 tile, deep-pipelined "8-phase" schedule for gfx950 (config C5, the wide transformer and conv
// GEMMs).  The two-barrier-per-K-tile kernel of gemm_bf16.hip.h stalls every K tile on the LDS-DMA it just
// issued (measured ceiling ~800 TFLOP/s, no-epilogue ablation); this one keeps loads in flight across barriers:
//
//   * one workgroup per CU: 8 waves (2 x 4), wave tile 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16;
//   * LDS = 2 K-tile buffers x (A 256 x 64 + B 256 x 64 bf16) = 128 KB, staged by global_load_lds in HALF-tiles
//     (128 rows, 2 DMA instructions per thread), same XOR chunk swizzle as gemm_bf16.hip.h (4 lanes per 16-byte
//     slot of a 256-byte bank row = the minimum for a 1 KB wave read);
//   * a K tile is 4 phases of 16 MFMAs (one 64 x 32 quadrant of the wave tile x K = 64); every phase issues one
//     half-tile of DMA for a LATER K tile, so 2-3 half-tiles are always in flight:
//         phase 1: read A rows 0..63 (8 x ds_read_b128) + B columns 0..31 (4),   DMA B-half0 of tile t+1
//         phase 2: read A rows 64..127 (8),                                       DMA B-half1 of tile t+1
//         phase 3: read B columns 32..63 (4)
//         phase 4: DMA A-half0 and A-half1 of tile t+2, then s_waitcnt vmcnt(4) = "tile t+1 has landed"
//   * the only waits are that counted vmcnt once per K tile and lgkmcnt(0) before each MFMA cluster; barriers are
//     raw s_barrier.  The two wave rows run one barrier apart (ping-pong): while one row's 16 MFMAs occupy the matrix
//     cores, the other row issues its LDS reads and DMA.
//   Hazards (with the one-barrier stagger a buffer may be re-staged no earlier than TWO phases after its last read,
//   and is read no earlier than the phase after the wait that retires its DMA):
//     A halves: read in phases 1-2 of tile t     -> re-staged in phase 4 of tile t (for t+2), waited in phase 4 of t+1
//     B halves: read in phases 1 and 3 of tile t -> re-staged in phases 1/2 of tile t+1 (for t+2), waited in phase 4 of t+1
//
// Requirements: N % 256 == 0, K % 128 == 0 (K tiles are processed in pairs so that buffer addresses are constants).
// Epilogue: fp32 slabs through LDS, bias / GELU / residual in fp32, 16-byte bf16 stores (as gemm_bf16_glds_kernel).
#pragma once
#include <hip/hip_runtime.h>

#include "dtypes.hip.h"
#include "gemm_f32.hip.h"

namespace nomad {

struct P8Cfg {
    static constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
    static constexpr int HALF_BYTES = 128 * 128;            // 128 rows x 64 bf16
    static constexpr int BUF_BYTES = 4 * HALF_BYTES;        // A0 A1 B0 B1
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;         // 128 KB
    static constexpr int ELD = 64 + 4;                      // epilogue slab row (floats)
};

// Epilogue shared by the 256 x 256 kernels (8 waves as 2 x 4, wave tile 128 x 64 = acc[8][4], accumulator: column fr,
// rows 4 fq + r of each 16 x 16 tile): 32-row fp32 slabs through LDS, bias / GELU / residual in fp32, 16-byte stores.
// OUT: 0 = bf16, 1 = split planes (p.c_plane; R split too), 2 = fp32 (R split).  NOSTORE: timing ablation.
// NT: bit 0 = output stores, bit 1 = residual loads carry the non-temporal hint (streamed through L2).
// NJ: 16-column accumulator tiles per wave (wave tile 128 x 16 NJ; 4 = the 256-column workgroup tile, 3 = the 192-column one).
// RPRE: residual prefetch (see below).
// PLAIN: C and R are plain row-major matrices (the caller checked: no ragged row maps, no column blocks) - the address arithmetic
//   of the general case (a binary search per row for ragged maps, two integer divisions for column blocks and frame limits),
//   unrolled over the 16 chunks of a lane, is 45 KB of the kernel's 58 KB of code; the plain epilogue is a fifth of that.  Code
//   size matters here: two large instantiations ALTERNATING between launches cost the launch after each switch 12-17 us
//   (instruction cache; gpurun_out/rpreprof2), which is what ate the isolated gains of the 256 x 192 tiles and of the residual
//   prefetch until every GEMM of the transformer layers ran the SAME instantiation.
template <bool NOSTORE, int X3, int NT = 0, int NJ = 4, bool RPRE = false, bool PLAIN = false>
__device__ __forceinline__ void p8_epilogue(const GemmParams& p, const f32x4 (&acc)[8][NJ], char* smem8, int grp, int m0, int n0,
                                            int wave, int wr, int wc, int lane, int fr, int fq) {
    using Cfg = P8Cfg;
    static_assert(NJ == 4 || NJ == 3, "wave tile of 64 or 48 columns");
    constexpr int ABL = NOSTORE ? 1 : 0;
    constexpr int WTN = 16 * NJ;  // columns of a wave tile
    bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C) + grp * p.c_goff;
    float* Cf = p.C + grp * p.c_goff;  // X3 == 2: fp32 output
    const bf16_t* Rg = p.R ? reinterpret_cast<const bf16_t*>(p.R) + grp * p.r_goff : nullptr;
    const float* biasg = p.bias ? p.bias + grp * p.bias_goff : nullptr;
    const bool c_plain = p.cmap.clip_rows >= p.M, r_plain = p.rmap.clip_rows >= p.M;
    constexpr int ELD = WTN + 4;
    float* slab = reinterpret_cast<float*>(smem8) + wave * (32 * ELD);
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wc * WTN + j * 16 + fr;
        bv[j] = (biasg && n < p.n_valid) ? biasg[n] : 0.f;
    }
    // Residual prefetch (plain bf16): C and R are not restrict-qualified (R may BE C), so the compiler keeps every residual load
    // behind the previous chunk's store and waits for it at once - 16 serial memory round trips per lane and tile (llvm-objdump:
    // global_load_dwordx4, s_waitcnt vmcnt(0), ..., global_store_dwordx4, 16 times).  A lane only ever stores the elements it
    // loaded, so all 16 loads can be issued up front, into the registers the operand fragments have just vacated; the counted
    // waits the compiler then places never cover a store.
    bf16x8 rpre[4][NJ];
    if (RPRE && X3 == 0 && Rg) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int it = 0; it < NJ; ++it) {
                const int id = lane + 64 * it, row = NJ == 4 ? id >> 3 : id / 6, cg = NJ == 4 ? id & 7 : id - 6 * row;
                const int m = m0 + wr * 128 + s4 * 32 + row;
                const int n = n0 + wc * WTN + cg * 8;
                if (m < p.M && n < p.n_valid) {
                    const bf16_t* rp = Rg + ((PLAIN || r_plain) ? p.rmap.off + (long long)m * p.rmap.ld : row_addr(p.rmap, m)) + n;
                    rpre[s4][it] = (NT & 2) ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(rp)) : *reinterpret_cast<const bf16x8*>(rp);
                } else {
                    rpre[s4][it] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
    }
    __syncthreads();  // every wave is done with the staging buffers
    // The slab is private to the wave and a wave's LDS operations execute in order, so inside the loop only the
    // compiler needs a fence: a workgroup barrier here would also wait (vmcnt) for the previous slab's global stores.
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {  // rows 32 s4 .. 32 s4 + 31 of the wave tile = accumulator row-tiles 2 s4, 2 s4 + 1
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[2 * s4 + ii][j][r] + bv[j];
                    if (p.gelu) v = X3 ? gelu_erf(v) : gelu_bf16out(v);   // (X3: fp32-class results, exact GELU)
                    slab[(ii * 16 + 4 * fq + r) * ELD + j * 16 + fr] = v;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ABL != 1) {
#pragma unroll
            for (int it = 0; it < NJ; ++it) {  // 32 rows x 2 NJ groups of 8 columns
                const int id = lane + 64 * it, row = NJ == 4 ? id >> 3 : id / 6, cg = NJ == 4 ? id & 7 : id - 6 * row;
                const int m = m0 + wr * 128 + s4 * 32 + row;
                const int n = n0 + wc * WTN + cg * 8;
                bool live = m < p.M && n < p.n_valid;
                if (!PLAIN && live && p.c_blk_step > 0 && p.c_colblk > 0) {  // column blocks are frames: drop those past the clip's end
                    int li, frames;
                    clip_pos(p.cmap, m, p.c_clip_frames, li, frames);
                    live = li * p.c_blk_step + n / p.c_colblk < frames;
                }
                if (live) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * ELD + cg * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * ELD + cg * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (RPRE && X3 == 0 && Rg) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rpre[s4][it][e];
                    } else if (Rg) {
                        const bf16_t* rp = Rg + ((PLAIN || r_plain) ? p.rmap.off + (long long)m * p.rmap.ld : row_addr(p.rmap, m)) + n;
                        const bf16x8 rv = (NT & 2) ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(rp)) : *reinterpret_cast<const bf16x8*>(rp);
                        if (X3) {
                            const bf16x8 rl = *reinterpret_cast<const bf16x8*>(rp + p.r_plane);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e] + (float)rl[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                        }
                    }
                    long long c_col = n;
                    if (!PLAIN && p.c_colblk > 0) {
                        const int blk = n / p.c_colblk;
                        c_col = (long long)blk * p.c_colblk_stride + (n - blk * p.c_colblk);
                    }
                    const long long ci = ((PLAIN || c_plain) ? p.cmap.off + (long long)m * p.cmap.ld : row_addr(p.cmap, m)) + c_col;
                    if (X3 == 2 || (X3 == 3 && p.c_plane == 0)) {   // fp32 output (X3 = 3: decided per problem)
                        if (NT & 1) {
                            __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(Cf + ci));
                            __builtin_nontemporal_store((f32x4){v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4*>(Cf + ci + 4));
                        } else {
                            *reinterpret_cast<f32x4*>(Cf + ci) = (f32x4){v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4*>(Cf + ci + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                        }
                    } else {
                        bf16x8 ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[e];
                        if (NT & 1) __builtin_nontemporal_store(ov, reinterpret_cast<bf16x8*>(Cg + ci));
                        else *reinterpret_cast<bf16x8*>(Cg + ci) = ov;
                        if (X3 == 1 || X3 == 3) {
                            bf16x8 ol;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ol[e] = (bf16_t)(v[e] - (float)ov[e]);
                            if (NT & 1) __builtin_nontemporal_store(ol, reinterpret_cast<bf16x8*>(Cg + ci + p.c_plane));
                            else *reinterpret_cast<bf16x8*>(Cg + ci + p.c_plane) = ol;
                        }
                    }
                }
            }
        }
    }
}

// ABL: 1 = no epilogue stores (timing only); 2 = no s_setprio around the MFMA clusters (A/B, same results);
//      3 = X3 timing probe: every third K tile keeps the previous tile's A fragments (skips its A reads from LDS);
//      4 / 5 / 6 = timing probes: no LDS-DMA / neither DMA nor LDS reads (MFMA + barriers only) / no LDS reads.
// BUFLD: LDS-DMA through buffer descriptors (dma16_buffer) instead of global_load_lds - an A/B switch, measured below.
// X3 ("bf16x3", fp32-class results on the bf16 matrix cores): A, W and R are SPLIT buffers (dtypes.hip.h: hi and lo
//   bf16 planes p.a_plane / p.w_plane / p.r_plane elements apart) and the K loop walks 3 K/64 tiles - tile 3 kk + s
//   multiplies k-range kk of (A_hi, W_hi), (A_hi, W_lo), (A_lo, W_hi) for s = 0, 1, 2 - into the same fp32
//   accumulators: a.w = (ah + al)(wh + wl) minus the al*wl term (2^-16 relative), i.e. the plain kernel run on
//   operands concatenated along K, with no other change to the schedule.  X3 = 1 stores the result split
//   (p.c_plane), X3 = 2 stores fp32.
//      8 / 9 / 10 = non-temporal output stores / stores + residual loads / residual loads only (A/B).
//      11 = timing probe: every workgroup stages A tile 0 (A always hits in L2; wrong results).
//      12 = timing probe: as 5 (no DMA, no LDS reads) and no barriers in the loop either - both wave rows issue MFMAs freely.
// timing probe ABL 7 (tools/gemm_timeline.py): per workgroup {entry, main loop start, main loop end, stores done} in 100 MHz
// wall-clock ticks + HW_ID + XCC_ID
// (kTimelineSlots / g_timeline live in dtypes.hip.h: the fp32 GEMM's probe writes the same buffer)

// NB = 3 (plain bf16 only): THREE B buffers.  With two, B of tile t+1 can only be issued in phases 1 / 2 of tile t (its buffer
//   is read until phase 3 of tile t-1) and is waited for in phase 4 of the same tile: 2-3 phases of lead, less than an L2
//   round trip under load once the loop runs near the MFMA rate.  With three, B of tile t+2 is issued in phases 1 / 2 of tile t
//   and waited for in phase 4 of tile t+1 (vmcnt(8): B and A of tile t+2 may be outstanding) - 1.75 K tiles of lead; A keeps
//   its two buffers and one tile of lead.  LDS: A0 A1 (32 KB each) | B0 B1 B2 = 160 KB; the B buffer of a K tile rotates
//   (t mod 3, a scalar offset).  Same MFMA order, same results.
// NJ = 3: a 256 x 192 workgroup tile (wave tile 128 x 48) for the N = 768 GEMMs (out_proj, fc2) whose 256 x 256 grids are a
//   little over two rounds of the 256 CUs (C5: 564 tiles = 2.2 rounds, so the launch takes three rounds for 2.2 rounds of work;
//   752 tiles of 256 x 192 = 2.94 rounds).  Same schedule: B tiles are 192 rows (the second B half is 64 rows = ONE DMA
//   instruction per thread), phases 3 / 4 multiply the third 16-column tile only (16 + 16 + 8 + 8 MFMAs per K tile), the counted
//   waits allow one DMA less.  Same MFMA order per output element, same results.
template <int ABL = 0, bool BUFLD = false, int X3 = 0, int NB = 2, int NJ = 4, bool RPRE = false, bool PLAIN = false>
__global__ __launch_bounds__(512) void gemm_bf16_8phase_kernel(const GemmParams p) {
    using Cfg = P8Cfg;
    static_assert(NB == 2 || (NB == 3 && X3 == 0 && !BUFLD), "three B buffers: plain bf16, global_load_lds");
    static_assert(NJ == 4 || (NJ == 3 && X3 == 0 && !BUFLD), "192-column tiles: plain bf16, global_load_lds");
    constexpr int BN = 64 * NJ;
    constexpr bool B3 = NB == 3;
    // cache-policy probes of the LDS-DMA (aux: 1 = sc0, 2 = nt, 16 = sc1): ABL 13 / 14 / 15 = nt on A / on B / on both, 16 = sc1 on both
    constexpr int AUX_A = (ABL == 13 || ABL == 15) ? 2 : ABL == 16 ? 16 : 0, AUX_B = (ABL == 14 || ABL == 15) ? 2 : ABL == 16 ? 16 : 0;
    constexpr int A_BUF = 2 * Cfg::HALF_BYTES;   // B3 layout: A buffers at 0 / 32 KB, B buffers from 64 KB on
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    const unsigned long long t_entry_ = ABL == 7 ? wall_clock64() : 0ull;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    if (PLAIN) {   // round 4, as gemm_f32.hip.h OPT bit 16384: the set-up's integer divisions as mulhi + shift with host-made magic numbers
        tile_m = p.tn_magic ? fast_div(wg, p.tn_magic, p.tn_shift) : wg;   // (the caller checks: uniform clip map, n-fastest tile walk)
        tile_n = wg - tile_m * p.tiles_n;
    } else {
        tile_coords(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
    }
    const int m0 = tile_m * Cfg::BM, n0 = tile_n * BN;
    auto row_addr_a = [&](int m) -> long long {
        if (PLAIN) {
            const int c = p.a_clip_magic ? fast_div(m, p.a_clip_magic, p.a_clip_shift) : 0;
            return p.amap.off + (long long)c * p.amap.clip_stride + (long long)(m - c * p.amap.clip_rows) * p.amap.ld;
        }
        return row_addr(p.amap, m);
    };
    const int grp = blockIdx.y;
    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(p.A) + grp * p.a_goff;
    const bf16_t* Wg = reinterpret_cast<const bf16_t*>(p.W) + grp * p.w_goff;

    // DMA sources as (uniform 64-bit base) + (per-thread 32-bit byte offset): one VGPR per address instead of two, so
    // that the whole working set stays inside the 256 registers a wave has at 2 waves/SIMD - a spilled address
    // would come back through a scratch load whose s_waitcnt drains the whole DMA queue.
    // Instruction i of a half-tile covers rows (tid + 512 i) / 8, physical chunk (tid + 512 i) % 8.
    // The 32-bit offsets are relative to the tile's FIRST row (row addresses grow with the row index, a 256-row tile
    // spans far less than 4 GB); the tensor itself may be larger than 4 GB (conv1 input at batch 512: 6.7 GB).
    const int m0_ld = ABL == 11 ? 0 : m0;  // ABL 11 (timing probe): every workgroup stages A tile 0 - always an L2 hit
    const long long tile_row0 = row_addr_a(m0_ld < p.M ? m0_ld : p.M - 1);  // wave-uniform
    unsigned a_off[2][2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 512, row = id >> 3, pc = id & 7;
        const int sw = (pc ^ ((row >> 1) & 7)) * 8;
        b_off[i] = (unsigned)(((long long)row * p.ldw + sw) * 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0_ld + h * 128 + row;
            m = m < p.M ? m : p.M - 1;
            a_off[h][i] = (unsigned)((row_addr_a(m) - tile_row0 + sw) * 2);
        }
    }
    const char* const a_base = reinterpret_cast<const char*>(Ag + tile_row0);
    const char* const b_base[2] = {reinterpret_cast<const char*>(Wg + (long long)n0 * p.ldw),
                                   reinterpret_cast<const char*>(Wg + (long long)(n0 + 128) * p.ldw)};
    // lo planes (X3): wave-uniform bases, selected per K tile by scalar code
    const char* const a_base_lo = a_base + (X3 ? p.a_plane * 2 : 0);
    const char* const b_base_lo[2] = {b_base[0] + (X3 ? p.w_plane * 2 : 0), b_base[1] + (X3 ? p.w_plane * 2 : 0)};
    char* const dma_dst = smem8 + wave * 1024;  // + lane * 16 implicit (lane-linear LDS-DMA destination)

    // LDS-DMA through buffer descriptors (dma16_buffer, gemm_f32.hip.h): wave-uniform base in SGPRs, one 32-bit VGPR
    // per lane, the K-tile term as the scalar offset - no 64-bit VGPR pointers for the compiler to hoist or spill.
#define NOMAD_P8_DMA_A(KT, H)                                                                                   \
    {                                                                                                           \
        int kt_ = (KT);                                                                                         \
        const char* ab_ = a_base;                                                                               \
        if (X3) {                                                                                               \
            const int kk_ = kt_ / 3;                                                                            \
            if (kt_ - 3 * kk_ == 2) ab_ = a_base_lo;                                                            \
            kt_ = kk_;                                                                                          \
        }                                                                                                       \
        const int k0_ = kt_ * 64;                                                                               \
        const int kq_ = k0_ / p.kchunk;                                                                         \
        const unsigned ko_ = (unsigned)((kq_ * p.kstride + (k0_ - kq_ * p.kchunk)) * 2);                        \
        char* d_ = dma_dst + ((KT)&1) * (B3 ? A_BUF : Cfg::BUF_BYTES) + (H)*Cfg::HALF_BYTES;                    \
        if (ABL == 4 || ABL == 5 || ABL == 12) {                                                                             \
        } else if (BUFLD) {                                                                                            \
            dma16_buffer(reinterpret_cast<const float*>(ab_), (lptr_t)(d_), (int)a_off[H][0], (int)ko_);          \
            dma16_buffer(reinterpret_cast<const float*>(ab_), (lptr_t)(d_ + 8192), (int)a_off[H][1], (int)ko_);   \
        } else {                                                                                                \
            __builtin_amdgcn_global_load_lds((gptr_t)(ab_ + (a_off[H][0] + ko_)), (lptr_t)(d_), 16, 0, AUX_A);         \
            __builtin_amdgcn_global_load_lds((gptr_t)(ab_ + (a_off[H][1] + ko_)), (lptr_t)(d_ + 8192), 16, 0, AUX_A);  \
        }                                                                                                       \
    }
#define NOMAD_P8_DMA_B(KT, H)                                                                                   \
    {                                                                                                           \
        int kt_ = (KT);                                                                                         \
        const char* bb_ = b_base[H];                                                                            \
        if (X3) {                                                                                               \
            const int kk_ = kt_ / 3;                                                                            \
            if (kt_ - 3 * kk_ == 1) bb_ = b_base_lo[H];                                                         \
            kt_ = kk_;                                                                                          \
        }                                                                                                       \
        const unsigned ko_ = (unsigned)(kt_ * 128);                                                             \
        char* d_ = B3 ? dma_dst + 2 * A_BUF + b3_dst_ + (H)*Cfg::HALF_BYTES                                    \
                      : dma_dst + ((KT)&1) * Cfg::BUF_BYTES + (2 + (H)) * Cfg::HALF_BYTES;                      \
        if (ABL == 4 || ABL == 5 || ABL == 12) {                                                                             \
        } else if (BUFLD) {                                                                                            \
            dma16_buffer(reinterpret_cast<const float*>(bb_), (lptr_t)(d_), (int)b_off[0], (int)ko_);             \
            dma16_buffer(reinterpret_cast<const float*>(bb_), (lptr_t)(d_ + 8192), (int)b_off[1], (int)ko_);      \
        } else {                                                                                                \
            __builtin_amdgcn_global_load_lds((gptr_t)(bb_ + (b_off[0] + ko_)), (lptr_t)(d_), 16, 0, AUX_B);            \
            if (NJ == 4 || (H) == 0)                                                                            \
                __builtin_amdgcn_global_load_lds((gptr_t)(bb_ + (b_off[1] + ko_)), (lptr_t)(d_ + 8192), 16, 0, AUX_B); \
        }                                                                                                       \
    }

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (X3 ? 3 : 1) * (p.K / 64);  // even
    // prologue: tile 0 complete, A of tile 1 on its way
    int b3_cur = 0;    // B3: byte offset of the B buffer of the current K tile (t mod 3) ...
    int b3_dst_ = 0;   // ... and of the buffer a NOMAD_P8_DMA_B fills
    NOMAD_P8_DMA_A(0, 0)
    NOMAD_P8_DMA_A(0, 1)
    NOMAD_P8_DMA_B(0, 0)
    NOMAD_P8_DMA_B(0, 1)
    NOMAD_P8_DMA_A(1, 0)
    NOMAD_P8_DMA_A(1, 1)
    if (B3) {  // tile 1 complete too: B of tile t+1 is never issued inside the loop
        b3_dst_ = A_BUF;
        NOMAD_P8_DMA_B(1, 0)
        NOMAD_P8_DMA_B(1, 1)
        if (NJ == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // A and B of tile 1 may be outstanding
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    unsigned long long ts_[4] = {0, 0, 0, 0};
    if (ABL == 7) {
        ts_[0] = t_entry_;
        ts_[1] = wall_clock64();
    }
    if (wr == 1 && ABL != 12) __builtin_amdgcn_s_barrier();  // second wave row runs one barrier behind (ping-pong)

    // fragment addresses: this lane's row fr of a 16-row tile, chunk (4 kh + fq) ^ swizzle
    const int sw = (fr >> 1) & 7;
    const int koff0 = ((0 + fq) ^ sw) * 16, koff1 = ((4 + fq) ^ sw) * 16;
    const int a_frag = wr * Cfg::HALF_BYTES + fr * 128;                                         // + i * 2048
    // B rows (= output columns) are contiguous over the two halves: row 16 NJ wc + 16 j + fr of the workgroup tile
    const int b_frag = (B3 ? 2 * A_BUF : 2 * Cfg::HALF_BYTES) + (wc * (16 * NJ) + fr) * 128;  // + j * 2048

    bf16x8 af[8][2], bf[2][2];
    if (ABL >= 3) {  // timing probes may skip fragment loads: keep the registers defined
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i][0] = af[i][1] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j][0] = bf[j][1] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1};
    }
#define NOMAD_P8_MMA(I0, J0)                                                                               \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
                if ((J0) + j < NJ)                                                                         \
                    acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[(I0) + i][kh], bf[j][kh], acc[(I0) + i][(J0) + j], 0, 0, 0);
#define NOMAD_P8_SYNC_COMPUTE(I0, J0)                   \
    if (ABL != 12) __builtin_amdgcn_s_barrier();        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    if (ABL != 2 && ABL != 12) __builtin_amdgcn_s_setprio(1);        \
    NOMAD_P8_MMA(I0, J0)                                \
    if (ABL != 2 && ABL != 12) __builtin_amdgcn_s_setprio(0);        \
    if (ABL != 12) __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");

    // one K tile (buffer BUF = KT & 1, a compile-time constant per call site)
#define NOMAD_P8_KTILE(KT, BUF)                                                                            \
    {                                                                                                      \
        const char* la_ = smem8 + (BUF) * (B3 ? A_BUF : Cfg::BUF_BYTES) + a_frag;                          \
        const char* lb_ = smem8 + (B3 ? b3_cur : (BUF)*Cfg::BUF_BYTES) + b_frag;                           \
        if (B3) b3_dst_ = b3_cur >= A_BUF ? b3_cur - A_BUF : b3_cur + 2 * A_BUF; /* buffer of tile t+2 */  \
        /* phase 1: B columns 0..31, A rows 0..63 */                                                       \
        if (ABL != 5 && ABL != 6 && ABL != 12) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                          \
            bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff0);                           \
            bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff1);                           \
        }                                                                                                  \
        if (!(ABL == 3 && (KT) % 3 == 1) && ABL != 5 && ABL != 6 && ABL != 12) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                  \
            af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                           \
            af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                           \
        }                                                                                                  \
        if (B3) {                                                                                          \
            if ((KT) + 2 < nk) NOMAD_P8_DMA_B((KT) + 2, 0)                                                 \
        } else if ((KT) + 1 < nk) NOMAD_P8_DMA_B((KT) + 1, 0)                                              \
        NOMAD_P8_SYNC_COMPUTE(0, 0)                                                                        \
        /* phase 2: A rows 64..127 */                                                                      \
        if (!(ABL == 3 && (KT) % 3 == 1) && ABL != 5 && ABL != 6 && ABL != 12) _Pragma("unroll") for (int i = 4; i < 8; ++i) {                  \
            af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                           \
            af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                           \
        }                                                                                                  \
        if (B3) {                                                                                          \
            if ((KT) + 2 < nk) NOMAD_P8_DMA_B((KT) + 2, 1)                                                 \
        } else if ((KT) + 1 < nk) NOMAD_P8_DMA_B((KT) + 1, 1)                                              \
        NOMAD_P8_SYNC_COMPUTE(4, 0)                                                                        \
        /* phase 3: B columns 32..63 */                                                                    \
        if (ABL != 5 && ABL != 6 && ABL != 12) _Pragma("unroll") for (int j = 0; j < NJ - 2; ++j) {                     \
            bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff0);                     \
            bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff1);                     \
        }                                                                                                  \
        NOMAD_P8_SYNC_COMPUTE(0, 2)                                                                        \
        /* phase 4: both A halves of tile t+2 (their last read was phase 2), then "tile t+1 has landed" */ \
        if ((KT) + 2 < nk) {                                                                               \
            NOMAD_P8_DMA_A((KT) + 2, 0)                                                                    \
            NOMAD_P8_DMA_A((KT) + 2, 1)                                                                    \
            if (B3 && NJ == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                            \
            else if (B3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                  \
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                          \
        } else {                                                                                           \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
        }                                                                                                  \
        NOMAD_P8_SYNC_COMPUTE(4, 2)                                                                        \
        if (B3) b3_cur = b3_cur >= 2 * A_BUF ? 0 : b3_cur + A_BUF;                                         \
    }

    for (int kt = 0; kt < nk; kt += 2) {
        NOMAD_P8_KTILE(kt, 0)
        NOMAD_P8_KTILE(kt + 1, 1)
    }
    if (wr == 0 && ABL != 12) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
#undef NOMAD_P8_KTILE
#undef NOMAD_P8_SYNC_COMPUTE
#undef NOMAD_P8_MMA
#undef NOMAD_P8_DMA_A
#undef NOMAD_P8_DMA_B

    if (ABL == 7) ts_[2] = wall_clock64();
    p8_epilogue<ABL == 1, X3, (ABL == 8 || (ABL >= 13 && ABL <= 16) ? 1 : ABL == 9 ? 3 : ABL == 10 ? 2 : 0), NJ, RPRE, PLAIN>(p, acc, smem8, grp, m0, n0, wave, wr, wc, lane, fr, fq);
    if (ABL == 7) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the output stores have left the CU
        ts_[3] = wall_clock64();
        if (tid == 0 && blockIdx.x < kTimelineSlots) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = g_timeline + (size_t)blockIdx.x * 6;
            o[0] = ts_[0]; o[1] = ts_[1]; o[2] = ts_[2]; o[3] = ts_[3]; o[4] = hw; o[5] = xcc;
        }
    }
}

template <int ABL = 0, bool BUFLD = false, int X3 = 0, int NB = 2, int NJ = 4, bool RPRE = false, bool PLAIN = false>
inline hipError_t launch_gemm_bf16_8phase(GemmParams p, int groups, hipStream_t s) {
    p.tiles_m = (p.M + P8Cfg::BM - 1) / P8Cfg::BM;
    p.tiles_n = p.N / (64 * NJ);
    if (PLAIN) {
        p.a_clip_magic = p.tn_magic = 0;
        p.a_clip_shift = p.tn_shift = 0;
        if (p.amap.clip_rows < p.M) fast_div_magic((unsigned)p.amap.clip_rows, &p.a_clip_magic, &p.a_clip_shift);
        if (p.tiles_n > 1) fast_div_magic((unsigned)p.tiles_n, &p.tn_magic, &p.tn_shift);
    }
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_bf16_8phase_kernel<ABL, BUFLD, X3, NB, NJ, RPRE, PLAIN>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_bf16_8phase_kernel<ABL, BUFLD, X3, NB, NJ, RPRE, PLAIN>), dim3(p.tiles_m * p.tiles_n, groups), dim3(P8Cfg::THREADS),
                       NB == 3 ? 160 * 1024 : P8Cfg::LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace nomad
