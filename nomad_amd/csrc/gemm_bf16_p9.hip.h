// bf16 GEMM, 256 x 256 tile, PERSISTENT form of the deep-pipelined "8-phase" schedule of gemm_bf16_8phase.hip.h (round 5,
// config C5: every plain transformer / conv GEMM of the bf16 forward).
//
// Why: the per-workgroup timeline of the one-tile-per-workgroup kernel (profiles/r02_gemm_timeline.txt, re-measured in
// profiles/r05_gemm_bf16_p9.txt) has a K = 768 tile spend 22.6 us in its K loop and ~9 us around it - 3 us from entry to the
// first MFMA (address set-up, the first two K tiles' round trip to L2 / HBM), 6 us of epilogue (a workgroup barrier that also
// waits for the bias / residual loads, fp32 slabs through LDS, store acknowledgements) - and with 128-160 KB of LDS per
// workgroup nothing else is resident on the CU to fill those gaps.  Here ONE workgroup per CU stays resident and walks a
// contiguous run of tiles of its XCD:
//   * the LDS-DMA stream never drains: the load cursor runs two K tiles ahead of the multiply cursor and crosses into the
//     next output tile first, so the next tile's K loop starts on operands that are already in LDS;
//   * the epilogue is direct from TRANSPOSED accumulators (W fragment as the first MFMA operand): a lane owns one output row and,
//     per pair of 16-column accumulator tiles, 8 consecutive columns - one 16-byte bf16 store, 64 contiguous bytes per row
//     and instruction - no LDS (it cannot collide with the staged K tiles), no barrier;  the W rows of a wave's 64-column
//     group are PERMUTED on their way into LDS (the per-lane DMA source address is free) so that the 8 values a lane holds for
//     accumulator tiles 2 jh and 2 jh + 1 are those 8 consecutive columns;
//   * bias and residual come through buffer descriptors whose size ends at row M: rows past the end load zero / store nothing.
// Same MFMA (v_mfma_f32_16x16x32_bf16), same k order per output element, (acc + bias) -> GELU -> + residual in fp32 as every
// other bf16 kernel: bit-identical to them (tests/test_gpu_bf16.py), so a clip's bits still do not depend on its batch.
//
// Requirements (the caller checks, p9_applies in nomad_hip.hip): C and R plain row-major matrices, one group, N % 256 == 0,
// n_valid == N, K % 128 == 0, contiguous K (kchunk == K), A a plain matrix or uniform clips of >= 2 rows, no split planes.
//
// Synchronisation is the 8-phase template's (cdna_hip_programming.md 5, gemm_bf16_8phase.hip.h): raw s_barrier, lgkmcnt(0)
// before each MFMA cluster, ONE counted vmcnt per K tile, the two wave rows one barrier apart.  vmcnt(8) in phase 4 means "all
// but the 8 newest vector-memory operations of this wave are complete"; the 8 newest are always the DMA of K tile t + 2 (4 B + 4
// A instructions), and loads complete in order, so A / B of tile t + 1 have landed whatever the epilogue's stores in between do.
#pragma once
#include <hip/hip_runtime.h>

#include "dtypes.hip.h"
#include "gemm_bf16_8phase.hip.h"
#include "gemm_f32.hip.h"

namespace nomad {

// s_waitcnt vmcnt(N) as the real instruction (__builtin_amdgcn_s_waitcnt), not as inline asm: the compiler's wait-count pass
// understands it - it knows every vector-memory operation older than the N newest complete behind it, so it adds no wait of its own
// (which would be vmcnt(0): a drained LDS-DMA queue) for an ordinary load that one of these counted waits already covers.
// gfx9 encoding of the immediate: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 (other counters left at "do not wait").
__device__ __forceinline__ constexpr int p9_vmcnt(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }
#define NOMAD_P9_WAIT_VM(N)                              \
    {                                                    \
        __builtin_amdgcn_s_waitcnt(p9_vmcnt(N));         \
        asm volatile("" ::: "memory");                   \
    }

// LDS position p (0..63) of a wave's 64-column W group holds W row perm(p): accumulator tile j = p >> 4, operand row i16 = p & 15
// (= 4 fq + r of the transposed result) -> column 32 (j >> 1) + 8 fq + 4 (j & 1) + r.
__host__ __device__ constexpr int p9_wperm(int p) { return 32 * (p >> 5) + 8 * ((p >> 2) & 3) + 4 * ((p >> 4) & 1) + (p & 3); }

// ABL: 0 = product; 7 = per-workgroup timeline probe (tools/p9_timeline.py); 1 = no output stores (timing probe)
// INTER (round 5, second step; reshaped in round 6): the epilogue of a tile WITHOUT a residual is interleaved into the first K tile of the
//   workgroup's NEXT tile.  The accumulators of the finished tile stay where they are; in PHASE 1 of the next tile's K tile 0 a wave adds the
//   bias, applies GELU, converts and stores its whole 128 x 64 wave tile and zeroes it - ONE hook (round 5 had four, one 64 x 32 quadrant per
//   phase, each beside the other wave row's MFMA cluster; round 6 measured a hook at 1 360 vector-issue cycles with GELU against the cluster's
//   256, and two waves of a SIMD in vector code issue at twice one wave's rate): wave row 1 runs its hook in its phase-1 load slot, wave row 0
//   - one barrier ahead - behind the phase's barrier, in front of its own MFMAs, i.e. in the SAME time slot.  In phase 1 the A fragments of rows
//   64 .. 127 (loaded in phase 2) are dead; their registers carry the hook's 16 bias values and temporaries.  The last tile of a workgroup is
//   flushed by ONE more K tile 0 (same code; its matrix work lands in zeroed accumulators that nobody stores).  Tiles WITH a residual keep the
//   epilogue between tiles (their 16 residual loads per lane need 64 registers, which only exist there).
//   The bias reaches the epilogue without a wait of its own: lane l of a wave holds bias[n0 + 64 wc + l] in ONE register, loaded in K
//   tile 1 of the tile - the counted vmcnt of that K tile's phase 4 covers it (it is older than the 8 newest operations), and because
//   that wait is a real s_waitcnt instruction (NOMAD_P9_WAIT_VM) the compiler knows so and never drains the LDS-DMA queue for it
//   (cdna_hip_programming.md 5, "mixing load kinds") - and distributed with ds_bpermute_b32.  (A first version loaded it by inline asm
//   and tied the register to an asm wait: the compiler copied the register BEFORE the wait - stale bias in some runs.)
//   vmcnt in the interleaved K tile: its phase 4 must see K tile 1 landed, whose DMA was issued before 4 + 16 + 4 = 24 newer
//   operations (2 + 2 B DMA, the hook's 16 stores, 4 A DMA): vmcnt(24); everywhere else vmcnt(8) as before.  (Round 5's four hooks had 12
//   stores in front of the wait: vmcnt(20).)
// DMAP: in which phase slots the B tile of K tile t + 2 is issued: 0 = phases 1 / 2 (next to the 12 / 8 fragment reads of those slots, as
//   the one-tile-per-workgroup kernel does), 1 = both halves in phase 3 (4 fragment reads), 2 = phases 2 / 3.  The microarchitecture guide
//   prices an LDS-DMA instruction at 100-185 issue cycles inside a slot that already carries many LDS reads and at 25-60 in a quiet one;
//   the counted waits do not change (the 8 newest operations at the phase-4 wait are the same two tiles either way).  3 = one half-tile per
//   MFMA cluster, issued between the cluster's two k-halves (B half 0 / 1, A half 0 / 1 in clusters 1 .. 4): the DMA issue sits in the shadow
//   of the cluster's own MFMAs (8 of every 16 cycles of a 16x16x32 MFMA hold the SIMD's vector issue; the rest is free).  A half may be
//   re-staged there: a wave in cluster 3 has passed barrier 5, so the other wave row has finished ITS phase-2 reads.  Phase 4 then waits
//   vmcnt(6) (B and A half 0 of tile t + 2 are newer than what must have landed).
template <int ABL = 0, bool INTER = true, int DMAP = 0>
__global__ __launch_bounds__(512) void gemm_bf16_p9_kernel(const GemmParams p) {
    using Cfg = P8Cfg;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int A_BUF = 2 * Cfg::HALF_BYTES;   // A buffers at 0 / 32 KB, B buffers (three) from 64 KB on: 160 KB
    extern __shared__ __attribute__((aligned(16))) char smem9[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    // SHORT tiles (round 6): p.p9_short = 1 makes the workgroup tile 192 x 256 - each wave row owns 96 rows (row tiles 0 .. 5 of its
    // eight), the LDS image keeps its 128-row halves with the last 32 rows of each unused.  A RUN-TIME mode of the one instantiation,
    // not a second kernel (two large kernels alternating between launches cost the launch after each switch 12-17 us of instruction
    // cache, which is what ate the 256 x 192 tiles' gain in round 3): the phases keep their shape, phases 2 and 4 multiply two row
    // tiles instead of four behind scalar branches.  Every wave still issues 8 LDS-DMA instructions per K tile and every hooked K tile
    // 12 stores: the compiler's wait-count pass is not path-sensitive, and with the instruction count (or the counted wait) behind a
    // branch it no longer sees that the K tile's counted vmcnt covers the bias load and drains the LDS-DMA queue (vmcnt(0)) in front
    // of every bias use.  So the surplus is made harmless by DATA instead: the second instruction of an A half in waves 4 .. 7 (rows
    // 96 .. 127 of the half, which nobody reads) fetches one 16-byte chunk for all lanes (one cache line instead of sixteen), the
    // stores of row tiles 6, 7 go past the end of the output descriptor.
    // For the N = 768 GEMMs of config C5 (out_proj, fc2): 188 x 3 = 564 tiles are 2.2 rounds of the 256 CUs, 250 x 3 = 750 are 2.93.
    const bool sh = p.p9_short != 0;
    const int bm_rows = sh ? 192 : 256, half_rows = sh ? 96 : 128;
    const bool a_skip = sh && wave >= 4;
    unsigned long long ts_[6] = {0, 0, 0, 0, 0, 0};
    if (ABL == 7) {
        // probe (round 6): are the epilogue's memory bursts expensive because every CU issues them at the same time?  Start the four groups
        // of workgroups p9_skew x 10 ns apart and compare the stamps of tile 1 (tools/p9_timeline.py --skew)
        if (p.p9_skew != 0) {   // (> 0: four groups; < 0: sixteen groups of 16 CUs, |p9_skew| x 10 ns apart)
            const unsigned long long until = wall_clock64() + (unsigned long long)(p.p9_skew < 0 ? -p.p9_skew : p.p9_skew) * (((unsigned)blockIdx.x >> 3) & (p.p9_skew < 0 ? 15u : 3u));
            while (wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
        }
        ts_[0] = wall_clock64();
    }

    // this workgroup's run of tiles: XCD x (blocks b with b % 8 == x share an XCD) owns a contiguous range of the n-fastest
    // tile walk, its workgroups take every wpx-th tile of it - tiles that are resident together share A row panels in L2
    const int ntiles = p.tiles_m * p.tiles_n;
    const int wpx = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3;
    const int tq = ntiles >> 3, trem = ntiles & 7;
    const int t_begin = xcd * tq + (xcd < trem ? xcd : trem) + local;
    const int t_end = xcd * tq + (xcd < trem ? xcd : trem) + tq + (xcd < trem ? 1 : 0);
    if (t_begin >= t_end) return;   // (uniform over the workgroup; before any barrier)

    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* Wg = reinterpret_cast<const bf16_t*>(p.W);
    auto row_addr_a = [&](int m) -> long long {
        const int c = p.a_clip_magic ? fast_div(m, p.a_clip_magic, p.a_clip_shift) : 0;
        return p.amap.off + (long long)c * p.amap.clip_stride + (long long)(m - c * p.amap.clip_rows) * p.amap.ld;
    };

    // tile-independent per-lane DMA geometry.  Instruction i of a half-tile covers LDS rows (tid + 512 i) / 8, physical chunk
    // (tid + 512 i) % 8; the source chunk is swizzled (gemm_bf16.hip.h).  B: LDS row R holds W row (R / 64) * 64 + perm(R % 64).
    unsigned b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 512, row = id >> 3, pc = id & 7;
        const int sw = (pc ^ ((row >> 1) & 7)) * 8;
        const int wrow = (row & ~63) + p9_wperm(row & 63);
        b_off[i] = (unsigned)(((long long)wrow * p.ldw + sw) * 2);
    }
    // load cursor: the output tile whose K tiles are being staged, and the byte offset of the next K tile within its rows
    int t_ld = t_begin, m0_ld, n0_ld;
    unsigned a_off[2][2];
    const char *a_base, *b_base0, *b_base1;
    auto setup = [&](int t) {
        const int tm = p.tn_magic ? fast_div(t, p.tn_magic, p.tn_shift) : t;
        m0_ld = tm * bm_rows;
        n0_ld = (t - tm * p.tiles_n) * Cfg::BN;
        const long long row0 = row_addr_a(m0_ld < p.M ? m0_ld : p.M - 1);   // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 512, row = id >> 3, pc = id & 7;
            const int sw = (pc ^ ((row >> 1) & 7)) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int m = m0_ld + h * half_rows + row;
                m = m < p.M ? m : p.M - 1;
                a_off[h][i] = (a_skip && i == 1) ? 0u : (unsigned)((row_addr_a(m) - row0 + sw) * 2);
            }
        }
        a_base = reinterpret_cast<const char*>(uniform_ptr(reinterpret_cast<const float*>(Ag + row0)));
        b_base0 = reinterpret_cast<const char*>(uniform_ptr(reinterpret_cast<const float*>(Wg + (long long)n0_ld * p.ldw)));
        b_base1 = b_base0 + (long long)128 * p.ldw * 2;
    };
    unsigned ko = 0;                         // byte offset of the load cursor's K tile (contiguous K: 128 bytes per K tile)
    const unsigned k_bytes = (unsigned)p.K * 2;
    char* const dma_dst = smem9 + wave * 1024;  // + lane * 16 implicit (lane-linear LDS-DMA destination)
    int b3_cur = 0;    // byte offset of the B buffer of the K tile being multiplied (rotates through three) ...
    int b3_dst = 0;    // ... and of the buffer the load cursor fills

#define NOMAD_P9_DMA_A(PAR, H)                                                                                            \
    {                                                                                                                     \
        char* d_ = dma_dst + (PAR)*A_BUF + (H)*Cfg::HALF_BYTES;                                                           \
        __builtin_amdgcn_global_load_lds((gptr_t)(a_base + (a_off[H][0] + ko)), (lptr_t)(d_), 16, 0, 0);                  \
        __builtin_amdgcn_global_load_lds((gptr_t)(a_base + (a_off[H][1] + ko)), (lptr_t)(d_ + 8192), 16, 0, 0);           \
    }
#define NOMAD_P9_DMA_B(H)                                                                                                 \
    {                                                                                                                     \
        char* d_ = dma_dst + 2 * A_BUF + b3_dst + (H)*Cfg::HALF_BYTES;                                                    \
        const char* bb_ = (H) ? b_base1 : b_base0;                                                                        \
        __builtin_amdgcn_global_load_lds((gptr_t)(bb_ + (b_off[0] + ko)), (lptr_t)(d_), 16, 0, 0);                        \
        __builtin_amdgcn_global_load_lds((gptr_t)(bb_ + (b_off[1] + ko)), (lptr_t)(d_ + 8192), 16, 0, 0);                 \
    }
    // the load cursor moves on by one K tile; past the last K tile of its output tile it crosses into the workgroup's next
    // output tile (the last one "crosses" into itself: two K tiles are staged that nobody reads - no has-next case in the loop)
#define NOMAD_P9_ADVANCE()                                                                                                \
    {                                                                                                                     \
        ko += 128;                                                                                                        \
        if (ko == k_bytes) {                                                                                              \
            ko = 0;                                                                                                       \
            if (t_ld + wpx < t_end) t_ld += wpx;                                                                          \
            setup(t_ld);                                                                                                  \
        }                                                                                                                 \
    }

    // prologue: K tiles 0 and 1 of the first output tile
    setup(t_ld);
    int m0 = m0_ld, n0 = n0_ld, t_cur = t_begin;
    NOMAD_P9_DMA_A(0, 0)
    NOMAD_P9_DMA_A(0, 1)
    NOMAD_P9_DMA_B(0)
    NOMAD_P9_DMA_B(1)
    NOMAD_P9_ADVANCE()
    b3_dst = A_BUF;
    NOMAD_P9_DMA_A(1, 0)
    NOMAD_P9_DMA_A(1, 1)
    NOMAD_P9_DMA_B(0)
    NOMAD_P9_DMA_B(1)
    NOMAD_P9_ADVANCE()
    NOMAD_P9_WAIT_VM(8)   // K tile 0 has landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ABL == 7) ts_[1] = wall_clock64();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // second wave row runs one barrier behind (ping-pong)

    // fragment addresses: this lane's row fr of a 16-row tile, chunk (4 kh + fq) ^ swizzle
    const int sw = (fr >> 1) & 7;
    const int koff0 = ((0 + fq) ^ sw) * 16, koff1 = ((4 + fq) ^ sw) * 16;
    const int a_frag = wr * Cfg::HALF_BYTES + fr * 128;                 // + i * 2048
    const int b_frag = 2 * A_BUF + (wc * 64 + fr) * 128;                // + j * 2048

    const int nk = p.K / 64;  // even
    const bool has_r = p.R != nullptr, has_b = p.bias != nullptr;
    const bool inter = INTER && !has_r && ABL != 1;   // this problem's epilogues are interleaved into the next tile's first K tile
    const bool late_hook = wr == 0 && p.p9_late != 0;   // (see NOMAD_P9_EPI_HOOK_LATE)
    // whole-line output stores (NOMAD_P9_LINE_SWAP): measured and left off - a run-time choice in libnomad_diag.so only.  As a run-time flag of
    // the product it cost every hook 8 v_mov per row tile (the merge of the two paths' store registers: 6 % of a GELU hook's vector
    // instructions, 20 % of a plain one's) and a branch per row tile.
#ifdef NOMAD_DIAG
    const bool wl = p.p9_wl != 0;
#else
    constexpr bool wl = false;
#endif
    int n_done = 0;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[8][2], bf[2][2];
    float bias_lane = 0.f;   // bias[n0 + 64 wc + lane] of the tile being multiplied (of the finished tile during its interleaved epilogue)
    bool pend = false;       // a finished tile's accumulators are waiting for their (interleaved) epilogue ...
    bool flush = false;      // the pass after the last tile: one more K tile 0 (on the operands staged past the end) for its hooks
    // ... whose output descriptor (base = first row of this wave's 128 x 64 part, size = up to row M) is:
    __amdgpu_buffer_rsrc_t rc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(p.C)), 0, 0, 0x00020000);
    auto clamp_bytes = [](long long v) { return (unsigned)(v < 0 ? 0 : (v > (1ll << 30) ? (1ll << 30) : v)); };
    auto out_rsrc = [&](int m0_, int n0_) {
        const int mw = m0_ + wr * half_rows, nw = n0_ + wc * 64;
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(uniform_ptr(reinterpret_cast<const float*>(reinterpret_cast<bf16_t*>(p.C) + p.cmap.off + (long long)mw * p.cmap.ld + nw))), 0,
            clamp_bytes((long long)(p.M - mw) * p.cmap.ld * 2), 0x00020000);
    };

    // epilogue of accumulator row tiles I0 .. I0 + 3 (64 rows x the wave's 64 columns) straight from the registers:
    // acc[i][j][r] = out[m0 + 128 wr + 16 i + fr][n0 + 64 wc + 32 (j >> 1) + 8 fq + 4 (j & 1) + r]; leaves them zero.
    // WHOLE-LINE stores (round 6, tools/micro/store_pattern.hip): as it stands an accumulator gives a store instruction of 16 rows x 64 bytes
    // - sixteen half cache lines, the other halves in a second instruction - and 256 CUs storing their tiles that way reach 4.2 TB/s
    // (8.0 us per 128 KB tile) where 8 rows x 128 bytes per instruction reach 5.6-5.9 (6.0 us).  A lane holds chunk fq of BOTH 64-byte
    // halves of its row fr; lanes fr and fr + 8 of a 16-lane row swap one chunk each (DPP row_ror:8 under a bank mask: two instructions and a
    // copy per register) so that instruction X carries rows 0 .. 7 of the row tile whole (lanes fr < 8: half 0, lanes fr >= 8: half 1 of row
    // fr - 8) and instruction Y rows 8 .. 15.  Same bytes, same places.
#define NOMAD_P9_LINE_SWAP(P0, P1)                                                                                        \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                                    \
        const int t_ = (int)(P1)[r_];                                                                                     \
        (P1)[r_] = (unsigned)__builtin_amdgcn_update_dpp((int)(P1)[r_], (int)(P0)[r_], 0x128, 0xF, 0x3, false);  /* Y: lanes 0-7 <- half 0 of rows 8-15 */ \
        (P0)[r_] = (unsigned)__builtin_amdgcn_update_dpp((int)(P0)[r_], t_, 0x128, 0xF, 0xC, false);              /* X: lanes 8-15 <- half 1 of rows 0-7 */ \
    }
#define NOMAD_P9_EPI_ROWS(I0, NI, GELU_)                                                                                   \
    {                                                                                                                     \
        int lane_e = lane;                                                                                                \
        asm volatile("" : "+v"(lane_e));  /* offsets recomputed here: hoisted, they would sit in registers through the K loop */ \
        const int fr_e = lane_e & 15, fq_e = lane_e >> 4;                                                                 \
        /* (wl = false, A/B: the accumulator's own shape - instruction X = the row's first 64 bytes, Y = its second) */   \
        const int c_voff = wl ? ((fr_e & 7) * p.cmap.ld) * 2 + 16 * (4 * (fr_e >> 3) + fq_e) : (fr_e * p.cmap.ld + 8 * fq_e) * 2; \
        const int y_step = wl ? 8 * p.cmap.ld * 2 : 64;                                                                   \
        float bq[2][8];                                                                                                   \
        _Pragma("unroll") for (int jh = 0; jh < 2; ++jh)                                                                  \
            _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                                 \
                bq[jh][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((32 * jh + 8 * fq_e + e) * 4, __builtin_bit_cast(int, bias_lane))); \
        _Pragma("unroll") for (int i = (I0); i < (I0) + (NI); ++i) {                                                      \
            bf16x8 o0, o1;                                                                                                \
            _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                                 \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                           \
                    float x0 = acc[i][g][r] + bq[0][4 * g + r], x1 = acc[i][2 + g][r] + bq[1][4 * g + r];                 \
                    if (GELU_) {                                                                                          \
                        x0 = gelu_bf16out(x0);                                                                            \
                        x1 = gelu_bf16out(x1);                                                                            \
                    }                                                                                                     \
                    o0[4 * g + r] = (bf16_t)x0;                                                                           \
                    o1[4 * g + r] = (bf16_t)x1;                                                                           \
                }                                                                                                         \
            u32x4 px = __builtin_bit_cast(u32x4, o0), py = __builtin_bit_cast(u32x4, o1);                                 \
            if (wl) NOMAD_P9_LINE_SWAP(px, py)                                                                            \
            /* (row offset in the VGPR offset, never in the scalar offset: the store-data hazard of DESIGN.md 5) */       \
            /* (short tiles: row tiles 6, 7 of a wave do not exist - their stores fall past the end of the descriptor) */ \
            const int o_ = c_voff + i * 16 * p.cmap.ld * 2 + ((i >= 6 && sh) ? 0x40000000 : 0);                           \
            __builtin_amdgcn_raw_buffer_store_b128(px, rc_p, o_, 0, 2);                                                   \
            __builtin_amdgcn_raw_buffer_store_b128(py, rc_p, o_ + y_step, 0, 2);                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};                        \
        }                                                                                                                 \
    }
    // Round 6, LATE hooks of the first wave row in GELU problems.  A hook with GELU is ~340 vector instructions (1 360 issue cycles for one
    // wave) against the 256 cycles of the partner wave's MFMA cluster beside it: hooked phases are vector-issue time, and with the two rows'
    // hooks in ALTERNATE slots (each next to the other row's matrix work) a SIMD issues them one wave at a time - 4 cycles per instruction.
    // Two waves issuing vector instructions together get 2 cycles per instruction.  So wave row 0 runs its hook of phase q not in its load
    // slot but right behind the barrier, in front of its own MFMAs of that phase - the slot in which wave row 1 (one barrier behind) runs
    // ITS hook of phase q: both hooks at once, then row 0's cluster.  Per hooked phase 1 360 + 256 + 256 instead of 2 x 1 360 cycles.
    // The count and order of a wave's vector-memory operations in front of the phase-4 wait do not change.  Without GELU a hook is shorter
    // than a cluster and the alternate placement stays.
    // (with whole-line stores a hook needs both column halves of a row at once, and 16 bias values instead of 8: the ONE hook of a K tile 0
    // sits in phase 1 and takes the whole 128 x 64 wave tile - there the A fragments of rows 64 .. 127, loaded in phase 2, are dead and
    // their 32 registers carry it; two hooks in phases 1 and 2 spilled.  JH != 0 / I0 != 0 mark the call sites without a hook.)
#define NOMAD_P9_EPI_HOOK(I0, JH)                                        \
    if ((I0) == 0 && (JH) == 0 && epi_now && !late_hook) {               \
        __builtin_amdgcn_sched_barrier(0);                               \
        if (p.gelu) NOMAD_P9_EPI_ROWS(0, 8, true)                        \
        else NOMAD_P9_EPI_ROWS(0, 8, false)                              \
        __builtin_amdgcn_sched_barrier(0);                               \
    }
#define NOMAD_P9_EPI_HOOK_LATE(I0, JH)                                   \
    if ((I0) == 0 && (JH) == 0 && epi_now && late_hook) {                \
        __builtin_amdgcn_sched_barrier(0);                               \
        if (p.gelu) NOMAD_P9_EPI_ROWS(0, 8, true)                        \
        else NOMAD_P9_EPI_ROWS(0, 8, false)                              \
        __builtin_amdgcn_sched_barrier(0);                               \
    }

    // transposed product: W fragment first -> a lane holds row fr (of A tile i) x W rows 4 fq + r (of LDS tile j)
#define NOMAD_P9_MMA_ROWS(I0, J0, KH, IB, IE)                                                              \
        _Pragma("unroll") for (int i = (IB); i < (IE); ++i)                                                \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
                acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][KH], af[(I0) + i][KH], acc[(I0) + i][(J0) + j], 0, 0, 0);
    // (short tiles: the second half of a wave's rows is two row tiles, not four - a scalar branch around the other two's MFMAs)
#define NOMAD_P9_MMA_KH(I0, J0, KH)                                                                        \
        NOMAD_P9_MMA_ROWS(I0, J0, KH, 0, 2)                                                                \
        if ((I0) == 0 || !sh) NOMAD_P9_MMA_ROWS(I0, J0, KH, 2, 4)
    // MID (DMAP == 3 only): a half-tile of LDS-DMA issued between the cluster's two k-halves, in the shadow of its MFMAs
    // (1 / 2 = B half 0 / 1, 3 / 4 = A half 0 / 1 of K tile t + 2; PAR = the A buffer's parity)
#define NOMAD_P9_SYNC_COMPUTE(I0, J0, MID, PAR, HOOKS, HI0, HJH) \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    if (HOOKS) NOMAD_P9_EPI_HOOK_LATE(HI0, HJH)         \
    __builtin_amdgcn_s_setprio(1);                      \
    NOMAD_P9_MMA_KH(I0, J0, 0)                          \
    if (DMAP == 3) {                                    \
        __builtin_amdgcn_sched_barrier(0);              \
        if ((MID) == 1) NOMAD_P9_DMA_B(0)               \
        if ((MID) == 2) NOMAD_P9_DMA_B(1)               \
        if ((MID) == 3) NOMAD_P9_DMA_A(PAR, 0)          \
        if ((MID) == 4) NOMAD_P9_DMA_A(PAR, 1)          \
        __builtin_amdgcn_sched_barrier(0);              \
    }                                                   \
    NOMAD_P9_MMA_KH(I0, J0, 1)                          \
    __builtin_amdgcn_s_setprio(0);                      \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");

    // one K tile (A buffer PAR = parity of the K tile: a compile-time constant per call site, K tiles go in pairs).
    // HOOKS: the K tile 0 call site - epilogue hooks of the pending tile (epi_now).
    // BIAS: the K tile 1 call site - the tile's bias register is loaded in the first pair (kt == 0).
#define NOMAD_P9_KTILE(PAR, HOOKS, BIAS)                                                                   \
    {                                                                                                      \
        const char* la_ = smem9 + (PAR)*A_BUF + a_frag;                                                    \
        const char* lb_ = smem9 + b3_cur + b_frag;                                                         \
        b3_dst = b3_cur >= A_BUF ? b3_cur - A_BUF : b3_cur + 2 * A_BUF; /* buffer of tile t+2 */           \
        if ((BIAS) && kt == 0 && has_b) bias_lane = p.bias[n0 + wc * 64 + lane];                           \
        /* phase 1: B columns 0..31, A rows 0..63 */                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
            bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff0);                           \
            bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + j * 2048 + koff1);                           \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
            af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                           \
            af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                           \
        }                                                                                                  \
        if (DMAP == 0) NOMAD_P9_DMA_B(0)                                                                   \
        if (HOOKS) NOMAD_P9_EPI_HOOK(0, 0)                                                                 \
        NOMAD_P9_SYNC_COMPUTE(0, 0, 1, PAR, HOOKS, 0, 0)                                                   \
        /* phase 2: A rows 64..127 */                                                                      \
        _Pragma("unroll") for (int i = 4; i < 6; ++i) {                                                    \
            af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                           \
            af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                           \
        }                                                                                                  \
        if (!sh) {                                                                                         \
            _Pragma("unroll") for (int i = 6; i < 8; ++i) {                                                \
                af[i][0] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff0);                       \
                af[i][1] = *reinterpret_cast<const bf16x8*>(la_ + i * 2048 + koff1);                       \
            }                                                                                              \
        }                                                                                                  \
        if (DMAP == 0) NOMAD_P9_DMA_B(1)                                                                   \
        if (DMAP == 2) NOMAD_P9_DMA_B(0)                                                                   \
        if (HOOKS) NOMAD_P9_EPI_HOOK(4, 0)                                                                 \
        NOMAD_P9_SYNC_COMPUTE(4, 0, 2, PAR, HOOKS, 4, 0)                                                   \
        /* phase 3: B columns 32..63 */                                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
            bf[j][0] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff0);                     \
            bf[j][1] = *reinterpret_cast<const bf16x8*>(lb_ + (2 + j) * 2048 + koff1);                     \
        }                                                                                                  \
        if (DMAP == 1) NOMAD_P9_DMA_B(0)                                                                   \
        if (DMAP == 1 || DMAP == 2) NOMAD_P9_DMA_B(1)                                                      \
        if (HOOKS) NOMAD_P9_EPI_HOOK(0, 1)                                                                 \
        NOMAD_P9_SYNC_COMPUTE(0, 2, 3, PAR, HOOKS, 0, 1)                                                   \
        /* phase 4: both A halves of tile t+2 (their last read was phase 2), then "tile t+1 has landed" */ \
        if (DMAP != 3) {                                                                                   \
            NOMAD_P9_DMA_A(PAR, 0)                                                                         \
            NOMAD_P9_DMA_A(PAR, 1)                                                                         \
            if ((HOOKS) && epi_now) NOMAD_P9_WAIT_VM(24)   /* 2 + 2 B DMA, 2 x 8 hook stores, 4 A DMA */            \
            else NOMAD_P9_WAIT_VM(8)                                                                       \
            NOMAD_P9_ADVANCE()                                                                             \
        } else {  /* B and A half 0 of tile t+2 were issued in clusters 1-3: 6 newer operations (+ 12 stores of the hooks) */ \
            if ((HOOKS) && epi_now) NOMAD_P9_WAIT_VM(22)                                                   \
            else NOMAD_P9_WAIT_VM(6)                                                                       \
        }                                                                                                  \
        if (HOOKS) NOMAD_P9_EPI_HOOK(4, 1)                                                                 \
        NOMAD_P9_SYNC_COMPUTE(4, 2, 4, PAR, HOOKS, 4, 1)                                                   \
        if (DMAP == 3) NOMAD_P9_ADVANCE()                                                                  \
        b3_cur = b3_cur >= 2 * A_BUF ? 0 : b3_cur + A_BUF;                                                 \
    }

    for (;;) {
        int kt = 0;
        do {   // (nk >= 2: no zero-trip copy of the loop's live ranges)
            const bool epi_now = pend && kt == 0;
            NOMAD_P9_KTILE(0, true, false)
            if (epi_now) pend = false;
            if (flush) break;
            NOMAD_P9_KTILE(1, false, true)
            kt += 2;
        } while (kt < nk);
        if (flush) break;
        if (ABL == 7 && n_done < 2) ts_[2 + 2 * n_done] = wall_clock64();

        if (inter) {   // the accumulators stay; the next tile's first K tile (or the flush pass) stores them
            pend = true;
            rc_p = out_rsrc(m0, n0);
        } else {
            // ---- epilogue between tiles (residual GEMMs; every GEMM of the INTER = false instantiation) ----
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int fr_e = lane_e & 15, fq_e = lane_e >> 4;
            const int mw = m0 + wr * half_rows, nw = n0 + wc * 64;
            const bf16_t* Rb = reinterpret_cast<const bf16_t*>(p.R);
            const __amdgpu_buffer_rsrc_t rc = out_rsrc(m0, n0);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(uniform_ptr(reinterpret_cast<const float*>(has_r ? Rb + p.rmap.off + (long long)mw * p.rmap.ld + nw : reinterpret_cast<const bf16_t*>(p.C)))), 0,
                has_r ? clamp_bytes((long long)(p.M - mw) * p.rmap.ld * 2) : 0u, 0x00020000);
            const int cl_voff = wl ? ((fr_e & 7) * p.cmap.ld) * 2 + 16 * (4 * (fr_e >> 3) + fq_e) : (fr_e * p.cmap.ld + 8 * fq_e) * 2;
            const int cl_ystep = wl ? 8 * p.cmap.ld * 2 : 64, r_voff = (fr_e * p.rmap.ld + 8 * fq_e) * 2;
            // one straight-line copy per (GELU, residual) combination: decided once per tile, not once per chunk
            auto epi = [&](auto gelu_c, auto res_c) {
                constexpr bool GELU = decltype(gelu_c)::value, RES = decltype(res_c)::value;
                // the whole residual (64 registers: the operand fragments are dead here) ahead of the first store: a load issued
                // behind a store is waited for together with that store's acknowledgement (in-order vmcnt)
                u32x4 rres[8][2];
                if (RES) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i >= 6 && sh) continue;   // short tiles: 6 row tiles per wave
#pragma unroll
                        for (int jh = 0; jh < 2; ++jh)
                            rres[i][jh] = __builtin_amdgcn_raw_buffer_load_b128(rr, r_voff + i * 16 * p.rmap.ld * 2 + jh * 64, 0, 0);
                    }
                    NOMAD_P9_WAIT_VM(0)   // (explicit: nothing is left pending for the compiler's pass to wait for at the K loop's header)
                }
                float bq[2][8];
#pragma unroll
                for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        bq[jh][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((32 * jh + 8 * fq_e + e) * 4, __builtin_bit_cast(int, bias_lane)));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i >= 6 && sh) continue;
                    u32x4 pxy[2];
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
                        float v[8];
#pragma unroll
                        for (int g = 0; g < 2; ++g)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float x = acc[i][2 * jh + g][r] + bq[jh][4 * g + r];
                                if (GELU) x = gelu_bf16out(x);
                                v[4 * g + r] = x;
                            }
                        if (RES) {
                            const bf16x8 rv = __builtin_bit_cast(bf16x8, rres[i][jh]);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                        }
                        bf16x8 ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[e];
                        pxy[jh] = __builtin_bit_cast(u32x4, ov);
                        acc[i][2 * jh] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        acc[i][2 * jh + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                    // whole-line stores (NOMAD_P9_EPI_ROWS): rows 0 .. 7 of the row tile in one instruction, rows 8 .. 15 in the other
                    if (wl) NOMAD_P9_LINE_SWAP(pxy[0], pxy[1])
                    // (row offset in the VGPR offset, never in the scalar offset: the store-data hazard of DESIGN.md 5)
                    if (ABL != 1 || p.M < 0) {
                        __builtin_amdgcn_raw_buffer_store_b128(pxy[0], rc, cl_voff + i * 16 * p.cmap.ld * 2, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b128(pxy[1], rc, cl_voff + i * 16 * p.cmap.ld * 2 + cl_ystep, 0, 2);
                    }
                }
            };
            if (INTER) {   // (only residual GEMMs - and the no-store probe - come here: one copy)
                if (has_r) epi(std::false_type{}, std::true_type{});
                else epi(std::false_type{}, std::false_type{});
            } else {
                if (p.gelu && has_r) epi(std::true_type{}, std::true_type{});
                else if (p.gelu) epi(std::true_type{}, std::false_type{});
                else if (has_r) epi(std::false_type{}, std::true_type{});
                else epi(std::false_type{}, std::false_type{});
            }
        }
        if (ABL == 7 && n_done < 2) ts_[3 + 2 * n_done] = wall_clock64();
        ++n_done;
        if (t_cur + wpx >= t_end) {
            if (!pend) break;
            flush = true;   // one more K tile 0 for its epilogue hooks: its matrix work runs on the K tiles the load cursor staged past
            continue;       // the end (the last tile's own, again) into the zeroed accumulators and is never stored - 1 K tile per launch
        }
        t_cur += wpx;
        {   // the next output tile's coordinates (the load cursor may already be one tile further on)
            const int tm = p.tn_magic ? fast_div(t_cur, p.tn_magic, p.tn_shift) : t_cur;
            m0 = tm * bm_rows;
            n0 = (t_cur - tm * p.tiles_n) * Cfg::BN;
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
    NOMAD_P9_WAIT_VM(0)   // the two K tiles staged past the end: LDS must not be released under them
#undef NOMAD_P9_KTILE
#undef NOMAD_P9_SYNC_COMPUTE
#undef NOMAD_P9_MMA_KH
#undef NOMAD_P9_MMA_ROWS
#undef NOMAD_P9_EPI_HOOK
#undef NOMAD_P9_EPI_HOOK_LATE
#undef NOMAD_P9_EPI_ROWS
#undef NOMAD_P9_LINE_SWAP
#undef NOMAD_P9_DMA_A
#undef NOMAD_P9_DMA_B
#undef NOMAD_P9_ADVANCE
    if (ABL == 7) {
        if (tid == 0 && blockIdx.x < kTimelineSlots) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned long long* o = g_timeline + (size_t)blockIdx.x * 6;
            // entry, first K loop start, K loop end / epilogue end of the first tile, K loop end of the second, tiles done << 32 | HW_ID
            o[0] = ts_[0]; o[1] = ts_[1]; o[2] = ts_[2]; o[3] = ts_[3]; o[4] = ts_[4]; o[5] = ((unsigned long long)n_done << 32) | hw;
        }
    }
}

// one workgroup per CU (160 KB of LDS), never more than there are tiles; the grid is a multiple of 8 (one share per XCD)
template <int ABL = 0, bool INTER = true, int DMAP = 0>
inline hipError_t launch_gemm_bf16_p9(GemmParams p, hipStream_t s, int num_cus) {
    const int bm = p.p9_short ? 192 : P8Cfg::BM;   // (short tiles: DMAP == 0 instantiations only)
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = p.N / 256;
    p.a_clip_magic = p.tn_magic = 0;
    p.a_clip_shift = p.tn_shift = 0;
    if (p.amap.clip_rows < p.M) fast_div_magic((unsigned)p.amap.clip_rows, &p.a_clip_magic, &p.a_clip_shift);
    if (p.tiles_n > 1) fast_div_magic((unsigned)p.tiles_n, &p.tn_magic, &p.tn_shift);
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_bf16_p9_kernel<ABL, INTER, DMAP>), 160 * 1024); e != hipSuccess) return e;
    const long long ntiles = (long long)p.tiles_m * p.tiles_n;
    long long per_xcd = (ntiles + 7) / 8;
    const int cap = num_cus >= 8 ? num_cus / 8 : 1;
    if (per_xcd > cap) per_xcd = cap;
    hipLaunchKernelGGL((gemm_bf16_p9_kernel<ABL, INTER, DMAP>), dim3((unsigned)(8 * per_xcd)), dim3(P8Cfg::THREADS), 160 * 1024, s, p);
    return hipGetLastError();
}

}  // namespace nomad
