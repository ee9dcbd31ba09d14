// fp32 MFMA GEMM / implicit-GEMM for gfx950 (CDNA4):  C = epilogue(A * W^T)
//
// One kernel template serves every dense contraction of the wav2vec 2.0 BASE forward
// (K3/K4 strided Conv1d, K6 post_extract_proj, K7 grouped pos-conv,
// K9 fused QKV, K11 out_proj, K12 FFN).  Activations are kept time-major ([clip][frame][channel]),
// so a k=3/stride=2 Conv1d output row is a dot product over 3*512 CONTIGUOUS floats starting at
// input frame 2t: the convolution is a plain GEMM whose A-row stride (1024) is smaller than K
// (1536).  The only extra machinery is the RowMap (rows restart per clip) and a chunked K map
// (pos-conv: 128 taps x 48 channels, taps 768 floats apart).
//
// Matrix core: v_mfma_f32_16x16x4_f32 in every production instantiation since the end of round 4 (exact fp32, 64 FLOP/clk/SIMD = the
// fp32 peak; half the accumulator-register traffic per multiply-add of v_mfma_f32_32x32x2_f32, under which the chip held a lower
// clock - DESIGN.md 4a "MFMA shape and the clock").  Operand maps of that shape (cdna_hip_programming.md section 3): lane l supplies
// A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D[i][j] lands at column j = l & 15, rows i = 4 (l >> 4) + reg.  A 32 x 32
// accumulator block is 2 x 2 such sub-blocks.  The contraction index may be visited in any order, so one ds_read_b128 per lane feeds
// four consecutive k-steps: lane (fi, g) reads logical chunk 4 kh + g of its row and element c of it feeds k-step c, which contracts
// k = 16 kh + c + {0, 4, 8, 12} (gemm_f32_glds_body, M16).  The legacy register-staged kernel gemm_f32_kernel (diag tiles 0-19) and the
// persistent experiment gemm_f32_pers_kernel are still on 32x32x2: they are NOT bit-identical to the production tiles any more.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dtypes.hip.h"

namespace nomad {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Address of logical row m: rows are grouped in clips of `clip_rows` rows.  Ragged batches (clips of different
// lengths packed back to back) set `pref`/`base` instead: clip c owns logical rows pref[c] .. pref[c+1]-1 and
// its row 0 sits at element off + base[c] * unit.
struct RowMap {
    long long off;          // element offset of (clip 0, row 0)
    long long clip_stride;  // elements between clips (uniform batches)
    int clip_rows;          // rows per clip in the logical M index (uniform batches); 0 for ragged maps
    int ld;                 // elements between consecutive rows of one clip
    const int* pref;        // ragged: nclips + 1 logical-row prefix sums (device), or nullptr
    const int* base;        // ragged: nclips + 1 per-clip base rows in the addressed buffer (device)
    int nclips;
    int unit;               // elements per base row
};

__device__ __forceinline__ long long row_addr(const RowMap& r, int m) {
    if (r.pref) {
        int lo = 0, hi = r.nclips;  // largest c with pref[c] <= m
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (r.pref[mid] <= m) lo = mid;
            else hi = mid;
        }
        return r.off + (long long)r.base[lo] * r.unit + (long long)(m - r.pref[lo]) * r.ld;
    }
    const int c = m / r.clip_rows;
    const int t = m - c * r.clip_rows;
    return r.off + (long long)c * r.clip_stride + (long long)t * r.ld;
}

struct GemmParams {
    const float* A;
    RowMap amap;
    int kchunk, kstride;  // logical k -> (k / kchunk) * kstride + k % kchunk   (BK divides kchunk)
    const float* W;       // [N_padded][ldw], k contiguous
    int ldw;
    float* C;
    RowMap cmap;
    const float* bias;    // [N] or nullptr
    const float* R;       // residual or nullptr, added AFTER the activation
    RowMap rmap;
    int M, N, K;          // N = padded N covered by the grid (multiple of BN)
    int n_valid;          // columns >= n_valid are not stored
    int gelu;
    // per-group (blockIdx.y) element offsets: grouped pos-conv
    long long a_goff, w_goff, c_goff, r_goff;
    int bias_goff;
    int tiles_m, tiles_n;
    int group_m;          // >0: walk tiles down M in groups of group_m row-tiles (L2 reuse of W), 0: n fastest
    // Optional column-block split of C: column n goes to block n / c_colblk (c_colblk_stride elements apart)
    // at column n % c_colblk.  post_extract_proj uses it to write the pos-conv input group-major.
    int c_colblk;
    long long c_colblk_stride;
    // Training support.  Upre (nullable): the pre-activation acc + bias is ALSO stored there, at the same
    // index as C (saved for the backward's GELU').  DG (nullable): the result is multiplied by
    // gelu'(DG[dgmap(m) + n]) after the optional GELU and before the residual: backward GEMMs whose output
    // feeds a GELU's input gradient.
    float* Upre;
    const float* DG;
    RowMap dgmap;
    long long dg_goff;
    // bf16x3 path (gemm_bf16_8phase.hip.h, X3): element distance from the hi to the lo plane of A, W, C and R
    long long a_plane, w_plane, c_plane, r_plane;
    // Column blocks that are FRAMES (bf16x3 pos-conv, p8_epilogue): with c_blk_step > 0 a row is a block of
    // c_blk_step consecutive frames of its clip and column block b (of c_colblk columns) is frame
    // local_row * c_blk_step + b, stored only while that is < the clip's frame count (c_clip_frames for uniform
    // batches, cmap.base[c + 1] - cmap.base[c] for ragged ones).
    int c_blk_step, c_clip_frames;
    // persistent kernel (gemm_f32_pers_kernel): exact division by amap.clip_rows and by tiles_n as mulhi + shift
    // (fast_div_magic; magic 0 = quotient 0 / divisor 1)
    unsigned a_clip_magic, tn_magic;
    int a_clip_shift, tn_shift;
    long long a_soff, w_soff, c_soff;   // gemm_f32_n48_kernel, K split over blockIdx.z (loss path only): element offsets per slice of A, W and C
    int tile_m_base;        // gemm_f32_mixed_kernel: the first row tile of this part of the problem (in units of its BM); 0 otherwise
    int p9_skew;            // gemm_bf16_p9_kernel, timeline probe only (ABL = 7): workgroup group (blockIdx.x >> 3) & 3 starts p9_skew x 10 ns x group late
    int p9_wl;              // gemm_bf16_p9_kernel: 1 = output stores of 8 rows x 128 bytes per instruction (lanes fr / fr + 8 swap a chunk), 0 = 16 rows x 64 bytes
    int p9_late;            // gemm_bf16_p9_kernel: 1 = wave row 0 runs its GELU epilogue hooks behind the phase's barrier, beside wave row 1's (round 6)
    int p9_short;           // gemm_bf16_p9_kernel: 1 = 192-row tiles (96 rows per wave row) instead of 256-row ones, same instantiation (run-time)
};

// q = n / d for 0 <= n < 2^31 as (n * magic) >> (32 + shift): magic = ceil(2^(31 + l) / d), l = ceil(log2 d), shift = l - 1
// (Granlund & Montgomery 1994, theorem 4.2 for 31-bit dividends: the magic number fits 32 bits).  d >= 2.
inline void fast_div_magic(unsigned d, unsigned* magic, int* shift) {
    int l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned long long num = 1ull << (31 + l);
    *magic = (unsigned)((num + d - 1) / d);
    *shift = l - 1;
}
__device__ __forceinline__ int fast_div(int n, unsigned magic, int shift) { return (int)(__umulhi((unsigned)n, magic) >> shift); }

// (local row, frames of its clip) of logical row m under map r - see GemmParams::c_blk_step
__device__ __forceinline__ void clip_pos(const RowMap& r, int m, int uniform_frames, int& local, int& frames) {
    if (r.pref) {
        int lo = 0, hi = r.nclips;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (r.pref[mid] <= m) lo = mid;
            else hi = mid;
        }
        local = m - r.pref[lo];
        frames = r.base[lo + 1] - r.base[lo];
    } else {
        local = m % r.clip_rows;
        frames = uniform_frames;
    }
}

__device__ __forceinline__ float dgelu_erf_(float u) {
    const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
    return cdf + u * 0.39894228040143267794f * expf(-0.5f * u * u);
}

// Exact (erf) GELU, gelu(x) = x * Phi(x), with erf from Abramowitz & Stegun 7.1.26
//   erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z >= 0,   |error| <= 1.5e-7,
// evaluated as  0.5 (x + |x|) - 0.5 |x| poly(t) exp(-x^2 / 2)  so that no "1 + erf" cancellation occurs.
// In fp32 its maximum absolute error against a float64 GELU is 3.3e-7 over [-12, 12] - below the 4.5e-7 of the
// textbook 0.5 x (1 + erff(x / sqrt 2)) evaluated in fp32 - at 13 VALU instructions instead of 38
// (v_rcp_f32 + v_exp_f32 + 9 FMA/MUL).  This function runs once per activation in seven conv layers, the
// pos-conv and every fc1 epilogue: 20 M times per 4 s clip.
// Round 6 (late) built the alternative the bf16 path now uses (gelu_bf16out below) at fp32 accuracy: max(x, 0) - |x| 2^q(min(|x|, 13)) with q a
// SEXTIC fitted to log2 of the Gaussian tail 1 - Phi - 2.8e-7 (the rounding of the result itself) at 10 instructions with one transcendental
// instead of 13 with two.  Alternating libraries on one box: fp32 headline 2391.1 -> 2393.7 clips/s, bf16x3 6314.8 -> 6347.8 (+0.1 % / +0.5 %):
// these paths' epilogues are not bound by vector-instruction issue the way the bf16 GEMM's are (16x / 3x the matrix time per activation), so the
// fp32 path keeps the bits it has had since round 1; the sextic stays buildable (form 2: `python -m nomad_amd.build --variant gelu_tail6`).
#ifndef NOMAD_GELU_F32_FORM
#define NOMAD_GELU_F32_FORM 1
#endif
#if NOMAD_GELU_F32_FORM == 1
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, ax, 1.0f));  // p / sqrt(2), p = 0.3275911
    // the polynomial carries the factor 0.5 (halved coefficients: exact scalings, the same bits as 0.5 * poly) and the positive
    // part is max(x, 0) == 0.5 (x + |x|) exactly: two instructions fewer per activation, results unchanged bit for bit
    float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    poly = fmaf(t, poly, 0.5f * 1.421413741f);
    poly = fmaf(t, poly, 0.5f * -0.284496736f);
    poly = fmaf(t, poly, 0.5f * 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-0.72134752044f * x * x);  // exp(-x^2/2) = 2^(-x^2 log2(e) / 2)
    return fmaf(-ax, poly * e, fmaxf(x, 0.0f));
}
#else
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fabsf(x);
    const float ac = fminf(a, 13.0f);
    float q = fmaf(ac, 3.30926559e-5f, -7.6921843e-4f);
    q = fmaf(q, ac, 8.08071252e-3f);
    q = fmaf(q, ac, -5.34120984e-2f);
    q = fmaf(q, ac, -4.5877099e-1f);
    q = fmaf(q, ac, -1.15120173f);
    q = fmaf(q, ac, -9.99993086e-1f);
    return fmaf(-a, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.0f));
}
#endif

// GELU for epilogues whose OUTPUT is bf16 (the bf16 path's GEMMs, conv0 and pos-conv, config C5; never the fp32 or bf16x3 paths).
// Round 6: gelu(x) = max(x, 0) - |x| T(|x|) with the tail T(a) = 1 - Phi(a) = 0.5 erfc(a / sqrt 2) evaluated as 2^q(a), q a polynomial fitted to
// log2 T (minimax on the error of the RESULT, a (2^q(a) - T(a)), Lawson iteration in float64 over [0, 6]; beyond, a T(a) < 6e-9).  The erf
// form's structure without its division: ONE transcendental.  Maximum absolute error against the exact erf GELU, evaluated in fp32:
//   cubic   5.5e-5  6 instructions (3 v_fma, v_exp, v_max, v_fma)          28 issue cycles (4 per instruction, 8 per transcendental)   <- shipped
//   quartic 6.2e-6  8 (+ v_min: the quartic turns upward at 13.5)         36
//   round 5's x sigmoid(g(x)): 2.55e-5, 9 instructions with v_exp AND v_rcp  44;   gelu_erf: 3.3e-7, 13 with two transcendentals, 60.
// The epilogues are bound by vector-instruction issue: configs[4] 1897 -> 1911 -> 1932 clips/s for sigmoid -> quartic -> cubic on one box, and
// the bf16 path's distance from the fp32 path does not move (embedding rms 2.95e-4, max 1.0e-3 with all three: tools/bf16_accuracy.py,
// profiles/r06_gelu_bf16_forms.txt) - 5.5e-5 is a 70th of the bf16 rounding step of an activation of size 1, and a tenth of the error of the
// tanh form that passes for GELU elsewhere (4.7e-4).  Non-finite activations (NaN, +inf, -inf) give NaN (inf * 0: an overflow stays visible; torch's fp32
// GELU does the same for -inf, and on the CPU for +inf); a hugely negative finite one gives the exact -0.
// Every bf16 kernel uses THIS function (their results stay bit-identical to each other); every fp32 / bf16x3 epilogue keeps gelu_erf.
#ifndef NOMAD_GELU_BF16_FORM
#define NOMAD_GELU_BF16_FORM 3
#endif
#if NOMAD_GELU_BF16_FORM == 1
// The first form (rounds 5-6), kept for A/B builds (python -m nomad_amd.build --variant gelu1, NOMAD_LIB_VARIANT=gelu1): x * sigmoid(g(x)),
// g(x) = x (c0 + c1 x^2 + c2 x^4) fitted to logit(Phi(x)), x^2 clamped at 50; v_mul_legacy_f32 so that -inf gives 0, not -inf * 0 = NaN.
extern "C" __device__ float nomad_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");   // this clang has no __builtin_amdgcn_fmul_legacy
__device__ __forceinline__ float gelu_bf16out(float x) {
    const float x2 = fminf(x * x, 50.0f);
    float p = fmaf(x2, 1.0153758e-3f, -1.0678258e-1f);     // -log2(e) * (c2 x^2 + c1)
    p = fmaf(p, x2, -2.3011138f);                            // -log2(e) * c0
    const float e = __builtin_amdgcn_exp2f(p * x);           // 2^(-g(x) log2 e) = exp(-g(x))
    return nomad_fmul_legacy(x, __builtin_amdgcn_rcpf(1.0f + e));
}
#elif NOMAD_GELU_BF16_FORM == 2
// The quartic tail (variant gelu2): 6.2e-6, 8 instructions.
__device__ __forceinline__ float gelu_bf16out(float x) {
    const float a = fabsf(x);
    const float ac = fminf(a, 13.0f);
    float q = fmaf(ac, 3.86565109e-3f, -4.40765619e-2f);
    q = fmaf(q, ac, -4.68018711e-1f);
    q = fmaf(q, ac, -1.14737022f);
    q = fmaf(q, ac, -1.00047994f);
    return fmaf(-a, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.0f));
}
#else
// The cubic tail (shipped): q falls monotonically for every a >= 0 (all three derivative coefficients negative), so nothing is clamped.
__device__ __forceinline__ float gelu_bf16out(float x) {
    const float a = fabsf(x);
    float q = fmaf(a, -2.48856321e-2f, -4.98820007e-1f);
    q = fmaf(q, a, -1.129246f);
    q = fmaf(q, a, -1.00353169f);
    return fmaf(-a, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.0f));
}
#endif

// XCD-aware bijective remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch),
// so give each XCD a contiguous run of tiles; consecutive tiles share the A row panel (n fastest).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// Linear tile id -> (tile_m, tile_n).  group_m > 0: groups of group_m row-tiles are walked m-fastest, so
// the ~64 workgroups that are resident on one XCD together cover a group_m x (64/group_m) block of tiles
// and each A / W panel fetched into that XCD's L2 is shared by several of them.
__device__ __forceinline__ void tile_coords(int wg, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    if (group_m <= 0) {
        tm = wg / tiles_n;
        tn = wg - tm * tiles_n;
        return;
    }
    const int per_group = group_m * tiles_n;
    const int g = wg / per_group, in = wg - g * per_group;
    const int rows = min(group_m, tiles_m - g * group_m);
    tn = in / rows;
    tm = g * group_m + (in - tn * rows);
}

template <int BM, int BN, int BK, int WM, int WN>
struct GemmCfg {
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int LD = BK + 4;  // +16 B row pad: conflict-free ds_read_b128 (144 B / 80 B rows)
    static constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    static constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
    static constexpr int A_CHUNKS = BM * BK / 4 / THREADS, B_CHUNKS = BN * BK / 4 / THREADS;
    static constexpr int LDS_BYTES = 2 * (BM + BN) * LD * 4;
};

// WM x WN waves per workgroup, each owning a (BM/WM) x (BN/WN) output tile.
// ABL != 0 are timing-only ablations (wrong results): 1 = no global loads / LDS stores in the K loop,
// 2 = additionally no barrier, 3 = loads and stores kept but no barrier.
template <int BM, int BN, int BK, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_f32_kernel(const GemmParams p) {
    using Cfg = GemmCfg<BM, BN, BK, WM, WN>;
    constexpr int LD = Cfg::LD, TM = Cfg::TM, TN = Cfg::TN, NT = Cfg::THREADS;
    constexpr int KC = BK / 4;  // float4 chunks per tile row
    static_assert(Cfg::A_CHUNKS >= 1 && Cfg::B_CHUNKS >= 1 && TM >= 1 && TN >= 1, "bad tile");
    static_assert(BM * BK / 4 % NT == 0 && BN * BK / 4 % NT == 0, "staging must divide evenly");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;   // [2][BN][LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_coords(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int grp = blockIdx.y;

    const float* Ag = p.A + grp * p.a_goff;
    const float* Wg = p.W + grp * p.w_goff;

    // per-thread staging sources (rows fixed for the whole K loop)
    const float* a_src[Cfg::A_CHUNKS];
    const float* b_src[Cfg::B_CHUNKS];
    int a_dst[Cfg::A_CHUNKS], b_dst[Cfg::B_CHUNKS];
#pragma unroll
    for (int i = 0; i < Cfg::A_CHUNKS; ++i) {
        const int id = tid + i * NT, row = id / KC, kc = id - row * KC;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        a_src[i] = Ag + row_addr(p.amap, m) + kc * 4;
        a_dst[i] = row * LD + kc * 4;
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        const int id = tid + i * NT, row = id / KC, kc = id - row * KC;
        b_src[i] = Wg + (long long)(n0 + row) * p.ldw + kc * 4;
        b_dst[i] = row * LD + kc * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // register staging uses native vector types: HIP's float4 class kept these arrays in scratch
    f32x4 a_reg[Cfg::A_CHUNKS], b_reg[Cfg::B_CHUNKS];
    const int nk = p.K / BK;

#define NOMAD_LOAD_TILE(KT)                                                                          \
    {                                                                                                \
        const int k0_ = (KT)*BK;                                                                     \
        const int kq_ = k0_ / p.kchunk;                                                              \
        const long long a_koff_ = (long long)kq_ * p.kstride + (k0_ - kq_ * p.kchunk);               \
        _Pragma("unroll") for (int i = 0; i < Cfg::A_CHUNKS; ++i) a_reg[i] =                         \
            *reinterpret_cast<const f32x4*>(a_src[i] + a_koff_);                                     \
        _Pragma("unroll") for (int i = 0; i < Cfg::B_CHUNKS; ++i) b_reg[i] =                         \
            *reinterpret_cast<const f32x4*>(b_src[i] + k0_);                                         \
    }
#define NOMAD_STORE_TILE(BUF)                                                                        \
    {                                                                                                \
        float* as_ = As + (BUF)*BM * LD;                                                             \
        float* bs_ = Bs + (BUF)*BN * LD;                                                             \
        _Pragma("unroll") for (int i = 0; i < Cfg::A_CHUNKS; ++i)                                    \
            *reinterpret_cast<f32x4*>(as_ + a_dst[i]) = a_reg[i];                                    \
        _Pragma("unroll") for (int i = 0; i < Cfg::B_CHUNKS; ++i)                                    \
            *reinterpret_cast<f32x4*>(bs_ + b_dst[i]) = b_reg[i];                                    \
    }

    NOMAD_LOAD_TILE(0)
    NOMAD_STORE_TILE(0)
    if (ABL != 0) NOMAD_STORE_TILE(1)
    __syncthreads();

    const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
    const int a_frag_off = (wm * Cfg::WTM + frag_row) * LD + frag_k;
    const int b_frag_off = (wn * Cfg::WTN + frag_row) * LD + frag_k;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk && ABL != 1 && ABL != 2) NOMAD_LOAD_TILE(kt + 1)  // global loads in flight under the MFMAs
        const float* as = As + cur * BM * LD + a_frag_off;
        const float* bs = Bs + cur * BN * LD + b_frag_off;
#pragma unroll
        for (int kq = 0; kq < BK / 8; ++kq) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LD + kq * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LD + kq * 8);
            // consecutive MFMAs go to different accumulators (no back-to-back dependent issue)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][c], bf[j][c], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk && ABL != 1 && ABL != 2) NOMAD_STORE_TILE(cur ^ 1)
        if (ABL < 2) __syncthreads();
    }

#undef NOMAD_LOAD_TILE
#undef NOMAD_STORE_TILE
    // epilogue: lane owns column n, 16 rows per 32x32 tile.  Plain (single-clip) maps avoid the
    // per-row integer division of row_addr().
    float* Cg = p.C + grp * p.c_goff;
    const float* Rg = p.R ? p.R + grp * p.r_goff : nullptr;
    const float* biasg = p.bias ? p.bias + grp * p.bias_goff : nullptr;
    const bool c_plain = p.cmap.clip_rows >= p.M, r_plain = p.rmap.clip_rows >= p.M;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * Cfg::WTN + j * 32 + (lane & 31);
        const bool n_ok = n < p.n_valid;
        const float bv = (biasg && n_ok) ? biasg[n] : 0.f;
        long long c_col = n;
        if (p.c_colblk > 0) {
            const int blk = n / p.c_colblk;
            c_col = (long long)blk * p.c_colblk_stride + (n - blk * p.c_colblk);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * Cfg::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n_ok) {
                    float v = acc[i][j][r] + bv;
                    const long long c_idx = (c_plain ? p.cmap.off + (long long)m * p.cmap.ld : row_addr(p.cmap, m)) + c_col;
                    if (p.Upre) p.Upre[grp * p.c_goff + c_idx] = v;
                    if (p.gelu) v = gelu_erf(v);
                    if (p.DG) v *= dgelu_erf_(p.DG[grp * p.dg_goff + row_addr(p.dgmap, m) + n]);
                    if (Rg) v += Rg[(r_plain ? p.rmap.off + (long long)m * p.rmap.ld : row_addr(p.rmap, m)) + n];
                    Cg[c_idx] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant: tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write).
// One wave-instruction fills 1 KiB of LDS linearly (wave-uniform base + lane*16 B), so the LDS image is
// unpadded [rows][BK]; the bank-conflict fix is an XOR swizzle applied on the per-lane SOURCE address and
// again on the fragment read (cdna_hip_programming.md rule 21): 16-B chunk c of row r lives at chunk
// c ^ ((r / RB) % KC), RB = rows per 256-B bank row.  A 16-lane ds_read_b128 group touches 16 rows that
// are distinct mod 16 at one logical chunk, which that map spreads over all 16 slots of the bank row.
template <int BM, int BN, int BK, int WM, int WN, int STAGES = 2>
struct GldsCfg {
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int KC = BK / 4, RB = 16 / KC;
    static constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static constexpr int A_CHUNKS = BM * KC / THREADS, B_CHUNKS = BN * KC / THREADS;
    static constexpr int STAGE_BYTES = STAGES * (BM + BN) * BK * 4;
    static constexpr int EPI_BYTES = WM * WN * 32 * (WTN + 4) * 4;  // one 32-row fp32 slab per wave
    static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// A pointer every lane of the wave holds the same value of, moved to SGPRs.  The tile base addresses below are uniform by
// construction but come out of 64-bit VALU arithmetic (row_addr's ragged branch loads through the vector memory path), and a
// buffer descriptor built from VGPRs makes the compiler wrap EVERY buffer_load ... lds in a waterfall loop (4 v_readfirstlane,
// 2 v_cmp_eq_u64, s_and_saveexec, a branch) - found in round 4 in the K loop of every fp32 GEMM, on the A operand.
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// 16 bytes per lane from `base + voff + soff` (bytes) straight into LDS through a raw buffer descriptor
// (buffer_load_dwordx4 ... offen lds): the base is wave-uniform (SGPRs), the per-lane part is one 32-bit VGPR.
__device__ __forceinline__ void dma16_buffer(const float* base, lptr_t dst, int voff, int soff) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

// X3 variant of the kernel below: hi = bf16(x), lo = bf16(x - hi) of 8 consecutive k values held by one lane - the operand
// fragments of v_mfma_f32_32x32x16_bf16, made in registers from the fp32 LDS image (nothing is stored split).
// The low halves x - hi come from v_dot2c_f32_bf16 (acc += a.lo * b.lo + a.hi * b.hi on packed bf16 pairs) with the constant
// pairs (-1, 0) / (0, -1) and x as the accumulator: one instruction per value instead of unpack + subtract (the difference is
// exactly representable, so the dot unit's internal rounding cannot matter).  16 VALU instructions per 8 values instead of 24:
// the kernel is VALU-bound on this split at one 32 x 32 accumulator per wave.
// The two constant pairs live in VGPRs filled from 32-bit literals (split_consts): written as compile-time constants the
// compiler encodes (-1, 0) as the INLINE constant -1.0, which the instruction reads as an fp16 -1.0 (0xBC00) in a bf16 operand
// - measured: embeddings off by 0.25.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
struct SplitConsts {
    bf16x2_t m0, m1;   // (-1, 0) and (0, -1)
};
__device__ __forceinline__ SplitConsts split_consts() {
    unsigned a, b;
    asm volatile("v_mov_b32 %0, 0x0000bf80\n\tv_mov_b32 %1, 0xbf800000" : "=v"(a), "=v"(b));
    SplitConsts c;
    c.m0 = __builtin_bit_cast(bf16x2_t, a);
    c.m1 = __builtin_bit_cast(bf16x2_t, b);
    return c;
}
__device__ __forceinline__ void split_frag8(const f32x4& a, const f32x4& b, const SplitConsts& k, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        bf16x2_t ha, hb;
        ha[0] = (bf16_t)a[e]; ha[1] = (bf16_t)a[e + 1];
        hb[0] = (bf16_t)b[e]; hb[1] = (bf16_t)b[e + 1];
        hi[e] = ha[0]; hi[e + 1] = ha[1];
        hi[4 + e] = hb[0]; hi[4 + e + 1] = hb[1];
        lo[e] = (bf16_t)__builtin_amdgcn_fdot2_f32_bf16(ha, k.m0, a[e], false);
        lo[e + 1] = (bf16_t)__builtin_amdgcn_fdot2_f32_bf16(ha, k.m1, a[e + 1], false);
        lo[4 + e] = (bf16_t)__builtin_amdgcn_fdot2_f32_bf16(hb, k.m0, b[e], false);
        lo[4 + e + 1] = (bf16_t)__builtin_amdgcn_fdot2_f32_bf16(hb, k.m1, b[e + 1], false);
    }
}

// STAGES = 2: double-buffered LDS, __syncthreads() (vmcnt(0) + barrier) once per K tile.
// STAGES = 3: the DMA of tile kt+2 is issued while tile kt is multiplied; the wait before the barrier is a
//             COUNTED vmcnt that leaves tile kt+1's loads in flight, and the barrier is the raw s_barrier
//             (cdna_hip_programming.md "Pipelining across barriers").
// NOEPI: timing-only ablation (no epilogue stores).
// OPT (experiments, results unchanged): bit 0 = epilogue slabs fenced per wave (lgkmcnt) instead of per workgroup
//      (__syncthreads also waits for the previous slab's global stores); bit 1 = s_setprio(1) around the MFMA cluster.
// X3 ("bf16x3 products on fp32 buffers"): same operands, staging, LDS image and epilogue, but every fp32 product a*w is
//      formed as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate; the dropped a_lo*w_lo is
//      2^-16 relative), the hi / lo halves made in registers from the fp32 fragments.  Per 32-deep K tile a 32 x 32
//      accumulator costs 6 MFMAs of 32 cycles instead of 16 of 64: the K loop of the small-M problems (config C4:
//      M = 1600 rows, one round of 64 x 64 tiles, where one wave's serial K loop IS the launch time) is ~5x shorter.
//      Every X3 instantiation contracts k in the same order, so the tile choice changes no result bit, as in fp32.
// (the kernel proper is gemm_f32_glds_kernel below; the body is a device function so that gemm_f32_mixed_kernel can run two tile
// shapes in one launch.  blk_x / grid_x / grp: the workgroup's index among, and the number of, workgroups that walk THIS problem's
// tiles, and the group index - the x block index, the x grid size and the y block index in the plain kernel.)
template <int BM, int BN, int BK, int WM, int WN, int STAGES = 2, bool NOEPI = false, int OPT = 0, bool X3 = false>
__device__ __forceinline__ void gemm_f32_glds_body(const GemmParams& p, const int blk_x, const int grid_x, const int grp) {
    using Cfg = GldsCfg<BM, BN, BK, WM, WN, STAGES>;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, NT = Cfg::THREADS, KC = Cfg::KC, RB = Cfg::RB;
    // OPT bit 1024 (round 4): TRANSPOSED accumulators + direct epilogue.  The two operands of the MFMA have the same lane layout, so
    // passing W as the first and A as the second operand leaves D^T in the same registers: a lane then owns ONE output row per
    // 16-row sub-block (lane & 15) and four CONSECUTIVE columns of it (4 (lane >> 4) .. + 3) - 16-byte stores straight from the
    // accumulators, no LDS slab, no barrier between the K loop and the stores, no per-store address arithmetic (buffer descriptor:
    // per-lane offset + immediate).  Products and their order per output element are unchanged (a*w == w*a): bit-identical
    // results.  Plain C matrices only (with bit 16), no residual.
    constexpr bool TR = (OPT & 1024) != 0;
    static_assert(!TR || ((OPT & 16) && !X3 && !(OPT & 32)), "transposed accumulators: plain scoring epilogue only");
    // M16 (round 4, every fp32-product instantiation): the products on v_mfma_f32_16x16x4_f32 instead of v_mfma_f32_32x32x2_f32 - the
    // same 64 FLOP / clk / SIMD with HALF the accumulator register traffic per multiply-add (4 + 4 registers per 1024 MACs against
    // 16 + 16 per 2048).  Why: on some boxes the chip holds 2313 MHz under the 32x32x2 kernel and 2392 MHz under hipBLASLt's
    // MI16x16x1 kernel (tools/clock_under_load.py), and tools/micro/mfma_shape_power.hip shows the 32x32x2 stream's clock sagging as
    // LDS reads and LDS-DMA are added while the 16x16x4 stream's does not.  A 32 x 32 accumulator block is 2 x 2 sub-blocks q = 2 si + sj
    // (registers 4q .. 4q+3); a lane (fi = lane & 15, g = lane >> 4) reads one ds_read_b128 per 16-row operand sub-tile and 16 k -
    // logical chunk 4 kh + g - whose element c feeds MFMA c: k-step c contracts k = 16 kh + {c, 4 + c, 8 + c, 12 + c}.  Per output
    // element the order is kt, kh, c ascending in every instantiation (BK = 16 or 32): the tile choice changes no bit, as before.
    constexpr bool M16 = !X3;
    static_assert(!M16 || BK % 16 == 0, "16x16x4 products: 16-deep k groups");
    static_assert(BK == 8 || BK == 16 || BK == 32, "swizzle is written for 32-, 64- and 128-B rows");
    static_assert(BM * KC % NT == 0 && BN * KC % NT == 0 && TM >= 1 && TN >= 1, "bad tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [STAGES][BM][BK]
    float* Bs = smem + STAGES * BM * BK;   // [STAGES][BN][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int nwg = p.tiles_m * p.tiles_n;
#ifdef NOMAD_DIAG
    // OPT bit 128 (timing probe, tools/gemm_timeline_f32.py): wave 0 stamps {entry, first barrier passed, K loop done, epilogue stores
    // issued, stores acknowledged} in 100 MHz wall-clock ticks + HW_ID | XCC_ID << 32 into g_timeline[6 * blockIdx.x]
    unsigned long long ts_[5] = {0, 0, 0, 0, 0};
    if (OPT & 128) ts_[0] = wall_clock64();
#endif
    // Persistent launch (gridDim.x < nwg): a workgroup walks tiles blockIdx.x, + gridDim.x, ...; its output stores
    // drain under the next tile's prologue instead of holding the wave slots until they are acknowledged.
    for (int t_ = blk_x; t_ < nwg; t_ += grid_x) {
    // OPT bit 4096 (round 4): the address set-up and the epilogue run at raised wave priority.  The timeline probe shows the
    // set-up of a workgroup that starts next to a peer in its K loop taking 13 us instead of 2 (its scalar / vector instructions
    // queue behind the peer's MFMA stream), and the lone peer fills only ~3/4 of the matrix pipe meanwhile.
    if (OPT & 4096) __builtin_amdgcn_s_setprio(3);
    // OPT bit 16384 (round 4): LEAN set-up.  Vector instructions of a freshly launched workgroup are starved by the older
    // peer workgroup's MFMA stream (tools/micro/issue_starve.hip: a dependent v_add chain next to two MFMA-streaming waves on
    // its SIMD makes NO progress until they stop - issue is oldest-wave-first, s_setprio does not change it - while scalar
    // ALU and scalar loads run at full speed), and the general set-up is ~250 vector instructions, most of them the
    // float-reciprocal sequences of four integer divisions.  Lean: every division is mulhi + shift with magic numbers from the
    // host (fast_div), the tile coordinates and bases stay on the scalar unit, ~60 vector instructions remain.  Uniform clip
    // maps only (no ragged prefix sums), tiles walked n-fastest (group_m == 0); the caller checks and fills the magic numbers.
    constexpr bool LEAN = (OPT & 16384) != 0;
    static_assert(!LEAN || (OPT & 4), "lean set-up: buffer-descriptor DMA only");
    const int wg = xcd_remap(t_, nwg);
    int tile_m, tile_n;
    if (LEAN) {
        tile_m = p.tn_magic ? fast_div(wg, p.tn_magic, p.tn_shift) : wg;
        tile_n = wg - tile_m * p.tiles_n;
    } else {
        tile_coords(wg, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);
    }
    tile_m += p.tile_m_base;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    auto row_addr_lean = [&](int m) -> long long {
        const int c = p.a_clip_magic ? fast_div(m, p.a_clip_magic, p.a_clip_shift) : 0;
        return p.amap.off + (long long)c * p.amap.clip_stride + (long long)(m - c * p.amap.clip_rows) * p.amap.ld;
    };
    const float* Ag = p.A + grp * p.a_goff;
    const float* Wg = p.W + grp * p.w_goff;
#ifdef NOMAD_DIAG
    if ((OPT & 128) && (OPT & 8192)) {   // set-up detail: kernel arguments read, tile coordinates known
        asm volatile("" :: "s"(m0), "s"(n0));
        ts_[1] = wall_clock64();
    }
#endif

    // Per-lane DMA sources; LDS chunk id = tid + i*NT is linear in the lane within each wave-instruction.
    // OPT & 4: buffer_load ... lds - an SGPR resource descriptor per operand (base = this tile's first row, so the
    // 32-bit per-lane offsets stay small whatever the tensor size), the K-tile offset as the scalar offset; otherwise
    // global_load_lds with 64-bit per-lane pointers.
    const float* a_src[Cfg::A_CHUNKS];
    const float* b_src[Cfg::B_CHUNKS];
    int a_voff[Cfg::A_CHUNKS], b_voff[Cfg::B_CHUNKS];
    const long long tile_row0 = LEAN ? row_addr_lean(m0 < p.M ? m0 : p.M - 1) : (OPT & 4) ? row_addr(p.amap, m0 < p.M ? m0 : p.M - 1) : 0;
#pragma unroll
    for (int i = 0; i < Cfg::A_CHUNKS; ++i) {
        const int id = tid + i * NT, row = id / KC, pc = id - row * KC;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        if (LEAN) {
            a_src[i] = nullptr;
            a_voff[i] = (int)((row_addr_lean(m) - tile_row0 + ((pc ^ ((row / RB) % KC)) * 4)) * 4);
        } else {
            a_src[i] = Ag + row_addr(p.amap, m) + ((pc ^ ((row / RB) % KC)) * 4);
            a_voff[i] = (int)((row_addr(p.amap, m) - tile_row0 + ((pc ^ ((row / RB) % KC)) * 4)) * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_CHUNKS; ++i) {
        const int id = tid + i * NT, row = id / KC, pc = id - row * KC;
        b_src[i] = Wg + (long long)(n0 + row) * p.ldw + ((pc ^ ((row / RB) % KC)) * 4);
        b_voff[i] = (int)(((long long)row * p.ldw + ((pc ^ ((row / RB) % KC)) * 4)) * 4);
    }
    const float* const a_tile = (OPT & 4) ? uniform_ptr(Ag + tile_row0) : Ag + tile_row0;
    const float* const b_tile = (OPT & 4) ? uniform_ptr(Wg + (long long)n0 * p.ldw) : Wg + (long long)n0 * p.ldw;
#ifdef NOMAD_DIAG
    if ((OPT & 128) && (OPT & 8192)) {   // set-up detail: per-lane offsets and tile bases done
        asm volatile("" :: "v"(a_voff[0]), "v"(b_voff[0]), "s"(a_tile), "s"(b_tile));
        ts_[2] = wall_clock64();
    }
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    // K tiles are staged strictly in order, so the chunked-K map (logical k -> (k / kchunk) * kstride + k % kchunk) is walked
    // with three running scalars instead of an integer division per tile (KT only documents which tile a call stages)
    int ld_k0 = 0, ld_in = 0;
    long long ld_chunk = 0;
#define NOMAD_GLDS_TILE(KT, BUF)                                                                         \
    {                                                                                                    \
        const int k0_ = ld_k0;                                                                           \
        const long long a_koff_ = ld_chunk + ld_in;                                                      \
        ld_k0 += BK;                                                                                     \
        ld_in += BK;                                                                                     \
        if (ld_in >= p.kchunk) {                                                                         \
            ld_in = 0;                                                                                   \
            ld_chunk += p.kstride;                                                                       \
        }                                                                                                \
        float* as_ = As + (BUF)*BM * BK + wave * 256;                                                    \
        float* bs_ = Bs + (BUF)*BN * BK + wave * 256;                                                    \
        if (OPT & 4) {                                                                                   \
            _Pragma("unroll") for (int i = 0; i < Cfg::A_CHUNKS; ++i)                                    \
                dma16_buffer(a_tile, (lptr_t)(as_ + i * NT * 4), a_voff[i], (int)(a_koff_ * 4));      \
            _Pragma("unroll") for (int i = 0; i < Cfg::B_CHUNKS; ++i)                                    \
                dma16_buffer(b_tile, (lptr_t)(bs_ + i * NT * 4), b_voff[i], k0_ * 4);                 \
        } else {                                                                                         \
            _Pragma("unroll") for (int i = 0; i < Cfg::A_CHUNKS; ++i)                                    \
                __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + a_koff_), (lptr_t)(as_ + i * NT * 4), 16, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < Cfg::B_CHUNKS; ++i)                                    \
                __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + k0_), (lptr_t)(bs_ + i * NT * 4), 16, 0, 0);     \
        }                                                                                                \
    }

#ifdef NOMAD_DIAG
    if ((OPT & 128) && (OPT & 2048)) ts_[1] = wall_clock64();   // prologue detail: address set-up done, first DMA about to be issued
    if ((OPT & 128) && (OPT & 8192)) ts_[3] = wall_clock64();   // set-up detail: accumulators zeroed, first DMA about to be issued
#endif
    NOMAD_GLDS_TILE(0, 0)
    if (STAGES == 3 && nk > 1) NOMAD_GLDS_TILE(1, 1)
    if (OPT & 4096) __builtin_amdgcn_s_setprio(0);

    // fragment read offsets (floats): row R, logical chunk 2*kq + h -> physical chunk ^ swz(R)
    const int frag_row = lane & 31, h = lane >> 5;
    const int swz = (frag_row / RB) % KC;  // WTM, 32 are multiples of RB*KC: independent of the tile offsets
    int koff[BK / 8];
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) koff[kq] = ((kq * 2 + h) ^ swz) * 4;
    const int a_row_off = (wm * Cfg::WTM + frag_row) * BK;
    const int b_row_off = (wn * Cfg::WTN + frag_row) * BK;
    // M16: row fi of a 16-row sub-tile, logical chunk 4 kh + g (16 si rows further on: the same swizzle, 16 is a multiple of RB * KC)
    const int fi16 = lane & 15, g16 = lane >> 4;
    const int swz16 = (fi16 / RB) % KC;
    int koff16[BK / 16 > 0 ? BK / 16 : 1];
#pragma unroll
    for (int kh = 0; kh < BK / 16; ++kh) koff16[kh] = ((kh * 4 + g16) ^ swz16) * 4;
    const int a_row_off16 = (wm * Cfg::WTM + fi16) * BK;
    const int b_row_off16 = (wn * Cfg::WTN + fi16) * BK;
    // acc[i][j] sub-block q (registers 4q .. 4q+3) += x (A side) * y (W side) over 4 k; TR: W as the first operand
    auto mfma16 = [&](f32x16& a, const int q, const float x, const float y) {
        f32x4 t = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        t = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, t, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, t, 0, 0, 0);
        a[4 * q] = t[0];
        a[4 * q + 1] = t[1];
        a[4 * q + 2] = t[2];
        a[4 * q + 3] = t[3];
    };

    SplitConsts sk{};
    if (X3) sk = split_consts();
    int cur = 0;  // LDS buffer of tile kt
    if constexpr ((OPT & 64) != 0) {
        static_assert(M16, "skewed schedule: fp32 products only");
        // OPT bit 64 (round 4): the skewed schedule, in QUARTERS of a K tile.  A wave waits for the barrier and for nothing else: all three
        // stages stay in flight, the fragments a quarter needs are read under the previous quarter's MFMAs.  The 2 TM x 2 TN operand sub-tiles of a tile are
        // read as halves A0 / A1 (sub-tile rows si = 0 / 1 of every 32-row block) and B0 / B1 (sj = 0 / 1); quarter (Aa, Bb) is the
        // TM x TN x 4 MFMAs of those sub-blocks, all four k-steps c.  Order (A0,B0) (A0,B1) | barrier | (A1,B1) (A1,B0): every quarter
        // reads the half the NEXT quarter needs - B1, A1, then the next tile's A0 and B0 (into the registers of the half that has just
        // died) - so 2 TM + 2 TN float4 are live, as in the 32x32x2 schedule; the per-tile vmcnt + barrier sits between the second and
        // the third quarter, behind this wave's last read of the tile, and the DMA of tile kt + 3 follows the third quarter's first
        // k-step.  The next tile's B0 lands in the registers of this tile's B1: tiles alternate the two B register sets (P).
        // Per accumulator sub-block the four k-steps of a tile stay together and ascend: bit-identical to the straight schedule.
        static_assert(STAGES == 3 && BK == 16, "skewed 16x16x4 schedule: 3 stages, one 16-deep k group per tile");
        constexpr int D = Cfg::A_CHUNKS + Cfg::B_CHUNKS;
#define NOMAD_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        if (nk > 2) {
            NOMAD_GLDS_TILE(2, 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");
        } else if (nk > 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        NOMAD_FENCE();
#ifdef NOMAD_DIAG
        if (OPT & 128) ts_[(OPT & 8192) ? 4 : (OPT & 2048) ? 3 : 1] = wall_clock64();
#endif
        f32x4 a0[TM], a1[TM], bq[2][TN];
        auto rdA = [&](f32x4 (&dst)[TM], int st, int si) {
            const float* as = As + st * BM * BK + a_row_off16 + 16 * si * BK + koff16[0];
#pragma unroll
            for (int i = 0; i < TM; ++i) dst[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * BK);
        };
        auto rdB = [&](f32x4 (&dst)[TN], int st, int sj) {
            const float* bs = Bs + st * BN * BK + b_row_off16 + 16 * sj * BK + koff16[0];
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * BK);
        };
        auto mmq = [&](const f32x4 (&a)[TM], const int si, const f32x4 (&b)[TN], const int sj, const int c0, const int c1) {
#pragma unroll
            for (int c = c0; c < c1; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mfma16(acc[i][j], 2 * si + sj, a[i][c], b[j][c]);
        };
        // one full tile (not the last): B0 of this tile sits in bq[P]
        auto tile_step = [&](auto par_c, const int kt) {
            constexpr int P = decltype(par_c)::value;
            rdB(bq[P ^ 1], cur, 1);
            NOMAD_FENCE();
            mmq(a0, 0, bq[P], 0, 0, 4);
            NOMAD_FENCE();
            rdA(a1, cur, 1);
            NOMAD_FENCE();
            mmq(a0, 0, bq[P ^ 1], 1, 0, 4);
            NOMAD_FENCE();
            const int nxt = cur + 1 == STAGES ? 0 : cur + 1;
            // this wave has read all of tile kt, and its share of tile kt+1 has landed (tile kt+2 may still be in flight)
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            NOMAD_FENCE();
            rdA(a0, nxt, 0);
            NOMAD_FENCE();
            mmq(a1, 1, bq[P ^ 1], 1, 0, 1);
            NOMAD_FENCE();
            if (kt + 3 < nk) NOMAD_GLDS_TILE(kt + 3, cur)
            NOMAD_FENCE();
            mmq(a1, 1, bq[P ^ 1], 1, 1, 4);
            NOMAD_FENCE();
            rdB(bq[P ^ 1], nxt, 0);
            NOMAD_FENCE();
            mmq(a1, 1, bq[P], 0, 0, 4);
            NOMAD_FENCE();
            cur = nxt;
        };
        auto last_tile = [&](auto par_c) {
            constexpr int P = decltype(par_c)::value;
            rdB(bq[P ^ 1], cur, 1);
            NOMAD_FENCE();
            mmq(a0, 0, bq[P], 0, 0, 4);
            NOMAD_FENCE();
            rdA(a1, cur, 1);
            NOMAD_FENCE();
            mmq(a0, 0, bq[P ^ 1], 1, 0, 4);
            NOMAD_FENCE();
            mmq(a1, 1, bq[P ^ 1], 1, 0, 4);
            mmq(a1, 1, bq[P], 0, 0, 4);
        };
        rdA(a0, 0, 0);
        rdB(bq[0], 0, 0);
        for (int kt = 0; kt + 1 < nk; ++kt) {
            tile_step(std::integral_constant<int, 0>{}, kt);
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[0][j] = bq[1][j];
        }
        last_tile(std::integral_constant<int, 0>{});
#undef NOMAD_FENCE
    } else
    for (int kt = 0; kt < nk; ++kt) {
        if (STAGES == 2) {
            __syncthreads();  // tile kt has landed (vmcnt(0)) and every wave is done with the other buffer
        } else {
            // this wave's loads of tile kt are done once at most one newer tile (kt+1) is still outstanding
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::A_CHUNKS + Cfg::B_CHUNKS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef NOMAD_DIAG
            if ((OPT & 128) && (OPT & 2048) && kt == 0) ts_[2] = wall_clock64();   // prologue detail: this wave's share of tile 0 has landed
#endif
            __builtin_amdgcn_s_barrier();  // every wave's share of tile kt has landed; buffer of tile kt-1 is free
            asm volatile("" ::: "memory");
#ifdef NOMAD_DIAG
            if ((OPT & 128) && kt == 0) ts_[(OPT & 8192) ? 4 : (OPT & 2048) ? 3 : 1] = wall_clock64();
#endif
        }
        const int nxt = kt + STAGES - 1;
        int nb = cur + STAGES - 1;
        nb = nb >= STAGES ? nb - STAGES : nb;
        // OPT & 8: the DMA of tile kt+STAGES-1 is issued after the first k-quarter's MFMAs instead of right behind the
        // barrier, where all eight waves would queue on the memory pipe before any of them reaches its MFMAs
        if (!(OPT & 8) && nxt < nk) NOMAD_GLDS_TILE(nxt, nb)
        const float* as = As + cur * BM * BK + a_row_off;
        const float* bs = Bs + cur * BN * BK + b_row_off;
        if (X3) {
            static_assert(!X3 || BK % 16 == 0, "X3 needs 16-deep k steps");
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {   // lane (row, h) holds k = 16 ks + 8 h .. + 7: logical chunks 4 ks + 2 h, + 1
                if ((OPT & 8) && ks == (BK / 16) - 1 && nxt < nk) NOMAD_GLDS_TILE(nxt, nb)
                const int c0 = ((4 * ks + 2 * h) ^ swz) * 4, c1 = ((4 * ks + 2 * h + 1) ^ swz) * 4;
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    split_frag8(*reinterpret_cast<const f32x4*>(as + i * 32 * BK + c0), *reinterpret_cast<const f32x4*>(as + i * 32 * BK + c1), sk, ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    split_frag8(*reinterpret_cast<const f32x4*>(bs + j * 32 * BK + c0), *reinterpret_cast<const f32x4*>(bs + j * 32 * BK + c1), sk, bh[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            const float* as16 = As + cur * BM * BK + a_row_off16;
            const float* bs16 = Bs + cur * BN * BK + b_row_off16;
#pragma unroll
            for (int kh = 0; kh < BK / 16; ++kh) {
                f32x4 af[2 * TM], bf[2 * TN];
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as16 + i * 16 * BK + koff16[kh]);
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs16 + j * 16 * BK + koff16[kh]);
                if (OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // the DMA of tile kt + STAGES - 1 behind the first quarter of the tile's MFMAs (OPT & 8), as in the 32x32x2 loop
                    if ((OPT & 8) && kh == BK / 16 - 1 && c == 1 && nxt < nk) NOMAD_GLDS_TILE(nxt, nb)
#pragma unroll
                    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2 * TN; ++j) mfma16(acc[i >> 1][j >> 1], 2 * (i & 1) + (j & 1), af[i][c], bf[j][c]);
                }
                if (OPT & 2) __builtin_amdgcn_s_setprio(0);
            }
        }
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
#undef NOMAD_GLDS_TILE
#ifdef NOMAD_DIAG
    if ((OPT & 128) && !(OPT & 8192)) ts_[(OPT & 2048) ? 4 : 2] = wall_clock64();
#endif
    if (OPT & 4096) __builtin_amdgcn_s_setprio(3);

    if constexpr (TR) {
        // ---- direct epilogue from transposed accumulators (see TR above) ----
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));   // keep the per-lane offsets below out of the K loop's live ranges
        const int mw = m0 + wm * Cfg::WTM, nw = n0 + wn * Cfg::WTN;   // this wave's first output row / column (uniform)
        auto clamp_bytes = [](long long v) { return (unsigned)(v < 0 ? 0 : (v > (1ll << 30) ? (1ll << 30) : v)); };
        // rows >= M lie beyond num_records and are dropped (stores) / read as zero (loads) by the buffer addressing itself
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(uniform_ptr(p.C + grp * p.c_goff + p.cmap.off + (long long)mw * p.cmap.ld + nw)), 0,
            clamp_bytes((long long)(p.M - mw) * p.cmap.ld * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(uniform_ptr(p.bias ? p.bias + grp * p.bias_goff + nw : p.C)), 0, p.bias ? (unsigned)(Cfg::WTN * 4) : 0u, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        // GEMMs WITHOUT a residual only (the caller checks; residual GEMMs keep the LDS epilogue).  All bias values of the wave tile are
        // loaded BEFORE the first store: loads and stores share one in-order vmcnt, so a bias load issued behind a group of stores
        // can only be waited for together with those stores' acknowledgements - the first version loaded the bias per 32-column
        // block and paid two such round trips (timeline: 9 us from the end of the K loop to the last store issued for ~130 instructions).
        {
            // 16x16x4 accumulators, W as the first operand: sub-block q = 2 si + sj of acc[i][j] holds output row 32 i + 16 si + fi,
            // columns 32 j + 16 sj + 4 g .. + 3 - one 16-byte store per sub-block, 64 contiguous bytes per row and instruction
            const int fi = lane_e & 15, gg = lane_e >> 4;
            const int c_voff16 = (fi * p.cmap.ld + 4 * gg) * 4;
            f32x4 b16[TN][2];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int sj = 0; sj < 2; ++sj)   // a zero-length descriptor (no bias) reads as 0.0f
                    b16[j][sj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (j * 32 + sj * 16 + 4 * gg) * 4, 0, 0));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int si = 0; si < 2; ++si)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int sj = 0; sj < 2; ++sj) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * (2 * si + sj) + e] + b16[j][sj][e];
                            if (p.gelu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                            }
                            // The row-block offset goes into the VGPR offset, NOT the scalar offset: with an SGPR soffset the compiler's hazard
                            // recogniser ("VMEM store of more than 8 bytes followed by a VALU write of the data registers") emits no wait state,
                            // and on gfx950 the v_add of the next chunk, issued right behind buffer_store_dwordx4 ... sN offen, corrupted
                            // element 0 of lanes 12-15 / 28-31 / 44-47 / 60-63 (tools/micro/store_hazard.hip, profiles/r04_store_hazard_micro.txt)
                            if (!NOEPI || p.M < 0)   // (NOEPI, a timing probe: the never-true condition keeps the arithmetic alive)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc,
                                                                       c_voff16 + (i * 32 + si * 16) * p.cmap.ld * 4 + (j * 32 + sj * 16) * 4, 0, 0);
                        }
        }
    } else {
    // Epilogue through LDS: an accumulator holds one output column per lane (4-byte stores, 64 per lane and
    // tile).  Each wave parks a 32-row slab (acc + bias) in LDS, then every lane owns 4 consecutive columns of
    // one row and does the rest - pre-activation copy, GELU, GELU' factor, residual - on 16-byte vectors:
    // 4x fewer store instructions, whole 256-byte row segments per 16 lanes.  Measured +3..6 % on the
    // transformer GEMMs (profiles/r01_gemm_sweep_epilogue.json).  The arithmetic per element is unchanged.
    float* Cg = p.C + grp * p.c_goff;
    const float* Rg = p.R ? p.R + grp * p.r_goff : nullptr;
    const float* biasg = p.bias ? p.bias + grp * p.bias_goff : nullptr;
    const float* DGg = p.DG ? p.DG + grp * p.dg_goff : nullptr;
    float* Ug = p.Upre ? p.Upre + grp * p.c_goff : nullptr;
    const bool c_plain = p.cmap.clip_rows >= p.M, r_plain = p.rmap.clip_rows >= p.M;
    constexpr int ELD = Cfg::WTN + 4, CG = Cfg::WTN / 4;
    // OPT bit 16: the epilogue for PLAIN C / R matrices (the caller checked: no ragged row maps, no column blocks; Upre / DG side
    // operands only with bit 32, plain too) with the residual PREFETCHED per slab.  (1) The general epilogue's address arithmetic - a binary search per
    // row for ragged maps, integer divisions for column blocks - unrolled over 16 chunks is most of this kernel's 70 KB of code,
    // and two such instantiations alternating between launches cost the launch after each switch 12-17 us of instruction-cache
    // misses (measured on the bf16 twin, gemm_bf16_8phase.hip.h); (2) C and R are not restrict-qualified, so the compiler keeps
    // every residual load behind the previous chunk's store and waits for it at once - a lane only stores what it loaded, so the
    // loads of a slab can all be issued before its first store.  Same arithmetic, same results.
    constexpr bool PL = (OPT & 16) != 0;
    // the lane index is laundered through an empty asm so that the per-lane row / column indices of the epilogue are computed HERE:
    // hoisted above the K loop they stayed live across it and were spilled to scratch in the 128-register instantiations (8 dwords
    // per lane and tile: +12 % on WRITE_SIZE of conv1, seen in the live traffic counters)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    constexpr int NIT = 32 * CG / 64;
    float* slab = smem + wave * (32 * ELD);
    float bv16[TN][2];   // M16: the bias of this lane's column in each 16-column sub-block
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) {
            const int n = n0 + wn * Cfg::WTN + j * 32 + sj * 16 + (lane_e & 15);
            bv16[j][sj] = (M16 && biasg && n < p.n_valid) ? biasg[n] : 0.f;
        }
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * Cfg::WTN + j * 32 + (lane_e & 31);
        bv[j] = (biasg && n < p.n_valid) ? biasg[n] : 0.f;
    }
    // OPT bit 32768 (round 4): residual AHEAD.  Loads and stores share one in-order vmcnt: a residual load issued behind a group of
    // stores can only be waited for together with those stores' acknowledgements (1-3 us each under load) - the grouped prefetch
    // below pays that four times per tile.  Here slab 0's residual is loaded at the very start, and chunk `it` of slab i + 1 is
    // loaded into the register chunk `it` of slab i has just been consumed from, BEFORE that chunk's store: the only stores in
    // front of a load a wave waits for were issued a whole slab earlier.  One slab of registers (NIT float4) next to the
    // accumulators still to be parked, which fits the 128-register budget (two slabs ahead spilled 58 registers).
    constexpr bool RA = (OPT & 32768) != 0 && PL && !(OPT & 32);
    f32x4 rahead[RA ? NIT : 1];
    // (through a buffer descriptor per slab: one VGPR offset per lane, rows >= M beyond num_records read as zero - eight 64-bit
    // address pairs and their bounds masks would not fit the 128-register budget next to the accumulators)
    auto slab_rsrc = [&](int i) {
        const int mw = m0 + wm * Cfg::WTM + i * 32, nw = n0 + wn * Cfg::WTN;
        long long nb = (long long)(p.M - mw) * p.rmap.ld * 4;
        nb = nb < 0 ? 0 : (nb > (1ll << 30) ? (1ll << 30) : nb);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(Rg + p.rmap.off + (long long)mw * p.rmap.ld + nw)), 0, (unsigned)nb, 0x00020000);
    };
    const int ra_v0 = RA ? ((lane_e / CG) * p.rmap.ld + (lane_e % CG) * 4) * 4 : 0;
    // (slab 0: RA0 chunks before the accumulators are parked, the rest right after slab 0's - every accumulator is still live here)
    constexpr int RA0 = NIT >= 8 ? NIT / 2 : NIT;
    auto ahead0 = [&](int lo, int hi) {
        const __amdgpu_buffer_rsrc_t rr = slab_rsrc(0);
#pragma unroll
        for (int it = lo; it < hi; ++it)
            rahead[RA ? it : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ra_v0 + it * (64 / CG) * p.rmap.ld * 4, 0, 0));
    };
    if (RA && Rg) ahead0(0, RA0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // the slab is private to the wave: after the first barrier (main loop done with the staging LDS) a wave-local
        // fence is enough, LDS operations of one wave execute in order
        // (in groups of NIT / 2 chunks, NIT / 4 with the training operands: the kernel must stay within 128 VGPRs for two workgroups per CU)
        constexpr int NH = (OPT & 32) ? (NIT >= 4 ? NIT / 4 : NIT) : (NIT >= 2 ? NIT / 2 : NIT);
        f32x4 rpre[NH];
        auto prefetch = [&](int it0) {
#pragma unroll
            for (int it = it0; it < it0 + NH; ++it) {
                const int id = lane_e + 64 * it, row = id / CG, cg = id - row * CG;
                const int m = m0 + wm * Cfg::WTM + i * 32 + row;
                const int n = n0 + wn * Cfg::WTN + cg * 4;
                rpre[it - it0] = (m < p.M && n < p.n_valid) ? *reinterpret_cast<const f32x4*>(Rg + p.rmap.off + (long long)m * p.rmap.ld + n)
                                                            : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        };
        if (PL && !RA && Rg) prefetch(0);
        if (!(OPT & 1) || i == 0) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (M16) {   // 16x16x4 accumulators: sub-block q = 2 si + sj, register r: row 16 si + 4 g + r, column 16 sj + fi
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        slab[(16 * (q >> 1) + 4 * (lane_e >> 4) + r) * ELD + j * 32 + 16 * (q & 1) + (lane_e & 15)] = acc[i][j][4 * q + r] + bv16[j][q & 1];
        } else
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * (lane_e >> 5)) * ELD + j * 32 + (lane_e & 31)] = acc[i][j][r] + bv[j];
        if (!(OPT & 1)) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (RA && Rg && i == 0 && RA0 < NIT) ahead0(RA0, NIT);
        if (!NOEPI) {
#pragma unroll
            for (int it = 0; it < 32 * CG / 64; ++it) {
                const int id = lane_e + 64 * it, row = id / CG, cg = id - row * CG;
                const int m = m0 + wm * Cfg::WTM + i * 32 + row;
                const int n = n0 + wn * Cfg::WTN + cg * 4;
                if (PL) {
                    if (!RA && NH < NIT && it > 0 && it % NH == 0 && Rg) prefetch(it);   // next group: after the previous group's stores have been issued
                    f32x4 ra = rahead[RA ? it : 0];
                    if (RA && Rg && i + 1 < TM)
                        rahead[RA ? it : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc(i + 1), ra_v0 + it * (64 / CG) * p.rmap.ld * 4, 0, 0));
                    if (m < p.M && n < p.n_valid) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * ELD + cg * 4);
                        const long long c_idx = p.cmap.off + (long long)m * p.cmap.ld + n;
                        if ((OPT & 32) && Ug) *reinterpret_cast<f32x4*>(Ug + c_idx) = v;   // OPT bit 32, training forward: the pre-activation copy
                        if (!RA && p.gelu) {   // (the residual-ahead instantiations are launched for GELU-free GEMMs only: erf's temporaries would not fit)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        }
                        if ((OPT & 32) && DGg) {                              // OPT bit 32, backward: GELU' of the saved pre-activation (plain map)
                            const f32x4 u = *reinterpret_cast<const f32x4*>(DGg + p.dgmap.off + (long long)m * p.dgmap.ld + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= dgelu_erf_(u[e]);
                        }
                        if (Rg) v += RA ? ra : rpre[it % NH];
                        *reinterpret_cast<f32x4*>(Cg + c_idx) = v;
                    }
                    // OPT bits 256 / 512 (experiment): pace the output stores (s_sleep 4 / 16 = 256 / 1024 cycles after each) so that the
                    // CU's vector memory pipe never holds a long queue of them in front of the other workgroup's operand loads
                    if (OPT & 256) __builtin_amdgcn_s_sleep(4);
                    if (OPT & 512) __builtin_amdgcn_s_sleep(16);
                } else if (m < p.M && n < p.n_valid) {
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
    }
    }   // !TR
    if (t_ + grid_x < nwg) __syncthreads();  // every wave has read its slab: the staging LDS may be refilled
    }
#ifdef NOMAD_DIAG
    if (OPT & 128) {
        if (!(OPT & (2048 | 8192))) ts_[3] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the output stores have been acknowledged
        if (!(OPT & (2048 | 8192))) ts_[4] = wall_clock64();
        if (tid == 0 && blockIdx.x < kTimelineSlots) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = g_timeline + (size_t)blockIdx.x * 6;
            o[0] = ts_[0]; o[1] = ts_[1]; o[2] = ts_[2]; o[3] = ts_[3]; o[4] = ts_[4]; o[5] = hw | ((unsigned long long)xcc << 32);
        }
    }
#endif
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES = 2, bool NOEPI = false, int OPT = 0, bool X3 = false>
__global__ __launch_bounds__(WM* WN * 64, ((OPT & 16) && WM * WN == 8) ? 4 : (((OPT & 16) && WM * WN == 4 && BM * BN == 256 * 128) ? (BK == 8 ? 3 : 2) : 1)) void gemm_f32_glds_kernel(const GemmParams p) {   // OPT bit 16: two 8-wave workgroups per CU = 128 VGPRs (8 values computed in the prologue for the epilogue are spilled over the K loop, none inside it)
    gemm_f32_glds_body<BM, BN, BK, WM, WN, STAGES, NOEPI, OPT, X3>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// ---- Two tile shapes in one launch (round 4) ------------------------------------------------------------------------------------
// N = 768 GEMMs of the bench batch are 2.33 rounds of 256 x 128 tiles on the 512 workgroup slots (fc1: 9.33): the last third of a
// round runs one workgroup per CU on 170 CUs while 86 idle, and costs ~2/3 of a full round.  Workgroups are dispatched in index order:
// here the first n_big workgroups take 256 x 128 tiles over the first rows (whole rounds' worth), the others 128 x 128 tiles (same 8
// waves, same K depth and stages, wave tile 32 x 64) over the remaining rows - twice as many, half as long, spread over all CUs.
// Every instantiation contracts k in the same order: which tile shape computes an element changes no bit of it.  pb / ps: the two
// row ranges as separate problems (plain A / C / R matrices: the second's pointers start at its first row).
template <int OPTB, int OPTS>
__global__ __launch_bounds__(512, 4) void gemm_f32_mixed_kernel(const GemmParams pb, const GemmParams ps, const int n_big) {
    if ((int)blockIdx.x < n_big) gemm_f32_glds_body<256, 128, 16, 4, 2, 3, false, OPTB, false>(pb, blockIdx.x, n_big, 0);
    else gemm_f32_glds_body<128, 128, 16, 4, 2, 3, false, OPTS, false>(ps, (int)blockIdx.x - n_big, (int)gridDim.x - n_big, 0);
}

// persist_blocks > 0: launch at most that many workgroups, each walking several tiles (see the kernel)
template <int BM, int BN, int BK, int WM, int WN, int STAGES = 2, bool NOEPI = false, int OPT = 0, bool X3 = false>
inline hipError_t launch_gemm_glds(GemmParams p, int groups, hipStream_t s, int extra_lds = 0, int persist_blocks = 0) {
    using Cfg = GldsCfg<BM, BN, BK, WM, WN, STAGES>;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.N / BN;
    if (OPT & 16384) {   // lean set-up: division magic (the caller has checked: no ragged A map, group_m == 0)
        p.a_clip_magic = p.tn_magic = 0;
        p.a_clip_shift = p.tn_shift = 0;
        if (p.amap.clip_rows < p.M) fast_div_magic((unsigned)p.amap.clip_rows, &p.a_clip_magic, &p.a_clip_shift);
        if (p.tiles_n > 1) fast_div_magic((unsigned)p.tiles_n, &p.tn_magic, &p.tn_shift);
    }
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_f32_glds_kernel<BM, BN, BK, WM, WN, STAGES, NOEPI, OPT, X3>), 160 * 1024); e != hipSuccess) return e;
    int gx = p.tiles_m * p.tiles_n;
    if (persist_blocks > 0 && gx > persist_blocks) gx = persist_blocks;
    dim3 grid(gx, groups);
    hipLaunchKernelGGL((gemm_f32_glds_kernel<BM, BN, BK, WM, WN, STAGES, NOEPI, OPT, X3>), grid, dim3(Cfg::THREADS),
                       Cfg::LDS_BYTES + extra_lds, s, p);
    return hipGetLastError();
}

// M1: rows of the 256 x 128 part (a multiple of 256, 0 < M1 < p.M); the caller has checked what the lean set-up and the plain
// epilogue need.  Both parts address the whole problem (row maps, bounds): the second only starts at row tile M1 / 128.
template <int OPTB, int OPTS>
inline hipError_t launch_gemm_mixed(GemmParams p, int M1, hipStream_t s) {
    using CfgB = GldsCfg<256, 128, 16, 4, 2, 3>;
    p.tiles_n = p.N / 128;
    p.a_clip_magic = p.tn_magic = 0;
    p.a_clip_shift = p.tn_shift = 0;
    if (p.amap.clip_rows < p.M) fast_div_magic((unsigned)p.amap.clip_rows, &p.a_clip_magic, &p.a_clip_shift);
    if (p.tiles_n > 1) fast_div_magic((unsigned)p.tiles_n, &p.tn_magic, &p.tn_shift);
    GemmParams pb = p, ps = p;
    pb.tiles_m = M1 / 256;
    pb.tile_m_base = 0;
    ps.tiles_m = (p.M - M1 + 127) / 128;
    ps.tile_m_base = M1 / 128;
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_f32_mixed_kernel<OPTB, OPTS>), 160 * 1024); e != hipSuccess) return e;
    const int n_big = pb.tiles_m * pb.tiles_n, n_small = ps.tiles_m * ps.tiles_n;
    hipLaunchKernelGGL((gemm_f32_mixed_kernel<OPTB, OPTS>), dim3(n_big + n_small), dim3(512), CfgB::LDS_BYTES, s, pb, ps, n_big);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Persistent 256 x 128 kernel (round 4).  The per-workgroup timeline of the kernel above (tools/gemm_timeline_f32.py,
// profiles/r04_gemm_timeline_f32.txt) shows a QKV tile spending 176 us in its K loop and 46 us around it - 13 us of address
// set-up that crawls next to a peer workgroup's MFMA stream, 9 us until the first K tile has landed and the first barrier is
// passed, 19 us of LDS-staged epilogue, 5 us until the slot is taken again - and the lone peer fills only ~3/4 of the matrix
// pipe meanwhile: two workgroups are inside their K loops for 58 % of a CU's time.  Here a workgroup stays resident and walks
// tiles blockIdx.x, + gridDim.x, ... (two workgroups per CU):
//  * the LDS-DMA stream never drains: while K tiles nk-2 and nk-1 of an output tile are multiplied, K tiles 0 and 1 of the NEXT
//    output tile are staged (its three per-lane offsets overwrite the current tile's after their last use, its bases are a few
//    scalar instructions with precomputed division magic), so the next K loop starts on data that is already in LDS;
//  * the epilogue is the direct one from transposed accumulators (OPT bit 1024 above): no LDS, so it cannot collide with the
//    staged K tiles, no barrier, and its vector-memory operations are accounted for in the counted vmcnt of the next two K tiles;
//  * nothing is recomputed per tile but ~40 scalar and ~20 vector instructions.
// Same tile, same K loop, same contraction order: bit-identical to every other fp32 instantiation.  Plain C / R matrices,
// one group, contiguous K (the caller checks); A may be a per-clip RowMap (the conv stack).
template <bool NOEPI = false, int OPT = 0>
__global__ __launch_bounds__(512, 4) void gemm_f32_pers_kernel(const GemmParams p) {
    constexpr int BM = 256, BN = 128, BK = 16, WN = 2, ST = 3, NT = 512, KC = 4, RB = 4, WTM = 64, WTN = 64, TM = 2, TN = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [ST][BM][BK]
    float* Bs = smem + ST * BM * BK;   // [ST][BN][BK]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int nwg = p.tiles_m * p.tiles_n, nk = p.K / BK;

    // tile-independent per-lane DMA geometry: LDS chunk id = tid + i * NT, source chunk swizzled (see the kernel above)
    int a_rowl[2], a_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * NT, row = id / KC, pc = id - row * KC;
        a_rowl[i] = row;
        a_sw[i] = (pc ^ ((row / RB) % KC)) * 4;
    }
    const int b_voff = [&] {
        const int row = tid / KC, pc = tid - row * KC;
        return (int)(((long long)row * p.ldw + ((pc ^ ((row / RB) % KC)) * 4)) * 4);
    }();
    // fragment read offsets
    const int frag_row = lane & 31, h = lane >> 5;
    const int swz = (frag_row / RB) % KC;
    int koff[2];
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) koff[kq] = ((kq * 2 + h) ^ swz) * 4;
    const int a_row_off = (wm * WTM + frag_row) * BK, b_row_off = (wn * WTN + frag_row) * BK;

    auto row_addr_fast = [&](int m) -> long long {
        const int c = p.a_clip_magic ? fast_div(m, p.a_clip_magic, p.a_clip_shift) : 0;
        return p.amap.off + (long long)c * p.amap.clip_stride + (long long)(m - c * p.amap.clip_rows) * p.amap.ld;
    };
    // tile t -> (m0, n0), scalar tile bases, per-lane A offsets
    auto setup = [&](int t, int& m0_, int& n0_, const float*& at, const float*& bt, int (&av)[2]) {
        const int wg = xcd_remap(t, nwg);
        const int tm = p.tn_magic ? fast_div(wg, p.tn_magic, p.tn_shift) : wg;
        m0_ = tm * BM;
        n0_ = (wg - tm * p.tiles_n) * BN;
        const long long row0 = row_addr_fast(m0_);
        at = uniform_ptr(p.A + row0);
        bt = uniform_ptr(p.W + (long long)n0_ * p.ldw);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0_ + a_rowl[i];
            m = m < p.M ? m : p.M - 1;
            av[i] = (int)((row_addr_fast(m) - row0 + a_sw[i]) * 4);
        }
    };
    auto issue = [&](const float* at, const float* bt, const int (&av)[2], int k_off_bytes, int stage) {
        float* as_ = As + stage * BM * BK + wave * 256;
        float* bs_ = Bs + stage * BN * BK + wave * 256;
        dma16_buffer(at, (lptr_t)(as_), av[0], k_off_bytes);
        dma16_buffer(at, (lptr_t)(as_ + NT * 4), av[1], k_off_bytes);
        dma16_buffer(bt, (lptr_t)(bs_), b_voff, k_off_bytes);
    };

    // Two cursors walk the same sequence of (output tile, K tile): the LOAD side (a_tile / b_tile / a_voff / k_ld) runs two K
    // tiles ahead of the multiply side and crosses into the next output tile first - its bases and per-lane offsets are
    // recomputed in place right after the current tile's last K tile has been issued.  The last output tile of a workgroup
    // "crosses" into itself: two K tiles are staged that nobody reads (no has-next special case anywhere in the loop).
    // OPT bit 1 (experiment): the second workgroup of every CU starts half a tile late, so that the two do not reach their
    // epilogues together for the rest of the launch (they process equal tiles at equal speed from a common start otherwise)
    if ((OPT & 1) && blockIdx.x >= gridDim.x / 2) {
        const int naps = nk / 4;   // ~ half a K loop: a K tile takes ~3.5 us when two workgroups share the CU, s_sleep 127 ~ 3.4 us
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int tile = blockIdx.x;
    int m0, n0, m0_ld, n0_ld, a_voff[2];
    const float *a_tile, *b_tile;
    setup(tile, m0_ld, n0_ld, a_tile, b_tile, a_voff);
    m0 = m0_ld;
    n0 = n0_ld;
    int k_ld = 0;         // byte offset of the next K tile to stage, within the load side's output tile
    const int k_bytes = p.K * 4;
    issue(a_tile, b_tile, a_voff, k_ld, 0);
    k_ld += BK * 4;
    issue(a_tile, b_tile, a_voff, k_ld, 1);
    k_ld += BK * 4;
    int cur = 0;          // LDS stage of the K tile being multiplied
    int fresh = 2;        // K tiles to go before an epilogue's vector-memory operations no longer sit between the DMA groups
    const bool has_r = p.R != nullptr;
    while (true) {
        const bool has_next = tile + (int)gridDim.x < nwg;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            // this wave's share of K tile kt has landed once only NEWER vector-memory operations are outstanding: the next K
            // tile's three DMA instructions and, for the first two K tiles after an epilogue, that epilogue's 8 bias loads, 16
            // stores and (with a residual) 16 residual loads, which were issued between the two DMA groups
            if (fresh < 2) {
                if (NOEPI) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                else if (has_r) asm volatile("s_waitcnt vmcnt(43)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
                ++fresh;
            } else {
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int nb = cur + 2;
            nb = nb >= ST ? nb - ST : nb;
            const float* as = As + cur * BM * BK + a_row_off;
            const float* bs = Bs + cur * BN * BK + b_row_off;
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                if (kq == 1) {   // stage the K tile two ahead in the stream, behind the first k-step's MFMAs
                    __builtin_amdgcn_sched_barrier(0);   // (unconditional here: without the fence the compiler hoists the DMA to the barrier)
                    issue(a_tile, b_tile, a_voff, k_ld, nb);
                    __builtin_amdgcn_sched_barrier(0);
                    k_ld += BK * 4;
                    if (k_ld == k_bytes) {   // the load side crosses into the next output tile
                        k_ld = 0;
                        setup(has_next ? tile + (int)gridDim.x : tile, m0_ld, n0_ld, a_tile, b_tile, a_voff);
                    }
                }
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * BK + koff[kq]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * BK + koff[kq]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][c], af[i][c], acc[i][j], 0, 0, 0);   // transposed: D^T
            }
            cur = cur + 1 == ST ? 0 : cur + 1;
        }
        fresh = 0;
        // ---- direct epilogue from transposed accumulators (as OPT bit 1024 of the kernel above) ----
        {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
            const int mrow = lane_e & 31, hh = lane_e >> 5;
            auto clamp_bytes = [](long long v) { return (unsigned)(v < 0 ? 0 : (v > (1ll << 30) ? (1ll << 30) : v)); };
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(uniform_ptr(p.C + p.cmap.off + (long long)mw * p.cmap.ld + nw)), 0, clamp_bytes((long long)(p.M - mw) * p.cmap.ld * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(uniform_ptr(has_r ? p.R + p.rmap.off + (long long)mw * p.rmap.ld + nw : p.C)), 0,
                has_r ? clamp_bytes((long long)(p.M - mw) * p.rmap.ld * 4) : 0u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(uniform_ptr(p.bias ? p.bias + nw : p.C)), 0, p.bias ? (unsigned)(WTN * 4) : 0u, 0x00020000);
            const int c_voff = (mrow * p.cmap.ld + 4 * hh) * 4, r_voff = (mrow * p.rmap.ld + 4 * hh) * 4, b_off = 16 * hh;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 rres[TM][4], b4[4];
                if (has_r) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            rres[i][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, r_voff + i * 32 * p.rmap.ld * 4 + (j * 32 + 8 * g) * 4, 0, 0));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) b4[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, b_off + (j * 32 + 8 * g) * 4, 0, 0));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + b4[g][e];
                        if (p.gelu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        }
                        if (has_r) v += rres[i][g];
                        // (VGPR offset only - see the store-data hazard note at OPT bit 1024)
                        if (!NOEPI || p.M < 0)   // (NOEPI, a timing probe: the never-true condition keeps the arithmetic alive)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, c_voff + i * 32 * p.cmap.ld * 4 + (j * 32 + 8 * g) * 4, 0, 0);
                    }
            }
        }
        if (!has_next) break;
        tile += (int)gridDim.x;
        m0 = m0_ld;
        n0 = n0_ld;
    }
}

// workgroups: two per CU (the kernel's residency), never more than there are tiles
inline hipError_t launch_gemm_pers(GemmParams p, hipStream_t s, int num_cus, bool noepi = false, bool one_tile_each = false, bool stagger = false) {
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 128;
    p.a_clip_magic = p.tn_magic = 0;
    p.a_clip_shift = p.tn_shift = 0;
    if (p.amap.clip_rows < p.M) fast_div_magic((unsigned)p.amap.clip_rows, &p.a_clip_magic, &p.a_clip_shift);
    if (p.tiles_n > 1) fast_div_magic((unsigned)p.tiles_n, &p.tn_magic, &p.tn_shift);
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_f32_pers_kernel<false>), 160 * 1024); e != hipSuccess) return e;
    static LdsAttrOnce attr_set_2;
    if (hipError_t e = attr_set_2.ensure(reinterpret_cast<const void*>(gemm_f32_pers_kernel<true>), 160 * 1024); e != hipSuccess) return e;
    const long long nwg = (long long)p.tiles_m * p.tiles_n;
    // one_tile_each: the same kernel launched with one workgroup per tile - no tile loop, only its lean set-up and direct epilogue
    const int grid = (int)((one_tile_each || nwg < 2ll * num_cus) ? nwg : 2ll * num_cus);
    constexpr int lds = 3 * (256 + 128) * 16 * 4;
    if (stagger) {
        static LdsAttrOnce attr2;
        if (hipError_t e = attr2.ensure(reinterpret_cast<const void*>(gemm_f32_pers_kernel<false, 1>), 160 * 1024); e != hipSuccess) return e;
        hipLaunchKernelGGL((gemm_f32_pers_kernel<false, 1>), dim3(grid), dim3(512), lds, s, p);
    } else if (noepi) hipLaunchKernelGGL(gemm_f32_pers_kernel<true>, dim3(grid), dim3(512), lds, s, p);
    else hipLaunchKernelGGL(gemm_f32_pers_kernel<false>, dim3(grid), dim3(512), lds, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// N = 48 variant for the grouped pos-conv (16 groups x [M x 48 x 6144], forward and backward): the 32x32
// MFMA kernels above must pad N to 64 and throw a quarter of their matrix-core work away.  Here a wave owns
// 32 rows x 48 columns as 2 x 3 tiles of v_mfma_f32_16x16x4_f32 (lane l: A[row l&15][k = l>>4], D: col = l&15,
// row = 4*(l>>4) + reg; same 64 FLOP/clk/SIMD), 8 waves = 256 rows per workgroup, BK = 16, the same 3-stage
// LDS-DMA pipeline, source-side swizzle and LDS-staged 16-byte epilogue.  One ds_read_b128 per 16-row operand
// tile feeds four MFMAs: lane group g reads k = 4g..4g+3 and MFMA j contracts {j, 4+j, 8+j, 12+j}.
struct N48Cfg {
    static constexpr int BM = 256, BN = 48, BK = 16, THREADS = 512, STAGES = 3, KC = 4, RB = 4;
    static constexpr int STAGE_BYTES = STAGES * (BM + BN) * BK * 4;
    static constexpr int ELD = BN + 4;
    static constexpr int EPI_BYTES = 8 * 32 * ELD * 4;
    static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
};

// BUFLD: LDS-DMA through buffer descriptors instead of global_load_lds (A/B switch).
template <bool BUFLD = false>
__global__ __launch_bounds__(512) void gemm_f32_n48_kernel(const GemmParams p) {
    using Cfg = N48Cfg;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NT = Cfg::THREADS, KC = Cfg::KC, RB = Cfg::RB, ST = Cfg::STAGES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [ST][256][16]
    float* Bs = smem + ST * BM * BK;  // [ST][48][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = xcd_remap(blockIdx.x, p.tiles_m) * BM;
    const int grp = blockIdx.y;
    const int sl = blockIdx.z;   // K slice (launch_gemm_n48 slices > 1: the loss path's small-M problems; p.K is the slice's depth)
    const float* Ag = p.A + grp * p.a_goff + sl * p.a_soff;
    const float* Wg = p.W + grp * p.w_goff + sl * p.w_soff;

    // LDS-DMA through buffer descriptors (see dma16_buffer): base = this tile's first row, 32-bit lane offsets
    const long long tile_row0 = row_addr(p.amap, m0 < p.M ? m0 : p.M - 1);
    const float* const a_tile = uniform_ptr(Ag + tile_row0);
    int a_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * NT, row = id / KC, pc = id - row * KC;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        a_voff[i] = (int)((row_addr(p.amap, m) - tile_row0 + ((pc ^ ((row / RB) % KC)) * 4)) * 4);
    }
    const bool loads_b = wave < 3;  // 48 rows x 4 chunks = 192 chunks = waves 0..2 (wave-uniform)
    int b_voff = 0;
    if (loads_b) {
        const int row = tid / KC, pc = tid - row * KC;
        b_voff = (int)(((long long)row * p.ldw + ((pc ^ ((row / RB) % KC)) * 4)) * 4);
    }
    f32x4 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
#define NOMAD_N48_TILE(KT, BUF)                                                                          \
    {                                                                                                    \
        const int k0_ = (KT)*BK;                                                                         \
        float* as_ = As + (BUF)*BM * BK + wave * 256;                                                    \
        if (BUFLD) {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
                dma16_buffer(a_tile, (lptr_t)(as_ + i * NT * 4), a_voff[i], k0_ * 4);                     \
            if (loads_b) dma16_buffer(Wg, (lptr_t)(Bs + (BUF)*BN * BK + wave * 256), b_voff, k0_ * 4);    \
        } else {                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(a_tile) + (unsigned)a_voff[i] + k0_ * 4), (lptr_t)(as_ + i * NT * 4), 16, 0, 0); \
            if (loads_b)                                                                                 \
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(Wg) + (unsigned)b_voff + k0_ * 4), (lptr_t)(Bs + (BUF)*BN * BK + wave * 256), 16, 0, 0); \
        }                                                                                                \
    }
    NOMAD_N48_TILE(0, 0)
    if (nk > 1) NOMAD_N48_TILE(1, 1)

    const int fi = lane & 15, g = lane >> 4;
    const int frag_off = ((g ^ ((fi >> 2) & 3)) * 4);  // this lane's chunk inside a 16-float row
    const int a_row_off = (wave * 32 + fi) * BK + frag_off;
    const int b_row_off = fi * BK + frag_off;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // waves 0..2 issue three DMA instructions per tile, the others two: wait for "all but the newest tile"
        if (kt + 1 < nk) {
            if (loads_b) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int nb = cur + 2;
        nb = nb >= ST ? nb - ST : nb;
        if (!BUFLD && kt + 2 < nk) NOMAD_N48_TILE(kt + 2, nb)
        const float* as = As + cur * BM * BK + a_row_off;
        const float* bs = Bs + cur * BN * BK + b_row_off;
        f32x4 af[2], bf[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 16 * BK);
#pragma unroll
        for (int j = 0; j < 3; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 16 * BK);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // production variant: the DMA is issued behind the first quarter of the tile's MFMAs (as in the big GEMM)
            if (BUFLD && c == 1 && kt + 2 < nk) NOMAD_N48_TILE(kt + 2, nb)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][c], bf[j][c], acc[i][j], 0, 0, 0);
        }
        cur = cur + 1 == ST ? 0 : cur + 1;
    }
#undef NOMAD_N48_TILE

    float* Cg = p.C + grp * p.c_goff + sl * p.c_soff;
    const float* Rg = p.R ? p.R + grp * p.r_goff : nullptr;
    const float* biasg = p.bias ? p.bias + grp * p.bias_goff : nullptr;
    const float* DGg = p.DG ? p.DG + grp * p.dg_goff : nullptr;
    float* Ug = p.Upre ? p.Upre + grp * p.c_goff : nullptr;
    const bool c_plain = p.cmap.clip_rows >= p.M, r_plain = p.rmap.clip_rows >= p.M;
    constexpr int ELD = Cfg::ELD;
    float* slab = smem + wave * (32 * ELD);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float bv = biasg ? biasg[j * 16 + fi] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(i * 16 + 4 * g + r) * ELD + j * 16 + fi] = acc[i][j][r] + bv;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 6; ++it) {  // 32 rows x 12 float4 groups = 384 items
        const int id = lane + 64 * it, row = id / 12, cg = id - row * 12;
        const int m = m0 + wave * 32 + row, n = cg * 4;
        if (m < p.M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * ELD + n);
            const long long c_idx = (c_plain ? p.cmap.off + (long long)m * p.cmap.ld : row_addr(p.cmap, m)) + n;
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
            if (Rg) v += *reinterpret_cast<const f32x4*>(
                        Rg + (r_plain ? p.rmap.off + (long long)m * p.rmap.ld : row_addr(p.rmap, m)) + n);
            *reinterpret_cast<f32x4*>(Cg + c_idx) = v;
        }
    }
}

template <bool BUFLD = false>
inline hipError_t launch_gemm_n48(GemmParams p, int groups, hipStream_t s, int slices = 1) {
    p.tiles_m = (p.M + N48Cfg::BM - 1) / N48Cfg::BM;
    p.tiles_n = 1;
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_f32_n48_kernel<BUFLD>), 160 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL(gemm_f32_n48_kernel<BUFLD>, dim3(p.tiles_m, groups, slices), dim3(N48Cfg::THREADS), N48Cfg::LDS_BYTES, s, p);
    return hipGetLastError();
}

template <int BM, int BN, int BK, int WM = 2, int WN = 2, int ABL = 0>
inline hipError_t launch_gemm(GemmParams p, int groups, hipStream_t s, int extra_lds = 0) {
    using Cfg = GemmCfg<BM, BN, BK, WM, WN>;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.N / BN;
    static LdsAttrOnce attr_set;
    if (hipError_t e = attr_set.ensure(reinterpret_cast<const void*>(gemm_f32_kernel<BM, BN, BK, WM, WN, ABL>), 160 * 1024); e != hipSuccess) return e;
    dim3 grid(p.tiles_m * p.tiles_n, groups);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, ABL>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES + extra_lds, s,
                       p);
    return hipGetLastError();
}

}  // namespace nomad
