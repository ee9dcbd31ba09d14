// The grouped positional convolution of the bf16 forward (fairseq pos_conv: Conv1d(768, 768, k = 128, padding 64, groups 16) + SamePad
// + GELU, added to its input; SURVEY.md section 3.2, called at nomad.py:226 / :245) with its input SLAB resident in LDS (round 5).
//
// As a GEMM the layer is 16 groups x [M x 48 x 6144]: output frame t of group g contracts the 6144 contiguous elements of the padded,
// group-major input xpad[g][clip][frame][48] that start at frame t (frontend.hip.h).  Rows t and t + 1 of that "A matrix" share 127 / 128
// of their elements, but a GEMM kernel cannot know: gemm_bf16_glds_kernel<128, 64> re-fetched every row's 12 KB K-tile by K-tile (32 KB of
// LDS-DMA per 128 x 64 x 64 step, 41 FLOP per byte - bound by the L2 -> LDS path at 645 TFLOP/s effective, a quarter of it spent on the
// padding of N = 48 to 64), 0.70 ms of config C5's 17.4 ms pass.
// Here a workgroup (4 waves) owns FR = 64 RT consecutive frames of one (clip, group) (of two short clips, 32 RT frames each: CPW below):
//   * it copies the (FR + 128) x 48 slab they read - ONE contiguous range of xpad, 61 KB for FR = 512 - into LDS once, by LDS-DMA;
//   * row m's K vector is then the 12 KB starting at LDS byte 96 m, so the fragment of row m for k-step ks (32 values) is the 16 bytes
//     at 96 m + 64 ks + 16 fq: one address per lane for the whole K loop, the k-step in the instruction's offset field.  The 96-byte row
//     stride is bank-conflict-free for 8 consecutive rows (24 banks apart, 4 banks each);
//   * the weights (48 x 6144 per group, the same for every workgroup of the group) never pass through LDS: each lane loads its 16 bytes
//     of a k-step straight into the MFMA operand registers, two k-steps ahead, from a copy of the panel in FRAGMENT order (a wave's
//     load is 1 KB of consecutive bytes); a workgroup's four waves read the same lines (L1 hits).  No barrier inside the K loop;
//   * v_mfma_f32_16x16x32_bf16, transposed product (W rows x frames): N = 48 is three 16-row tiles exactly - no padded columns - and a
//     lane ends up with 4 consecutive output channels of one frame: 8-byte stores, the residual (the input itself, frame 64 + t of the
//     slab) read back from LDS.
// The contraction order of an output element (k-steps 0 .. 191 in sequence) does not depend on RT, the batch or the clip's position:
// one kernel for uniform and ragged batches, a clip's bits are the same in every batch.
#pragma once
#include <hip/hip_runtime.h>

#include "dtypes.hip.h"
#include "gemm_f32.hip.h"

namespace nomad {

// The weights in FRAGMENT ORDER: wfrag[g][ks][j][lane][8] = W[g][16 j + (lane & 15)][32 ks + 8 (lane >> 4) ..] - what lane `lane` feeds
// v_mfma_f32_16x16x32_bf16 for k-step ks and W row tile j, so that a wave's load instruction reads 1 KB of CONSECUTIVE bytes.  Loaded from
// the row-major panel the same instruction touched 16 cache lines (64 bytes of each) and the texture-address units were 93 % busy with
// the kernel at 45 % of its matrix time (profiles/r05_pmc_posconv_bf16_slab.txt).  grid: 16 * 192 blocks of 192 threads.
__global__ __launch_bounds__(192) void posconv_wfrag_kernel(const float* __restrict__ w /* [16][64][6144] */, bf16_t* __restrict__ wfrag) {
    const int g = blockIdx.x / 192, ks = blockIdx.x - g * 192;
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* src = w + ((long long)g * 64 + 16 * j + (lane & 15)) * 6144 + 32 * ks + 8 * (lane >> 4);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)src[e];
    *reinterpret_cast<bf16x8*>(wfrag + (((long long)g * 192 + ks) * 3 + j) * 512 + lane * 8) = v;
}
constexpr size_t posconv_wfrag_elems() { return (size_t)16 * 192 * 3 * 512; }

constexpr int posconv_slab_frames(int rt, int cpw) { return (4 / cpw) * 16 * rt; }                  // frames of one clip per workgroup
constexpr int posconv_slab_lds(int rt, int cpw) { return cpw * (posconv_slab_frames(rt, cpw) + 128) * 96; }

// A workgroup of 4 waves takes FRC = (4 / CPW) 16 RT consecutive frames of each of CPW consecutive clips for one group: CPW = 1 for
// long clips (512 frames); 2 or more clips share a workgroup when they are short, so that the weight fragments a wave streams through
// its registers still meet 8 (RT) row tiles - one clip of T = 199 per workgroup streamed the 590 KB panel for 199 rows and ran slower
// than the GEMM it replaces.  An output's contraction order does not depend on any of this.
// grid: (ceil(max T / FRC), ceil(B / CPW), 16 groups), 256 threads, dynamic LDS posconv_slab_lds(RT, CPW).
// xpad: [16][frames of all clips + 128 each][48] bf16 (uniform: clip b at frame b (T + 128); ragged: at ppref[b], ppref[B] per group);
// W: posconv_wfrag_kernel's fragment-ordered weights; bias: [768] fp32; y: [M][768] bf16, rows row0(b) + t, columns 48 g ..
template <int RT, int CPW>
__global__ __launch_bounds__(256, 2) void posconv_bf16_slab_kernel(const bf16_t* __restrict__ xpad, const bf16_t* __restrict__ W,
                                                                   const float* __restrict__ bias, bf16_t* __restrict__ y, int T_uniform, int B,
                                                                   const int* __restrict__ tpref, const int* __restrict__ ppref) {
    constexpr int WPC = 4 / CPW;                       // waves per clip
    constexpr int FRC = posconv_slab_frames(RT, CPW);
    constexpr int SLAB_BYTES = (FRC + 128) * 96;       // per clip
    static_assert(CPW == 1 || CPW == 2 || CPW == 4, "clips per workgroup");
    static_assert(SLAB_BYTES % 4096 == 0, "a slab is copied in rounds of 256 threads x 16 bytes");
    extern __shared__ __attribute__((aligned(16))) char ps_lds[];
    const int g = blockIdx.z, t0 = blockIdx.x * FRC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int fr = lane & 15, fq = lane >> 4;
    const int slot = wave_u / WPC;                     // the clip (of this workgroup's CPW) this wave multiplies
    long long row0 = 0;                                // ... its first output row, its frames
    int T = 0;
    const long long grp_frames = tpref ? (long long)ppref[B] : (long long)B * (T_uniform + 128);

    // ---- the slabs: padded frames t0 .. t0 + FRC + 127 of each clip for this group, as far as the clip's T + 128 padded frames go
    // (chunks past them repeat its last one: they only feed rows >= T, which are never stored) ----
    bool any = false;
#pragma unroll
    for (int cs = 0; cs < CPW; ++cs) {
        const int b = blockIdx.y * CPW + cs;
        if (b >= B) break;
        const int Tb = tpref ? tpref[b + 1] - tpref[b] : T_uniform;
        if (cs == slot) {
            T = Tb;
            row0 = tpref ? (long long)tpref[b] : (long long)b * T_uniform;
        }
        if (t0 >= Tb) continue;
        any = true;
        const long long pad0 = tpref ? (long long)ppref[b] : (long long)b * (T_uniform + 128);
        const char* src = reinterpret_cast<const char*>(xpad + (g * grp_frames + pad0 + t0) * 48);
        const int valid = (Tb + 128 - t0 < FRC + 128 ? Tb + 128 - t0 : FRC + 128) * 96;
#pragma unroll
        for (int i = 0; i < SLAB_BYTES / 4096; ++i) {
            int off = (i * 256 + tid) * 16;
            off = off < valid ? off : valid - 16;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lptr_t)(ps_lds + cs * SLAB_BYTES + i * 4096 + wave_u * 1024), 16, 0, 0);
        }
    }
    if (!any) return;  // whole workgroup (every condition above is uniform over it), before any barrier
    // ---- this lane's weight fragments: wfrag[g][ks][j][lane], two k-steps in flight (a wave-uniform base + one 32-bit lane offset) ----
    const char* wg = reinterpret_cast<const char*>(W + (long long)g * 192 * 3 * 512);   // (blockIdx.z: already scalar)
    const unsigned wlane = (unsigned)lane * 16u;
    bf16x8 w0[3], w1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        w0[j] = *reinterpret_cast<const bf16x8*>(wg + (wlane + (unsigned)j * 1024u));
        w1[j] = *reinterpret_cast<const bf16x8*>(wg + (wlane + 3072u + (unsigned)j * 1024u));
    }
    f32x4 acc[RT][3];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wrow0 = (wave_u % WPC) * (16 * RT);   // this wave's first frame inside its clip's FRC
    const bool wave_live = t0 + wrow0 < T;          // wave-uniform (T = 0 for a slot past the batch's end)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!wave_live) return;                         // (no barrier below)

    const char* slab = ps_lds + slot * SLAB_BYTES;
    const char* ap = slab + (wrow0 + fr) * 96 + 16 * fq;   // + 1536 i (row tile) + 64 ks (k-step)
    // Software pipeline, written out: the fragments of k-step ks + 1 are read while the 3 RT MFMAs of k-step ks run (left alone, hipcc
    // sinks every read to just in front of its three MFMAs and waits for it there: LDS latency per row tile instead of per K loop);
    // the weights of k-steps ks + 2, ks + 3 are loaded while ks, ks + 1 are multiplied.
    bf16x8 a0[RT], a1[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) a0[i] = *reinterpret_cast<const bf16x8*>(ap + i * 1536);
    // two k-steps: multiply with (wa, wb) - loaded two calls ago - while (wn0, wn1), the weights of k-steps ks + 2 / ks + 3, are fetched
    auto two_steps = [&](const bf16x8 (&wa)[3], const bf16x8 (&wb)[3], bf16x8 (&wn0)[3], bf16x8 (&wn1)[3], int ks) {
        const int kn = ks + 2 < 192 ? ks + 2 : 190;   // (the last call re-reads the panel's end: in bounds, unused)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wn0[j] = *reinterpret_cast<const bf16x8*>(wg + (wlane + (unsigned)kn * 3072u + (unsigned)j * 1024u));
            wn1[j] = *reinterpret_cast<const bf16x8*>(wg + (wlane + (unsigned)kn * 3072u + 3072u + (unsigned)j * 1024u));
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) a1[i] = *reinterpret_cast<const bf16x8*>(ap + i * 1536 + (ks + 1) * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], a0[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int k2 = ks + 2 < 192 ? ks + 2 : 191;   // (past the end: a valid slab address, unused)
#pragma unroll
            for (int i = 0; i < RT; ++i) a0[i] = *reinterpret_cast<const bf16x8*>(ap + i * 1536 + k2 * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], a1[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    bf16x8 w2[3], w3[3];
#pragma unroll 1
    for (int ks = 0; ks < 192; ks += 4) {   // (the two weight register sets swap roles: no copies)
        two_steps(w0, w1, w2, w3, ks);
        two_steps(w2, w3, w0, w1, ks + 2);
    }

    // ---- epilogue: acc[i][j][r] = conv[frame wrow0 + 16 i + fr][channel 16 j + 4 fq + r]; x + gelu(conv + bias), the same operations in
    // the same order as the GEMM epilogue this replaces (gemm_bf16.hip.h) ----
    float bq[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[j][r] = bias[g * 48 + 16 * j + 4 * fq + r];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int t = t0 + wrow0 + 16 * i + fr;
        if (t < T) {
            bf16_t* dst = y + (row0 + t) * 768 + g * 48 + 4 * fq;
            const char* res = slab + (wrow0 + 16 * i + fr + 64) * 96 + 8 * fq;   // the layer's input at frame t: padded frame t + 64
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + 32 * j);
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bq[j][r];
                    v = gelu_bf16out(v);
                    v += (float)rv[r];
                    ov[r] = (bf16_t)v;
                }
                *reinterpret_cast<bf16x4*>(dst + 16 * j) = ov;
            }
        }
    }
}

template <int RT, int CPW>
inline hipError_t launch_posconv_bf16_slab(const bf16_t* xpad, const bf16_t* W, const float* bias, bf16_t* y, int max_t, int B,
                                           const int* tpref, const int* ppref, hipStream_t s) {
    static LdsAttrOnce configured;
    auto kern = posconv_bf16_slab_kernel<RT, CPW>;
    constexpr int lds = posconv_slab_lds(RT, CPW);
    constexpr int frc = posconv_slab_frames(RT, CPW);
    if (hipError_t e = configured.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((max_t + frc - 1) / frc, (B + CPW - 1) / CPW, 16), dim3(256), lds, s, xpad, W, bias, y, max_t, B, tpref, ppref);
    return hipGetLastError();
}

}  // namespace nomad
