// Waveform front end: Conv1d(1->512, k=10, s=5, no bias) -> GroupNorm(512 groups) -> GELU
// (fairseq ConvFeatureExtractionModel layer 0, extractor_mode=default).
//
// GroupNorm needs per-(clip, channel) statistics over all L0 = (N-10)/5+1 frames before any
// output element can be normalised.  Because the layer has a single input channel, those
// statistics are quadratic forms of the waveform's 10x10 frame autocorrelation:
//     y[c,t] = sum_j w[c,j] x[5t+j]
//     sum_t y      = sum_j   w[c,j]        S[j],      S[j]   = sum_t x[5t+j]
//     sum_t y^2    = sum_jk  w[c,j] w[c,k] R[j][k],   R[j][k] = sum_t x[5t+j] x[5t+k]
// so one fp64 pass over the 256 KB waveform (not the 26 MB conv output) yields mean and variance
// of all 512 channels; the conv itself is then evaluated once, with normalise+affine folded into
// a per-(clip,channel) scale/shift, GELU applied in registers, and written time-major.
#pragma once
#include <hip/hip_runtime.h>

#include "dtypes.hip.h"
#include "gemm_f32.hip.h"

namespace nomad {

constexpr int kStatsPerClip = 65;  // 10 sums + 55 upper-triangular products

// The 65 sums of a clip are taken in chunks of kStatsChunk conv-0 frames, one workgroup per (chunk, clip) - a 30 s clip
// is 12 workgroups instead of one - and folded per clip in chunk order (wav_stats_fold_kernel): the order depends on the
// clip's length only, so the statistics are batch-invariant and deterministic.
// grid: (ceil(max L0 / kStatsChunk), B) blocks of 256 threads.  part[(b * gridDim.x + j)][0..9] = S, [10..64] = R
// (j<=k, row-major) over frames [j * kStatsChunk, ...).  Ragged batches: lens != nullptr gives each clip's sample
// count, clips are `n_samples` (the row stride) apart.
// Round 6: clips of at most 8192 frames (2.6 s) are summed in chunks of 1024 frames - a 1 s clip (3 276 frames, configs[3]) was ONE
// workgroup per clip: 32 workgroups, 13 dependent load rounds each, 49 us per launch for 2 MB of waveform; four chunks per such clip fill
// 128 workgroups.  Longer clips keep 8192 (with 1024 the 65 wave reductions per chunk tripled the kernel's time on 30 s clips: 38 -> 118 us
// per configs[4] pass).  The chunk size, hence the summation order, is a function of the CLIP's length only.
constexpr int kStatsChunk = 8192, kStatsChunkShort = 1024;
__host__ __device__ constexpr int stats_chunk(int L0) { return L0 <= kStatsChunk ? kStatsChunkShort : kStatsChunk; }
// chunk slots a batch needs per clip when its longest clip has max_l0 frames (a short clip beside a long one may need 8 of its own)
__host__ __device__ constexpr int stats_chunk_slots(int max_l0) {
    return max_l0 <= kStatsChunk ? (max_l0 + kStatsChunkShort - 1) / kStatsChunkShort
                                 : ((max_l0 + kStatsChunk - 1) / kStatsChunk > kStatsChunk / kStatsChunkShort ? (max_l0 + kStatsChunk - 1) / kStatsChunk
                                                                                                            : kStatsChunk / kStatsChunkShort);
}
__global__ __launch_bounds__(256) void wav_stats_kernel(const float* __restrict__ wav, int n_samples, int L0,
                                                        double* __restrict__ part, const int* __restrict__ lens) {
    const int b = blockIdx.y, j = blockIdx.x;
    const float* x = wav + (long long)b * n_samples;
    if (lens) L0 = (lens[b] - 10) / 5 + 1;
    const int chunk = stats_chunk(L0);
    if (j * chunk >= L0) return;
    const int t_end = min(L0, (j + 1) * chunk);
    double acc[kStatsPerClip];
#pragma unroll
    for (int i = 0; i < kStatsPerClip; ++i) acc[i] = 0.0;
    for (int t = j * chunk + threadIdx.x; t < t_end; t += 256) {
        double v[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) v[q] = (double)x[5 * t + q];
        int idx = 10;
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            acc[q] += v[q];
#pragma unroll
            for (int k = q; k < 10; ++k) acc[idx++] += v[q] * v[k];
        }
    }
    __shared__ double red[4][kStatsPerClip];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kStatsPerClip; ++i) {
        double s = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < kStatsPerClip)
        part[((long long)b * gridDim.x + j) * kStatsPerClip + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// grid: B blocks of 128 threads.  stats[b][i] = sum over the clip's chunks, in chunk order.
__global__ __launch_bounds__(128) void wav_stats_fold_kernel(const double* __restrict__ part, int nchunk_max, int L0,
                                                             double* __restrict__ stats, const int* __restrict__ lens) {
    const int b = blockIdx.x, i = threadIdx.x;
    if (i >= kStatsPerClip) return;
    if (lens) L0 = (lens[b] - 10) / 5 + 1;
    const int chunk = stats_chunk(L0), nchunk = (L0 + chunk - 1) / chunk;
    double s = 0.0;
    for (int j = 0; j < nchunk; ++j) s += part[((long long)b * nchunk_max + j) * kStatsPerClip + i];
    stats[(long long)b * kStatsPerClip + i] = s;
}

// grid: B blocks of 512 threads (one per channel).  scale = rstd*gamma, shift = beta - mean*rstd*gamma.
__global__ __launch_bounds__(512) void gn_fold_kernel(const double* __restrict__ stats, const float* __restrict__ w0,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int L0, float* __restrict__ scale, float* __restrict__ shift,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      const int* __restrict__ lens) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (lens) L0 = (lens[b] - 10) / 5 + 1;
    const double* st = stats + (long long)b * kStatsPerClip;
    double w[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) w[j] = (double)w0[c * 10 + j];
    double s1 = 0.0, s2 = 0.0;
    int idx = 10;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        s1 += w[j] * st[j];
#pragma unroll
        for (int k = j; k < 10; ++k) {
            const double r = st[idx++];
            s2 += (k == j ? 1.0 : 2.0) * w[j] * w[k] * r;
        }
    }
    const double mean = s1 / L0;
    double var = s2 / L0 - mean * mean;  // biased variance, as torch group_norm
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + 1e-5);
    const double g = (double)gamma[c];
    scale[b * 512 + c] = (float)(rstd * g);
    shift[b * 512 + c] = (float)((double)beta[c] - mean * rstd * g);
    if (mean_out) {  // saved for the GroupNorm backward
        mean_out[b * 512 + c] = (float)mean;
        rstd_out[b * 512 + c] = (float)rstd;
    }
}

// grid: (ceil(L0/FR), B), 256 threads.  Thread = (4 channels) x (frame parity); samples staged in LDS.
// out[b][t][c] time-major, c contiguous: each frame is one 2 KB coalesced row.
constexpr int kConv0Frames = 64;
// VAR (libnomad_diag.so, nomad_diag_conv0_bf16 / tools/race_hunt_conv0.py; 0 everywhere else): variants of the kernel that
// located the packed-FP32 hazard in a build WITH those instructions - bit 0: samples straight from global memory
// (no LDS: still fails), bit 1: the tap loop kept scalar (no v_pk_fma_f32: never fails).
template <typename TOut, int VAR = 0>
__global__ __launch_bounds__(256) void conv0_gn_gelu_kernel(const float* __restrict__ wav, int n_samples, int L0,
                                                            const float* __restrict__ w0, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, TOut* __restrict__ out,
                                                            const int* __restrict__ lens, const int* __restrict__ pref0,
                                                            long long out_plane = 0) {
    __shared__ float xs[kConv0Frames * 5 + 8];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * kConv0Frames;
    long long out_row0 = (long long)b * L0;
    if (lens) {  // ragged: this clip's own frame count and packed output position
        L0 = (lens[b] - 10) / 5 + 1;
        out_row0 = pref0[b];
        if (t0 >= L0) return;
    }
    const int nfr = min(kConv0Frames, L0 - t0);
    const float* x = wav + (long long)b * n_samples + 5 * t0;
    const int nx = 5 * nfr + 5;
    for (int i = threadIdx.x; i < nx; i += 256) xs[i] = x[i];
    const int cq = threadIdx.x & 127, par = threadIdx.x >> 7;
    float w[4][10], sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cq * 4 + q;
        sc[q] = scale[b * 512 + c];
        sh[q] = shift[b * 512 + c];
#pragma unroll
        for (int j = 0; j < 10; ++j) w[q][j] = w0[c * 10 + j];
    }
    __syncthreads();
    TOut* o = out + (out_row0 + t0) * 512 + cq * 4;
    for (int t = par; t < nfr; t += 2) {
        float xv[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) xv[j] = (VAR & 1) ? x[5 * t + j] : xs[5 * t + j];   // VAR bit 0: race-hunt variant without LDS reads
        float4 r;
        float* rp = reinterpret_cast<float*>(&r);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                y = fmaf(w[q][j], xv[j], y);
                if (VAR & 2) asm volatile("" : "+v"(y));   // VAR bit 1: keeps the compiler from pairing channels into v_pk_fma_f32
            }
            rp[q] = gelu_erf(fmaf(y, sc[q], sh[q]));
        }
        store4p<TOut>(o + (long long)t * 512, out_plane, r);
    }
}

// conv0 on the matrix cores, for the bf16 output path (config C5): the 10-tap, 1-input-channel convolution is a GEMM with K = 10,
// too thin for MFMA in bf16 alone - but K = 32 of v_mfma_f32_16x16x32_bf16 holds the THREE bf16 products of the hi / lo split
// at once: k 0..9 = x_hi w_hi, 10..19 = x_hi w_lo, 20..29 = x_lo w_hi (x_lo w_lo, 2^-16 relative, dropped; 30, 31 zero), fp32
// accumulation - an fp32-class conv for ONE matrix instruction per 16 channels x 16 frames, where conv0_gn_gelu_kernel spends
// 40 of its 110 vector instructions per 4 outputs on the taps (the rest - GroupNorm affine, GELU, convert - stays on the VALU).
// Computed transposed, D[channel][frame] = W[channel][k] X[k][frame], with the channel rows of the four MFMAs of a 64-channel
// group interleaved (row 4q + r of MFMA j = channel 64 g + 16 q + 4 j + r) so that a lane (frame lane & 15, q = lane >> 4) ends
// up with 16 CONSECUTIVE channels: two 16-byte stores per group, four lanes fill a 128-byte line of the time-major output.
// wfrag: the A operands in fragment order [8 groups][4][64 lanes][8 bf16] (conv0_wfrag_kernel).  grid: (ceil(L0 / 256), B),
// 256 threads = 4 waves x 64 frames (four 16-frame B operands per wave share every A fragment).
constexpr int kConv0MfmaFrames = 256;   // per workgroup: 4 waves x 4 B operands of 16 frames
constexpr int kConv0MfmaOcc = 4, kConv0MfmaUf = 1;   // the instantiation the forward uses (tools/conv0_time.py)
__global__ __launch_bounds__(256) void conv0_wfrag_kernel(const float* __restrict__ w0, bf16_t* __restrict__ wfrag) {
    // one thread per (group g, j, lane): blockIdx.x = g * 4 + j, threadIdx.x = lane (64)
    const int g = blockIdx.x >> 2, j = blockIdx.x & 3, lane = threadIdx.x;
    const int row = lane & 15, kb = lane >> 4;
    const int c = 64 * g + 16 * (row >> 2) + 4 * j + (row & 3);
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * kb + e, seg = k / 10, tap = k - 10 * seg;
        const float w = k < 30 ? w0[c * 10 + tap] : 0.f;
        const bf16_t hi = (bf16_t)w;
        a[e] = (k >= 30) ? (bf16_t)0.f : (seg == 1 ? (bf16_t)(w - (float)hi) : hi);
    }
    *reinterpret_cast<bf16x8*>(wfrag + ((size_t)blockIdx.x * 64 + lane) * 8) = a;
}

// OCC: waves per SIMD the register allocation must allow; UF: 16-frame tiles whose 16 GELUs each are issued together (more
// independent rcp / exp chains in flight per wave, more registers) - measured in tools/conv0_time.py.
// ABL (libnomad_diag.so timing probe): 2 = no GELU (the affine result is stored): 0.68 ms of the kernel's 0.88 at 32 x 30 s are
// its 3.1 GB of output, the GELU adds the rest (gpurun_out/conv0mfma4; the VALU kernel: 1.02-1.13 ms).
template <int OCC, int UF, int ABL = 0>
__global__ __launch_bounds__(256, OCC) void conv0_mfma_gn_gelu_kernel(const float* __restrict__ wav, int n_samples, int L0,
                                                                      const bf16_t* __restrict__ wfrag, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, bf16_t* __restrict__ out,
                                                                      const int* __restrict__ lens, const int* __restrict__ pref0) {
    constexpr int NFT = kConv0MfmaFrames / 64;                            // 16-frame B operands per wave
    static_assert(NFT % UF == 0, "tiles per wave must divide by the unroll factor");
    __shared__ __attribute__((aligned(16))) float scs[512], shs[512];
    __shared__ float xs[kConv0MfmaFrames * 5 + 8];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * kConv0MfmaFrames;
    long long out_row0 = (long long)b * L0;
    if (lens) {  // ragged: this clip's own frame count and packed output position
        L0 = (lens[b] - 10) / 5 + 1;
        out_row0 = pref0[b];
        if (t0 >= L0) return;
    }
    const int nfr = min(kConv0MfmaFrames, L0 - t0);
    const float* x = wav + (long long)b * n_samples + 5 * t0;
    const int nx = 5 * nfr + 5;
    for (int i = threadIdx.x; i < kConv0MfmaFrames * 5 + 8; i += 256) xs[i] = i < nx ? x[i] : 0.f;
    for (int i = threadIdx.x; i < 512; i += 256) {
        scs[i] = scale[b * 512 + i];
        shs[i] = shift[b * 512 + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int f_base = wave * (16 * NFT);         // this wave's first frame inside the block
    if (f_base >= nfr) return;                    // wave-uniform, after the only barrier
    // B operands: frame f_base + 16 ft + col, k = 8 q + e
    bf16x8 xb[NFT];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) {
        const float* xp = xs + 5 * (f_base + 16 * ft + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * q + e, seg = k / 10, tap = k - 10 * seg;
            const float v = xp[k < 30 ? tap : 0];
            const bf16_t hi = (bf16_t)v;
            xb[ft][e] = (k >= 30) ? (bf16_t)0.f : (seg == 2 ? (bf16_t)(v - (float)hi) : hi);
        }
    }
    bf16_t* const o_lane = out + (out_row0 + t0 + f_base + col) * 512 + 16 * q;
    // A fragments straight from global memory (32 KB for the whole layer: L1 / L2 hits), the next group's loaded one group ahead;
    // staging them in LDS (32 KB per workgroup) capped the kernel at 3 waves per SIMD
    const bf16_t* wl = wfrag + lane * 8;
    bf16x8 an[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) an[j] = *reinterpret_cast<const bf16x8*>(wl + j * 512);
    for (int g = 0; g < 8; ++g) {
        bf16x8 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = an[j];
        if (g < 7) {
#pragma unroll
            for (int j = 0; j < 4; ++j) an[j] = *reinterpret_cast<const bf16x8*>(wl + ((g + 1) * 4 + j) * 512);
        }
        f32x4 sc4[4], sh4[4];   // this lane's 16 channels of the group
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc4[j] = *reinterpret_cast<const f32x4*>(scs + 64 * g + 16 * q + 4 * j);
            sh4[j] = *reinterpret_cast<const f32x4*>(shs + 64 * g + 16 * q + 4 * j);
        }
#pragma unroll
        for (int f0 = 0; f0 < NFT; f0 += UF) {
            if (f_base + 16 * f0 >= nfr) break;   // wave-uniform
            f32x4 acc[UF][4];
#pragma unroll
            for (int u = 0; u < UF; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], xb[f0 + u], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                bf16x8 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = fmaf(acc[u][j][r], sc4[j][r], sh4[j][r]);
                        // (round 6: the bf16-output GELU of the bf16 GEMM epilogues here too - 6 instead of 13 instructions, |error| <= 5.5e-5
                        // against the erf form, a 70th of the rounding step of an output of size 1; ABL 3 = the erf form, A/B)
                        const float y = ABL == 2 ? z : (ABL == 3 ? gelu_erf(z) : gelu_bf16out(z));
                        if (j < 2) lo[4 * j + r] = (bf16_t)y;
                        else hi[4 * (j - 2) + r] = (bf16_t)y;
                    }
                if (f_base + 16 * (f0 + u) + col < nfr) {
                    bf16_t* o = o_lane + (long long)(f0 + u) * (16 * 512) + 64 * g;
                    *reinterpret_cast<bf16x8*>(o) = lo;
                    *reinterpret_cast<bf16x8*>(o + 8) = hi;
                }
            }
        }
    }
}

// Pos-conv input buffer, GROUP-MAJOR: xg[16 groups][B clips][T+128 frames][48 channels], 64 zero frames on
// each side of every clip (SamePad of the k=128 grouped conv).  With 192-B rows a group's K vector
// (128 taps x 48 channels) for output frame t is 6144 CONTIGUOUS floats starting at frame t, so the
// grouped conv is a plain GEMM with lda = 48 < K and every byte of every fetched line is used.
// grid: 16*B blocks of 256 threads; zeroes the two 64-frame pads of one (group, clip).
// Ragged batches (tpref != nullptr): blockIdx.x = group * B + clip, clip c has tpref[c+1]-tpref[c] frames and starts
// at padded frame ppref[c]; a group spans ppref[B] frames.
template <typename T_>
__global__ __launch_bounds__(256) void zero_pad_rows_kernel(T_* __restrict__ xg, int T, const int* __restrict__ tpref,
                                                            const int* __restrict__ ppref, int B) {
    constexpr int V = 48 * sizeof(T_) / 16;  // 16-byte vectors per 48-channel frame
    long long first = (long long)blockIdx.x * (T + 128);
    if (tpref) {
        const int g = blockIdx.x / B, b = blockIdx.x - g * B;
        T = tpref[b + 1] - tpref[b];
        first = (long long)g * ppref[B] + ppref[b];
    }
    float4* base = reinterpret_cast<float4*>(xg + first * 48);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* tail = base + (long long)(T + 64) * V;
    for (int i = threadIdx.x; i < 64 * V; i += 256) {
        base[i] = z;
        tail[i] = z;
    }
}

// bf16x3 pos-conv as a dense GEMM (nomad_hip.hip, forward_x3_run): kPosBlk consecutive output frames of one group
// become the N dimension.  Row (j, co) of the Toeplitz weight of group g is the filter of output channel co shifted
// by j taps over a window of 128 + kPosBlk - 1 input frames:
//   Wt[g][j * 48 + co][tap * 48 + ci] = pos_w[g][co][(tap - j) * 48 + ci]  for 0 <= tap - j < 128, else 0
// (rows kPosBlk * 48 .. 255 are zero padding up to the 256-wide tile).  pos_w: [16][64][6144] as in the fp32 path.
// grid: 16 * 256 blocks; output written as split planes.
constexpr int kPosBlk = 5;
constexpr int kPosKt = (128 + kPosBlk - 1) * 48;  // 6336
__global__ __launch_bounds__(256) void posconv_toeplitz_kernel(const float* __restrict__ pos_w, bf16s_t* __restrict__ out,
                                                               long long plane) {
    const int g = blockIdx.x >> 8, r = blockIdx.x & 255;
    const int j = r / 48, co = r - j * 48;
    bf16s_t* o = out + ((long long)g * 256 + r) * kPosKt;
    const float* w = pos_w + ((long long)g * 64 + co) * 6144;
    for (int k4 = threadIdx.x; k4 < kPosKt / 4; k4 += 256) {
        const int k = k4 * 4, tap = k / 48;  // 48 % 4 == 0: the four elements share their tap
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < kPosBlk && tap >= j && tap - j < 128) v = *reinterpret_cast<const float4*>(w + k - j * 48);
        store4p<bf16s_t>(o + k, plane, v);
    }
}
// bias tiled to the Toeplitz rows: bt[g][j * 48 + co] = pos_b[g * 48 + co], zero beyond kPosBlk * 48
__global__ __launch_bounds__(256) void posconv_toeplitz_bias_kernel(const float* __restrict__ pos_b, float* __restrict__ bt) {
    const int g = blockIdx.x, r = threadIdx.x;
    bt[g * 256 + r] = r < kPosBlk * 48 ? pos_b[g * 48 + r % 48] : 0.f;
}

}  // namespace nomad
