// Row-wise kernels: LayerNorm (SURVEY.md K5/K8 and the two post-LN norms of every encoder layer),
// the embedding head (K13: mean_t -> ReLU -> Linear(768,256) -> L2 normalise, nomad.py:228-230)
// and the NomadLoss L1 reduction (K15, nomad.py:267-282).  All are HBM-bound: one wave per row,
// 16-byte loads, statistics by wavefront (64-lane) shuffles.
#pragma once
#include <hip/hip_runtime.h>

#include "dtypes.hip.h"

namespace nomad {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over the S slices of a split-K GEMM's partial products for one float4: ((p[0] + p[stride4]) + p[2 stride4]) + ... in slice order,
// with the loads of the first kSliceBurst slices issued BEFORE the first add.  Round 6: written as `for (s = 1; s < S; ++s) a += p[s * stride4]`
// with a run-time S hipcc emits load - s_waitcnt vmcnt(0) - add per slice, and the epilogue kernels of configs[3] (1632 rows: one wave of
// work per CU, nothing else to hide a round trip behind) spent 6 dependent memory round trips per column chunk.  S is wave-uniform.
constexpr int kSliceBurst = 8;
__device__ __forceinline__ void load_slices4(float4 (&t)[kSliceBurst], const float4* __restrict__ p, long long stride4, int S) {
#pragma unroll
    for (int s = 0; s < kSliceBurst; ++s)
        if (s < S) t[s] = p[s * stride4];
}
__device__ __forceinline__ float4 add_slices4(const float4 (&t)[kSliceBurst], const float4* __restrict__ p, long long stride4, int S) {
    float4 a = t[0];
#pragma unroll
    for (int s = 1; s < kSliceBurst; ++s)
        if (s < S) {
            a.x += t[s].x; a.y += t[s].y; a.z += t[s].z; a.w += t[s].w;
        }
    for (int s = kSliceBurst; s < S; ++s) {
        const float4 b = p[s * stride4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    return a;
}
__device__ __forceinline__ float4 sum_slices4(const float4* __restrict__ p, long long stride4, int S) {
    float4 t[kSliceBurst];
    load_slices4(t, p, stride4, S);
    return add_slices4(t, p, stride4, S);
}

// One row of LayerNorm by one wave: lane l holds elements 4 (l + 64 i) .. + 3 of the row, i < VPT (N = 256 VPT).  layernorm_kernel below
// and the split-K epilogue that normalises its own rows (splitk_epilogue_ln_kernel, train.hip.h) run exactly this code - one summation
// order, so a row's result does not depend on which kernel normalised it.
template <int VPT, typename TOut>
__device__ __forceinline__ void ln_row_finish(const float4 (&v)[VPT], const float4 (&g)[VPT], const float4 (&bb)[VPT], TOut* __restrict__ o,
                                              long long out_plane, float4* __restrict__ o2, int lane) {
    constexpr int N = 256 * VPT;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / N) + 1e-5f);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        float4 r;
        r.x = (v[i].x - mean) * rstd * g[i].x + bb[i].x;
        r.y = (v[i].y - mean) * rstd * g[i].y + bb[i].y;
        r.z = (v[i].z - mean) * rstd * g[i].z + bb[i].z;
        r.w = (v[i].w - mean) * rstd * g[i].w + bb[i].w;
        store4p<TOut>(o + 4 * (lane + 64 * i), out_plane, r);
        if (o2) o2[lane + 64 * i] = r;
    }
}

// out[m][:] = (in[m][:] - mean) * rstd * gamma + beta; optional second copy out2 (layer_results).
// VPT float4 per lane: N = 256 * VPT (512 -> 2, 768 -> 3).  grid: ceil(M / (4 ROWS)) blocks of 256 threads; a wave normalises ROWS
// consecutive rows with one fetch of gamma / beta (round 6: at one bf16 row per wave the 6 KB of gamma and beta a wave pulls through the
// vector cache are twice the 3 KB of the row it reads and writes; the bf16 forward takes 4 rows per wave, all their loads in flight at once).
// A row's arithmetic is ln_row_finish whatever ROWS is.
template <int VPT, typename TIn = float, typename TOut = float, int ROWS = 1>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIn* __restrict__ in, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TOut* __restrict__ out,
                                                        float* __restrict__ out2, int M, long long in_plane = 0,
                                                        long long out_plane = 0) {
    constexpr int N = 256 * VPT;
    const int lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (m0 >= M) return;
    float4 v[ROWS][VPT], g[VPT], bb[VPT];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const TIn* row = in + (long long)min(m0 + r, M - 1) * N;
#pragma unroll
        for (int i = 0; i < VPT; ++i) v[r][i] = load4p<TIn>(row + 4 * (lane + 64 * i), in_plane);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        g[i] = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
        bb[i] = reinterpret_cast<const float4*>(beta)[lane + 64 * i];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int m = m0 + r;
        if (m < M)
            ln_row_finish<VPT, TOut>(v[r], g, bb, out + (long long)m * N, out_plane, out2 ? reinterpret_cast<float4*>(out2 + (long long)m * N) : nullptr, lane);
    }
}

// ---- head: mean over time -> ReLU -> Linear(768, 256) -> L2 normalise (nomad.py:228-230) ---------------------------
// Stage 1, head_pool_kernel: the time sum of a clip in chunks of kHeadChunk frames, one workgroup per (chunk, clip), so
// that a 30 s clip (T = 1499) is summed by 24 workgroups instead of one thread column walking 1499 rows (452 us for a
// batch of 32 such clips with the single-stage kernel, 40 x the time of reading the 74 MB once).
// Wave w of the workgroup takes frames w, w + 4, ... of the chunk, lane l the columns 4l + 256j (j = 0..2); the four
// waves are folded in fixed order.  Chunk j of clip b (rows r0 .. r0 + T - 1 of x) goes to slot r0 / 64 + b + j of
// `pool` - unique for packed clips of any lengths, at most M / 64 + B slots.  The summation order depends on T only:
// batch-invariant and deterministic.
constexpr int kHeadChunk = 64;
template <typename TIn = float>
__global__ __launch_bounds__(256) void head_pool_kernel(const TIn* __restrict__ x, int T, float* __restrict__ pool,
                                                        const int* __restrict__ tpref = nullptr) {
    __shared__ float4 part[4][3][64];
    const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long row0 = (long long)b * T;
    if (tpref) {  // ragged batch: this clip's own frame range
        row0 = tpref[b];
        T = tpref[b + 1] - tpref[b];
    }
    if (j * kHeadChunk >= T) return;
    const int t_end = min(T, (j + 1) * kHeadChunk);
    float4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = j * kHeadChunk + wave; t < t_end; t += 4) {
        const TIn* r = x + (row0 + t) * 768 + 4 * lane;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 v = load4<TIn>(r + 256 * q);
            acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) part[wave][q][lane] = acc[q];
    __syncthreads();
    if (wave == 0) {
        float* dst = pool + (row0 / kHeadChunk + b + j) * 768 + 4 * lane;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float4 a = part[0][q][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 o = part[w][q][lane];
                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            }
            *reinterpret_cast<float4*>(dst + 256 * q) = a;
        }
    }
}

// Stage 2.  grid: B blocks of 1024 threads.  pool (stage 1) -> emb [B][256].
// Round 6: 16 waves instead of 4 - with one 4-wave workgroup per clip a batch of 32 clips (configs[3]) kept 128 waves busy streaming the
// 786 KB of W each: 66 us per launch, all of it load latency.  Wave w now takes output rows 16 w .. 16 w + 15, four rows (48 loads) in
// flight per trip.  A row's dot product is still one wave's: lane l multiplies elements l + 64 i in the order i = 0 .. 11, then the
// wave sum - the same bits as before; the squared norm is summed by waves 0 .. 3 over e[64 w ..] and folded as before.
__global__ __launch_bounds__(1024) void head_kernel(const float* __restrict__ pool, int T, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ emb,
                                                    const int* __restrict__ tpref = nullptr) {
    __shared__ float pooled[768];
    __shared__ float e[256];
    __shared__ float wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    long long row0 = (long long)b * T;
    if (tpref) {
        row0 = tpref[b];
        T = tpref[b + 1] - tpref[b];
    }
    if (tid < 256) {
        const float* pb = pool + (row0 / kHeadChunk + b) * 768;
        const int nchunk = (T + kHeadChunk - 1) / kHeadChunk;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < nchunk; ++j) {  // chunks in order
            const float* r = pb + (long long)j * 768;
            s0 += r[tid];
            s1 += r[tid + 256];
            s2 += r[tid + 512];
        }
        const float inv = 1.0f / (float)T;
        pooled[tid] = fmaxf(s0 * inv, 0.f);
        pooled[tid + 256] = fmaxf(s1 * inv, 0.f);
        pooled[tid + 512] = fmaxf(s2 * inv, 0.f);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    float p[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = pooled[lane + 64 * i];
#pragma unroll 1
    for (int o0 = wave * 16; o0 < wave * 16 + 16; o0 += 4) {
        float wv[4][12];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 12; ++i) wv[u][i] = w[(long long)(o0 + u) * 768 + lane + 64 * i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) d = fmaf(wv[u][i], p[i], d);
            d = wave_sum(d);
            if (lane == 0) e[o0 + u] = d + bias[o0 + u];
        }
    }
    __syncthreads();
    if (tid < 256) {
        const float v = e[tid];
        const float ss = wave_sum(v * v);
        if (lane == 0) wsum[wave] = ss;
    }
    __syncthreads();
    if (tid < 256) {
        const float nrm = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
        emb[(long long)b * 256 + tid] = e[tid] / fmaxf(nrm, 1e-12f);  // F.normalize eps
    }
}

// NomadLoss.  Stage 1: per-block fp64 partial sums of |a-b| over the 12 layer tensors (n_layer
// float4s in total) and over the embeddings; stage 2: one block folds the partials in fixed order.
constexpr int kL1Blocks = 1024;
__global__ __launch_bounds__(256) void l1_partial_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                         long long n4, const float* __restrict__ ea,
                                                         const float* __restrict__ eb, int ne,
                                                         double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)kL1Blocks * 256) {
        const float4 x = a[i], y = b[i];
        s += (double)((fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w)));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0) {  // embedding term, one block
        __syncthreads();
        double se = 0.0;
        for (int i = threadIdx.x; i < ne; i += 256) se += (double)fabsf(ea[i] - eb[i]);
        se = wave_sum(se);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
        __syncthreads();
        if (threadIdx.x == 0) partial[kL1Blocks] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

__global__ __launch_bounds__(256) void l1_final_kernel(const double* __restrict__ partial, double layer_elems,
                                                       double emb_elems, float* __restrict__ loss) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < kL1Blocks; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double layers = (red[0] + red[1]) + (red[2] + red[3]);
        // 12 terms each a mean over layer_elems elements, plus the embedding term
        loss[0] = (float)(layers / layer_elems + partial[kL1Blocks] / emb_elems);
    }
}

}  // namespace nomad
