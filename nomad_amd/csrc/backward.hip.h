// Backward (d loss / d waveform) kernels for Nomad.forward() used as a training loss
// (/root/reference/src/nomad_audio/nomad.py:142-146 + autograd).
// Weights are frozen (the user's purpose is the gradient w.r.t. `estimate`), so only dX is propagated:
// every dense contraction of the backward is the SAME fp32 MFMA GEMM kernel as the forward, fed with
// transposed weight copies; this file holds what is not a GEMM:
//   LayerNorm backward, GELU' multiplies (fused into GEMM epilogues or as row kernels), head / L1-loss
//   backward, GroupNorm + conv0 backward, and the one-time weight transposes.  The attention backward
//   (MFMA, flash style, no atomics) is in attention_bwd.hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include "dropout.hip.h"
#include "gemm_f32.hip.h"
#include "rowops.hip.h"

namespace nomad {

// d/du gelu_erf(u) = Phi(u) + u * phi(u)
__device__ __forceinline__ float dgelu_erf(float u) {
    const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
    return cdf + u * pdf;
}

// ---- one-time weight transposes (nomad_enable_backward) ---------------------------------------------
// out[c * ld_out + r] = in[r * ld_in + c], r < R, c < C.  grid: (ceil(C/32), ceil(R/32)), block (32, 8).
__global__ void transpose_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int ld_out, int R,
                                 int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < C) ? in[(long long)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < C) out[(long long)c * ld_out + r] = tile[threadIdx.x][i];
    }
}

// Pos-conv backward weights: wb[g][ci (64 rows, 48 valid)][tp*48 + n] = w[g][n][(127 - tp)*48 + ci]
// (w = forward repack [16][64][6144]).  grid: 16*64 blocks of 256 threads.
__global__ void posconv_bwd_weight_kernel(const float* __restrict__ w, float* __restrict__ wb) {
    const int g = blockIdx.x >> 6, ci = blockIdx.x & 63;
    float* dst = wb + ((long long)g * 64 + ci) * 6144;
    for (int k = threadIdx.x; k < 6144; k += 256) {
        const int tp = k / 48, n = k - tp * 48;
        dst[k] = ci < 48 ? w[((long long)g * 64 + n) * 6144 + (127 - tp) * 48 + ci] : 0.f;
    }
}

// ---- LayerNorm backward -----------------------------------------------------------------------------
// x = LN input, g (+ g2) = d loss / d LN output; dx = rstd * (gg - mean(gg) - xhat * mean(gg * xhat)),
// gg = (g + g2) * gamma.  One wave per row; N = 256 * VPT.
// SPLITK (round 6): g is not read but formed here as the epilogue of a split-K GEMM - g[m][:] = sum_s partial[s][m][:] + R[m][:], slices in
// order, then the residual: splitk_epilogue_kernel's arithmetic, element for element - so that the dX GEMM in front of a LayerNorm
// backward (fc1^T before LN1, qkv^T before the previous layer's LN2 / the encoder LN) needs no epilogue launch of its own and g never
// goes to memory.  The same bits as the two launches (tests/test_gpu_backward.py); dx may alias R (a wave reads its whole row first).
template <int VPT, bool SPLITK = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ g2,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            int M, int S = 0, const float* __restrict__ R = nullptr) {
    constexpr int N = 256 * VPT;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)m * N);
    const float4* gr = reinterpret_cast<const float4*>(g + (long long)m * N);   // (SPLITK: slice 0 of the partial products)
    const float4* rr = (SPLITK && R) ? reinterpret_cast<const float4*>(R + (long long)m * N) : nullptr;
    const float4* g2r = g2 ? reinterpret_cast<const float4*>(g2 + (long long)m * N) : nullptr;
    const float4* gm4 = reinterpret_cast<const float4*>(gamma);
    float xv[VPT][4], gv[VPT][4];
    float s = 0.f;
    // every load of the row first (round 6: behind run-time branches and a run-time slice loop they went out one memory round trip at a time)
    float4 t[SPLITK ? VPT : 1][kSliceBurst];   // (SPLITK: the slices of the whole row in flight at once, rowops.hip.h sum_slices4)
    float4 xa[VPT], ga[VPT], ra[VPT], g2a[VPT], gma[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (SPLITK) load_slices4(t[i], gr + lane + 64 * i, (long long)M * (N / 4), S);
        else ga[i] = gr[lane + 64 * i];
        xa[i] = xr[lane + 64 * i];
        gma[i] = gm4[lane + 64 * i];
        if (rr) ra[i] = rr[lane + 64 * i];
        if (g2r) g2a[i] = g2r[lane + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const float4 a = xa[i];
        float4 b;
        if (!SPLITK) b = ga[i];
        if (SPLITK) {
            b = add_slices4(t[i], gr + lane + 64 * i, (long long)M * (N / 4), S);
            if (rr) {
                const float4 c = ra[i];
                b.x += c.x; b.y += c.y; b.z += c.z; b.w += c.w;
            }
        }
        if (g2r) {
            const float4 c = g2a[i];
            b.x += c.x; b.y += c.y; b.z += c.z; b.w += c.w;
        }
        const float4 gm = gma[i];
        xv[i][0] = a.x; xv[i][1] = a.y; xv[i][2] = a.z; xv[i][3] = a.w;
        gv[i][0] = b.x * gm.x; gv[i][1] = b.y * gm.y; gv[i][2] = b.z * gm.z; gv[i][3] = b.w * gm.w;
        s += (a.x + a.y) + (a.z + a.w);
    }
    const float mean = wave_sum(s) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[i][j] -= mean;
            q += xv[i][j] * xv[i][j];
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / N) + 1e-5f);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[i][j] *= rstd;  // xhat
            sg += gv[i][j];
            sgx += gv[i][j] * xv[i][j];
        }
    const float mg = wave_sum(sg) * (1.0f / N), mgx = wave_sum(sgx) * (1.0f / N);
    float4* o = reinterpret_cast<float4*>(dx + (long long)m * N);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        float4 r;
        r.x = rstd * (gv[i][0] - mg - xv[i][0] * mgx);
        r.y = rstd * (gv[i][1] - mg - xv[i][1] * mgx);
        r.z = rstd * (gv[i][2] - mg - xv[i][2] * mgx);
        r.w = rstd * (gv[i][3] - mg - xv[i][3] * mgx);
        o[lane + 64 * i] = r;
    }
}

// out[map(m)][c] = g[m][c] * gelu'(u[m][c]) for rows of 512 floats: conv6 gradient into the padded dU layout.
__global__ __launch_bounds__(128) void dgelu_rows512_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                            float* __restrict__ out, RowMap omap, int M, float scale) {
    const int m = blockIdx.x;
    if (m >= M) return;
    float4 a = reinterpret_cast<const float4*>(g + (long long)m * 512)[threadIdx.x];
    const float4 b = reinterpret_cast<const float4*>(u + (long long)m * 512)[threadIdx.x];
    a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;  // GradMultiply's backward: grad * scale, then GELU'
    float4 r;
    r.x = a.x * dgelu_erf(b.x); r.y = a.y * dgelu_erf(b.y); r.z = a.z * dgelu_erf(b.z); r.w = a.w * dgelu_erf(b.w);
    reinterpret_cast<float4*>(out + row_addr(omap, m))[threadIdx.x] = r;
}

// Rows [a0, a1) and [b0, b1) of every clip's [rows][512] block set to zero: the pad rows (and the frames no output frame reaches) of a
// padded dU buffer of the conv stack's backward - the GEMMs write every other row, so the buffer as a whole needs no memset (round 6:
// the full memsets were 8 launches / 89 us of a configs[3] step, 215 MB for conv0's output gradient alone).  grid: B blocks of 256.
__global__ __launch_bounds__(256) void zero_rows512_kernel(float* __restrict__ base, long long clip_stride, int a0, int a1, int b0, int b1) {
    float* p = base + (long long)blockIdx.x * clip_stride;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < (a1 - a0) * 128; i += 256) reinterpret_cast<float4*>(p + (long long)a0 * 512)[i] = z;
    for (int i = threadIdx.x; i < (b1 - b0) * 128; i += 256) reinterpret_cast<float4*>(p + (long long)b0 * 512)[i] = z;
}

// Pos-conv: dug[grp][clip][64 + t][48] = dy0[m][grp*48 + c] * gelu'(upc[m][grp*48 + c])   (group-major, padded)
// grid: M blocks of 192 threads (one float4 each).
__global__ __launch_bounds__(192) void dgelu_to_groups_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                              float* __restrict__ dug, int T, long long grp_stride) {
    const int m = blockIdx.x, b = m / T, t = m - b * T;
    const float4 a = reinterpret_cast<const float4*>(g + (long long)m * 768)[threadIdx.x];
    const float4 c = reinterpret_cast<const float4*>(u + (long long)m * 768)[threadIdx.x];
    float4 r;
    r.x = a.x * dgelu_erf(c.x); r.y = a.y * dgelu_erf(c.y); r.z = a.z * dgelu_erf(c.z); r.w = a.w * dgelu_erf(c.w);
    const int col = threadIdx.x * 4, grp = col / 48, cc = col - grp * 48;
    float* dst = dug + grp * grp_stride + ((long long)b * (T + 128) + 64 + t) * 48 + cc;
    *reinterpret_cast<float4*>(dst) = r;
}

// ---- head backward ----------------------------------------------------------------------------------
// e = normalize(W relu(mean_t x) + b).  Given de -> gx[b][t][:] = relu'(mean) * (W^T dz) / T for every t.
// grid: B blocks of 1024 threads (round 6: 16 waves - with 4 the kernel took 55 us for the 32 clips of configs[3], all of it the latency
// of streaming W twice through 128 waves).  The time sum keeps the forward's order (waves 0 .. 3 take frames w, w + 4, ..: for T <= 64 the
// same bits as head_pool_kernel, so the ReLU mask is the forward's); z = W pooled + b as in head_kernel (a row's dot product is one
// wave's, same order); W^T dz in four fixed slices of 64 rows, folded in slice order.
__global__ __launch_bounds__(1024) void head_bwd_kernel(const float* __restrict__ x, int T, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ de,
                                                        float* __restrict__ gx, float* __restrict__ pooled_out = nullptr,
                                                        float* __restrict__ dz_out = nullptr) {
    __shared__ float pooled[768], mask[768], z[256], dz[256], red[4], red2[4];
    __shared__ __attribute__((aligned(16))) float psum[4][768];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (long long)b * T * 768;
    if (wave < 4) {   // time sum: wave w takes frames w, w+4, ... (16-byte loads), the four partial sums are combined in fixed order
        float4 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = wave; t < T; t += 4) {
            const float4* r = reinterpret_cast<const float4*>(xb + (long long)t * 768);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float4 v = r[lane + 64 * i];
                acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) reinterpret_cast<float4*>(psum[wave])[lane + 64 * i] = acc[i];
    }
    __syncthreads();
    if (tid < 256) {
        const float s0 = (psum[0][tid] + psum[1][tid]) + (psum[2][tid] + psum[3][tid]);
        const float s1 = (psum[0][tid + 256] + psum[1][tid + 256]) + (psum[2][tid + 256] + psum[3][tid + 256]);
        const float s2 = (psum[0][tid + 512] + psum[1][tid + 512]) + (psum[2][tid + 512] + psum[3][tid + 512]);
        const float inv = 1.0f / (float)T;
        const float m0 = s0 * inv, m1 = s1 * inv, m2 = s2 * inv;
        pooled[tid] = fmaxf(m0, 0.f); pooled[tid + 256] = fmaxf(m1, 0.f); pooled[tid + 512] = fmaxf(m2, 0.f);
        mask[tid] = m0 > 0.f ? inv : 0.f; mask[tid + 256] = m1 > 0.f ? inv : 0.f; mask[tid + 512] = m2 > 0.f ? inv : 0.f;
    }
    __syncthreads();
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
            if (lane == 0) z[o0 + u] = d + bias[o0 + u];
        }
    }
    __syncthreads();
    float zv = 0.f, dev = 0.f;
    if (tid < 256) {
        zv = z[tid];
        dev = de[(long long)b * 256 + tid];
        const float ss = wave_sum(zv * zv);
        if (lane == 0) red[wave] = ss;
    }
    __syncthreads();
    const float nrm = fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f);
    const float ev = zv / nrm;
    if (tid < 256) {
        const float dot = wave_sum(ev * dev);
        if (lane == 0) red2[wave] = dot;
    }
    __syncthreads();
    if (tid < 256) {
        const float edot = (red2[0] + red2[1]) + (red2[2] + red2[3]);
        dz[tid] = (dev - ev * edot) / nrm;
        if (dz_out) {  // what the head's own parameter gradients need (train.hip.h: head_param_grad_kernel)
            dz_out[(long long)b * 256 + tid] = dz[tid];
            pooled_out[(long long)b * 768 + tid] = pooled[tid];
            pooled_out[(long long)b * 768 + tid + 256] = pooled[tid + 256];
            pooled_out[(long long)b * 768 + tid + 512] = pooled[tid + 512];
        }
    }
    __syncthreads();
    // dp[c] = sum_o W[o][c] dz[o]: slice q = tid / 256 takes o = 64 q .. 64 q + 63 for the columns c = tid % 256 + 256 j (coalesced over c);
    // psum is free again and carries the four slices' sums
    {
        const int q = tid >> 8, c0 = tid & 255;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll 8
        for (int o = 64 * q; o < 64 * q + 64; ++o) {
            const float* wr = w + (long long)o * 768;
            const float dzo = dz[o];
            d0 = fmaf(wr[c0], dzo, d0); d1 = fmaf(wr[c0 + 256], dzo, d1); d2 = fmaf(wr[c0 + 512], dzo, d2);
        }
        psum[q][c0] = d0; psum[q][c0 + 256] = d1; psum[q][c0 + 512] = d2;
    }
    __syncthreads();
    if (tid < 768) {
        const float d = (psum[0][tid] + psum[1][tid]) + (psum[2][tid] + psum[3][tid]);
        pooled[tid] = d * mask[tid];     // (pooled is dead: it carries the per-column gradient to the broadcast below)
    }
    __syncthreads();
    float4 gv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) gv[i] = reinterpret_cast<const float4*>(pooled)[lane + 64 * i];
    float* gb = gx + (long long)b * T * 768;
    for (int t = wave; t < T; t += 16) {
        float4* r = reinterpret_cast<float4*>(gb + (long long)t * 768);
#pragma unroll
        for (int i = 0; i < 3; ++i) r[lane + 64 * i] = gv[i];
    }
}

// ---- L1 loss backward: d/da sum_i mean|a_i - b_i| = sign(a - b) / numel_i, times the upstream scalar ------
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                     long long n4, float inv_numel, const float* __restrict__ upstream,
                                                     float4* __restrict__ out) {
    const float sc = inv_numel * upstream[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        float4 r;
        r.x = x.x > y.x ? sc : (x.x < y.x ? -sc : 0.f);
        r.y = x.y > y.y ? sc : (x.y < y.y ? -sc : 0.f);
        r.z = x.z > y.z ? sc : (x.z < y.z ? -sc : 0.f);
        r.w = x.w > y.w ? sc : (x.w < y.w ? -sc : 0.f);
        out[i] = r;
    }
}

// (attention backward: attention_bwd.hip.h)

// ---- conv0 + GroupNorm backward ---------------------------------------------------------------------
// Forward: y[t,c] = sum_j w[c,j] x[5t+j];  z = (y - mean_c) * rstd_c * gamma_c + beta_c;  out = gelu(z).
// Given G = d loss / d out (time-major [B][L0][512]):
//   dz = G * gelu'(z);  s1[c] = sum_t dz,  s2[c] = sum_t dz * yhat,  yhat = (y - mean) * rstd
//   dy = gamma * rstd * (dz - s1/L0 - yhat * s2/L0);  dx[5t+j] += sum_c dy[t,c] w[c,j]
// Pass 1 (gn_bwd_stats_kernel): per-(clip, frame chunk) partial s1, s2 -> fixed-order fold in pass 2's prologue.
// Pass 2 (conv0_bwd_kernel): dy, then the 512-channel contraction per frame (tiled through LDS, registers per frame).
constexpr int kGnChunk = 256;  // frames per block of the parameter-gradient pass (train.hip.h)
// Round 6: the statistics pass takes 64 frames per block (256 before: 13 x 32 = 416 workgroups for configs[3], 1.6 per CU, each a serial
// loop of 256 frames x ~100 vector instructions - 156 us for a pass whose 215 MB take 45 us to read); its per-chunk sums are folded ONCE per
// clip by gn_bwd_fold_kernel (every block of pass 2 used to fold all of them itself), which also lays out the 16 per-channel constants of
// pass 2 as one float4-loadable table.
constexpr int kGnStatsChunk = 64;

__device__ __forceinline__ void conv0_frame(const float* xs, int t, const float (&w)[2][10], const float (&sc)[2],
                                            const float (&sh)[2], const float (&mean)[2], const float (&rstd)[2],
                                            float (&z)[2], float (&yhat)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) y = fmaf(w[q][j], xs[5 * t + j], y);
        z[q] = fmaf(y, sc[q], sh[q]);
        yhat[q] = (y - mean[q]) * rstd[q];
    }
}

// grid: (chunks, B), 256 threads: thread owns channels c = tid and tid + 256.  partial[b][chunk][2][512]
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const float* __restrict__ wav, int n_samples, int L0,
                                                           const float* __restrict__ w0, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ gmean,
                                                           const float* __restrict__ grstd, const float* __restrict__ G,
                                                           float* __restrict__ partial) {
    __shared__ float xs[kGnStatsChunk * 5 + 8];
    const int b = blockIdx.y, t0 = blockIdx.x * kGnStatsChunk, nfr = min(kGnStatsChunk, L0 - t0), tid = threadIdx.x;
    const float* x = wav + (long long)b * n_samples + 5 * t0;
    for (int i = tid; i < 5 * nfr + 5; i += 256) xs[i] = x[i];
    float w[2][10], sc[2], sh[2], mean[2], rstd[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = tid + 256 * q;
        sc[q] = scale[b * 512 + c]; sh[q] = shift[b * 512 + c];
        mean[q] = gmean[b * 512 + c]; rstd[q] = grstd[b * 512 + c];
#pragma unroll
        for (int j = 0; j < 10; ++j) w[q][j] = w0[c * 10 + j];
    }
    __syncthreads();
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const float* g = G + ((long long)b * L0 + t0) * 512;
    for (int tb = 0; tb < nfr; tb += 4) {   // four frames' loads in flight; the sums keep the frame order
        float gv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) gv[u][q] = tb + u < nfr ? g[(long long)(tb + u) * 512 + tid + 256 * q] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (tb + u >= nfr) break;
            float z[2], yh[2];
            conv0_frame(xs, tb + u, w, sc, sh, mean, rstd, z, yh);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float dz = gv[u][q] * dgelu_erf(z[q]);
                s1[q] += dz;
                s2[q] = fmaf(dz, yh[q], s2[q]);
            }
        }
    }
    float* p = partial + ((long long)b * gridDim.x + blockIdx.x) * 1024;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        p[tid + 256 * q] = s1[q];
        p[512 + tid + 256 * q] = s2[q];
    }
}

// Fold.  grid: B blocks of 512 threads (one per channel): the chunk sums of pass 1 in chunk order -> fold[b][0..511] = s1, [512..1023] = s2
// (raw sums: the parameter-gradient kernels of train.hip.h read them too), and the table pass 2 stages with one float4 per thread:
// cwtab[b][c][0..9] = w0[c][:], [10] = scale, [11] = shift, [12] = mean, [13] = rstd, [14] = s1 / L0, [15] = s2 / L0.
__global__ __launch_bounds__(512) void gn_bwd_fold_kernel(const float* __restrict__ partial, int nchunks, int L0,
                                                          const float* __restrict__ w0, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ gmean,
                                                          const float* __restrict__ grstd, float* __restrict__ fold,
                                                          float* __restrict__ cwtab) {
    const int b = blockIdx.x, c = threadIdx.x;
    float a1 = 0.f, a2 = 0.f;
    for (int k = 0; k < nchunks; ++k) {
        const float* p = partial + ((long long)b * nchunks + k) * 1024;
        a1 += p[c];
        a2 += p[512 + c];
    }
    fold[(long long)b * 1024 + c] = a1;
    fold[(long long)b * 1024 + 512 + c] = a2;
    float4* t = reinterpret_cast<float4*>(cwtab + ((long long)b * 512 + c) * 16);
    const float* w = w0 + c * 10;
    t[0] = make_float4(w[0], w[1], w[2], w[3]);
    t[1] = make_float4(w[4], w[5], w[6], w[7]);
    t[2] = make_float4(w[8], w[9], scale[b * 512 + c], shift[b * 512 + c]);
    t[3] = make_float4(gmean[b * 512 + c], grstd[b * 512 + c], a1 / (float)L0, a2 / (float)L0);
}

// Pass 2.  grid: (ceil(L0 / 64), B), 256 threads = 64 frames x 4 channel groups; a block walks the 512 channels in chunks of 64:
// the chunk's slice of G goes through LDS (coalesced 256-byte rows in, conflict-free column reads out), its per-channel
// constants sit in LDS as broadcasts, and thread (frame f, group g) accumulates its 10 tap contributions over channels
// g, g + 4, ... in registers.  The four groups and the two frames that touch a sample are then folded in fixed order in
// LDS; only that sum goes to memory.  dwav must be zero-initialised: a sample on a block boundary receives one more
// contribution from the neighbouring block, and two float atomics commute.  (The version this replaces made 10 wave
// reductions, two barriers and an atomic per FRAME: 687 us for config C4's 32 x 16384 samples; this one ~10 x less.)
// Round 6: the next chunk's G slice and constants (one float4 of cwtab per thread) are in flight while the current chunk is
// multiplied - before, every chunk began with 16 dependent scalar loads per thread and a branchy gather of the constants.
constexpr int kC0Frames = 64;
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const float* __restrict__ wav, int n_samples, int L0,
                                                        const float* __restrict__ cwtab, const float* __restrict__ G,
                                                        float* __restrict__ dwav) {
    __shared__ float xs[kC0Frames * 5 + 8];
    __shared__ float gt[kC0Frames][65];        // G chunk: [frame][channel of the chunk]
    __shared__ __attribute__((aligned(16))) float cw[64][16];   // per channel of the chunk: w[0..9], sc, sh, mean, rstd, m1, m2
    __shared__ float red[4][kC0Frames][10];
    const int b = blockIdx.y, t0 = blockIdx.x * kC0Frames, nfr = min(kC0Frames, L0 - t0), tid = threadIdx.x;
    const int f = tid & 63, g = tid >> 6;
    const float* x = wav + (long long)b * n_samples + 5 * t0;
    for (int i = tid; i < 5 * nfr + 5; i += 256) xs[i] = x[i];
    const float* gbase = G + ((long long)b * L0 + t0) * 512;
    const float4* ctab = reinterpret_cast<const float4*>(cwtab + (long long)b * 512 * 16);
    // staging map: G slice as float4 - thread -> (frame tid / 16 + 16 i, channels 4 (tid % 16) .. + 3); constants: float4 tid of the chunk
    float4 greg[4], creg;
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fr = (tid >> 4) + 16 * i;
            greg[i] = fr < nfr ? *reinterpret_cast<const float4*>(gbase + (long long)fr * 512 + c0 + 4 * (tid & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        creg = ctab[c0 * 4 + tid];
    };
    fetch(0);
    __syncthreads();
    float xr[10], contrib[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        xr[j] = f < nfr ? xs[5 * f + j] : 0.f;
        contrib[j] = 0.f;
    }
    for (int c0 = 0; c0 < 512; c0 += 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fr = (tid >> 4) + 16 * i, cc = 4 * (tid & 15);
            gt[fr][cc] = greg[i].x; gt[fr][cc + 1] = greg[i].y; gt[fr][cc + 2] = greg[i].z; gt[fr][cc + 3] = greg[i].w;
        }
        reinterpret_cast<float4*>(&cw[0][0])[tid] = creg;
        __syncthreads();
        if (c0 + 64 < 512) fetch(c0 + 64);
        if (f < nfr) {
            for (int cc = g; cc < 64; cc += 4) {
                const float* q = cw[cc];
                float y = 0.f;
#pragma unroll
                for (int j = 0; j < 10; ++j) y = fmaf(q[j], xr[j], y);
                const float z = fmaf(y, q[10], q[11]);
                const float yhat = (y - q[12]) * q[13];
                const float dz = gt[f][cc] * dgelu_erf(z);
                const float dy = q[10] * (dz - q[14] - yhat * q[15]);  // sc = gamma * rstd
#pragma unroll
                for (int j = 0; j < 10; ++j) contrib[j] = fmaf(dy, q[j], contrib[j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) red[g][f][j] = f < nfr ? contrib[j] : 0.f;
    __syncthreads();
    float* dx = dwav + (long long)b * n_samples + 5 * t0;
    for (int s = tid; s < 5 * nfr + 5; s += 256) {   // sample s: tap s % 5 of frame s / 5 and tap s % 5 + 5 of the frame before
        const int fa = s / 5, j = s - 5 * fa;
        float v = 0.f;
        if (fa < nfr) v += (red[0][fa][j] + red[1][fa][j]) + (red[2][fa][j] + red[3][fa][j]);
        if (fa >= 1) v += (red[0][fa - 1][j + 5] + red[1][fa - 1][j + 5]) + (red[2][fa - 1][j + 5] + red[3][fa - 1][j + 5]);
        atomicAdd(&dx[s], v);
    }
}

}  // namespace nomad
