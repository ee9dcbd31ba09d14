// Parameter-gradient and optimiser kernels for the triplet fine-tuning step
// (/root/reference/src/training/train_triplet.py:112-133: A/P/N forward, nn.TripletMarginLoss, loss.backward(),
// Adam step).  The reference trains with freeze_convnet: True
// (src/config/train_triplet.yaml), so the trainable set is everything after the conv feature extractor.
//
// Every weight gradient dW[out][in] = sum_m dY[m][out] X[m][in] is the SAME fp32 MFMA GEMM kernel as the forward:
// the two operands are transposed (zero-padded along the contraction) by transpose_pad_kernel and the long
// contraction is split across workgroups as "groups" whose partial products are folded in fixed order by
// splitk_reduce_kernel - deterministic, no atomics.  This file holds the rest: bias / LayerNorm / head / pos-conv
// (weight-norm) parameter gradients, the triplet loss, dropout, and Adam.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "backward.hip.h"
#include "dropout.hip.h"
#include "gemm_f32.hip.h"
#include "rowops.hip.h"

namespace nomad {

// out[c][m] = f(in[m][c]) for m < M, 0 for M <= m < ld_out (ld_out % 64 == 0, C % 4 == 0).  ACT: 0 identity, 1 GELU
// (the fc2 input is recomputed from the saved pre-activation).  64 x 64 tiles, 16-byte loads and stores.
// grid: (ld_out/64, ceil(C/64)), 256 threads.
template <int ACT>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                            int ld_out, int M, int C) {
    __shared__ float tile[64][65];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
    const int r = tid >> 4, q = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r + 16 * i, c = c0 + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && c < C) v = *reinterpret_cast<const float4*>(in + (long long)m * ld_in + c);
        if (ACT == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        tile[r + 16 * i][q] = v.x; tile[r + 16 * i][q + 1] = v.y; tile[r + 16 * i][q + 2] = v.z; tile[r + 16 * i][q + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + r + 16 * i;
        if (c < C) {
            float4 v;
            v.x = tile[q][r + 16 * i]; v.y = tile[q + 1][r + 16 * i]; v.z = tile[q + 2][r + 16 * i]; v.w = tile[q + 3][r + 16 * i];
            *reinterpret_cast<float4*>(out + (long long)c * ld_out + m0 + q) = v;
        }
    }
}

// out[i] += scale * sum_s partial[s * stride4 + i], s in fixed order, i < count4 (float4 units).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float4* __restrict__ partial, int S, long long stride4,
                                                            long long count4, float4* __restrict__ out, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    float4 o = out[i];
    const float4 a = sum_slices4(partial + i, stride4, S);
    o.x = fmaf(scale, a.x, o.x); o.y = fmaf(scale, a.y, o.y); o.z = fmaf(scale, a.z, o.z); o.w = fmaf(scale, a.w, o.w);
    out[i] = o;
}

// Epilogue of a split-K dense GEMM (run_gemm_splitk): C[m][n] = act(sum_s partial[s][m][n] + bias[n]) + R[m][n], s in
// fixed order, plain row-major C / R with ld = N.  Float4 units: count4 = M * N / 4, n4 = N / 4.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float4* __restrict__ partial, int S, long long count4, int n4,
                                                              const float4* __restrict__ bias, const float4* __restrict__ R,
                                                              float4* __restrict__ C, int gelu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    // (bias and residual fetched beside the slices, not behind them)
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), rv = bv;
    if (bias) bv = bias[i % n4];
    if (R) rv = R[i];
    float4 a = sum_slices4(partial + i, count4, S);
    if (bias) {
        a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
    }
    if (gelu) {
        a.x = gelu_erf(a.x); a.y = gelu_erf(a.y); a.z = gelu_erf(a.z); a.w = gelu_erf(a.w);
    }
    if (R) {
        a.x += rv.x; a.y += rv.y; a.z += rv.z; a.w += rv.w;
    }
    C[i] = a;
}

// The same epilogue for the N = 768 residual GEMMs of a transformer layer (out_proj, fc2) TOGETHER with the LayerNorm that follows
// them: y[m][:] = sum_s partial[s][m][:] + bias + R[m][:] (stored: the backward needs it), x[m][:] = LayerNorm(y[m][:]) (+ the copy x2
// for layer_results).  One wave per row; every element's sum is the stand-alone epilogue's (slice order, then bias, then residual) and
// the row code is the stand-alone LayerNorm's (ln_row_finish): the same bits as splitk_epilogue_kernel + layernorm_kernel<3>, one
// launch and one pass over y less.  configs[3] runs 48 such pairs per step.  grid: ceil(M / 4) blocks of 256 threads.
__global__ __launch_bounds__(256) void splitk_epilogue_ln_kernel(const float* __restrict__ partial, int S, int M, const float* __restrict__ bias,
                                                                 const float* __restrict__ R, float* __restrict__ y,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ x, float* __restrict__ x2) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const long long row = (long long)m * 768, plane = (long long)M * 768;
    float4 v[3], g[3], bb[3], bv[3], rv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // everything that does not depend on the slices first: one memory round trip for the row, not 2 + S per chunk
        const int c = 4 * (lane + 64 * i);
        bv[i] = rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv[i] = *reinterpret_cast<const float4*>(bias + c);
        if (R) rv[i] = *reinterpret_cast<const float4*>(R + row + c);
        g[i] = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
        bb[i] = reinterpret_cast<const float4*>(beta)[lane + 64 * i];
    }
    float4 t[3][kSliceBurst];
#pragma unroll
    for (int i = 0; i < 3; ++i) load_slices4(t[i], reinterpret_cast<const float4*>(partial + row + 4 * (lane + 64 * i)), plane / 4, S);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = 4 * (lane + 64 * i);
        float4 a = add_slices4(t[i], reinterpret_cast<const float4*>(partial + row + c), plane / 4, S);
        if (bias) {
            a.x += bv[i].x; a.y += bv[i].y; a.z += bv[i].z; a.w += bv[i].w;
        }
        if (R) {
            a.x += rv[i].x; a.y += rv[i].y; a.z += rv[i].z; a.w += rv[i].w;
        }
        *reinterpret_cast<float4*>(y + row + c) = a;
        v[i] = a;
    }
    ln_row_finish<3, float>(v, g, bb, x + row, 0, x2 ? reinterpret_cast<float4*>(x2 + row) : nullptr, lane);
}

// Epilogue of the K-split grouped pos-conv GEMM (loss path, run_posconv_splitk): per row m and column c = 48 g + n
//   v = sum_s partial[s][m][c] (+ bias[c]);  Upre[m][c] = v (training forward);  v = gelu(v) (forward);  C[m][c] = v + R(m, g, n)
// R through its row map (the forward's residual is the padded per-group buffer: rmap + g * r_goff; the backward's a plain matrix).
__global__ __launch_bounds__(192) void posconv_splitk_epilogue_kernel(const float* __restrict__ partial, int S, int M, const float* __restrict__ bias,
                                                                      const float* __restrict__ R, RowMap rmap, long long r_goff,
                                                                      float* __restrict__ Upre, float* __restrict__ C, int gelu) {
    const int m = blockIdx.x, c4 = threadIdx.x;          // 192 float4 per row
    const long long i = (long long)m * 768 + c4 * 4;
    float4 a = sum_slices4(reinterpret_cast<const float4*>(partial + i), (long long)M * 192, S);
    if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + c4 * 4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (Upre) *reinterpret_cast<float4*>(Upre + i) = a;
    if (gelu) {
        a.x = gelu_erf(a.x); a.y = gelu_erf(a.y); a.z = gelu_erf(a.z); a.w = gelu_erf(a.w);
    }
    if (R) {
        const int g = c4 / 12, n = (c4 - g * 12) * 4;
        const float4 r = *reinterpret_cast<const float4*>(R + g * r_goff + row_addr(rmap, m) + n);
        a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    *reinterpret_cast<float4*>(C + i) = a;
}

// Bias gradient from the transposed dY: out[r] += scale * sum_m in[r][m].  One wave per row; ld % 4 == 0.
__global__ __launch_bounds__(256) void rowsum_acc_kernel(const float* __restrict__ in, int ld, int rows,
                                                         float* __restrict__ out, float scale) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float4* p = reinterpret_cast<const float4*>(in + (long long)r * ld);
    float s = 0.f;
    for (int i = lane; i < ld / 4; i += 64) {
        const float4 v = p[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = fmaf(scale, s, out[r]);
}

// LayerNorm parameter gradients: dgamma[n] = sum_m g[m][n] * xhat[m][n], dbeta[n] = sum_m g[m][n]  (g = g1 + g2).
// Stage 1: a block of 4 waves walks kLnRows rows (wave w takes rows w, w+4, ...), partial[blk][2][N].
constexpr int kLnRows = 16;
template <int VPT>
__global__ __launch_bounds__(256) void ln_param_partial_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                               const float* __restrict__ g2, float* __restrict__ partial,
                                                               int M) {
    constexpr int N = 256 * VPT;
    __shared__ float red[4][2][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[VPT][4], db[VPT][4];
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dg[i][j] = db[i][j] = 0.f;
    const int m_end = min(M, (int)(blockIdx.x + 1) * kLnRows);
    for (int m = blockIdx.x * kLnRows + wave; m < m_end; m += 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + (long long)m * N);
        const float4* gr = reinterpret_cast<const float4*>(g + (long long)m * N);
        const float4* g2r = g2 ? reinterpret_cast<const float4*>(g2 + (long long)m * N) : nullptr;
        float xv[VPT][4], gv[VPT][4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const float4 a = xr[lane + 64 * i];
            float4 b = gr[lane + 64 * i];
            if (g2r) {
                const float4 c = g2r[lane + 64 * i];
                b.x += c.x; b.y += c.y; b.z += c.z; b.w += c.w;
            }
            xv[i][0] = a.x; xv[i][1] = a.y; xv[i][2] = a.z; xv[i][3] = a.w;
            gv[i][0] = b.x; gv[i][1] = b.y; gv[i][2] = b.z; gv[i][3] = b.w;
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
#pragma unroll
        for (int i = 0; i < VPT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dg[i][j] = fmaf(gv[i][j], xv[i][j] * rstd, dg[i][j]);
                db[i][j] += gv[i][j];
            }
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[wave][0][4 * (lane + 64 * i) + j] = dg[i][j];
            red[wave][1][4 * (lane + 64 * i) + j] = db[i][j];
        }
    __syncthreads();
    float* p = partial + (long long)blockIdx.x * 2 * N;
    for (int i = threadIdx.x; i < 2 * N; i += 256) {
        const int which = i / N, n = i - which * N;
        p[i] = (red[0][which][n] + red[1][which][n]) + (red[2][which][n] + red[3][which][n]);
    }
}

// Stage 2: dgamma[n] += sum_blk partial[blk][0][n], dbeta[n] += sum_blk partial[blk][1][n].  A block owns 64 of the
// 2N outputs; its 4 waves each fold every 4th partial block, then the 4 sums are combined in fixed order.
// grid: 2N/64 blocks of 256 threads.
__global__ __launch_bounds__(256) void ln_param_final_kernel(const float* __restrict__ partial, int nblk, int N,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
#pragma unroll 8
    for (int b = wave; b < nblk; b += 4) s += partial[(long long)b * 2 * N + i];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (i < N) dgamma[i] += t;
        else dbeta[i - N] += t;
    }
}

// Head parameter gradients from what head_bwd_kernel leaves behind: pooled[B][768] (after ReLU), dz[B][256].
// dW[o][c] += sum_b dz[b][o] pooled[b][c]; db[o] += sum_b dz[b][o].  grid: 256 blocks (o) of 256 threads.
__global__ __launch_bounds__(256) void head_param_grad_kernel(const float* __restrict__ pooled,
                                                              const float* __restrict__ dz, int B,
                                                              float* __restrict__ dW, float* __restrict__ db) {
    const int o = blockIdx.x, tid = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dz[b * 256 + o];
        const float* p = pooled + (long long)b * 768;
        a0 = fmaf(d, p[tid], a0); a1 = fmaf(d, p[tid + 256], a1); a2 = fmaf(d, p[tid + 512], a2);
        sb += d;
    }
    float* w = dW + (long long)o * 768;
    w[tid] += a0; w[tid + 256] += a1; w[tid + 512] += a2;
    if (tid == 0) db[o] += sb;
}

// ---- pos-conv weight gradient ----------------------------------------------------------------------------
// Forward (group g): out[b][tau][n] = sum_{t,ci} w[g][n][t*48+ci] * xg[g][b][tau + t][ci]  (x at frames 64..64+T).
// dw[g][t][n][ci] = sum_{b,tau} dU[b][tau][n] * xg[g][b][tau + t][ci], dU stored at frame 64 + tau of dug.
// One workgroup: one group, kPdwTaps taps, a slice of the clips; 16x16x4 fp32 MFMA with the frames as the
// contraction.  Wave w owns taps 2w, 2w+1 (3 x 3 tiles of 16 x 16 each).  Pad frames of both buffers are zero,
// so chunks may run past the clip's last frame.  partial[split][g][t][n][ci].
constexpr int kPdwTaps = 8, kPdwChunk = 64;
__global__ __launch_bounds__(256) void posconv_dw_kernel(const float* __restrict__ dug, const float* __restrict__ xg,
                                                         float* __restrict__ partial, int B, int T,
                                                         int clips_per_split) {
    __shared__ __attribute__((aligned(16))) float dUs[kPdwChunk * 48];
    __shared__ __attribute__((aligned(16))) float Xs[(kPdwChunk + kPdwTaps) * 48];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, g4 = lane >> 4;
    const int t0 = blockIdx.x * kPdwTaps, grp = blockIdx.y, sp = blockIdx.z;
    const int P = T + 128;
    f32x4 acc[2][3][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int b_end = min(B, (sp + 1) * clips_per_split);
    for (int b = sp * clips_per_split; b < b_end; ++b) {
        const float* dub = dug + ((long long)grp * B + b) * P * 48;
        const float* xb = xg + ((long long)grp * B + b) * P * 48;
        for (int tau0 = 0; tau0 < T; tau0 += kPdwChunk) {
            __syncthreads();
            // dU frames 64+tau0 .. +63 (always inside the clip's padded block), X frames tau0+t0 .. +71
            for (int i = tid; i < kPdwChunk * 12; i += 256) {
                const int r = i / 12, c4 = i - r * 12;
                reinterpret_cast<float4*>(dUs)[i] = (64 + tau0 + r < P)
                    ? reinterpret_cast<const float4*>(dub + (long long)(64 + tau0 + r) * 48)[c4]
                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int i = tid; i < (kPdwChunk + kPdwTaps) * 12; i += 256) {
                const int r = i / 12, c4 = i - r * 12;
                reinterpret_cast<float4*>(Xs)[i] = (tau0 + t0 + r < P)
                    ? reinterpret_cast<const float4*>(xb + (long long)(tau0 + t0 + r) * 48)[c4]
                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
#pragma unroll 4
            for (int ks = 0; ks < kPdwChunk / 4; ++ks) {
                const int k = ks * 4 + g4;
                float af[3], bf[2][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) af[i] = dUs[k * 48 + i * 16 + fi];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 3; ++j) bf[a][j] = Xs[(k + 2 * wave + a) * 48 + j * 16 + fi];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            acc[a][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[a][j], acc[a][i][j], 0, 0, 0);
            }
        }
    }
    float* out = partial + (((long long)sp * 16 + grp) * 128 + t0 + 2 * wave) * 2304;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(long long)a * 2304 + (i * 16 + 4 * g4 + r) * 48 + j * 16 + fi] = acc[a][i][j][r];
}

// Fold the split partials into the checkpoint layout of weight_v: dwe[o = g*48+n][ci][t].
// grid: 768 blocks (o) of 256 threads.
__global__ __launch_bounds__(256) void posconv_dw_gather_kernel(const float* __restrict__ partial, int S,
                                                                float* __restrict__ dwe) {
    __shared__ float tile[128 * 49];
    const int o = blockIdx.x, grp = o / 48, n = o - grp * 48;
    for (int i = threadIdx.x; i < 128 * 48; i += 256) {
        const int t = i / 48, ci = i - t * 48;
        float s = 0.f;
        for (int sp = 0; sp < S; ++sp) s += partial[((((long long)sp * 16 + grp) * 128 + t) * 48 + n) * 48 + ci];
        tile[t * 49 + ci] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 48 * 128; i += 256) {
        const int ci = i >> 7, t = i & 127;
        dwe[(long long)o * 6144 + i] = tile[t * 49 + ci];
    }
}

// Per-tap sums over the (o, ci) rows of two [36864][128] arrays: out_partial[blk][t] = sum_rows a*b (b == nullptr:
// a*a).  grid: 576 blocks of 256 threads, 64 rows each; folded by tap_sum_final_kernel.  fp64 accumulation.
__global__ __launch_bounds__(256) void tap_dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              double* __restrict__ partial) {
    __shared__ double red[128];
    const int t = threadIdx.x & 127, half = threadIdx.x >> 7;
    double s = 0.0;
    for (int r = half; r < 64; r += 2) {
        const long long idx = ((long long)blockIdx.x * 64 + r) * 128 + t;
        const float av = a[idx];
        s += (double)av * (double)(b ? b[idx] : av);
    }
    if (half == 1) red[t] = s;
    __syncthreads();
    if (half == 0) partial[(long long)blockIdx.x * 128 + t] = s + red[t];
}
// one block of 1024 threads: 8 parts x 128 taps, part p folds blocks p, p+8, ...; parts combined in fixed order
__global__ __launch_bounds__(1024) void tap_sum_final_kernel(const double* __restrict__ partial, int nblk,
                                                             double* __restrict__ out) {
    __shared__ double red[8][128];
    const int t = threadIdx.x & 127, part = threadIdx.x >> 7;
    double s = 0.0;
    for (int b = part; b < nblk; b += 8) s += partial[(long long)b * 128 + t];
    red[part][t] = s;
    __syncthreads();
    if (part == 0)
        out[t] = ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) + ((red[4][t] + red[5][t]) + (red[6][t] + red[7][t]));
}

// weight_norm(dim=2): w[o][ci][t] = v[o][ci][t] * g[t] / ||v[:, :, t]||.
// Fold into the forward's layout pos_w[grp][64][t*48 + ci] (rows 48..63 stay zero).  grid: 768 blocks (o).
__global__ __launch_bounds__(256) void posconv_fold_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                           const double* __restrict__ nrm2, float* __restrict__ w) {
    __shared__ float sc[128];
    const int o = blockIdx.x, grp = o / 48, n = o - grp * 48;
    if (threadIdx.x < 128) sc[threadIdx.x] = (float)((double)g[threadIdx.x] / sqrt(nrm2[threadIdx.x]));
    __syncthreads();
    float* dst = w + ((long long)grp * 64 + n) * 6144;
    for (int i = threadIdx.x; i < 6144; i += 256) {  // i = ci*128 + t (coalesced read), scattered write
        const int ci = i >> 7, t = i & 127;
        dst[t * 48 + ci] = v[(long long)o * 6144 + i] * sc[t];
    }
}

// Weight-norm backward: dot[t] = sum dwe*v, n = ||v_t||:  dg[t] += dot/n;  dv += g/n * (dwe - v * dot / n^2).
__global__ __launch_bounds__(256) void posconv_wn_bwd_kernel(const float* __restrict__ dwe, const float* __restrict__ v,
                                                             const float* __restrict__ g, const double* __restrict__ nrm2,
                                                             const double* __restrict__ dot, float* __restrict__ dv,
                                                             float* __restrict__ dg) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over 768*48*128
    const int t = (int)(i & 127);
    const double n2 = nrm2[t], n = sqrt(n2);
    const float a = (float)((double)g[t] / n), c = (float)(dot[t] / n2);
    dv[i] += a * (dwe[i] - v[i] * c);
    if (i < 128) dg[i] += (float)(dot[i] / sqrt(nrm2[i]));
}

// Column sums of the group-major padded pos-conv gradient (bias gradient): db[grp*48 + c] += sum_frames dug.
// rows = B * (T + 128) frames of 48 floats per group (pad frames are zero).  Stage 1: grid (kPbChunks, 16), block =
// 5 row-parts x 48 columns; partial[grp][chunk][48].  Stage 2: one block of 768 threads folds the chunks in order.
constexpr int kPbChunks = 64;
__global__ __launch_bounds__(256) void posconv_bias_partial_kernel(const float* __restrict__ dug, long long rows,
                                                                   float* __restrict__ partial) {
    __shared__ float red[5][48];
    const int grp = blockIdx.y, c = threadIdx.x % 48, part = threadIdx.x / 48;
    const long long per = (rows + kPbChunks - 1) / kPbChunks, r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    const float* base = dug + (long long)grp * rows * 48;
    float s = 0.f;
    if (part < 5)
        for (long long r = r0 + part; r < r1; r += 5) s += base[r * 48 + c];
    if (part < 5) red[part][c] = s;
    __syncthreads();
    if (threadIdx.x < 48)
        partial[((long long)grp * kPbChunks + blockIdx.x) * 48 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) + red[4][threadIdx.x];
}
__global__ __launch_bounds__(768) void posconv_bias_final_kernel(const float* __restrict__ partial, float* __restrict__ db) {
    const int grp = threadIdx.x / 48, c = threadIdx.x - grp * 48;
    float s = 0.f;
    for (int k = 0; k < kPbChunks; ++k) s += partial[((long long)grp * kPbChunks + k) * 48 + c];
    db[threadIdx.x] += s;
}

// Fused q/k/v weight in the forward's layout from the master copy: rows 0..767 (q) scaled by head_dim^-0.5.
__global__ __launch_bounds__(256) void scale_rows_kernel(const float4* __restrict__ in, float4* __restrict__ out,
                                                         long long n4, long long n4_scaled, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = in[i];
    if (i < n4_scaled) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
    out[i] = v;
}

// ---- nn.TripletMarginLoss(margin, p=2, eps=1e-6, reduction='mean') forward + backward -------------------------
// d(x, y) = ||x - y + eps||_2 (torch.pairwise_distance);  loss = mean_i max(d(a,p) - d(a,n) + margin, 0).
// One block; wave w takes rows w, w+4, ...; gradients are optional (validation pass).
__global__ __launch_bounds__(256) void triplet_loss_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                           const float* __restrict__ n, int B, float margin,
                                                           float* __restrict__ loss, float* __restrict__ da,
                                                           float* __restrict__ dp, float* __restrict__ dn) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float eps = 1e-6f, invB = 1.0f / (float)B;
    float total = 0.f;
    for (int i = wave; i < B; i += 4) {
        const float4 av = reinterpret_cast<const float4*>(a + (long long)i * 256)[lane];
        const float4 pv = reinterpret_cast<const float4*>(p + (long long)i * 256)[lane];
        const float4 nv = reinterpret_cast<const float4*>(n + (long long)i * 256)[lane];
        float ap[4] = {av.x - pv.x + eps, av.y - pv.y + eps, av.z - pv.z + eps, av.w - pv.w + eps};
        float an[4] = {av.x - nv.x + eps, av.y - nv.y + eps, av.z - nv.z + eps, av.w - nv.w + eps};
        const float dap = sqrtf(wave_sum((ap[0] * ap[0] + ap[1] * ap[1]) + (ap[2] * ap[2] + ap[3] * ap[3])));
        const float dan = sqrtf(wave_sum((an[0] * an[0] + an[1] * an[1]) + (an[2] * an[2] + an[3] * an[3])));
        const float l = dap - dan + margin;
        total += fmaxf(l, 0.f);
        if (da) {
            const float on = l > 0.f ? invB : 0.f;
            const float ip = dap > 0.f ? on / dap : 0.f, in_ = dan > 0.f ? on / dan : 0.f;
            float4 ga, gp, gn;
            gp.x = -ap[0] * ip; gp.y = -ap[1] * ip; gp.z = -ap[2] * ip; gp.w = -ap[3] * ip;
            gn.x = an[0] * in_; gn.y = an[1] * in_; gn.z = an[2] * in_; gn.w = an[3] * in_;
            ga.x = -gp.x - gn.x; ga.y = -gp.y - gn.y; ga.z = -gp.z - gn.z; ga.w = -gp.w - gn.w;
            reinterpret_cast<float4*>(da + (long long)i * 256)[lane] = ga;
            reinterpret_cast<float4*>(dp + (long long)i * 256)[lane] = gp;
            reinterpret_cast<float4*>(dn + (long long)i * 256)[lane] = gn;
        }
    }
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * invB;
}

// ---- torch.optim.Adam (amsgrad=False, weight_decay=0), two learning rates (train_triplet.py:98-107) ------------
// theta[i >= head_begin] uses lr_head.  bc1 = 1 - beta1^t, bc2 = 1 - beta2^t computed on the host in double.
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ theta, const float4* __restrict__ grad,
                                                   float4* __restrict__ m, float4* __restrict__ v, long long n4,
                                                   long long head_begin4, float lr_body, float lr_head, float beta1,
                                                   float beta2, float eps, float bc1, float sqrt_bc2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float step = (i >= head_begin4 ? lr_head : lr_body) / bc1;
    const float4 g = grad[i];
    float4 mi = m[i], vi = v[i], th = theta[i];
#define NOMAD_ADAM(c)                                                   \
    mi.c = mi.c + (g.c - mi.c) * (1.0f - beta1);                        \
    vi.c = vi.c * beta2 + (1.0f - beta2) * g.c * g.c;                   \
    th.c = th.c - step * (mi.c / (sqrtf(vi.c) / sqrt_bc2 + eps));
    NOMAD_ADAM(x) NOMAD_ADAM(y) NOMAD_ADAM(z) NOMAD_ADAM(w)
#undef NOMAD_ADAM
    m[i] = mi; v[i] = vi; theta[i] = th;
}

// ---- dropout (dropout.hip.h) -------------------------------------------------------------------------------
// y[i] = (resid ? resid[i] : 0) + keep(idx0 + i) * scale * x[i];  x == y allowed.  idx0: element index of x[0] in the
// whole tensor (a call on one branch of a merged batch keeps the whole batch's mask).
__global__ __launch_bounds__(256) void dropout_add_kernel(const float4* __restrict__ x, const float4* __restrict__ resid,
                                                          float4* __restrict__ y, long long n4, DropCfg d,
                                                          uint32_t site, unsigned long long idx0) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = x[i];
    const unsigned long long e = idx0 + (unsigned long long)i * 4;
    v.x *= drop_mult(d, site, e); v.y *= drop_mult(d, site, e + 1);
    v.z *= drop_mult(d, site, e + 2); v.w *= drop_mult(d, site, e + 3);
    if (resid) {
        const float4 r = resid[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    y[i] = v;
}

// In-place dropout of the group-major pos-conv input xg[grp][clip][64 + t][48] (element index m*768 + c as if the
// tensor were the plain [M][768] post_extract_proj output).  grid: M blocks of 192 threads.
__global__ __launch_bounds__(192) void dropout_groups_kernel(float* __restrict__ xg, int T, long long grp_stride,
                                                             DropCfg d, uint32_t site) {
    const int m = blockIdx.x, b = m / T, t = m - b * T;
    const int col = threadIdx.x * 4, grp = col / 48, cc = col - grp * 48;
    float4* p = reinterpret_cast<float4*>(xg + grp * grp_stride + ((long long)b * (T + 128) + 64 + t) * 48 + cc);
    const unsigned long long e = (unsigned long long)m * 768 + col;
    float4 v = *p;
    v.x *= drop_mult(d, site, e); v.y *= drop_mult(d, site, e + 1);
    v.z *= drop_mult(d, site, e + 2); v.w *= drop_mult(d, site, e + 3);
    *p = v;
}

}  // namespace nomad

// ---- parameter gradients of the conv feature extractor (freeze_convnet: False, train_triplet.py:71-73) --------------
namespace nomad {

// Per-clip transpose for the conv dW GEMMs: out[c][b * Lp + m] = f(in[b * in_clip + m * ld_in + c]) for m < L, 0 for
// L <= m < Lp (Lp % 64 == 0, C % 4 == 0).  With ld_in = stride * 512 and C = taps * 512 the rows are the im2col rows
// of a time-major activation (taps of consecutive frames are contiguous).  grid: (Lp/64, ceil(C/64), B), 256 threads.
template <int ACT>
__global__ __launch_bounds__(256) void transpose_clips_kernel(const float* __restrict__ in, long long in_clip, int ld_in,
                                                              float* __restrict__ out, long long ld_out, int Lp, int L, int C) {
    __shared__ float tile[64][65];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
    const int r = tid >> 4, q = (tid & 15) * 4;
    in += (long long)blockIdx.z * in_clip;
    out += (long long)blockIdx.z * Lp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r + 16 * i, c = c0 + q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < L && c < C) v = *reinterpret_cast<const float4*>(in + (long long)m * ld_in + c);
        if (ACT == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        tile[r + 16 * i][q] = v.x; tile[r + 16 * i][q + 1] = v.y; tile[r + 16 * i][q + 2] = v.z; tile[r + 16 * i][q + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + r + 16 * i;
        if (c < C) {
            float4 v;
            v.x = tile[q][r + 16 * i]; v.y = tile[q + 1][r + 16 * i]; v.z = tile[q + 2][r + 16 * i]; v.w = tile[q + 3][r + 16 * i];
            *reinterpret_cast<float4*>(out + (long long)c * ld_out + m0 + q) = v;
        }
    }
}

// Checkpoint layout [co][ci][tap] -> kernel layout [co][tap * 512 + ci] (K taps).  grid: 512 * K * 2 blocks of 256.
__global__ __launch_bounds__(256) void conv_repack_kernel(const float* __restrict__ w, float* __restrict__ out, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // index into out
    if (i >= 512LL * 512 * K) return;
    const int ci = (int)(i & 511), tap = (int)((i >> 9) % K), co = (int)(i / (512LL * K));
    out[i] = w[((long long)co * 512 + ci) * K + tap];
}

// grad[co][ci][tap] += dw[co][tap * 512 + ci]: the dW GEMM's output back into the checkpoint layout.
__global__ __launch_bounds__(256) void conv_grad_permute_kernel(const float* __restrict__ dw, float* __restrict__ grad, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // index into grad
    if (i >= 512LL * 512 * K) return;
    const int tap = (int)(i % K), ci = (int)((i / K) & 511), co = (int)(i / (512LL * K));
    grad[i] += dw[((long long)co * K + tap) * 512 + ci];
}

// conv0 + GroupNorm parameter gradients, pass 2 (pass 1 = gn_bwd_stats_kernel, backward.hip.h): with
//   dz = G gelu'(z), dy = gamma rstd (dz - s1/L0 - yhat s2/L0):   d w0[c][j] = sum_{b,t} dy[t,c] x[5t+j]
// per (clip, frame chunk) partials cpart[b][chunk][512][10], folded in fixed order by conv0_param_final_kernel together
// with d gamma[c] = sum_b s2[b][c], d beta[c] = sum_b s1[b][c] from pass 1's partials.  grid: (chunks, B), 256 threads.
__global__ __launch_bounds__(256) void conv0_param_partial_kernel(const float* __restrict__ wav, int n_samples, int L0,
                                                                  const float* __restrict__ w0, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ gmean,
                                                                  const float* __restrict__ grstd, const float* __restrict__ G,
                                                                  const float* __restrict__ fold, int nchunks,
                                                                  float* __restrict__ cpart) {   // fold: gn_bwd_fold_kernel's [B][2][512] raw sums
    __shared__ float xs[kGnChunk * 5 + 8];
    const int b = blockIdx.y, t0 = blockIdx.x * kGnChunk, nfr = min(kGnChunk, L0 - t0), tid = threadIdx.x;
    const float* x = wav + (long long)b * n_samples + 5 * t0;
    for (int i = tid; i < 5 * nfr + 5; i += 256) xs[i] = x[i];
    float w[2][10], sc[2], sh[2], mean[2], rstd[2], m1[2], m2[2], dw[2][10];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = tid + 256 * q;
        sc[q] = scale[b * 512 + c]; sh[q] = shift[b * 512 + c];
        mean[q] = gmean[b * 512 + c]; rstd[q] = grstd[b * 512 + c];
#pragma unroll
        for (int j = 0; j < 10; ++j) { w[q][j] = w0[c * 10 + j]; dw[q][j] = 0.f; }
        m1[q] = fold[(long long)b * 1024 + c] / (float)L0;
        m2[q] = fold[(long long)b * 1024 + 512 + c] / (float)L0;
    }
    __syncthreads();
    const float* g = G + ((long long)b * L0 + t0) * 512;
    for (int t = 0; t < nfr; ++t) {
        float z[2], yh[2];
        conv0_frame(xs, t, w, sc, sh, mean, rstd, z, yh);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float dz = g[(long long)t * 512 + tid + 256 * q] * dgelu_erf(z[q]);
            const float dy = sc[q] * (dz - m1[q] - yh[q] * m2[q]);
#pragma unroll
            for (int j = 0; j < 10; ++j) dw[q][j] = fmaf(dy, xs[5 * t + j], dw[q][j]);
        }
    }
    float* o = cpart + ((long long)b * nchunks + blockIdx.x) * 5120;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 10; ++j) o[(tid + 256 * q) * 10 + j] = dw[q][j];
}

// grid: 22 blocks of 256: elements 0..5119 = d w0, 5120..5631 = d gamma, 5632..6143 = d beta (accumulated into the
// gradient vector; (b, chunk) folded in fixed order).
__global__ __launch_bounds__(256) void conv0_param_final_kernel(const float* __restrict__ cpart, const float* __restrict__ fold,
                                                                int B, int nchunks, float* __restrict__ dw0,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 5120) {
        float a = 0.f;
        for (int k = 0; k < B * nchunks; ++k) a += cpart[(long long)k * 5120 + i];
        dw0[i] += a;
    } else if (i < 6144) {
        const int c = (i - 5120) & 511, which = (i - 5120) >> 9;  // 0: gamma <- s2, 1: beta <- s1
        float a = 0.f;
        for (int k = 0; k < B; ++k) a += fold[(long long)k * 1024 + (which == 0 ? 512 : 0) + c];   // per-clip sums (gn_bwd_fold_kernel), clip order
        (which == 0 ? dgamma : dbeta)[c] += a;
    }
}

}  // namespace nomad
