// fp32 flash-style self-attention for wav2vec 2.0 BASE (12 heads x 64, no mask, eval mode):
//   ctx[b,t,h*64:(h+1)*64] = softmax_j(q[b,t,h] . k[b,j,h]) v[b,j,h]          
// Input is the fused QKV GEMM output qkv[B*T][2304] = [q*64^-0.5 | k | v] (the 1/8 scale of
// fairseq MultiheadAttention is folded into Wq/bq at repack time).
//
// One workgroup = 64 query rows of one (clip, head); wave w owns 16 of them.  K/V are streamed
// in 64-key tiles through LDS with an online softmax, so any T works (199 frames for 4 s,
// 1499 for 30 s) without materialising the T x T score matrix.
//
// Matrix core: v_mfma_f32_16x16x4_f32.  Both products are computed TRANSPOSED so that the score
// accumulator is directly the next MFMA's B operand (no LDS round trip, no lane shuffles):
//   S^T[key][q] = sum_d K[key][d] Q[q][d]      A = K (LDS), B = Q (registers)
//   O^T[d][q]  += sum_key V[key][d] P[key][q]  A = V (LDS), B = P (the S^T accumulator itself)
// D layout: col = lane&15, row = 4*(lane>>4) + reg.  Lane (qi = lane&15, g = lane>>4) therefore
// holds, for its ONE query row, keys {16*sub + 4g + r} - and an MFMA k-step wants lane group g to
// supply contraction index g.  The contraction order is free, so k-step (sub, r) contracts keys
// {16*sub + 4g + r : g = 0..3}: each lane's own register r.  The same trick orders d for S^T
// (one ds_read_b128 of K feeds 4 MFMAs).  Softmax statistics are per lane (all 4 lane groups of
// a query share them after two xor-shuffles).
#pragma once
#include <hip/hip_runtime.h>

#include "dropout.hip.h"
#include "dtypes.hip.h"
#include "gemm_f32.hip.h"

namespace nomad {

constexpr int kAttnLD = 68;  // 64 + 4 floats: 272-B rows

// exp(x) = 2^(x log2 e) on the hardware exp2 (v_exp_f32, ~1 ulp); x <= 0 here, exp(-inf) = 0.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// One 64-key tile for one wave: NSUB = number of 16-key sub-tiles that hold at least one valid key,
// `valid` = number of valid keys in the tile (keys >= valid are masked to -inf).
// DROP: attention dropout (training) - the probabilities that multiply V are masked and rescaled, the softmax
// normaliser l_run is not (fairseq: dropout(softmax(scores)) @ v).  drow = ((b*12+h)*T + q)*T + first key of the tile.
template <int NSUB, bool DROP = false>
__device__ __forceinline__ void attn_tile(const float* __restrict__ Ks, const float* __restrict__ Vs,
                                          const float4 (&qf)[4], f32x4 (&o)[4], float& m_run, float& l_run, int qi,
                                          int g, int valid, const DropCfg* dc = nullptr, uint32_t site = 0,
                                          unsigned long long drow = 0) {
    f32x4 s[NSUB];
#pragma unroll
    for (int i = 0; i < NSUB; ++i) s[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        float4 kf[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
            kf[sub] = *reinterpret_cast<const float4*>(Ks + (sub * 16 + qi) * kAttnLD + dd * 16 + g * 4);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[sub].x, qf[dd].x, s[sub], 0, 0, 0);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[sub].y, qf[dd].y, s[sub], 0, 0, 0);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[sub].z, qf[dd].z, s[sub], 0, 0, 0);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[sub].w, qf[dd].w, s[sub], 0, 0, 0);
    }
    float m_tile = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (sub * 16 + g * 4 + r >= valid) s[sub][r] = -INFINITY;
            m_tile = fmaxf(m_tile, s[sub][r]);
        }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16));
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
    const float m_new = fmaxf(m_run, m_tile);  // finite: every tile holds at least one valid key
    const float alpha = fast_exp(m_run - m_new);   // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = fast_exp(s[sub][r] - m_new);
            psum += pv;
            s[sub][r] = DROP ? pv * drop_mult(*dc, site, drow + (sub * 16 + g * 4 + r)) : pv;
        }
    l_run = l_run * alpha + psum;  // per-lane partial (this lane's keys); folded across g at the end
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] *= alpha;
    // O^T += V^T P^T : k-step (sub, r) contracts keys 16*sub + 4g + r
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vrow = Vs + (sub * 16 + g * 4 + r) * kAttnLD + qi;
            const float pv = s[sub][r];
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) o[ds] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[ds * 16], pv, o[ds], 0, 0, 0);
        }
}

// This is the kernel for SHORT clips (fewer than kAttnV2MinT frames: config C4's T = 50, short files of a ragged batch)
// and for the training forward with attention dropout; longer clips take attention_f32_v2_kernel (attention_f32_v2.hip.h).
// Which kernel a clip gets depends on ITS frame count only, never on the batch it is in.
// lse (nullable): [B*12][T] log-sum-exp of every score row, saved for the backward pass.
// T_ = storage type of qkv / out (fp32 or bf16); the arithmetic is fp32 MFMA either way.
// tpref (nullable): ragged batches - clip b owns rows tpref[b] .. tpref[b+1]-1 of qkv / out.
// TO_ = storage type of out when it differs from qkv's (split planes `out_plane` apart in the bf16x3 path).
template <typename T_ = float, bool DROP = false, typename TO_ = T_>
__global__ __launch_bounds__(256) void attention_f32_kernel(const T_* __restrict__ qkv, TO_* __restrict__ out,
                                                            float* __restrict__ lse, int T,
                                                            const int* __restrict__ tpref = nullptr,
                                                            DropCfg dc = DropCfg{}, uint32_t site = 0, int bh0 = 0,
                                                            long long out_plane = 0, int t_below = 0) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * kAttnLD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * kAttnLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / 12, h = bh - b * 12;
    long long row0 = (long long)b * T;
    if (tpref) {
        row0 = tpref[b];
        T = tpref[b + 1] - tpref[b];
        if ((int)blockIdx.x * 64 >= T) return;  // whole workgroup: no barrier has been reached yet
        if (t_below && T >= t_below) return;     // ragged batch: clips of t_below frames or more belong to attention_f32_v2_kernel
    }
    const long long base = row0 * 2304 + h * 64;
    const int q_row = blockIdx.x * 64 + wave * 16 + qi;
    const int q_ld = q_row < T ? q_row : T - 1;

    // Q fragment: Q[q][16*dd + 4g + j]
    float4 qf[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd)
        qf[dd] = load4<T_>(qkv + base + (long long)q_ld * 2304 + dd * 16 + g * 4);

    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // Work is skipped at 16-row granularity in both dimensions (T = 199 is 12.4 sub-tiles of 16): a wave
    // whose 16 query rows are all past T only helps with the K/V loads, and the last key tile multiplies only
    // its NSUB sub-tiles that hold a valid key (straight-line code per NSUB, no branches among the MFMAs).
    const bool wave_active = blockIdx.x * 64 + wave * 16 < T;  // wave-uniform
    const int ntiles = (T + 63) / 64;
    // K/V tiles are prefetched into registers one tile ahead, so the global-load latency of tile kt+1 runs
    // under the MFMAs of tile kt instead of in front of them.
    float4 kreg[4], vreg[4];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256, row = id >> 4, c4 = id & 15;
            int key = kt * 64 + row;
            key = key < T ? key : T - 1;
            const T_* src = qkv + base + (long long)key * 2304 + c4 * 4;
            kreg[i] = load4<T_>(src + 768);
            vreg[i] = load4<T_>(src + 1536);
        }
    };
    fetch(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256, row = id >> 4, c4 = id & 15;
            *reinterpret_cast<float4*>(Ks + row * kAttnLD + c4 * 4) = kreg[i];
            *reinterpret_cast<float4*>(Vs + row * kAttnLD + c4 * 4) = vreg[i];
        }
        __syncthreads();
        if (kt + 1 < ntiles) fetch(kt + 1);
        if (!wave_active) continue;
        const int valid = T - kt * 64;  // valid keys in this tile (>= 1)
        const unsigned long long drow = ((unsigned long long)(bh0 + bh) * T + q_ld) * T + kt * 64;
        if (valid >= 64) attn_tile<4, DROP>(Ks, Vs, qf, o, m_run, l_run, qi, g, 64, &dc, site, drow);
        else if (valid > 48) attn_tile<4, DROP>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid, &dc, site, drow);
        else if (valid > 32) attn_tile<3, DROP>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid, &dc, site, drow);
        else if (valid > 16) attn_tile<2, DROP>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid, &dc, site, drow);
        else attn_tile<1, DROP>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid, &dc, site, drow);
    }

    float l_tot = l_run + __shfl_xor(l_run, 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float inv = 1.0f / l_tot;
    if (lse && q_row < T && g == 0) lse[(long long)bh * T + q_row] = m_run + logf(l_tot);
    if (q_row < T) {
        TO_* dst = out + (row0 + q_row) * 768 + h * 64 + g * 4;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            float4 r;
            r.x = o[ds][0] * inv; r.y = o[ds][1] * inv; r.z = o[ds][2] * inv; r.w = o[ds][3] * inv;
            store4p<TO_>(dst + ds * 16, out_plane, r);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Building blocks of the bf16 MFMA attention tiles (the bf16x3 kernels below; the plain bf16 kernel of config C5 is in
// attention_bf16_v2.hip.h).  Same transposed structure with v_mfma_f32_16x16x32_bf16 (lane l: A[row l&15][k = 8(l>>4)+j], B likewise):
//   S^T = K Q^T   : A = K rows from LDS (ds_read_b128, 144-B padded rows), B = Q from registers
//   O^T += V^T P^T: B = the S^T accumulators converted to bf16 in place - k-slot (g, j) of a 32-key block is
//                   key 16*s0 + 4g + j (j < 4) or 16*s1 + 4g + j - 4 (j >= 4), i.e. the lane's own registers of the
//                   two 16-key sub-tiles; A = V^T gathered by ds_read_b64_tr_b16 from the ROW-major V tile
//                   (hardware transpose: lane i of a 16-lane group gets column i of 4 key rows), two reads per MFMA.
// Softmax statistics, masks and the output accumulator are fp32, exactly as in the fp32 kernel.
constexpr int kAttn16LD = 160;  // bytes per LDS row: 64 bf16 + 32 B pad.  ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): a group reads fragment rows 0-3 / 12-15 at one
// 16-byte chunk and rows 4-11 at its neighbour.  With a 10-slot row stride the first set lands on the even slots of the
// 256-byte bank row and the second on the odd ones; the 9-slot stride (144 B) that suits contiguous groups is 2-way.
// The transposing V reads (8 rows x 32 B per 32-lane half) tile the 64 banks exactly at this stride too.

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

// ---------------------------------------------------------------------------------------------------
// bf16x3 attention: the bf16 kernel above on split operands (dtypes.hip.h: hi / lo bf16 planes `plane` elements apart).
// Both products run as three bf16 MFMAs with fp32 accumulation,
//   S^T  = K_hi Q_hi^T + K_hi Q_lo^T + K_lo Q_hi^T
//   O^T += V_hi^T P_hi^T + V_hi^T P_lo^T + V_lo^T P_hi^T      (P split in registers: p_hi = bf16(p), p_lo = bf16(p - p_hi))
// so logits and the output see 2^-16-relative product errors instead of bf16's 2^-8; softmax statistics, masks and the
// accumulators are fp32 as everywhere.  6 bf16 MFMAs (16 cycles each) replace 8 fp32 MFMAs (32 cycles each) per
// 16 x 16 x 64 block.  LDS: K_hi, K_lo, V_hi, V_lo tiles, 36 KB.
template <int NSUB>
__device__ __forceinline__ void attn_tile_x3(const char* __restrict__ Kh, const char* __restrict__ Kl,
                                             const char* __restrict__ Vh, const char* __restrict__ Vl,
                                             const bf16x8 (&qh)[2], const bf16x8 (&ql)[2], f32x4 (&o)[4], float& m_run,
                                             float& l_run, int qi, int g, int valid) {
    f32x4 s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 kfh[NSUB], kfl[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            kfh[sub] = *reinterpret_cast<const bf16x8*>(Kh + (sub * 16 + qi) * kAttn16LD + (4 * ks + g) * 16);
            kfl[sub] = *reinterpret_cast<const bf16x8*>(Kl + (sub * 16 + qi) * kAttn16LD + (4 * ks + g) * 16);
        }
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            s[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfl[sub], qh[ks], s[sub], 0, 0, 0);
            s[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh[sub], ql[ks], s[sub], 0, 0, 0);
            s[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh[sub], qh[ks], s[sub], 0, 0, 0);
        }
    }
    float m_tile = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (sub >= NSUB || sub * 16 + g * 4 + r >= valid) s[sub][r] = -INFINITY;
            m_tile = fmaxf(m_tile, s[sub][r]);
        }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16));
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
    const float m_new = fmaxf(m_run, m_tile);
    const float alpha = fast_exp(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = fast_exp(s[sub][r] - m_new);
            s[sub][r] = pv;
            psum += pv;
        }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] *= alpha;
    const int tr_row = (qi >> 2), tr_col = (qi & 3) * 4;
#pragma unroll
    for (int pb = 0; pb < (NSUB + 1) / 2; ++pb) {
        bf16x8 ph, pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ph[r] = (bf16_t)s[2 * pb][r];
            ph[4 + r] = (bf16_t)s[2 * pb + 1][r];
            pl[r] = (bf16_t)(s[2 * pb][r] - (float)ph[r]);
            pl[4 + r] = (bf16_t)(s[2 * pb + 1][r] - (float)ph[4 + r]);
        }
        const int voff = ((2 * pb) * 16 + g * 4 + tr_row) * kAttn16LD + tr_col * 2;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(Vh + voff + ds * 32));
            const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(Vh + voff + 16 * kAttn16LD + ds * 32));
            const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(Vl + voff + ds * 32));
            const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(Vl + voff + 16 * kAttn16LD + ds * 32));
            bf16x8 vh, vl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vh[r] = h0[r];
                vh[4 + r] = h1[r];
                vl[r] = l0[r];
                vl[4 + r] = l1[r];
            }
            o[ds] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph, o[ds], 0, 0, 0);
            o[ds] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[ds], 0, 0, 0);
            o[ds] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[ds], 0, 0, 0);
        }
    }
}

// qkv: split [M][2304] (q pre-scaled), out: split [M][768].  tpref (nullable): ragged batches, as above.
__global__ __launch_bounds__(256) void attention_x3_kernel(const bf16s_t* __restrict__ qkv_s, long long in_plane,
                                                           bf16s_t* __restrict__ out, long long out_plane, int T,
                                                           const int* __restrict__ tpref = nullptr) {
    __shared__ __attribute__((aligned(16))) char Kh[64 * kAttn16LD];
    __shared__ __attribute__((aligned(16))) char Kl[64 * kAttn16LD];
    __shared__ __attribute__((aligned(16))) char Vh[64 * kAttn16LD];
    __shared__ __attribute__((aligned(16))) char Vl[64 * kAttn16LD];
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_s);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / 12, h = bh - b * 12;
    long long row0 = (long long)b * T;
    if (tpref) {
        row0 = tpref[b];
        T = tpref[b + 1] - tpref[b];
        if ((int)blockIdx.x * 64 >= T) return;  // whole workgroup: no barrier has been reached yet
    }
    const long long base = row0 * 2304 + h * 64;
    const int q_row = blockIdx.x * 64 + wave * 16 + qi;
    const int q_ld = q_row < T ? q_row : T - 1;
    bf16x8 qh[2], ql[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16_t* src = qkv + base + (long long)q_ld * 2304 + ks * 32 + g * 8;
        qh[ks] = *reinterpret_cast<const bf16x8*>(src);
        ql[ks] = *reinterpret_cast<const bf16x8*>(src + in_plane);
    }
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const bool wave_active = blockIdx.x * 64 + wave * 16 < T;
    const int ntiles = (T + 63) / 64;
    bf16x8 kh[2], kl[2], vh[2], vl[2];  // next tile, prefetched under the current tile's MFMAs
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256, row = id >> 3, ch = id & 7;
            int key = kt * 64 + row;
            key = key < T ? key : T - 1;
            const bf16_t* src = qkv + base + (long long)key * 2304 + ch * 8;
            kh[i] = *reinterpret_cast<const bf16x8*>(src + 768);
            kl[i] = *reinterpret_cast<const bf16x8*>(src + 768 + in_plane);
            vh[i] = *reinterpret_cast<const bf16x8*>(src + 1536);
            vl[i] = *reinterpret_cast<const bf16x8*>(src + 1536 + in_plane);
        }
    };
    fetch(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256, row = id >> 3, ch = id & 7;
            *reinterpret_cast<bf16x8*>(Kh + row * kAttn16LD + ch * 16) = kh[i];
            *reinterpret_cast<bf16x8*>(Kl + row * kAttn16LD + ch * 16) = kl[i];
            *reinterpret_cast<bf16x8*>(Vh + row * kAttn16LD + ch * 16) = vh[i];
            *reinterpret_cast<bf16x8*>(Vl + row * kAttn16LD + ch * 16) = vl[i];
        }
        __syncthreads();
        if (kt + 1 < ntiles) fetch(kt + 1);
        if (!wave_active) continue;  // wave-uniform: the transposing reads need a full EXEC mask
        const int valid = T - kt * 64;
        if (valid > 48) attn_tile_x3<4>(Kh, Kl, Vh, Vl, qh, ql, o, m_run, l_run, qi, g, valid);
        else if (valid > 32) attn_tile_x3<3>(Kh, Kl, Vh, Vl, qh, ql, o, m_run, l_run, qi, g, valid);
        else if (valid > 16) attn_tile_x3<2>(Kh, Kl, Vh, Vl, qh, ql, o, m_run, l_run, qi, g, valid);
        else attn_tile_x3<1>(Kh, Kl, Vh, Vl, qh, ql, o, m_run, l_run, qi, g, valid);
    }
    float l_tot = l_run + __shfl_xor(l_run, 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float inv = 1.0f / l_tot;
    if (q_row < T) {
        bf16s_t* dst = out + (row0 + q_row) * 768 + h * 64 + g * 4;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
            store4p<bf16s_t>(dst + ds * 16, out_plane, make_float4(o[ds][0] * inv, o[ds][1] * inv, o[ds][2] * inv, o[ds][3] * inv));
    }
}

// The same attention with the head's whole K / V (split planes) resident in LDS: one workgroup per (clip, head), K / V
// loaded once instead of once per 64-query block, no barrier inside the loop - waves walk the 16-query sub-tiles on
// their own.  For short clips (4 x ceil(T / 32) * 32 rows of 160 B <= 160 KB, i.e. T <= 256; a 4 s clip has T = 199).
// Same tile function, same key order: bit-identical to attention_x3_kernel.  Dynamic LDS: attn_x3_resident_lds(max T).
__host__ __device__ inline int attn_x3_resident_rows(int T) { return (T + 31) / 32 * 32; }
__host__ __device__ inline size_t attn_x3_resident_lds(int T) { return (size_t)4 * attn_x3_resident_rows(T) * kAttn16LD; }
constexpr int kAttnResidentMaxT = 256;
constexpr int kAttnResidentWaves = 8;   // waves per (clip, head) workgroup (measured: profiles/r01_attention_x3_resident.txt)

__global__ __launch_bounds__(512) void attention_x3_resident_kernel(const bf16s_t* __restrict__ qkv_s, long long in_plane,
                                                                    bf16s_t* __restrict__ out, long long out_plane, int T,
                                                                    int rows_alloc, const int* __restrict__ tpref = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char attn_lds[];
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_s);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / 12, h = bh - b * 12;
    long long row0 = (long long)b * T;
    if (tpref) {
        row0 = tpref[b];
        T = tpref[b + 1] - tpref[b];
    }
    const int rows = attn_x3_resident_rows(T);
    char* Kh = attn_lds;
    char* Kl = Kh + (size_t)rows_alloc * kAttn16LD;
    char* Vh = Kl + (size_t)rows_alloc * kAttn16LD;
    char* Vl = Vh + (size_t)rows_alloc * kAttn16LD;
    const long long base = row0 * 2304 + h * 64;
    // stage K and V of this head: rows >= T repeat the last key (finite data under the masks)
    for (int id = tid; id < rows * 8; id += blockDim.x) {
        const int row = id >> 3, ch = id & 7;
        const int key = row < T ? row : T - 1;
        const bf16_t* src = qkv + base + (long long)key * 2304 + ch * 8;
        *reinterpret_cast<bf16x8*>(Kh + row * kAttn16LD + ch * 16) = *reinterpret_cast<const bf16x8*>(src + 768);
        *reinterpret_cast<bf16x8*>(Kl + row * kAttn16LD + ch * 16) = *reinterpret_cast<const bf16x8*>(src + 768 + in_plane);
        *reinterpret_cast<bf16x8*>(Vh + row * kAttn16LD + ch * 16) = *reinterpret_cast<const bf16x8*>(src + 1536);
        *reinterpret_cast<bf16x8*>(Vl + row * kAttn16LD + ch * 16) = *reinterpret_cast<const bf16x8*>(src + 1536 + in_plane);
    }
    __syncthreads();
    const int ntiles = (T + 63) / 64;
    for (int qs = wave; qs * 16 < T; qs += nwaves) {  // wave-uniform: the transposing reads need a full EXEC mask
        const int q_row = qs * 16 + qi;
        const int q_ld = q_row < T ? q_row : T - 1;
        bf16x8 qh[2], ql[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16_t* src = qkv + base + (long long)q_ld * 2304 + ks * 32 + g * 8;
            qh[ks] = *reinterpret_cast<const bf16x8*>(src);
            ql[ks] = *reinterpret_cast<const bf16x8*>(src + in_plane);
        }
        f32x4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        for (int kt = 0; kt < ntiles; ++kt) {
            const int off = kt * 64 * kAttn16LD;
            const int valid = T - kt * 64;
            if (valid > 48) attn_tile_x3<4>(Kh + off, Kl + off, Vh + off, Vl + off, qh, ql, o, m_run, l_run, qi, g, valid);
            else if (valid > 32) attn_tile_x3<3>(Kh + off, Kl + off, Vh + off, Vl + off, qh, ql, o, m_run, l_run, qi, g, valid);
            else if (valid > 16) attn_tile_x3<2>(Kh + off, Kl + off, Vh + off, Vl + off, qh, ql, o, m_run, l_run, qi, g, valid);
            else attn_tile_x3<1>(Kh + off, Kl + off, Vh + off, Vl + off, qh, ql, o, m_run, l_run, qi, g, valid);
        }
        float l_tot = l_run + __shfl_xor(l_run, 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = 1.0f / l_tot;
        if (q_row < T) {
            bf16s_t* dst = out + (row0 + q_row) * 768 + h * 64 + g * 4;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds)
                store4p<bf16s_t>(dst + ds * 16, out_plane, make_float4(o[ds][0] * inv, o[ds][1] * inv, o[ds][2] * inv, o[ds][3] * inv));
        }
    }
}

}  // namespace nomad
