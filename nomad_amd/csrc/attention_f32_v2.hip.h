// fp32 self-attention, second generation (the reference's arithmetic: exact fp32 products, fp32 softmax):
//   ctx[b,t,h*64:(h+1)*64] = softmax_j(q[b,t,h] . k[b,j,h]) v[b,j,h]        (fairseq MultiheadAttention)
// on qkv[B*T][2304] = [q*64^-0.5 | k | v] fp32.  Same structure as attention_bf16_v2.hip.h, on v_mfma_f32_32x32x2_f32:
//   * 128 queries per workgroup, 32 per wave; both products TRANSPOSED (S^T = K Q^T, O^T += V^T P^T), so a lane owns
//     ONE query (column lane & 31) and 16 of a 32-key block's scores (rows (i&3) + 8(i>>2) + 4(lane>>5)): maximum and
//     sum are register trees + one v_permlane32_swap, and the score accumulator's register i IS the B operand of
//     P.V's k-step i (keys (i&3) + 8(i>>2) of lanes 0-31 with keys +4 of lanes 32-63) - no conversion, no LDS trip;
//   * the contraction order of S^T is free as well: a lane reads K and Q as float4 chunks d = 8j + 4h .. + 3 and
//     k-step (j, r) contracts d = 8j + r with 8j + 4 + r; V is staged TRANSPOSED ([d][key]) so that one float4 holds a
//     lane's A operands of four k-steps.  Eight ds_read_b128 per operand and 32-key block, 65 MFMAs;
//   * scores in log2 units (q * log2 e at load), reference maximum subtracted by the matrix core (one extra k-step
//     ones x (-m_ref)), moved only when a block exceeds it by 2^kA2Thr: per score one v_exp and one v_add remain;
//   * K / V^T tiles of 32 keys double-buffered in LDS (32 KB per workgroup), ONE barrier per tile, rows XOR-swizzled
//     by 16-byte chunk (conflict-free ds_read_b128 lane groups); the workgroups of a head share an XCD.
// Round 1's kernel (attention_f32_kernel, attention.hip.h: 16x16x4 tiles, two barriers per 64 keys) measured 80 TFLOP/s,
// matrix pipe busy 54 %; it stays for the training forward with attention dropout.
#pragma once
#include <hip/hip_runtime.h>

#include "attention_bf16_v2.hip.h"

namespace nomad {

constexpr int kAttnV2MinT = 128;                  // clips with at least this many frames take this kernel (a property of the
                                                  // clip, not of the batch: results stay batch-invariant)
constexpr int kF2KT = 32;                         // keys per LDS tile
constexpr int kF2Buf = kF2KT * 256 + 64 * 128;    // K [32][64 f32] + V^T [64][32 f32] = 16 KB
constexpr int attn_f32_v2_lds() { return 2 * kF2Buf; }

// lse (nullable): [B*12][T] natural-log log-sum-exp of every score row (the backward recomputes P from it).
// grid: 1-D, ceil(T / 128) * B * 12 workgroups of 256 threads; dynamic LDS attn_f32_v2_lds().
// tpref (nullable): ragged batches - clip b owns rows tpref[b] .. tpref[b+1]-1; T is then the longest clip's.
__global__ __launch_bounds__(256, 3) void attention_f32_v2_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  float* __restrict__ lse, int T, int nqblk,
                                                                  const int* __restrict__ tpref, int t_min) {
    extern __shared__ __attribute__((aligned(16))) char f2_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    const int virt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int bh = virt / nqblk, qb = virt - bh * nqblk;
    const int b = bh / 12, hd = bh - b * 12;
    long long row0 = (long long)b * T;
    if (tpref) {
        row0 = tpref[b];
        T = tpref[b + 1] - tpref[b];
    }
    if (qb * 128 >= T || T < t_min) return;  // whole workgroup, before any barrier (t_min: ragged batches leave short clips
                                             // to attention_f32_kernel)
    const float* __restrict__ src_bh = qkv + row0 * 2304 + hd * 64;
    const int q_row = qb * 128 + wave * 32 + r;
    float4 qf[8];  // Q[q][8j + 4h .. + 3] * log2(e)
    {
        const float* qp = src_bh + (long long)(q_row < T ? q_row : T - 1) * 2304 + 4 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(qp + 8 * j);
            qf[j] = make_float4(v.x * kLog2e, v.y * kLog2e, v.z * kLog2e, v.w * kLog2e);
        }
    }
    f32x16 o0, o1;  // O^T: d = 32*dblk + (i&3) + 8(i>>2) + 4h, this lane's query
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        o0[i] = 0.f;
        o1[i] = 0.f;
    }
    float m_ref = 0.f, l_run = 0.f;
    const float ones_a = h == 0 ? 1.f : 0.f;  // A[key][k = h] of the extra k-step
    float negm_b = 0.f;                        // B[k = h][query] = (h == 0) ? -m_ref : 0
    const bool wave_active = qb * 128 + wave * 32 < T;
    const int ntiles = (T + kF2KT - 1) / kF2KT;

    // ---- staging: thread -> (key row, float4 chunk) x 2 of the 32 x 64 tile, for K and for V ----
    // Round 4: K goes global -> LDS by LDS-DMA (the instruction fills 1 KB per wave linearly, so the thread fetches the LOGICAL chunk
    // that the XOR swizzle maps to its slot: chunk ^ (row & 15)); only V, which is stored transposed, still passes through registers.
    // Before, the eight staging registers of K pushed the loop over its 168-register budget: the compiler spilled and reloaded 8
    // registers through scratch in every key tile (llvm-objdump of the round-3 kernel: scratch_store / scratch_load_dwordx4 x 2).
    float4 vreg[2];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto fetch = [&](int kt) {
        char* B0 = f2_lds + (kt & 1) * kF2Buf;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cid = tid + i * 256, row = cid >> 4, pc = cid & 15;
            int key = kt * kF2KT + row;
            key = key < T ? key : T - 1;
            const float* src = src_bh + (long long)key * 2304;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 768 + 4 * (pc ^ (row & 15))), (lptr_t)(B0 + i * 4096 + wave_u * 1024), 16, 0, 0);
            vreg[i] = *reinterpret_cast<const float4*>(src + 1536 + pc * 4);
        }
    };
    auto stage = [&](int buf) {
        char* B0 = f2_lds + buf * kF2Buf;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // V[key = row][d = 4*c16 + c] -> V^T[d][key]: row d is 128 B, chunk (key >> 2) ^ ((d >> 1) & 7), element key & 3
            const int cid = tid + i * 256, row = cid >> 4, c16 = cid & 15;
            const int kq = row >> 2, ke = row & 3;
            char* vt = B0 + kF2KT * 256 + 4 * ke;
            const int d0 = 4 * c16, x0 = (d0 >> 1) & 7, x1 = x0 + 1;  // d0, d0+1 share x0; d0+2, d0+3 share x0 + 1 (d0 % 4 == 0)
            *reinterpret_cast<float*>(vt + (d0 + 0) * 128 + 16 * (kq ^ x0)) = vreg[i].x;
            *reinterpret_cast<float*>(vt + (d0 + 1) * 128 + 16 * (kq ^ x0)) = vreg[i].y;
            *reinterpret_cast<float*>(vt + (d0 + 2) * 128 + 16 * (kq ^ x1)) = vreg[i].z;
            *reinterpret_cast<float*>(vt + (d0 + 3) * 128 + 16 * (kq ^ x1)) = vreg[i].w;
        }
    };
    // ---- fragment addresses inside a buffer ----
    const int k_base = r * 256 + 16 * (h ^ (r & 1)), k_x = (r >> 1) & 7;       // chunk 2j + h of row r at 32 * (j ^ k_x)
    const int v_x = (r >> 1) & 7;                                               // V^T row d = r (+ 32): chunk 2m + h
    const int v_base = kF2KT * 256 + r * 128 + 16 * (h ^ (v_x & 1)), v_xm = v_x >> 1;  // at 32 * (m ^ v_xm)

    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) fetch(kt + 1);
        if (wave_active) {
            const char* B0 = f2_lds + (kt & 1) * kF2Buf;
            // ---- S^T - m_ref (log2 units) for 32 keys x 32 queries ----
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ones_a, negm_b, s, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 kf = *reinterpret_cast<const float4*>(B0 + k_base + 32 * (j ^ k_x));
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[j].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[j].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[j].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[j].w, s, 0, 0, 0);
            }
            const int valid = T - kt * kF2KT;
            if (valid < 32) {  // the clip's last, partial block
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) + 4 * h >= valid) s[i] = -1e30f;
            }
            // ---- block maximum; first / last operations compiler-visible (MFMA -> VALU and VALU -> permlane wait states) ----
            float pm = fmaxf(s[0], s[1]);
            pm = a2_max3(pm, s[2], s[3]);
            pm = a2_max3(pm, s[4], s[5]);
            pm = a2_max3(pm, s[6], s[7]);
            pm = a2_max3(pm, s[8], s[9]);
            pm = a2_max3(pm, s[10], s[11]);
            pm = a2_max3(pm, s[12], s[13]);
            pm = fmaxf(pm, fmaxf(s[14], s[15]));
            float plo, phi;
            a2_halves(pm, plo, phi);
            const float pmax = fmaxf(plo, phi);  // relative to m_ref
            if (kt == 0 || __any(pmax > kA2Thr)) {  // rare after the first block: move the reference maximum
                const float delta = kt == 0 ? pmax : fmaxf(pmax, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                o0 *= alpha;
                o1 *= alpha;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] -= delta;
                m_ref += delta;
                negm_b = h == 0 ? -m_ref : 0.f;
            }
            // ---- p = 2^(s - m_ref), row sums; register i of s is k-step i of P.V ----
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
            float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                ls0 += s[i];
                ls1 += s[i + 1];
            }
            l_run += ls0 + ls1;
            // ---- O^T += V^T P^T: k-step i contracts keys (i&3) + 8(i>>2) (+4 in lanes 32-63); the groups of four k-steps
            // (8 keys) past the clip's end in its last, partial block are skipped (their p are exact zeros) ----
            const int ngrp = valid >= 32 ? 4 : (valid + 7) >> 3;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < ngrp) {
                    const float4 v0 = *reinterpret_cast<const float4*>(B0 + v_base + 32 * (m ^ v_xm));
                    const float4 v1 = *reinterpret_cast<const float4*>(B0 + v_base + 4096 + 32 * (m ^ v_xm));
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.x, s[4 * m], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.x, s[4 * m], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.y, s[4 * m + 1], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.y, s[4 * m + 1], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.z, s[4 * m + 2], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.z, s[4 * m + 2], o1, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.w, s[4 * m + 3], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.w, s[4 * m + 3], o1, 0, 0, 0);
                }
            }
        }
        if (kt + 1 < ntiles) stage((kt + 1) & 1);
        __syncthreads();
    }
    float llo, lhi;
    a2_halves(l_run, llo, lhi);
    const float l_tot = llo + lhi;
    const float inv = 1.0f / l_tot;
    if (q_row < T) {
        if (lse && h == 0)  // natural-log units; the two terms are large and nearly cancel in fp32: one float64 expression per query
            lse[(long long)bh * T + q_row] = (float)(((double)m_ref + log2((double)l_tot)) * 0.69314718055994531);
        float* dst = out + (row0 + q_row) * 768 + hd * 64 + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            *reinterpret_cast<float4*>(dst + 8 * g4) =
                make_float4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
            *reinterpret_cast<float4*>(dst + 32 + 8 * g4) =
                make_float4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
        }
    }
}

inline hipError_t launch_attention_f32_v2(const float* qkv, float* out, float* lse, int B, int T, const int* tpref, hipStream_t s,
                                          int t_min = 0) {
    const int nqblk = (T + 127) / 128;
    hipLaunchKernelGGL(attention_f32_v2_kernel, dim3(nqblk * B * 12), dim3(256), attn_f32_v2_lds(), s, qkv, out, lse, T, nqblk, tpref, t_min);
    return hipGetLastError();
}

}  // namespace nomad
