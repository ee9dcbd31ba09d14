// fp32 self-attention, second generation (the reference's arithmetic: exact fp32 products, fp32 softmax):
//   ctx[b,t,h*64:(h+1)*64] = softmax_j(q[b,t,h] . k[b,j,h]) v[b,j,h]        (fairseq MultiheadAttention)
// on qkv[B*T][2304] = [q*64^-0.5 | k | v] fp32.  Same structure as attention_bf16_v2.hip.h.  Matrix shape: v_mfma_f32_16x16x4_f32
// since the end of round 4 (the operand maps of THAT shape are in the second comment block below; this first block describes the
// structure, which the shape change kept):
//   * 128 queries per workgroup, 32 per wave; both products TRANSPOSED (S^T = K Q^T, O^T += V^T P^T), so a lane owns its
//     queries' scores in registers: maximum and sum are register trees + lane swaps, and the score accumulator's registers ARE the
//     B operand of P.V - no conversion, no LDS trip;
//   * the contraction order of S^T is free as well: a lane reads K and Q as float4 chunks and one float4 feeds four k-steps;
//     V is staged TRANSPOSED ([d][key]) so that one float4 holds a lane's A operands of four k-steps.  Eight ds_read_b128 per
//     operand and 32-key block, 128 MFMAs;
//   * scores in log2 units (q * log2 e at load), reference maximum subtracted by the matrix core (the C operand of a block's first
//     MFMA is -m_ref), moved only when a block exceeds it by 2^kA2Thr: per score one v_exp and one v_add remain;
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

// Round 4 (end): the products moved from v_mfma_f32_32x32x2_f32 to v_mfma_f32_16x16x4_f32 - the same rate with half the accumulator
// register traffic per multiply-add; under the 32x32x2 version the chip held 2.03-2.14 GHz (profiles/r04_attention_f32_timeline.txt),
// and the fp32 GEMM gained 3 % of clock from the same change (gemm_f32.hip.h, M16).  The structure above carries over with
// lane = (fi = lane & 15, g = lane >> 4): a 32-key x 32-query block is 2 x 2 sub-blocks s[sk][sq], register r of which is
// S^T[key 16 sk + 4 g + r][query 16 sq + fi]: a lane owns TWO queries (sq = 0, 1) and 8 of a block's scores for each, the four lanes
// fi + 16 g share a query (maximum / sum: register tree + v_permlane16_swap + v_permlane32_swap), and register r of s[sk][sq] IS the B
// operand of P.V's k-step (sk, r), which contracts keys 16 sk + r + {0, 4, 8, 12}.  K and Q are read as float4 chunks d = 16 j + 4 g .. + 3
// (k-step (j, e) contracts d = 16 j + e + {0, 4, 8, 12}), V^T as chunks of keys 16 sk + 4 g .. + 3: eight ds_read_b128 per operand and
// 32-key tile as before, 64 + 64 MFMAs of half the length (+ 4 for the reference maximum until round 5).  The clip's last, partial key tile skips its second 16-key half
// entirely (scores and P.V) when it is empty.
// lse (nullable): [B*12][T] natural-log log-sum-exp of every score row (the backward recomputes P from it).
// grid: 1-D, ceil(T / 128) * B * 12 workgroups of 256 threads; dynamic LDS attn_f32_v2_lds().
// tpref (nullable): ragged batches - clip b owns rows tpref[b] .. tpref[b+1]-1; T is then the longest clip's.
__device__ __forceinline__ float f2_max4(float x) {   // maximum over the four lanes fi, fi + 16, fi + 32, fi + 48, in all of them
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    float lo, hi;
    a2_halves(y, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float f2_sum4(float x) {   // sum over the same four lanes (fixed order: rows 0+1, 2+3, then the halves)
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    float lo, hi;
    a2_halves(y, lo, hi);
    return lo + hi;
}

// EXTV (true in every product launch): the K / V fragments as ext-vector loads; false = float4 struct copies, in front of which hipcc
// waits for the next tile's LDS-DMA (A/B runs of libnomad_diag.so, NOMAD_F32_ATTN_STRUCT_LOADS=1).
template <bool EXTV = true, bool VT4 = true>
__global__ __launch_bounds__(256, 3) void attention_f32_v2_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  float* __restrict__ lse, int T, int nqblk,
                                                                  const int* __restrict__ tpref, int t_min) {
    extern __shared__ __attribute__((aligned(16))) char f2_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, g = lane >> 4;
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
    const int q_base = qb * 128 + wave * 32;
    float4 qf[2][4];  // Q[query 16 sq + fi][16 j + 4 g .. + 3] * log2(e)
#pragma unroll
    for (int sq = 0; sq < 2; ++sq) {
        const int q_row = q_base + 16 * sq + fi;
        const float* qp = src_bh + (long long)(q_row < T ? q_row : T - 1) * 2304 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(qp + 16 * j);
            qf[sq][j] = make_float4(v.x * kLog2e, v.y * kLog2e, v.z * kLog2e, v.w * kLog2e);
        }
    }
    f32x4 o[4][2];  // O^T: d = 16 sd + 4 g + r, query 16 sq + fi
#pragma unroll
    for (int sd = 0; sd < 4; ++sd)
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) o[sd][sq] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_ref[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
    // -m_ref enters the score chain as the C operand of its first MFMA (round 5; a lane's four accumulator registers of a sub-block
    // belong to ONE query): the accumulator comes out as s - m_ref exactly as with the extra k-step ones x (-m_ref) this replaced
    // (0 + 1 x (-m_ref) is exact), at four MFMAs less per 32-key block
    f32x4 negm[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const bool wave_active = q_base < T;
    // a clip's last query wave may own fewer than 17 queries (T = 199: queries 192 .. 198): its second sub-block then has no query at
    // all and is left out of both products (round 5; wave-uniform, and the other sub-block's arithmetic is untouched: bit-identical)
    const int nsq = q_base + 16 < T ? 2 : 1;
    const int ntiles = (T + kF2KT - 1) / kF2KT;

    // ---- staging: thread -> (key row, float4 chunk) x 2 of the 32 x 64 tile, for K and for V (unchanged) ----
    // K goes global -> LDS by LDS-DMA (the instruction fills 1 KB per wave linearly, so the thread fetches the LOGICAL chunk that the XOR
    // swizzle maps to its slot: chunk ^ (row & 15)); only V, which is stored transposed, passes through registers.
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
            if (VT4) {
                // Round 6: lanes l, l + 16, l + 32, l + 48 of a wave hold keys 4 kq .. 4 kq + 3 of the same four d: a 4 x 4 transpose across
                // them (two v_permlane16_swap, two v_permlane32_swap) leaves lane group k with the four KEYS of d = 4 c16 + k - one chunk
                // of V^T, one ds_write_b128 instead of four ds_write_b32 whose 16 lanes of a key hit 4 banks (PMC: 3.9 % of the kernel's
                // cycles were LDS bank conflicts, all from these stores).  Data movement only: the same bytes land in the same places.
                const auto ab = __builtin_amdgcn_permlane16_swap(__float_as_uint(vreg[i].x), __float_as_uint(vreg[i].y), false, false);
                const auto cd = __builtin_amdgcn_permlane16_swap(__float_as_uint(vreg[i].z), __float_as_uint(vreg[i].w), false, false);
                const auto ac = __builtin_amdgcn_permlane32_swap(ab[0], cd[0], false, false);
                const auto bd = __builtin_amdgcn_permlane32_swap(ab[1], cd[1], false, false);
                const int kq = row >> 2, k = row & 3, d = 4 * c16 + k;
                f32x4 v4 = {__uint_as_float(ac[0]), __uint_as_float(bd[0]), __uint_as_float(ac[1]), __uint_as_float(bd[1])};
                *reinterpret_cast<f32x4*>(B0 + kF2KT * 256 + d * 128 + 16 * (kq ^ ((d >> 1) & 7))) = v4;
                continue;
            }
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
    // K row (16 sk + fi): 256 B, logical chunk 4 j + g at physical chunk ^ fi; V^T row d = 16 sd + fi: 128 B, logical chunk 4 sk + g
    // (keys 16 sk + 4 g .. + 3) at physical chunk ^ ((fi >> 1) & 7)
    const int k_base = fi * 256, v_base = kF2KT * 256 + fi * 128, v_x = (fi >> 1) & 7;

    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) fetch(kt + 1);
        if (wave_active) {
            const char* B0 = f2_lds + (kt & 1) * kF2Buf;
            const int valid = T - kt * kF2KT;
            const int nsk = valid > 16 ? 2 : 1;   // the clip's last block may have an empty second half
            // ---- S^T - m_ref (log2 units): 2 x 2 sub-blocks of 16 keys x 16 queries ----
            f32x4 s[2][2];
#pragma unroll
            for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                for (int sq = 0; sq < 2; ++sq) s[sk][sq] = negm[sq];
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                if (sk < nsk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (an ext-vector load, not a float4 struct copy: the struct's load carries no alias information and hipcc then
                        // puts s_waitcnt vmcnt(0) in front of it - "the LDS-DMA issued above may alias" - which serialises the next
                        // tile's fetch with this tile's products; profiles/NOTEBOOK.md, round 5)
                        float4 kf;
                        if (EXTV) {
                            const f32x4 kv = *reinterpret_cast<const f32x4*>(B0 + k_base + sk * 4096 + 16 * ((4 * j + g) ^ fi));
                            kf = make_float4(kv[0], kv[1], kv[2], kv[3]);
                        } else {
                            kf = *reinterpret_cast<const float4*>(B0 + k_base + sk * 4096 + 16 * ((4 * j + g) ^ fi));
                        }
#pragma unroll
                        for (int sq = 0; sq < 2; ++sq) {
                            if (sq < nsq) {
                                s[sk][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[sq][j].x, s[sk][sq], 0, 0, 0);
                                s[sk][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[sq][j].y, s[sk][sq], 0, 0, 0);
                                s[sk][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[sq][j].z, s[sk][sq], 0, 0, 0);
                                s[sk][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[sq][j].w, s[sk][sq], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (valid < 32) {  // the clip's last, partial block (an empty second half stays out of everything below through nsk)
#pragma unroll
                for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * sk + 4 * g + r >= valid) {
                            s[sk][0][r] = -1e30f;
                            s[sk][1][r] = -1e30f;
                        }
            }
            // ---- block maximum per query (8 scores in this lane, 4 lanes per query).  Whether ANY query's maximum is above the threshold is
            // decided on the lanes' own maxima (round 5: no lane exchange in the common block); a rescale reduces over the query's lanes ----
            float pm_l[2];
#pragma unroll
            for (int sq = 0; sq < 2; ++sq) {
                float pm = fmaxf(s[0][sq][0], s[0][sq][1]);
                pm = a2_max3(pm, s[0][sq][2], s[0][sq][3]);
                pm = a2_max3(pm, s[1][sq][0], s[1][sq][1]);
                pm_l[sq] = fmaxf(pm, fmaxf(s[1][sq][2], s[1][sq][3]));
            }
            if (kt == 0 || __any(fmaxf(pm_l[0], pm_l[1]) > kA2Thr)) {  // rare after the first block: move the reference maxima
#pragma unroll
                for (int sq = 0; sq < 2; ++sq) {
                    const float pmax_q = f2_max4(pm_l[sq]);   // relative to m_ref, the same in the query's four lanes
                    const float delta = kt == 0 ? pmax_q : fmaxf(pmax_q, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int sd = 0; sd < 4; ++sd) o[sd][sq] *= alpha;
                    l_run[sq] *= alpha;
#pragma unroll
                    for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[sk][sq][r] -= delta;
                    m_ref[sq] += delta;
                    negm[sq] = (f32x4){-m_ref[sq], -m_ref[sq], -m_ref[sq], -m_ref[sq]};
                }
            }
            // ---- p = 2^(s - m_ref), row sums; register r of s[sk][sq] is k-step (sk, r) of P.V ----
#pragma unroll
            for (int sq = 0; sq < 2; ++sq) {
#pragma unroll
                for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[sk][sq][r] = __builtin_amdgcn_exp2f(s[sk][sq][r]);
                l_run[sq] += ((s[0][sq][0] + s[0][sq][1]) + (s[0][sq][2] + s[0][sq][3])) + ((s[1][sq][0] + s[1][sq][1]) + (s[1][sq][2] + s[1][sq][3]));
            }
            // ---- O^T += V^T P^T: k-step (sk, r) contracts keys 16 sk + r + {0, 4, 8, 12} ----
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                if (sk < nsk) {
#pragma unroll
                    for (int sd = 0; sd < 4; ++sd) {
                        float4 vf;
                        if (EXTV) {
                            const f32x4 vv = *reinterpret_cast<const f32x4*>(B0 + v_base + sd * 2048 + 16 * ((4 * sk + g) ^ v_x));
                            vf = make_float4(vv[0], vv[1], vv[2], vv[3]);
                        } else {
                            vf = *reinterpret_cast<const float4*>(B0 + v_base + sd * 2048 + 16 * ((4 * sk + g) ^ v_x));
                        }
#pragma unroll
                        for (int sq = 0; sq < 2; ++sq) {
                            if (sq < nsq) {
                                o[sd][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, s[sk][sq][0], o[sd][sq], 0, 0, 0);
                                o[sd][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, s[sk][sq][1], o[sd][sq], 0, 0, 0);
                                o[sd][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, s[sk][sq][2], o[sd][sq], 0, 0, 0);
                                o[sd][sq] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, s[sk][sq][3], o[sd][sq], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        if (kt + 1 < ntiles) stage((kt + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int sq = 0; sq < 2; ++sq) {
        const float l_tot = f2_sum4(l_run[sq]);
        const float inv = 1.0f / l_tot;
        const int q_row = q_base + 16 * sq + fi;
        if (q_row < T) {
            if (lse && g == 0)  // natural-log units; the two terms are large and nearly cancel in fp32: one float64 expression per query
                lse[(long long)bh * T + q_row] = (float)(((double)m_ref[sq] + log2((double)l_tot)) * 0.69314718055994531);
            float* dst = out + (row0 + q_row) * 768 + hd * 64 + 4 * g;
#pragma unroll
            for (int sd = 0; sd < 4; ++sd)
                *reinterpret_cast<float4*>(dst + 16 * sd) = make_float4(o[sd][sq][0] * inv, o[sd][sq][1] * inv, o[sd][sq][2] * inv, o[sd][sq][3] * inv);
        }
    }
}

template <bool EXTV = true, bool VT4 = true>
inline hipError_t launch_attention_f32_v2(const float* qkv, float* out, float* lse, int B, int T, const int* tpref, hipStream_t s,
                                          int t_min = 0) {
    const int nqblk = (T + 127) / 128;
    hipLaunchKernelGGL((attention_f32_v2_kernel<EXTV, VT4>), dim3(nqblk * B * 12), dim3(256), attn_f32_v2_lds(), s, qkv, out, lse, T, nqblk, tpref, t_min);
    return hipGetLastError();
}

}  // namespace nomad
