// bf16 self-attention for long clips (BASELINE config C5: 30 s, T = 1499; 12 heads x 64, no mask, eval mode):
//   ctx[b,t,h*64:(h+1)*64] = softmax_j(q[b,t,h] . k[b,j,h]) v[b,j,h]        (fairseq MultiheadAttention)
// on qkv[B*T][2304] = [q*64^-0.5 | k | v] bf16 (the fused QKV GEMM's output), fp32 accumulation and softmax statistics.
//
// Round 1's kernel (now tools/micro/attention_bf16_r1.hip.h, kept for A/B runs) gave 64 query rows to a workgroup, so at
// T = 1499 every head's K and V (384 KB) were staged 24 times, and its 16x16x32 tiles left each lane 16 scores of FOUR
// different statistics groups; measured 520-550 TFLOP/s = 0.22 of the bf16 peak, 21 % of the C5 step.  Here
//   * a workgroup is NW waves x 32 query rows of one (clip, head), and the workgroups of a head are placed on one XCD
//     (1-D grid, XCD-aware remap) so that their K / V re-reads are L2 hits;
//   * both products are v_mfma_f32_32x32x16_bf16, computed TRANSPOSED (S^T = K Q^T, O^T += V^T P^T): a lane owns ONE
//     query (column lane & 31) and 16 of a 32-key block's scores (rows (i&3) + 8(i>>2) + 4(lane>>5)), so the row maximum
//     and sum are plain register trees plus one v_permlane32_swap between the two lane halves, and the score
//     accumulator, converted to bf16 in place (registers 8s..8s+7 = k-step s), IS the B operand of the second product -
//     no LDS round trip; the matching A operand V^T is gathered from the row-major V tile by ds_read_b64_tr_b16;
//   * scores are in log2 units (q carries log2 e) and the reference maximum m_ref, kept bf16-exact, is subtracted by the
//     matrix core (a fifth k-step ones x (-m_ref) in the score chain), so p = 2^acc needs no VALU arithmetic; m_ref moves
//     only when a block's maximum exceeds it by more than kA2Thr (deferred rescale: p <= 2^kA2Thr instead of <= 1,
//     harmless in fp32 accumulation).  The common block costs per score one v_exp, one add (row sum), a third of a
//     v_max3 and half a v_cvt_pk;
//   * keys past the end of the clip are set to -1e30 in the one partial block only (a wave-uniform branch): no masked
//     copy of the block code;
//   * K / V tiles of KT keys are double-buffered in LDS with ONE barrier per tile: the next tile's global loads are in
//     flight during the whole compute phase and are written to the other buffer just before the barrier (DMA = false), or go
//     straight into it by LDS-DMA (DMA = true: what the forward uses since round 3, with KT = 128).  Rows are
//     128 B, unpadded, with the 16-byte chunk index XOR-swizzled by the row ((row>>1)&7 for K, 4*((row>>1)&1) for V):
//     the ds_read_b128 lane groups of the K fragments and the 4-row x 64-B blocks of the transposing V reads each
//     cover all 64 banks once.
#pragma once
#include <hip/hip_runtime.h>

#include "attention.hip.h"

namespace nomad {

constexpr float kA2Thr = 8.0f;  // deferred-rescale threshold in log2 units (p <= 256: exact enough in bf16 P / fp32 accumulation)
constexpr float kLog2e = 1.44269504088896341f;

constexpr int attn_bf16_v2_lds(int KT) { return 2 * KT * 256; }  // two buffers of K [KT][128 B] + V [KT][128 B]

__device__ __forceinline__ float a2_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float a2_add(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float a2_mul(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// p = 2^(s * log2e - mc)
__device__ __forceinline__ float a2_p(float s, float mc) {
    float t;
    asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(t) : "v"(s), "s"(kLog2e), "v"(mc));
    return __builtin_amdgcn_exp2f(t);
}

// {x of lanes 0-31, x of lanes 32-63}, each in all lanes of... the caller combines the two (max / sum)
__device__ __forceinline__ void a2_halves(float x, float& lo, float& hi) {
    const unsigned u = __float_as_uint(x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lo = __uint_as_float(sw[0]);
    hi = __uint_as_float(sw[1]);
}

// grid: 1-D, nqblk * B * 12 workgroups of 64 * NW threads (nqblk = ceil(T / (32 NW))); dynamic LDS attn_bf16_v2_lds(KT).
// tpref (nullable): ragged batches - clip b owns rows tpref[b] .. tpref[b+1]-1 of qkv / out; T is then the longest clip's.
// OCC: waves per SIMD the register allocation must allow (__launch_bounds__' second argument).
// LOG2E: q already carries the factor log2(e) (folded into the bf16 q weights: nomad_enable_bf16); otherwise the Q
// fragments are scaled (and re-rounded to bf16) when they are loaded.
// DMA: K / V tiles go global -> LDS by global_load_lds (no VGPR round trip, no ds_write): a wave's instruction fills 1 KB = 8 rows
// linearly, so lane (row l >> 3, physical chunk l & 7) fetches the LOGICAL chunk the swizzle maps there; the next tile's DMA is
// issued at the top of an iteration into the buffer the previous iteration's barrier released and waited for (vmcnt(0)) just
// before this iteration's barrier.
template <int NW, int KT, int OCC, bool LOG2E, bool DMA = false>
__global__ __launch_bounds__(64 * NW, OCC) void attention_bf16_v2_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                         int T, int nqblk, const int* __restrict__ tpref) {
    extern __shared__ __attribute__((aligned(16))) char a2_lds[];
    constexpr int NT = 64 * NW;       // threads
    constexpr int QB = 32 * NW;       // queries per workgroup
    constexpr int NB = KT / 32;       // 32-key blocks per tile
    constexpr int NCH = KT * 8 / NT;  // 16-byte chunks of K (and of V) each thread stages per tile
    static_assert(NCH >= 1 && NCH * NT == KT * 8, "tile rows must divide over the threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware placement: workgroups id, id + 8, ... share an XCD (round-robin dispatch); give each XCD a contiguous
    // run of (head, query-block) pairs so that the query blocks of one head re-read its K / V from the same L2
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
    if (qb * QB >= T) return;  // whole workgroup, before any barrier
    const bf16_t* __restrict__ src_bh = qkv + row0 * 2304 + hd * 64;
    const int q_row = qb * QB + wave * 32 + r;
    bf16x8 qf[4];
    {
        const bf16_t* qp = src_bh + (long long)(q_row < T ? q_row : T - 1) * 2304 + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
            if (!LOG2E) {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16_t)((float)qf[ks][j] * kLog2e);
            }
        }
    }
    f32x16 o0, o1;  // O^T: d = 32*dblk + (i&3) + 8(i>>2) + 4h, this lane's query
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        o0[i] = 0.f;
        o1[i] = 0.f;
    }
    // Reference maximum m_ref of this lane's query, in log2 units and exactly representable in bf16: it enters the score
    // chain as a FIFTH k-step, ones[key][k=0] x (-m_ref)[k=0][query], so the accumulator comes out as s - m_ref and the
    // exponential needs no subtraction.  (Any reference close to the row maximum will do - the softmax is invariant to
    // it - as long as the same one is used for every block between two rescales.)
    float m_ref = 0.f;
    float l_run = 0.f;  // sum of p over this lane half's keys
    bf16x8 onesf, mf;   // A: ones[key r][k = 8h + j] = (k == 0); B: (-m_ref)[k = 8h + j][query] = (k == 0) ? -m_ref : 0
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        onesf[j] = (bf16_t)0.f;
        mf[j] = (bf16_t)0.f;
    }
    if (h == 0) onesf[0] = (bf16_t)1.f;
    const bool wave_active = qb * QB + wave * 32 < T;  // wave-uniform: the transposing reads need a full EXEC mask
    const int ntiles = (T + KT - 1) / KT;

    // ---- staging: thread -> (row, chunk) of the tile; LDS position of chunk c of row w is w*128 + 16*(c ^ swz(w)) ----
    bf16x8 kreg[NCH], vreg[NCH];
    int st_k[NCH], st_v[NCH];  // byte offsets inside a buffer's K / V image
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cid = tid + i * NT, row = cid >> 3, ch = cid & 7;
        st_k[i] = row * 128 + 16 * (ch ^ ((row >> 1) & 7));
        st_v[i] = KT * 128 + row * 128 + 16 * (ch ^ (4 * ((row >> 1) & 1)));
    }
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cid = tid + i * NT, row = cid >> 3, ch = cid & 7;
            int key = kt * KT + row;
            key = key < T ? key : T - 1;
            if (DMA) {   // ch is the PHYSICAL chunk of this lane's 16 bytes; destination: wave-uniform base + lane * 16
                const bf16_t* src = src_bh + (long long)key * 2304;
                char* dk = a2_lds + ((kt & 1) * (KT * 256)) + (i * NT + wave * 64) * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + 768 + 8 * (ch ^ ((row >> 1) & 7))), (lptr_t)dk, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(src + 1536 + 8 * (ch ^ (4 * ((row >> 1) & 1)))), (lptr_t)(dk + KT * 128), 16, 0, 0);
            } else {
                const bf16_t* src = src_bh + (long long)key * 2304 + ch * 8;
                kreg[i] = *reinterpret_cast<const bf16x8*>(src + 768);
                vreg[i] = *reinterpret_cast<const bf16x8*>(src + 1536);
            }
        }
    };
    auto stage = [&](int buf) {
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the next tile has landed
            return;
        }
        char* B0 = a2_lds + buf * (KT * 256);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *reinterpret_cast<bf16x8*>(B0 + st_k[i]) = kreg[i];
            *reinterpret_cast<bf16x8*>(B0 + st_v[i]) = vreg[i];
        }
    };
    // ---- fragment read addresses (bytes inside a buffer, block 0) ----
    // K (A operand of S^T): lane (key r, h) reads chunk 2*ks + h of row r -> position (2*ks + h) ^ ((r>>1)&7)
    const int kswz = (r >> 1) & 7;
    const int k_base = r * 128 + 16 * (h ^ (kswz & 1));
    const int k_x = kswz >> 1;  // k-step ks sits at 32 * (ks ^ k_x)
    // V (A operand of O^T): lane (d = 32*dblk + lane&31, h), element j <-> key 16*s2 + 8(j>>2) + 4h + (j&3): two
    // transposing reads (4 key rows x 16 d columns per 16-lane group; lane 4q+p addresses row q, columns 4p..4p+3).
    // chunk of (dblk, G&1, p) = 4*dblk + 2*(G&1) + (p>>1), swizzled by 4*((row>>1)&1) = 4*(q>>1): dblk ^ (q>>1)
    const int i16 = lane & 15, G = lane >> 4, q4 = i16 >> 2, p4 = i16 & 3;
    const int v_base = KT * 128 + (4 * h + q4) * 128 + 32 * (G & 1) + 8 * p4;
    const int v_d0 = 64 * (q4 >> 1), v_d1 = 64 - v_d0;  // byte offset of d-block 0 / 1 in this lane's rows

    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) fetch(kt + 1);  // in flight during the whole compute phase
        if (wave_active) {
            const char* B0 = a2_lds + (kt & 1) * (KT * 256);
            const int left = T - kt * KT;  // valid keys from this tile on
            const int nb = left >= KT ? NB : (left + 31) >> 5;
            // scores of one 32-key block: S^T - m_ref = K Q^T + ones (-m_ref)^T for 32 keys x 32 queries (log2 units)
            auto scores = [&](int blk) -> f32x16 {
                f32x16 s;
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = 0.f;
                const char* kp = B0 + blk * 4096 + k_base;
                bf16x8 kf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 32 * (ks ^ k_x));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(onesf, mf, s, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
                return s;
            };
            // softmax + P.V of one block
            auto block = [&](f32x16& s, int blk) {
                const int valid = left - blk * 32;
                if (valid < 32) {  // the clip's last, partial block: keys past its end drop out of the softmax
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if ((i & 3) + 8 * (i >> 2) + 4 * h >= valid) s[i] = -1e30f;
                }
                // ---- block maximum (this lane's query) + the other lane half.  The FIRST and LAST operations on the fresh
                // MFMA results / around the lane exchange are compiler-visible, so that hipcc places the MFMA -> VALU and
                // VALU -> permlane wait states itself (it pads nothing around inline asm) ----
                {
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
                    const bool first = (kt == 0 && blk == 0);
                    if (first || __any(pmax > kA2Thr)) {  // rare after the first block: move the reference maximum
                        const float m_new = (float)(bf16_t)(m_ref + (first ? pmax : fmaxf(pmax, 0.f)));  // bf16-exact
                        const float delta = m_new - m_ref;
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        o0 *= alpha;
                        o1 *= alpha;
                        l_run *= alpha;
#pragma unroll
                        for (int i = 0; i < 16; ++i) s[i] -= delta;
                        m_ref = m_new;
                        mf[0] = (bf16_t)(h == 0 ? -m_new : 0.f);
                    }
                }
                // ---- p = 2^(s - m_ref), row sums, P^T as the B operand (k-step s2 = registers 8*s2 .. 8*s2+7) ----
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
                float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    ls0 += s[i];
                    ls1 += s[i + 1];
                }
                l_run += ls0 + ls1;
                bf16x8 pf0, pf1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pf0[j] = (bf16_t)s[j];
                    pf1[j] = (bf16_t)s[8 + j];
                }
                // ---- O^T += V^T P^T ----
                const char* vp = B0 + blk * 4096 + v_base;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const char* p0 = vp + s2 * 2048;
                    const bf16x4 a00 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0 + v_d0));
                    const bf16x4 a01 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0 + 1024 + v_d0));
                    const bf16x4 a10 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0 + v_d1));
                    const bf16x4 a11 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p0 + 1024 + v_d1));
                    bf16x8 vf0, vf1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf0[j] = a00[j];
                        vf0[4 + j] = a01[j];
                        vf1[j] = a10[j];
                        vf1[4 + j] = a11[j];
                    }
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, s2 ? pf1 : pf0, o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, s2 ? pf1 : pf0, o1, 0, 0, 0);
                }
            };
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if (blk < nb) {
                    f32x16 s = scores(blk);
                    block(s, blk);
                }
            }
        }
        if (kt + 1 < ntiles) stage((kt + 1) & 1);
        __syncthreads();
    }
    float llo, lhi;
    a2_halves(l_run, llo, lhi);
    const float inv = 1.0f / (llo + lhi);
    if (q_row < T) {
        bf16_t* dst = out + (row0 + q_row) * 768 + hd * 64 + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            store4<bf16_t>(dst + 8 * g4, make_float4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv));
            store4<bf16_t>(dst + 32 + 8 * g4, make_float4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv));
        }
    }
}

// host-side launcher of one instantiation.  Returns the hipError of the launch.
template <int NW, int KT, int OCC, bool LOG2E, bool DMA = false>
inline hipError_t launch_attention_bf16_v2(const bf16_t* qkv, bf16_t* out, int B, int T, const int* tpref, hipStream_t s) {
    static LdsAttrOnce configured;
    auto kern = attention_bf16_v2_kernel<NW, KT, OCC, LOG2E, DMA>;
    constexpr int lds = attn_bf16_v2_lds(KT);
    if (hipError_t e = configured.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int nqblk = (T + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL(kern, dim3(nqblk * B * 12), dim3(64 * NW), lds, s, qkv, out, T, nqblk, tpref);
    return hipGetLastError();
}

}  // namespace nomad
