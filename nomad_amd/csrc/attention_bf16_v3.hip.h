// bf16 self-attention for long clips on the 16-wide matrix shape (round 5; BASELINE config C5: 30 s, T = 1499).
//
// attention_bf16_v2.hip.h computes both products with v_mfma_f32_32x32x16_bf16; the chip holds 1994 MHz of its 2400 under that
// kernel (profiles/NOTEBOOK.md, "Open at the end of round 4") - the 32-wide shape moves twice the accumulator registers per
// multiply-add, the same effect that cost the fp32 GEMM and the fp32 attention 3-4 % of clock until they moved to 16x16x4.
// Same algorithm here on v_mfma_f32_16x16x32_bf16, with lane = (fr = lane & 15, fq = lane >> 4):
//   * a wave owns 32 queries as two 16-query sub-blocks qs; a 32-key block is two 16-key sub-blocks kb.  Both products are
//     TRANSPOSED: S^T[kb][qs] = K[kb] Q[qs]^T leaves lane (fr, fq) the scores of query fr for keys 4 fq + r, and those four registers of
//     kb = 0 and kb = 1, converted to bf16 in place, ARE the B operand of O^T[db][qs] += V^T[db] P^T[qs] once the contraction slot
//     (fq, j) of that product is DEFINED as key 16 (j >> 2) + 4 fq + (j & 3): no lane exchange, no LDS round trip;
//   * the matching A operand V^T (rows d = 16 db + fr, slots (fq, j)) is two ds_read_b64_tr_b16 per db: lane group fq gathers
//     4 key rows x 16 d columns, which is exactly keys 4 fq .. 4 fq + 3 (j < 4) and 16 + 4 fq .. (j >= 4) of the row-major V tile;
//   * the reference maximum m_ref enters as the C operand of the first score MFMA (a register quad holding -m_ref): the
//     accumulator comes out as s - m_ref at no cost - the 32-wide kernel spent a fifth k-step (ones x (-m_ref)) on that, 11 % of
//     its matrix time;
//   * the deferred rescale is decided on the lanes' OWN maxima (any lane above the threshold <=> some query's maximum above it), so
//     the common block has no cross-lane operation at all; only a rescale reduces over the four lanes of a query
//     (v_permlane16_swap + v_permlane32_swap);
//   * K rows keep the GEMM's chunk swizzle ((row >> 1) & 7); V rows are swizzled in 32-byte pairs by (row >> 1) & 3, which spreads
//     the 8 rows x 32 bytes a 32-lane half gathers per transposing read over all 64 banks once.
// Scores in log2 units (q carries log2 e), p = 2^(s - m_ref) <= 2^kA2Thr between rescales, fp32 accumulation of O and of the row sums, K / V
// tiles of KT keys double-buffered by LDS-DMA with one barrier per tile: as attention_bf16_v2.hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "attention_bf16_v2.hip.h"
#include "attention_f32_v2.hip.h"
#include "gemm_f32.hip.h"

namespace nomad {

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) (the transposing reads below take
// their LDS offset as an instruction immediate, so the block index has to be a constant expression).
template <class F, int... I>
__device__ __forceinline__ void a3_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void a3_static_for(F&& f) {
    a3_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Two transposing reads (keys 4 fq .. and 16 + 4 fq .. of one d block) through inline asm, and the wait that goes with them.
// WHY asm (round 5, last finding): __builtin_amdgcn_ds_read_tr16_b64 carries no alias information, so hipcc orders it against every
// LDS-DMA in flight - "the DMA issued at the top of the tile may write what this reads" - with s_waitcnt vmcnt(0) in front of the first
// V read of the tile: the NEXT tile's fetch, issued to overlap this tile's products, was waited for before this tile's first P.V (the
// double buffer makes the wait unnecessary; the K reads, plain ext-vector loads with alias information, never got it).  The compiler
// does not see an asm's LDS traffic: the waits are explicit - LDS operations return in order, so "all but the N newest" covers every
// older read whatever else the scheduler has put in between; the compiler's own lgkmcnt values for its own reads can only over-wait.
template <int OFF>
__device__ __forceinline__ void a3_tr_pair(bf16x4& lo, bf16x4& hi, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(OFF), "n"(OFF + 2048));
}
template <int N>
__device__ __forceinline__ void a3_lds_wait(bf16x4& lo, bf16x4& hi) {   // (the operands tie the MFMAs that use them behind the wait)
    static_assert(N == 0 || N == 2, "lgkmcnt");
    if (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo), "+v"(hi));
}

// grid: 1-D, workgroups of 64 * NW threads, one per work item; dynamic LDS attn_bf16_v2_lds(KT).  The work items of a batch are the
// nqblk * B * 12 pairs (clip-head bh, query block qb), item = bh * nqblk + qb, nqblk = ceil(T / (16 QS NW)); a launch covers items
// virt0 .. virt0 + gridDim.x - 1 (round 6: the items of a sparsely filled last round go to a second launch with half-size workgroups,
// nomad_hip.hip run_attention_bf16).
// tpref (nullable): ragged batches - clip b owns rows tpref[b] .. tpref[b+1]-1 of qkv / out; T is then the longest clip's.
// q must carry the factor log2(e) (nomad_enable_bf16 folds it into the q rows of the QKV weight).
// QS: 16-query sub-blocks per wave (2 or 4).  Round 5, second step: with 32 queries per wave the kernel is co-limited by LDS bandwidth - a
// 32-key block costs a wave 8 KB of fragment reads (4 ds_read_b128 of K, 8 transposing reads of V) for 16 MFMAs, and sixteen such waves
// per CU read 128 KB per block step = 1024 cycles at 128 bytes per clock, exactly the 4 x 256 cycles the four waves of a SIMD spend in
// MFMAs - which is why the 16-wide shape alone (QS = 2: 11 % fewer matrix cycles, no per-block lane exchange) changed nothing in the
// forward (profiles/r05_c5_layer_table.txt).  QS = 4 reuses every K / V fragment for twice the queries: half the LDS bytes per MFMA, at
// 2 waves per SIMD instead of 4 (about 190 registers).
// ASMV: the V reads through a3_tr_pair (false: the builtin, A/B runs of libnomad_diag.so).
// HOLD (round 6): -m_ref moved in place and the ones operand / V addresses held in registers (false: round 5's form, A/B runs).
template <int NW, int KT, int OCC, int QS = 2, bool ASMV = true, bool HOLD = true>
__global__ __launch_bounds__(64 * NW, OCC) void attention_bf16_v3_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                         int T, int nqblk, const int* __restrict__ tpref, int virt0) {
    extern __shared__ __attribute__((aligned(16))) char a3_lds[];
    constexpr int NT = 64 * NW;       // threads
    constexpr int QW = 16 * QS;       // queries per wave
    constexpr int QB = QW * NW;       // queries per workgroup
    constexpr int NB = KT / 32;       // 32-key blocks per tile
    constexpr int NCH = KT * 8 / NT;  // 16-byte chunks of K (and of V) each thread stages per tile
    static_assert(NCH >= 1 && NCH * NT == KT * 8, "tile rows must divide over the threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // (the LDS-DMA destination goes through M0: a scalar)
    const int fr = lane & 15, fq = lane >> 4;
    // XCD-aware placement: the query blocks of one head run on one XCD (their K / V re-reads are L2 hits)
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    const int virt = virt0 + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int bh = virt / nqblk, qb = virt - bh * nqblk;
    const int b = bh / 12, hd = bh - b * 12;
    long long row0 = (long long)b * T;
    if (tpref) {
        row0 = tpref[b];
        T = tpref[b + 1] - tpref[b];
    }
    if (qb * QB >= T) return;  // whole workgroup, before any barrier
    const bf16_t* __restrict__ src_bh = qkv + row0 * 2304 + hd * 64;
    const int q_row0 = qb * QB + wave * QW + fr;   // + 16 qs
    bf16x8 qf[QS][2];   // B operand of S^T: query fr of sub-block qs, d = 32 ks + 8 fq .. + 7
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int qr = q_row0 + 16 * qs;
        const bf16_t* qp = src_bh + (long long)(qr < T ? qr : T - 1) * 2304 + 8 * fq;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qs][ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
    }
    f32x4 o[4][QS];   // O^T[db][qs]: d = 16 db + 4 fq + r, query fr of sub-block qs
    // -m_ref (m_ref: reference maximum of this lane's queries in log2 units, the same in the four lanes of a query): the C operand of the
    // first score MFMA.  Moved IN PLACE by a rescale (negm -= delta, which is -(m_ref + delta) bit for bit): round 6 - rebuilt from a scalar
    // m_ref the quad was a fresh value on the rare path, and hipcc kept a second copy of both quads for the blocks behind the branch,
    // refreshed by four v_mov_b64 in EVERY block, and for want of those 8 registers re-made the ones operand and the four V addresses per
    // block as well (10 of the 48 vector instructions of a block, in a loop bound by vector-instruction issue).
    f32x4 negm[QS];
    float m_ref[QS];    // (HOLD = false only)
    // Row sums on the matrix core (round 5): lsum[qs] += ones[16 x 32] P^T[qs] - every row of the result is the sum of the block's 32
    // values of p for query fr, over ALL four lanes' contraction slots.  Two MFMAs per block instead of 16 v_add_f32 + a final lane
    // reduction: the kernel is bound by vector-instruction ISSUE (an MFMA holds the SIMD's issue for 8 cycles, a v_add for 4; see the
    // NOTEBOOK), so this is 48 issue cycles less per block and wave.  The sum is over the bf16-rounded p the numerator uses as well.
    f32x4 lsum[QS];
    bf16x8 ones_f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones_f[j] = (bf16_t)1.0f;
    // (laundered, like va[] below: hipcc re-makes a constant operand and an address sum in every block rather than hold them - 2 v_mov_b64 and
    // 4 v_add_u32 per block, in a loop bound by vector-instruction issue, with registers to spare since the -m_ref quads are moved in place)
    if (HOLD) asm volatile("" : "+v"(ones_f));
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db][qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
        negm[qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m_ref[qs] = 0.f;
        lsum[qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float inf_v;   // +inf, opaque to the compiler (see the block maxima below)
    asm volatile("v_mov_b32 %0, 0x7f800000" : "=v"(inf_v));
    const bool wave_active = qb * QB + wave * QW < T;  // wave-uniform: the transposing reads need a full EXEC mask
    const int ntiles = (T + KT - 1) / KT;

    // ---- staging by LDS-DMA: a wave's instruction fills 1 KB = 8 rows linearly; lane (row l >> 3, physical chunk l & 7) fetches the
    // LOGICAL chunk the swizzle maps there (K: chunk ^ ((row >> 1) & 7); V: chunk ^ 2 ((row >> 1) & 3)) ----
    // (tiles that lie wholly inside the clip - all but the last - take a wave-uniform tile base + per-thread 32-bit offsets computed once:
    // the per-tile 64-bit address arithmetic of the general form is vector-instruction issue the kernel is short of)
    unsigned k_voff[NCH], v_voff[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cid = tid + i * NT, row = cid >> 3, ch = cid & 7;
        k_voff[i] = (unsigned)((row * 2304 + 768 + 8 * (ch ^ ((row >> 1) & 7))) * 2);
        v_voff[i] = (unsigned)((row * 2304 + 1536 + 8 * (ch ^ (2 * ((row >> 1) & 3)))) * 2);
    }
    auto fetch = [&](int kt) {
        if (kt * KT + KT <= T) {
            const char* tile = reinterpret_cast<const char*>(uniform_ptr(reinterpret_cast<const float*>(src_bh + (long long)kt * KT * 2304)));
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                char* dk = a3_lds + ((kt & 1) * (KT * 256)) + (i * NT + wave_u * 64) * 16;
                // (laundered: hoisted out of the loop as 64-bit values the offsets no longer match the scalar-base + 32-bit-offset form of
                // the instruction, and every DMA costs a 64-bit vector add)
                unsigned ko = k_voff[i], vo = v_voff[i];
                asm volatile("" : "+v"(ko), "+v"(vo));
                __builtin_amdgcn_global_load_lds((gptr_t)(tile + ko), (lptr_t)dk, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(tile + vo), (lptr_t)(dk + KT * 128), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cid = tid + i * NT, row = cid >> 3, ch = cid & 7;
            int key = kt * KT + row;
            key = key < T ? key : T - 1;
            const bf16_t* src = src_bh + (long long)key * 2304;
            char* dk = a3_lds + ((kt & 1) * (KT * 256)) + (i * NT + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 768 + 8 * (ch ^ ((row >> 1) & 7))), (lptr_t)dk, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 1536 + 8 * (ch ^ (2 * ((row >> 1) & 3)))), (lptr_t)(dk + KT * 128), 16, 0, 0);
        }
    };
    // ---- fragment read addresses (bytes inside a buffer, 32-key block 0) ----
    // K (A operand of S^T): lane (key row 16 kb + fr, chunk 4 ks + fq) -> physical chunk (4 ks + fq) ^ ((fr >> 1) & 7)
    const int kswz = (fr >> 1) & 7;
    const int k_off0 = fr * 128 + 16 * ((0 + fq) ^ kswz), k_off1 = fr * 128 + 16 * ((4 + fq) ^ kswz);   // ks = 0 / 1; + 2048 kb
    // V (A operand of O^T): within the 16-lane group fq, lane 4 q4 + p4 addresses key row 16 half + 4 fq + q4, d columns 16 db + 4 p4 ..:
    // byte 32 (db ^ x) + 8 p4 of the row, x = (2 (fq & 1) + (q4 >> 1)) & 3 the row's pair swizzle
    const int q4 = fr >> 2, p4 = fr & 3;
    const int vx = (2 * (fq & 1) + (q4 >> 1)) & 3;
    const int v_row = KT * 128 + (4 * fq + q4) * 128 + 8 * p4;   // + 2048 half + 4096 blk
    int v_db[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) v_db[db] = v_row + 32 * (db ^ vx);

    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) fetch(kt + 1);  // in flight during the whole compute phase, into the buffer the last barrier released
        if (wave_active) {
            const char* B0 = a3_lds + (kt & 1) * (KT * 256);
            const int left = T - kt * KT;  // valid keys from this tile on
            const int nb = left >= KT ? NB : (left + 31) >> 5;
            unsigned va[4];   // LDS byte addresses of this lane's V rows in the tile (+ 4096 blk + 2048 half as instruction offsets)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                va[db] = (unsigned)(size_t)(lptr_t)(a3_lds) + (unsigned)((kt & 1) * (KT * 256) + v_db[db]);
                if (HOLD) asm volatile("" : "+v"(va[db]));
            }
            a3_static_for<NB>([&](auto blk_c) {
                constexpr int blk = decltype(blk_c)::value;
                if (blk < nb) {
                    // ---- scores: S^T[kb][qs] - m_ref = K[kb] Q[qs]^T + (-m_ref) ----
                    f32x4 s[2][QS];
                    {
                        const char* kp = B0 + blk * 4096;
                        bf16x8 kf[2][2];
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            kf[kb][0] = *reinterpret_cast<const bf16x8*>(kp + kb * 2048 + k_off0);
                            kf[kb][1] = *reinterpret_cast<const bf16x8*>(kp + kb * 2048 + k_off1);
                        }
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int qs = 0; qs < QS; ++qs) {
                                s[kb][qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][0], qf[qs][0], negm[qs], 0, 0, 0);
                                s[kb][qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][1], qf[qs][1], s[kb][qs], 0, 0, 0);
                            }
                    }
                    const int valid = left - blk * 32;
                    if (valid < 32) {  // the clip's last, partial block: keys past its end drop out of the softmax
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (16 * kb + 4 * fq + r >= valid) {
#pragma unroll
                                    for (int qs = 0; qs < QS; ++qs) s[kb][qs][r] = -1e30f;
                                }
                    }
                    // ---- this lane's maxima (relative to m_ref); a rescale only when some lane is above the threshold ----
                    float pm[QS], pall = 0.f;
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs) {
                        // The FIRST read of each accumulator quad is a compiler-visible instruction (v_med3_f32 with +inf = the maximum of
                        // two; the +inf sits in a register the compiler cannot see through, or it folds the median back into a maxnum): hipcc places the MFMA -> VALU wait states for it and pads nothing around inline asm; the rest are v_max3
                        // through asm.  (fmaxf() there would cost two more instructions per quad: the compiler canonicalises each input of
                        // a maxnum on fresh MFMA results with v_max_f32 x, x - in a loop bound by vector-instruction issue.)
                        const float t0 = __builtin_amdgcn_fmed3f(s[0][qs][0], s[0][qs][1], inf_v);
                        const float t1 = __builtin_amdgcn_fmed3f(s[1][qs][0], s[1][qs][1], inf_v);
                        float m = a2_max3(t0, s[0][qs][2], s[0][qs][3]);
                        m = a2_max3(m, t1, s[1][qs][2]);
                        pm[qs] = a2_max3(m, s[1][qs][3], s[1][qs][3]);
                        pall = qs == 0 ? pm[0] : a2_max3(pall, pm[qs], pm[qs]);
                    }
                    const bool first = (kt == 0 && blk == 0);
                    if (first || __any(pall > kA2Thr)) {  // rare after the first block: move the reference maxima
#pragma unroll
                        for (int qs = 0; qs < QS; ++qs) {
                            const float pmax = f2_max4(pm[qs]);   // over the four lanes of the query: the same value in all of them
                            const float delta = first ? pmax : fmaxf(pmax, 0.f);
                            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                            for (int db = 0; db < 4; ++db) o[db][qs] *= alpha;
                            lsum[qs] *= alpha;
#pragma unroll
                            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                                for (int r = 0; r < 4; ++r) s[kb][qs][r] -= delta;
                            if (HOLD) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) negm[qs][r] -= delta;
                            } else {
                                m_ref[qs] += delta;
                                negm[qs] = (f32x4){-m_ref[qs], -m_ref[qs], -m_ref[qs], -m_ref[qs]};
                            }
                        }
                    }
                    // ---- p = 2^(s - m_ref), row sums, P^T[qs] as the B operand (slot (fq, j) = key 16 (j >> 2) + 4 fq + (j & 3)) ----
                    bf16x8 pf[QS];
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pf[qs][r] = (bf16_t)__builtin_amdgcn_exp2f(s[0][qs][r]);
                            pf[qs][4 + r] = (bf16_t)__builtin_amdgcn_exp2f(s[1][qs][r]);
                        }
                        lsum[qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pf[qs], lsum[qs], 0, 0, 0);
                    }
                    // ---- O^T[db][qs] += V^T[db] P^T[qs] ----
                    if (ASMV) {   // two d blocks' reads in flight ahead of the MFMAs that consume them
                        bf16x4 vlo[4], vhi[4];
                        a3_tr_pair<blk * 4096>(vlo[0], vhi[0], va[0]);
                        a3_tr_pair<blk * 4096>(vlo[1], vhi[1], va[1]);
#pragma unroll
                        for (int db = 0; db < 4; ++db) {
                            if (db < 3) a3_lds_wait<2>(vlo[db], vhi[db]);   // the pair issued after this one may still be in flight
                            else a3_lds_wait<0>(vlo[db], vhi[db]);
                            const bf16x8 vf = __builtin_shufflevector(vlo[db], vhi[db], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                            for (int qs = 0; qs < QS; ++qs) o[db][qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qs], o[db][qs], 0, 0, 0);
                            if (db == 0) a3_tr_pair<blk * 4096>(vlo[2], vhi[2], va[2]);
                            if (db == 1) a3_tr_pair<blk * 4096>(vlo[3], vhi[3], va[3]);
                        }
                    } else {
                        const char* vp = B0 + blk * 4096;
#pragma unroll
                        for (int db = 0; db < 4; ++db) {
                            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vp + v_db[db]));
                            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vp + 2048 + v_db[db]));
                            bf16x8 vf;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                vf[j] = v0[j];
                                vf[4 + j] = v1[j];
                            }
#pragma unroll
                            for (int qs = 0; qs < QS; ++qs) o[db][qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qs], o[db][qs], 0, 0, 0);
                        }
                    }
                }
            });
        }
        if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the next tile has landed
        __syncthreads();
    }
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const float inv = 1.0f / lsum[qs][0];
        const int qr = q_row0 + 16 * qs;
        if (qr < T) {
            bf16_t* dst = out + (row0 + qr) * 768 + hd * 64 + 4 * fq;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                store4<bf16_t>(dst + 16 * db, make_float4(o[db][qs][0] * inv, o[db][qs][1] * inv, o[db][qs][2] * inv, o[db][qs][3] * inv));
        }
    }
}

// Items item0 .. item0 + nitems - 1 of a batch cut into query blocks of 16 QS NW queries, nqblk per clip-head (nqblk may exceed
// ceil(T / (16 QS NW)): blocks past the clip's end return at once).
template <int NW, int KT, int OCC, int QS = 2, bool ASMV = true, bool HOLD = true>
inline hipError_t launch_attention_bf16_v3_items(const bf16_t* qkv, bf16_t* out, int T, int nqblk, const int* tpref, hipStream_t s, int item0,
                                                 int nitems) {
    static LdsAttrOnce configured;
    auto kern = attention_bf16_v3_kernel<NW, KT, OCC, QS, ASMV, HOLD>;
    constexpr int lds = attn_bf16_v2_lds(KT);
    if (hipError_t e = configured.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    if (nitems <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(nitems), dim3(64 * NW), lds, s, qkv, out, T, nqblk, tpref, item0);
    return hipGetLastError();
}

template <int NW, int KT, int OCC, int QS = 2, bool ASMV = true, bool HOLD = true>
inline hipError_t launch_attention_bf16_v3(const bf16_t* qkv, bf16_t* out, int B, int T, const int* tpref, hipStream_t s, int item0 = 0,
                                           int nitems = -1) {
    const int nqblk = (T + 16 * QS * NW - 1) / (16 * QS * NW);
    if (nitems < 0) nitems = nqblk * B * 12 - item0;
    return launch_attention_bf16_v3_items<NW, KT, OCC, QS, ASMV, HOLD>(qkv, out, T, nqblk, tpref, s, item0, nitems);
}

}  // namespace nomad
