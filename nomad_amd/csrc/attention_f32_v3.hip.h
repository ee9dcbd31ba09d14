// EXPERIMENT (libnomad_diag.so only; NOMAD_ATTN_PIPE=1 routes the fp32 attention of diagnostic builds here): a persistent,
// software-pipelined variant of attention_f32_v2_kernel.  Correct (tests/test_gpu_kernels.py pass with it), NOT faster: 0.46 vs 0.43 ms
// event-timed on the bench shape (256 clips x T = 199), slower at few long clips (T = 1499, B = 8: 0.66 vs 0.55 ms).  Kept for its timing
// probes - ablation switches and per-workgroup stamps (tools/attn_one.py, tools/attn_timeline.py; profiles/r04_attention_f32_ablations.txt,
// r04_attention_f32_timeline.txt) - which are what is known about the kernel's bound:
//   * the CUs run at 2.03-2.14 GHz under this kernel, not 2.4: its MFMAs alone are 0.27-0.29 ms of the 0.43;
//   * the rest does not overlap with them: a kernel without MFMAs takes 0.13-0.16 ms, without one of the two products 0.28, and the
//     matrix pipe of a CU is 53 % busy with ONE workgroup resident and only 71 % with three;
//   * issue is oldest-wave-first: of a CU's three persistent workgroups the first finishes its equal share after 280 us, the last
//     after 425 us; handing the items out dynamically evens the lives out (364-443 us) without shortening the kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "attention_f32_v2.hip.h"

namespace nomad {

// The same arithmetic per score; persistent workgroups, software-pipelined across key tiles AND across work items.
// Ablations of attention_f32_v2_kernel on the bench shape (profiles/r04_attention_f32_ablations.txt): its MFMAs alone account
// for 0.29 of its 0.43 ms (event-timed); a kernel with every MFMA, the softmax and the key loop's staging removed still takes
// 0.087 ms - the life of a workgroup outside its key loop (launch, the Q loads, the first K / V tile, the output stores: a chain of
// memory latencies) - and the parts ADD UP: the three workgroups of a CU start together, take equally long and so stay in phase,
// idling the matrix pipe through their prologues together.  Here:
//   * one workgroup per occupancy slot (3 per CU) walks a list of (clip, head, query block) items; inside an item's LAST key tile
//     (which needs neither Q nor a K buffer any more) it issues the next item's Q loads, its first two K tiles (LDS-DMA) and its first
//     V tile, so the next item's score products start right behind the current item's output stores;
//   * a wave overlaps its softmax of tile t with the 32 product MFMAs of S(t + 1) (a dependent accumulation chain: one MFMA per 64
//     cycles with the VALU work in its shadow); products first, "- m_ref" last: S(t+1) = (sum_d k_d q_d) + (-m_ref) with the
//     reference maximum as it stands AFTER tile t (v2 put that k-step first: the sums differ from v2's in the last bit; the kernel
//     is as batch-invariant as v2 - a query's arithmetic depends on its clip alone);
//   * K tiles triple-buffered (tile t + 2 lands while t + 1 is read), V^T double-buffered: 40 KB per workgroup;
//   * the items are handed out DYNAMICALLY, from one atomic counter per XCD (a head pair's query blocks stay on one XCD, its K / V
//     in that L2): the issue arbiter serves the oldest wave first, so of a CU's three workgroups the first launched runs ~1.5x faster
//     than the last (per-workgroup stamps, profiles/r04_attention_f32_timeline.txt) - with equal static shares the kernel ended 25 %
//     after its average workgroup.  Wave w takes the 32-query sub-block (w + i) & 3 of the workgroup's i-th item: the idle wave of a
//     clip's partial last query block (T = 199: 71 queries) moves over the SIMDs.
constexpr int kF3K = kF2KT * 256, kF3V = 64 * 128;
constexpr int attn_f32_v3_lds() { return 3 * kF3K + 2 * kF3V; }

// ABL (diagnostic builds, NOMAD_ATTN_ABLATE): 1 no softmax arithmetic, 2 no P.V products, 4 no barrier / staging inside the loop,
// 8 no global loads inside the loop, 16 no score products - timing probes, wrong results.
// n_items = B * 12 * nqblk; grid: min(n_items, 3 * CUs) workgroups of 256 threads; queue: 8 counters, zero at launch.
template <int ABL>
__global__ __launch_bounds__(256, 3) void attention_f32_v3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  float* __restrict__ lse, int T_uniform, int nqblk,
                                                                  const int* __restrict__ tpref, int t_min, int n_items, int* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) char f2_lds[];
    __shared__ int q_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float ones_a = h == 0 ? 1.f : 0.f;
    char* const vt_lds = f2_lds + 3 * kF3K;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // ---- the item queue of this workgroup's XCD (workgroup id & 7): whole head pairs, n_items / 8 rounded up ----
    const int xq = blockIdx.x & 7;
    const int per = ((n_items / nqblk + 7) >> 3) * nqblk;
    const int seg_lo = xq * per;
    const int seg_n = n_items - seg_lo < per ? n_items - seg_lo : per;
    struct Item {
        const float* src;   // qkv row 0 of the clip, this head's q columns
        long long row0;
        int T, qb, bh;
    };
    auto decode = [&](int idx, Item& it) -> bool {
        const int bh = idx / nqblk, qb = idx - bh * nqblk;
        const int b = bh / 12, hd = bh - b * 12;
        it.T = T_uniform;
        it.row0 = (long long)b * T_uniform;
        if (tpref) {
            it.row0 = tpref[b];
            it.T = tpref[b + 1] - tpref[b];
        }
        it.qb = qb;
        it.bh = bh;
        it.src = qkv + it.row0 * 2304 + hd * 64;
        return qb * 128 < it.T && it.T >= t_min;   // (ragged batches: the clip has no such query block / is left to attention_f32_kernel)
    };
    // thread 0: `got` is a ticket already drawn; returns the first valid item at or after it (drawing more tickets), or -1
    auto settle = [&](int got) -> int {
        Item tmp;
        while (got < seg_n && !decode(seg_lo + got, tmp)) got = atomicAdd(&queue[xq], 1);
        return got < seg_n ? seg_lo + got : -1;
    };

    float4 qf[8];  // Q[q][8j + 4h .. + 3] * log2(e)
    float4 vreg[2];
    auto load_q = [&](const Item& it, int sub) {
        const int q_row = it.qb * 128 + sub * 32 + r;
        const float* qp = it.src + (long long)(q_row < it.T ? q_row : it.T - 1) * 2304 + 4 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = *reinterpret_cast<const float4*>(qp + 8 * j);
    };
    auto scale_q = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = make_float4(qf[j].x * kLog2e, qf[j].y * kLog2e, qf[j].z * kLog2e, qf[j].w * kLog2e);
    };
    auto fetch_k = [&](const Item& it, int kt, int kbuf) {   // tile kt -> K buffer kbuf, by LDS-DMA (source-side XOR swizzle, as in v2)
        char* B0 = f2_lds + kbuf * kF3K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cid = tid + i * 256, row = cid >> 4, pc = cid & 15;
            int key = kt * kF2KT + row;
            key = key < it.T ? key : it.T - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(it.src + (long long)key * 2304 + 768 + 4 * (pc ^ (row & 15))), (lptr_t)(B0 + i * 4096 + wave_u * 1024), 16, 0, 0);
        }
    };
    auto fetch_v = [&](const Item& it, int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cid = tid + i * 256, row = cid >> 4, pc = cid & 15;
            int key = kt * kF2KT + row;
            key = key < it.T ? key : it.T - 1;
            vreg[i] = *reinterpret_cast<const float4*>(it.src + (long long)key * 2304 + 1536 + pc * 4);
        }
    };
    auto stage_v = [&](int vbuf) {
        char* B0 = vt_lds + vbuf * kF3V;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cid = tid + i * 256, row = cid >> 4, c16 = cid & 15;
            const int kq = row >> 2, ke = row & 3;
            char* vt = B0 + 4 * ke;
            const int d0 = 4 * c16, x0 = (d0 >> 1) & 7, x1 = x0 + 1;
            *reinterpret_cast<float*>(vt + (d0 + 0) * 128 + 16 * (kq ^ x0)) = vreg[i].x;
            *reinterpret_cast<float*>(vt + (d0 + 1) * 128 + 16 * (kq ^ x0)) = vreg[i].y;
            *reinterpret_cast<float*>(vt + (d0 + 2) * 128 + 16 * (kq ^ x1)) = vreg[i].z;
            *reinterpret_cast<float*>(vt + (d0 + 3) * 128 + 16 * (kq ^ x1)) = vreg[i].w;
        }
    };
    const int k_base = r * 256 + 16 * (h ^ (r & 1)), k_x = (r >> 1) & 7;
    const int v_x = (r >> 1) & 7;
    const int v_base = r * 128 + 16 * (h ^ (v_x & 1)), v_xm = v_x >> 1;

    // four chunks (16 k-steps) of the product chain of one tile
    auto products = [&](f32x16& acc, const char* KB, int j0) {
#pragma unroll
        for (int j = j0; j < j0 + 4; ++j) {
            const float4 kf = *reinterpret_cast<const float4*>(KB + k_base + 32 * (j ^ k_x));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[j].w, acc, 0, 0, 0);
        }
    };

    if (tid == 0) q_next = settle(atomicAdd(&queue[xq], 1));
    __syncthreads();
    const int first = __builtin_amdgcn_readfirstlane(q_next);
    if (first < 0) return;   // whole workgroup
    Item cur;
    decode(first, cur);
    int seq = 0;             // items this workgroup has begun
    unsigned long long ts_wall0 = 0, ts_clk0 = 0;
    if (ABL & 32) {   // timing probe: {wall start, wall end, shader clock start, shader clock end, HW_ID | XCC_ID << 32, SIMD id per wave (bytes)}
        ts_wall0 = wall_clock64();
        ts_clk0 = clock64();
    }
    int vflip = 0;         // V^T buffer of the current item's tile kt: (kt + vflip) & 1
    // ---- the first item's prologue (every later item's is issued inside its predecessor's last tile) ----
    load_q(cur, (wave + seq) & 3);
    fetch_k(cur, 0, 0);
    if (cur.T > kF2KT) fetch_k(cur, 1, 1);
    fetch_v(cur, 0);
    stage_v(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    scale_q();
    __syncthreads();

    for (;;) {
        const int T = cur.T;
        const int sub = (wave + seq) & 3;
        const int q_row = cur.qb * 128 + sub * 32 + r;
        const bool wave_active = cur.qb * 128 + sub * 32 < T;
        const int ntiles = (T + kF2KT - 1) / kF2KT;
        f32x16 o0, o1;  // O^T: d = 32*dblk + (i&3) + 8(i>>2) + 4h, this lane's query
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o0[i] = 0.f;
            o1[i] = 0.f;
        }
        float m_ref = 0.f, l_run = 0.f;
        float negm_b = 0.f;
        f32x16 s;  // S^T(kt) - m_ref, log2 units: keys (i&3) + 8(i>>2) + 4h of the tile, this lane's query
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        if (wave_active) {
            products(s, f2_lds, 0);
            products(s, f2_lds, 4);
        }
        int kb1 = 1;   // K buffer of tile kt + 1
        // the next item's ticket: drawn now, settled (and published through LDS) behind the first key tile, read at the last one
        int ticket = 0;
        if (tid == 0) ticket = atomicAdd(&queue[xq], 1);
        // one key tile: softmax of s (tile kt) next to the products of tile kt + 1 (NEXT), then P.V of tile kt
        auto tile = [&](int kt, auto next_c) {
            constexpr bool NEXT = decltype(next_c)::value;
            const char* KB = f2_lds + kb1 * kF3K;
            const char* VB = vt_lds + ((kt + vflip) & 1) * kF3V;
            f32x16 sn;
#pragma unroll
            for (int i = 0; i < 16; ++i) sn[i] = 0.f;
            if (NEXT && !(ABL & 16)) products(sn, KB, 0);
            const int valid = T - kt * kF2KT;
            if (!NEXT && valid < 32) {  // the clip's last, partial block
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) + 4 * h >= valid) s[i] = -1e30f;
            }
            if (!(ABL & 1)) {
                // block maximum; first / last operations compiler-visible (MFMA -> VALU and VALU -> permlane wait states)
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
            }
            if (NEXT && !(ABL & 16)) products(sn, KB, 4);
            if (!(ABL & 1)) {
                // p = 2^(s - m_ref), row sums; register i of s is k-step i of P.V
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
                float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    ls0 += s[i];
                    ls1 += s[i + 1];
                }
                l_run += ls0 + ls1;
            } else {
                l_run += s[0];
            }
            if (NEXT) sn = __builtin_amdgcn_mfma_f32_32x32x2f32(ones_a, negm_b, sn, 0, 0, 0);   // ... - m_ref, as it stands after this tile
            // O^T += V^T P^T: k-step i contracts keys (i&3) + 8(i>>2) (+4 in lanes 32-63); the groups of four k-steps (8 keys) past
            // the clip's end in its last, partial block are skipped (their p are exact zeros)
            const int ngrp = (ABL & 2) ? 0 : ((NEXT || valid >= 32) ? 4 : (valid + 7) >> 3);
            if (ABL & 2) o0[0] += s[3] + s[7];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < ngrp) {
                    const float4 v0 = *reinterpret_cast<const float4*>(VB + v_base + 32 * (m ^ v_xm));
                    const float4 v1 = *reinterpret_cast<const float4*>(VB + v_base + 4096 + 32 * (m ^ v_xm));
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
            s = sn;
        };
        for (int kt = 0; kt + 1 < ntiles; ++kt) {
            const int kb2 = kb1 == 2 ? 0 : kb1 + 1;
            if (!(ABL & 12)) {
                if (kt + 2 < ntiles) fetch_k(cur, kt + 2, kb2);   // (the buffer tile kt - 1 was read from, two barriers ago)
                fetch_v(cur, kt + 1);                              // (behind the DMA: waiting for these registers waits for it too)
            }
            if (wave_active) tile(kt, std::true_type{});
            if (kt == 0 && tid == 0) q_next = settle(ticket);
            if (!(ABL & 4)) {
                stage_v((kt + 1 + vflip) & 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            kb1 = kb2;
        }
        // ---- the last key tile, with the next item's first loads in flight: the score products of this item are done (no wave reads
        // Q or a K buffer any more), its last V^T tile sits in buffer (ntiles - 1 + vflip) & 1 - the next item's first goes to the other ----
        if (ntiles == 1 || (ABL & 4)) {   // (no barrier since the ticket was drawn)
            if (ntiles == 1 && tid == 0) q_next = settle(ticket);
            __syncthreads();
        }
        Item nxt;
        const int nidx = __builtin_amdgcn_readfirstlane(q_next);
        const int nseq = nidx >= 0 ? seq + 1 : -1;
        if (nidx >= 0) decode(nidx, nxt);
        const int nflip = (ntiles + vflip) & 1;
        if (nseq >= 0) {
            load_q(nxt, (wave + nseq) & 3);
            fetch_k(nxt, 0, 0);
            if (nxt.T > kF2KT) fetch_k(nxt, 1, 1);
            fetch_v(nxt, 0);
        }
        if (wave_active) tile(ntiles - 1, std::false_type{});
        if (nseq >= 0) {
            stage_v(nflip);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (before this item's output stores are issued: nothing but the loads above to wait for)
            scale_q();
        }
        float llo, lhi;
        a2_halves(l_run, llo, lhi);
        const float l_tot = llo + lhi;
        const float inv = 1.0f / l_tot;
        if (q_row < T) {
            if (lse && h == 0)  // natural-log units; the two terms are large and nearly cancel in fp32: one float64 expression per query
                lse[(long long)cur.bh * T + q_row] = (float)(((double)m_ref + log2((double)l_tot)) * 0.69314718055994531);
            const int hd = cur.bh % 12;
            float* dst = out + (cur.row0 + q_row) * 768 + hd * 64 + 4 * h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                *reinterpret_cast<float4*>(dst + 8 * g4) =
                    make_float4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
                *reinterpret_cast<float4*>(dst + 32 + 8 * g4) =
                    make_float4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
            }
        }
        if (nseq < 0) break;
        __syncthreads();   // the next item's K tiles 0 / 1 and V^T tile 0 are in LDS
        cur = nxt;
        seq = nseq;
        vflip = nflip;
    }
    if ((ABL & 32) && blockIdx.x < kTimelineSlots) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = g_timeline + (size_t)blockIdx.x * 6;
        if (lane == 0) reinterpret_cast<unsigned char*>(o + 5)[wave] = (unsigned char)((hw >> 4) & 3);
        if (tid == 0) {
            o[0] = ts_wall0;
            o[1] = wall_clock64();
            o[2] = ts_clk0;
            o[3] = clock64();
            o[4] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        }
    }
}

// queue: 8 ints of device memory that no other launch in flight uses (zeroed here, on the stream)
inline hipError_t launch_attention_f32_v3(const float* qkv, float* out, float* lse, int B, int T, const int* tpref, hipStream_t s,
                                          int* queue, int t_min = 0) {
    const int nqblk = (T + 127) / 128;
    static const int slots = [] {   // 3 workgroups per CU (168 VGPRs, 40 KB of LDS)
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const char* v = getenv("NOMAD_ATTN_SLOTS");   // (experiment: workgroups per CU)
        return (v && atoi(v) > 0 ? atoi(v) : 3) * cus;
    }();
    const int n_items = nqblk * B * 12;
    const int grid = n_items < slots ? n_items : slots;
    hipError_t qe = hipMemsetAsync(queue, 0, 8 * sizeof(int), s);
    if (qe != hipSuccess) return qe;
    static const int abl = [] { const char* v = getenv("NOMAD_ATTN_ABLATE"); return v ? atoi(v) : 0; }();
#define NOMAD_ABL_CASE(A) case A: hipLaunchKernelGGL(attention_f32_v3_kernel<A>, dim3(grid), dim3(256), attn_f32_v3_lds(), s, qkv, out, lse, T, nqblk, tpref, t_min, n_items, queue); return hipGetLastError();
    switch (abl) {
        NOMAD_ABL_CASE(1) NOMAD_ABL_CASE(2) NOMAD_ABL_CASE(16) NOMAD_ABL_CASE(18) NOMAD_ABL_CASE(19) NOMAD_ABL_CASE(23) NOMAD_ABL_CASE(32)   // (every instantiation is ~10 s of build time: add the combination you need)
        default: break;
    }
#undef NOMAD_ABL_CASE
    hipLaunchKernelGGL(attention_f32_v3_kernel<0>, dim3(grid), dim3(256), attn_f32_v3_lds(), s, qkv, out, lse, T, nqblk, tpref, t_min, n_items, queue);
    return hipGetLastError();
}

}  // namespace nomad
