// Round 1's bf16 attention kernel (64 query rows per workgroup, v_mfma_f32_16x16x32_bf16, two barriers per 64-key tile),
// kept only as the A/B reference of tools/micro/attn_bf16.hip; the library uses nomad_amd/csrc/attention_bf16_v2.hip.h.
#pragma once
#include "../../nomad_amd/csrc/attention.hip.h"

namespace nomad {

template <int NSUB>
__device__ __forceinline__ void attn_tile_bf16(const char* __restrict__ Ks, const char* __restrict__ Vs,
                                               const bf16x8 (&qf)[2], f32x4 (&o)[4], float& m_run, float& l_run,
                                               int qi, int g, int valid) {
    f32x4 s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 kf[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
            kf[sub] = *reinterpret_cast<const bf16x8*>(Ks + (sub * 16 + qi) * kAttn16LD + (4 * ks + g) * 16);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[sub], qf[ks], s[sub], 0, 0, 0);
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
    // transposing-read address of this lane inside its 16-lane group: row (lane>>2)&3, columns 4*(lane&3)..+3
    const int tr_row = (qi >> 2), tr_col = (qi & 3) * 4;
#pragma unroll
    for (int pb = 0; pb < (NSUB + 1) / 2; ++pb) {
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (bf16_t)s[2 * pb][r];
            pf[4 + r] = (bf16_t)s[2 * pb + 1][r];
        }
        const char* v0 = Vs + ((2 * pb) * 16 + g * 4 + tr_row) * kAttn16LD + tr_col * 2;
        const char* v1 = v0 + 16 * kAttn16LD;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(v0 + ds * 32));
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(v1 + ds * 32));
            bf16x8 vf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vf[r] = a0[r];
                vf[4 + r] = a1[r];
            }
            o[ds] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[ds], 0, 0, 0);
        }
    }
}

// tpref (nullable): ragged batches, as in attention_f32_kernel.
__global__ __launch_bounds__(256) void attention_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                             int T, const int* __restrict__ tpref = nullptr) {
    __shared__ __attribute__((aligned(16))) char Ks[64 * kAttn16LD];
    __shared__ __attribute__((aligned(16))) char Vs[64 * kAttn16LD];
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
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(qkv + base + (long long)q_ld * 2304 + ks * 32 + g * 8);
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const bool wave_active = blockIdx.x * 64 + wave * 16 < T;
    const int ntiles = (T + 63) / 64;
    bf16x8 kreg[2], vreg[2];  // next tile, prefetched under the current tile's MFMAs
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256, row = id >> 3, ch = id & 7;
            int key = kt * 64 + row;
            key = key < T ? key : T - 1;
            const bf16_t* src = qkv + base + (long long)key * 2304 + ch * 8;
            kreg[i] = *reinterpret_cast<const bf16x8*>(src + 768);
            vreg[i] = *reinterpret_cast<const bf16x8*>(src + 1536);
        }
    };
    fetch(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256, row = id >> 3, ch = id & 7;
            *reinterpret_cast<bf16x8*>(Ks + row * kAttn16LD + ch * 16) = kreg[i];
            *reinterpret_cast<bf16x8*>(Vs + row * kAttn16LD + ch * 16) = vreg[i];
        }
        __syncthreads();
        if (kt + 1 < ntiles) fetch(kt + 1);
        if (!wave_active) continue;  // wave-uniform: the transposing reads below need a full EXEC mask
        const int valid = T - kt * 64;
        if (valid > 48) attn_tile_bf16<4>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid);
        else if (valid > 32) attn_tile_bf16<3>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid);
        else if (valid > 16) attn_tile_bf16<2>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid);
        else attn_tile_bf16<1>(Ks, Vs, qf, o, m_run, l_run, qi, g, valid);
    }
    float l_tot = l_run + __shfl_xor(l_run, 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float inv = 1.0f / l_tot;
    if (q_row < T) {
        bf16_t* dst = out + (row0 + q_row) * 768 + h * 64 + g * 4;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
            store4<bf16_t>(dst + ds * 16, make_float4(o[ds][0] * inv, o[ds][1] * inv, o[ds][2] * inv, o[ds][3] * inv));
    }
}

}  // namespace nomad
