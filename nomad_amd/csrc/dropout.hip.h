// Counter-based dropout for the fine-tuning step (model.train() in /root/reference/src/training/train_triplet.py:113).
// fairseq applies dropout 0.1 after the encoder LayerNorm, after out_proj and fc2, on the attention probabilities
// (attention_dropout 0.1) and after post_extract_proj (dropout_input 0.1).  The keep decision of element `idx` at
// `site` is a pure function of (seed, site, idx): the backward recomputes every mask instead of storing it, and
// the CPU oracle restates the same hash to reproduce the masks bit for bit (torch's own RNG stream cannot be
// reproduced on a different device, here or in the reference).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace nomad {

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t dropout_bits(uint32_t seed_lo, uint32_t seed_hi, uint32_t site,
                                                          unsigned long long idx) {
    uint32_t h = fmix32((uint32_t)idx ^ seed_lo ^ (site * 0x9E3779B9u));
    h = fmix32(h + (uint32_t)(idx >> 32) * 0x85EBCA77u + seed_hi);
    return h;
}

// keep <=> bits >= threshold, threshold = round(p * 2^32); scale = 1 / (1 - p)
struct DropCfg {
    uint32_t seed_lo, seed_hi, threshold;
    float scale;
};
// multiplier of element idx: scale if kept, 0 if dropped
__device__ __forceinline__ float drop_mult(const DropCfg& d, uint32_t site, unsigned long long idx) {
    return dropout_bits(d.seed_lo, d.seed_hi, site, idx) >= d.threshold ? d.scale : 0.f;
}

// Site numbering: 0 = dropout_input, 1 = after the encoder LayerNorm, layer l: 2+3l attention probabilities
// (idx = ((b*12+h)*T + q)*T + k), 3+3l after out_proj, 4+3l after fc2 (idx = m*768 + c for [M][768] tensors).
constexpr uint32_t kSiteInput = 0, kSiteEncoder = 1;
__host__ __device__ constexpr uint32_t site_attn(int l) { return 2 + 3 * l; }
__host__ __device__ constexpr uint32_t site_proj(int l) { return 3 + 3 * l; }
__host__ __device__ constexpr uint32_t site_ffn(int l) { return 4 + 3 * l; }

}  // namespace nomad
