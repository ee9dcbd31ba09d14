// Host-side definitions shared by the translation units of libnomad_hip.so (round 6: the library is built from three of them in
// parallel - nomad_hip.hip: engine, forwards, backward, training, C ABI; nomad_gemm_f32.hip: every fp32 GEMM instantiation and its
// dispatch; nomad_gemm_bf16.hip: every bf16 / bf16x3 GEMM instantiation and its dispatch - half of the device code is fp32 GEMM
// kernels, a third bf16 GEMM kernels).  The context, the tuning switches, error reporting and the profiling scope; what crosses
// translation units is declared at the end (hidden visibility: the dynamic symbol table stays the C ABI of include/nomad_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nomad_hip.h"

#include "dtypes.hip.h"
#include "gemm_f32.hip.h"   // GemmParams, RowMap (templates: nothing is instantiated by including it)

#define NOMAD_INTERNAL __attribute__((visibility("hidden")))

using namespace nomad;

inline thread_local char g_err[512] = "";   // ONE buffer per thread for all translation units (nomad_last_error reads it)

namespace {

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) return fail(NOMAD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

constexpr int kConvK[7] = {10, 3, 3, 3, 3, 2, 2};
constexpr int kConvS[7] = {5, 2, 2, 2, 2, 2, 2};
constexpr int kSplitKLayersMaxM = 4096;   // a forward that returns the layer outputs of fewer frames than this may split K (forward_impl)
constexpr size_t kSplitKPartFloats = (size_t)4 * 512 * 64 * 64;  // 4 slices of the largest problem that is split (< 512 tiles of 64 x 64)
constexpr long long kPairScratchDoubles = 1 << 21;  // 16 MB: e.g. 16 ref tiles x 131 072 deg rows per launch pair
constexpr int kEventChunk = 8192;  // the profiling event pool grows by this many events whenever it runs out
const int* const kNoInts = nullptr;  // "uniform batch" for the kernels' optional ragged-metadata pointers
const hipStream_t kNoStream = reinterpret_cast<hipStream_t>(~uintptr_t(0));  // nomad_pairwise: a scratch block bound to no stream

}  // namespace

struct LayerDev {
    float *qkv_w, *qkv_b, *o_w, *o_b, *ln1_w, *ln1_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ln2_w, *ln2_b;
};


// Kernel-choice switches.  Every default below is the shipped configuration, and the product library (libnomad_hip.so) never
// reads the environment: "nothing but the arguments" decides what a call does.  Only libnomad_diag.so (-DNOMAD_DIAG) fills a
// context's copy from NOMAD_* environment variables, once, in nomad_create (tuning_from_env) - the A/B runs of tools/ and profiles/.
struct Tuning {
    bool splitk_lnb_fuse = true;   // NOMAD_SPLITK_LNB: a split-K dX GEMM in front of a LayerNorm backward has that kernel form its output (no epilogue launch)
    bool splitk_ln_fuse = true;    // NOMAD_SPLITK_LN: a split-K out_proj / fc2 normalises its rows in its own epilogue (splitk_epilogue_ln_kernel)
    bool splitk_posconv = true;    // NOMAD_SPLITK_POSCONV: the grouped pos-conv of the loss path splits K four ways
    bool splitk_layers = true;     // NOMAD_SPLITK_LAYERS: so do the dense GEMMs of a small layer-output forward
    bool f32_plain_epi = true;     // NOMAD_F32_PLAIN_EPI: small epilogue for plain C / R matrices
    bool f32_lean = true, f32_direct_epi = true, f32_skew = true, f32_res_ahead = true;   // NOMAD_F32_LEAN / _DIRECT_EPI / _SKEW / _RES_AHEAD
    int f32_mixed = 1;             // NOMAD_F32_MIXED: two tile shapes in one launch
    int f32_mixed_m1 = 0;          // NOMAD_F32_MIXED_M1: forced row split (diagnostics)
    int f32_mixed_slots = 0;       // NOMAD_F32_MIXED_SLOTS: 0 = two workgroup slots per CU
    double f32_mixed_min = 0.05, f32_mixed_max = 0.70;   // NOMAD_F32_MIXED_MIN / _MAX: fill of the last round that takes the split
    bool f32_mixed_prefer = false; // NOMAD_F32_MIXED_PREFER
    bool f32_quant_tile = true;    // NOMAD_F32_QUANT_TILE: tile choice by the largest tile count any CU gets
    double f32_quant_penalty = 0.0;  // NOMAD_F32_QUANT_PENALTY (percent): 0 = 8 % with two concurrent parts, 3 % alone
    bool f32_longk_33 = false;     // NOMAD_F32_LONGK_33
    int f32_mid_tile = 31;         // NOMAD_F32_MID_TILE
    bool f32_attn_vt4 = true;      // NOMAD_F32_ATTN_VT4 (diag): the fp32 attention transposes V across lanes and stores 16-byte chunks (0: four ds_write_b32, A/B)
    bool f32_attn_struct_loads = false;  // NOMAD_F32_ATTN_STRUCT_LOADS (diag): the fp32 attention's LDS fragments as float4 struct copies (A/B)
    bool bf16_posconv_slab = true;  // NOMAD_BF16_POSCONV_SLAB: the bf16 pos-conv with its input slab resident in LDS (posconv_bf16_slab.hip.h);
                                    // false: the grouped GEMM on 128 x 64 tiles it replaces (A/B)
    int bf16_attn_dma = 2;         // NOMAD_BF16_ATTN_DMA: K / V of the bf16 attention by LDS-DMA in 128-key tiles
    int bf16_attn_v3 = 2;          // NOMAD_BF16_ATTN_V3: the bf16 attention on v_mfma_f32_16x16x32_bf16 with 32 queries per wave (2, shipped);
                                   // 3: its V reads through the builtin; 5: round 5's register use; 4 / 8: 64 queries per wave, 4 / 8 waves per workgroup (A/B: no faster); 0: the 32x32x16 kernel
    int bf16_attn_tail = 0;        // NOMAD_BF16_ATTN_TAIL (diag): a last round of 256-query workgroups that is at most this many eighths full runs
                                   // as 128-query workgroups in a second launch (0: never; run_attention_bf16 - measured slower, A/B only)
    int bf16_ln_rows = 4;          // NOMAD_BF16_LN_ROWS: rows per wave of the bf16 forward's LayerNorm (4: one gamma / beta fetch per 4 rows; 1: A/B)
    bool bf16_conv0_mfma = true;   // NOMAD_BF16_CONV0_MFMA
    bool bf16_conv0_gelu_erf = false;   // NOMAD_BF16_CONV0_GELU_ERF (diag): the matrix-core conv0 with the erf GELU instead of the bf16-output one (A/B)
    int p8_min_tiles = 256;        // NOMAD_BF16_8PHASE_MIN_TILES: smallest grid (256 x 256 tiles) for the deep-pipelined bf16 kernels
    bool p8_nt_stores = true;      // NOMAD_BF16_NT_STORES
    int p8_rpre = 3;               // NOMAD_BF16_RPRE
    bool x3_plain_epi = true;      // NOMAD_X3_PLAIN_EPI
    bool p8_three_b = true;        // NOMAD_BF16_B3
    int p8_n192 = 0;               // NOMAD_BF16_N192
    bool p9 = true;                // NOMAD_BF16_P9: the persistent 256 x 256 bf16 kernel wherever it applies
    bool p9_res = true;            // NOMAD_BF16_P9_RES: residual GEMMs on the persistent kernel too (0: the one-tile-per-workgroup kernel, A/B)
    bool p9_share = false;         // NOMAD_BF16_P9_SHARE: persistent launches of concurrent batch parts share the CUs (1 / parts each)
    bool p9_tail_split = false;    // NOMAD_BF16_P9_TAIL: rows of a sparse last round of its tiles go to the 128 x 128 kernel (+4-17 % on the
                                   // N = 768 GEMMs alone, -3 % in the two-stream forward, where the other half's kernels fill that round)
    int p9_short = 1;              // NOMAD_BF16_P9_SHORT: 192-row tiles of the persistent kernel (a run-time mode of the same instantiation) where they
                                   // save more than they cost: 1 = by the round count, batches that run alone only (the N = 768 GEMMs of config C5 on one
                                   // stream), 2 = every problem, 0 = never
    bool attn_bwd_small = true;    // NOMAD_ATTN_BWD_SMALL: clips of at most 64 frames take the fused attention backward (one launch; 0: rowdot + dkv + dq)
    bool p9_wl = false;            // NOMAD_BF16_P9_WL: whole-line output stores of the persistent kernel (8 rows x 128 bytes per instruction instead of the
                                   // accumulator's 16 x 64: 5.6 against 4.2 TB/s in tools/micro/store_pattern.hip, QKV -3.5 % alone, configs[4] -0.8 %: off)
    bool p9_late = true;           // NOMAD_BF16_P9_LATE: GELU epilogue hooks of wave row 0 behind the phase barrier, concurrent with wave row 1's
    int p9_skew = 0;               // NOMAD_BF16_P9_SKEW (diag, timeline probe tile 61 only): start skew between workgroup groups, units of 10 ns
    int concurrent_parts = 1;      // nomad_set_concurrent_parts: batches the host layer runs concurrently on separate streams
};

#ifdef NOMAD_DIAG
static void tuning_from_env(Tuning& t) {
    auto geti = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
    auto getb = [](const char* n, bool d) { const char* e = getenv(n); return e ? atoi(e) != 0 : d; };
    auto getd = [](const char* n, double d) { const char* e = getenv(n); return e ? atof(e) : d; };
    t.splitk_ln_fuse = getb("NOMAD_SPLITK_LN", t.splitk_ln_fuse);
    t.splitk_posconv = getb("NOMAD_SPLITK_POSCONV", t.splitk_posconv);
    t.splitk_layers = getb("NOMAD_SPLITK_LAYERS", t.splitk_layers);
    t.f32_plain_epi = getb("NOMAD_F32_PLAIN_EPI", t.f32_plain_epi);
    t.f32_lean = getb("NOMAD_F32_LEAN", t.f32_lean);
    t.f32_direct_epi = getb("NOMAD_F32_DIRECT_EPI", t.f32_direct_epi);
    t.f32_skew = getb("NOMAD_F32_SKEW", t.f32_skew);
    t.f32_res_ahead = getb("NOMAD_F32_RES_AHEAD", t.f32_res_ahead);
    t.f32_mixed = geti("NOMAD_F32_MIXED", t.f32_mixed);
    t.f32_mixed_m1 = geti("NOMAD_F32_MIXED_M1", t.f32_mixed_m1);
    t.f32_mixed_slots = geti("NOMAD_F32_MIXED_SLOTS", t.f32_mixed_slots);
    t.f32_mixed_min = getd("NOMAD_F32_MIXED_MIN", t.f32_mixed_min);
    t.f32_mixed_max = getd("NOMAD_F32_MIXED_MAX", t.f32_mixed_max);
    t.f32_mixed_prefer = getb("NOMAD_F32_MIXED_PREFER", t.f32_mixed_prefer);
    t.f32_quant_tile = getb("NOMAD_F32_QUANT_TILE", t.f32_quant_tile);
    t.f32_quant_penalty = getd("NOMAD_F32_QUANT_PENALTY", t.f32_quant_penalty);
    t.f32_longk_33 = getb("NOMAD_F32_LONGK_33", t.f32_longk_33);
    t.f32_mid_tile = geti("NOMAD_F32_MID_TILE", t.f32_mid_tile);
    t.bf16_attn_dma = geti("NOMAD_BF16_ATTN_DMA", t.bf16_attn_dma);
    t.f32_attn_struct_loads = geti("NOMAD_F32_ATTN_STRUCT_LOADS", t.f32_attn_struct_loads) != 0;
    t.bf16_posconv_slab = geti("NOMAD_BF16_POSCONV_SLAB", t.bf16_posconv_slab) != 0;
    t.bf16_attn_v3 = geti("NOMAD_BF16_ATTN_V3", t.bf16_attn_v3);
    t.bf16_attn_tail = geti("NOMAD_BF16_ATTN_TAIL", t.bf16_attn_tail);
    t.bf16_ln_rows = geti("NOMAD_BF16_LN_ROWS", t.bf16_ln_rows);
    t.bf16_conv0_mfma = getb("NOMAD_BF16_CONV0_MFMA", t.bf16_conv0_mfma);
    t.p8_min_tiles = geti("NOMAD_BF16_8PHASE_MIN_TILES", t.p8_min_tiles);
    t.p8_nt_stores = getb("NOMAD_BF16_NT_STORES", t.p8_nt_stores);
    t.p8_rpre = geti("NOMAD_BF16_RPRE", t.p8_rpre);
    t.x3_plain_epi = getb("NOMAD_X3_PLAIN_EPI", t.x3_plain_epi);
    t.p8_three_b = getb("NOMAD_BF16_B3", t.p8_three_b);
    t.p8_n192 = geti("NOMAD_BF16_N192", t.p8_n192);
    t.p9 = getb("NOMAD_BF16_P9", t.p9);
    t.p9_tail_split = getb("NOMAD_BF16_P9_TAIL", t.p9_tail_split);
    t.p9_share = getb("NOMAD_BF16_P9_SHARE", t.p9_share);
    t.p9_res = getb("NOMAD_BF16_P9_RES", t.p9_res);
    t.p9_short = geti("NOMAD_BF16_P9_SHORT", t.p9_short);
    t.p9_skew = geti("NOMAD_BF16_P9_SKEW", t.p9_skew);
    t.p9_late = getb("NOMAD_BF16_P9_LATE", t.p9_late);
    t.p9_wl = getb("NOMAD_BF16_P9_WL", t.p9_wl);
    t.splitk_lnb_fuse = getb("NOMAD_SPLITK_LNB", t.splitk_lnb_fuse);
    t.f32_attn_vt4 = getb("NOMAD_F32_ATTN_VT4", t.f32_attn_vt4);
    t.bf16_conv0_gelu_erf = getb("NOMAD_BF16_CONV0_GELU_ERF", t.bf16_conv0_gelu_erf);
    t.attn_bwd_small = getb("NOMAD_ATTN_BWD_SMALL", t.attn_bwd_small);
}
#endif

struct nomad_ctx {
    int device = 0;
    int num_cus = 256;   // multiProcessorCount (the persistent GEMM launches two workgroups per CU)
    Tuning tune;
    bool keep = false;
    // repacked weights (device)
    float* conv0_w = nullptr;            // [512][10]
    float* conv_w[7] = {};               // i>=1: [512][k*512] with k index = tap*512 + cin
    float *gn_w = nullptr, *gn_b = nullptr, *fln_w = nullptr, *fln_b = nullptr;
    float *proj_w = nullptr, *proj_b = nullptr;
    float *pos_w = nullptr, *pos_b = nullptr;  // [16][64][6144] (rows 48..63 zero), k = tap*48 + cin
    float *eln_w = nullptr, *eln_b = nullptr;
    LayerDev layers[NOMAD_NUM_LAYERS] = {};
    float *emb_w = nullptr, *emb_b = nullptr;
    // Split-K for the small-M GEMMs of Nomad.forward()'s loss forward / backward (config C4: M = 1600 rows): partial
    // products of up to 4 K-slices, allocated by nomad_enable_backward; splitk_ok is raised for the duration of such a
    // call only (never for scoring forwards, whose bits must not depend on the batch; never in fine-tuning mode)
    float* splitk_part = nullptr;
    // (the no-gradient branches of Nomad.forward() - layer outputs wanted, nothing saved - take their partial-sum block from the
    // call's own workspace: Layout::splitk)
    float* splitk_cur = nullptr;                            // the block of the call being enqueued
    bool splitk_ok = false;
    // A LayerNorm(768) the caller will apply to the output of the NEXT dense GEMM (run_layer: out_proj -> LN, fc2 -> LN): when that GEMM
    // splits K, its epilogue normalises the rows itself (splitk_epilogue_ln_kernel) and sets `done`; otherwise the caller launches the
    // stand-alone LayerNorm.  Per context, set and consumed inside one forward call.
    struct PendingLn {
        const float* gamma = nullptr;
        const float* beta = nullptr;
        float* out = nullptr;
        float* out2 = nullptr;
        bool armed = false, done = false;
    } pending_ln;
    // The backward's mirror (round 6): the LayerNorm BACKWARD the caller will apply to the output of the next dX GEMM (+ g2, the layer-output
    // gradient).  When that GEMM splits K, the LayerNorm-backward kernel forms the GEMM's output row itself from the partial products
    // (layernorm_bwd_kernel<3, true>) and sets `done`: no epilogue launch, the GEMM's output never goes to memory.  dX-only backward.
    struct PendingLnBwd {
        const float* x = nullptr;      // the LayerNorm's input (saved by the forward)
        const float* g2 = nullptr;
        const float* gamma = nullptr;
        float* out = nullptr;
        bool armed = false, done = false;
    } pending_lnb;
    // nomad_pairwise: row sums per (64-ref tile, deg row), kPairScratchDoubles doubles PER LAUNCH STREAM - calls on
    // different streams of one context may be in flight together (Engine / ShardedScorer are driven from side streams), so
    // each stream gets its own block: the first at nomad_create, further ones on a stream's first call
    std::vector<std::pair<hipStream_t, double*>> pair_scratch;   // (kNoStream: block not bound to a stream)
    std::mutex pair_mu;   // guards pair_scratch: nomad_pairwise may be called from several host threads (one stream each)
    // transposed copies for the dX-only backward (built by nomad_enable_backward)
    bool bwd_ready = false;
    float* conv_bw_even[7] = {};  // k=3 layers 1..4: [512][1024] = [W_tap2^T | W_tap0^T]
    float* conv_bw_odd[7] = {};   // k=3 layers 1..4: [512][512]  = W_tap1^T
    float* conv_bw2[7] = {};      // k=2 layers 5,6: [1024][512]
    float* proj_wT = nullptr;     // [512][768]
    float* pos_wb = nullptr;      // [16][64][6144], taps flipped
    float *qkv_wT[NOMAD_NUM_LAYERS] = {}, *o_wT[NOMAD_NUM_LAYERS] = {}, *fc1_wT[NOMAD_NUM_LAYERS] = {},
          *fc2_wT[NOMAD_NUM_LAYERS] = {};
    // bf16 weight copies for the bf16 path (built by nomad_enable_bf16); biases and norm parameters stay fp32
    bool bf16_ready = false;
    bf16_t* conv_w16[7] = {};
    bf16_t* conv0_wfrag = nullptr;       // conv0's MFMA A operands [8][4][64][8] (conv0_wfrag_kernel): bf16 path
    bf16_t *proj_w16 = nullptr, *pos_w16 = nullptr;
    bf16_t* pos_wfrag16 = nullptr;        // the pos-conv weights in MFMA fragment order (posconv_wfrag_kernel): bf16 path
    bf16_t *qkv_w16[NOMAD_NUM_LAYERS] = {}, *o_w16[NOMAD_NUM_LAYERS] = {}, *fc1_w16[NOMAD_NUM_LAYERS] = {},
           *fc2_w16[NOMAD_NUM_LAYERS] = {};
    // the bf16 path's q rows of the fused QKV weight and bias also carry log2(e): its attention kernel works in log2
    // units (p = 2^(s - m), attention_bf16_v2.hip.h); qkv_b16 is the matching fp32 bias
    float* qkv_b16[NOMAD_NUM_LAYERS] = {};
    // split (hi | lo bf16 planes) weight copies for the bf16x3 path (built by nomad_enable_bf16x3)
    bool x3_ready = false;
    bf16s_t* conv_wx[7] = {};
    bf16s_t* proj_wx = nullptr;
    bf16s_t* pos_wx = nullptr;   // Toeplitz form [16][256][kPosKt] (posconv_toeplitz_kernel)
    float* pos_bx = nullptr;     // [16][256]
    bf16s_t *qkv_wx[NOMAD_NUM_LAYERS] = {}, *o_wx[NOMAD_NUM_LAYERS] = {}, *fc1_wx[NOMAD_NUM_LAYERS] = {},
            *fc2_wx[NOMAD_NUM_LAYERS] = {};
    // fine-tuning state (nomad_train_enable): master parameters, gradients, Adam moments; see ParamOffsets
    bool train_ready = false;
    float *theta = nullptr, *grad = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    double *pos_nrm2 = nullptr, *tap_partial = nullptr, *tap_dot = nullptr;
    long long adam_t = 0;
    // model.train() regularisation applied by nomad_embed_train / nomad_train_backward (nomad_train_set_stochastic)
    float p_drop = 0.f, p_attn = 0.f, p_input = 0.f;
    // fairseq Wav2Vec2Model.feature_grad_mult: the gradient entering the conv feature extractor is scaled by this
    // (GradMultiply on the extractor's output); 0.1 in the wav2vec 2.0 BASE config that wav2vec_small.pt carries
    float feature_grad_mult = 0.1f;
    // nomad_set_gemm_precision: 1 = the fp32-layout GEMMs (forward, nomad_embed_train, backward, dW) form their products as
    // three bf16 MFMA products over hi / lo halves split in registers (gemm_f32_glds_kernel<..., X3>); buffers stay fp32
    int gemm_x3 = 0;
    unsigned long long drop_seed = 0;
    unsigned layer_mask = 0xFFFu;  // bit l set: encoder layer l runs (LayerDrop clears bits)
    // A training batch may be several equal groups of clips ("branches": anchor | positive | negative), each with
    // its own LayerDrop mask, as if each had been its own forward call (nomad_train_set_branches)
    bool train_convnet = false;    // config freeze_convnet: False - the conv feature extractor's parameters get gradients (nomad_train_set_convnet)
    bool freeze_encoder = false;   // config freeze_all: the encoder's parameters get no gradient (nomad_train_set_frozen)
    int branches = 1;
    unsigned branch_mask[4] = {0xFFFu, 0xFFFu, 0xFFFu, 0xFFFu};
    std::vector<void*> allocs;
    // host copies of the last ragged batches' metadata (sources of asynchronous H2D copies); a ring, so that two
    // forwards enqueued back to back on different streams do not share a staging vector
    std::vector<int> ragged_meta_ring[4];
    unsigned ragged_seq = 0;
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev;   // grows on demand (Scope): a long timed region is never silently truncated
    std::vector<int> ev_class;
    int ev_used = 0;
    double p_ms[NOMAD_K_COUNT] = {};
    long long p_n[NOMAD_K_COUNT] = {};
    double p_fl[NOMAD_K_COUNT] = {};
    bool ev_ready = false;
    bool prof_overflow = false;   // an event could not be created: the counters are incomplete and profile_read says so
    // libnomad_diag.so only (nomad_diag_set_cksum): per-stage, per-clip checksums of the NEXT bf16 forward's intermediates
    unsigned long long* cksum = nullptr;
    int cksum_stages = 0, cksum_segs = 0;
    // ... and device-to-device copies of up to 4 of those stages' buffers (nomad_diag_set_snapshot)
    int snap_stage[4] = {-1, -1, -1, -1};
    void* snap_dst[4] = {};
    size_t snap_cap[4] = {};
};

namespace {

// Brackets one launch with events when profiling is on.
struct Scope {
    nomad_ctx* c;
    hipStream_t s;
    int slot = -1;
    // cls2 (optional): a sub-class that receives the same time / launch / FLOP counts
    Scope(nomad_ctx* c_, hipStream_t s_, int cls, double flops, int cls2 = -1) : c(c_), s(s_) {
        if (!c->prof) return;
        if (c->ev_used + 2 > (int)c->ev.size()) {  // pool used up: grow it (event creation is host-only work)
            const size_t old = c->ev.size();
            c->ev.resize(old + kEventChunk);
            c->ev_class.resize((old + kEventChunk) / 2);
            for (size_t i = old; i < c->ev.size(); ++i)
                if (hipEventCreate(&c->ev[i]) != hipSuccess) {  // out of events: stop profiling LOUDLY (profile_read fails)
                    for (size_t j = old; j < i; ++j) (void)hipEventDestroy(c->ev[j]);
                    c->ev.resize(old);
                    c->ev_class.resize(old / 2);
                    c->prof_overflow = true;
                    return;
                }
        }
        c->p_fl[cls] += flops;
        c->p_n[cls] += 1;
        if (cls2 >= 0) {
            c->p_fl[cls2] += flops;
            c->p_n[cls2] += 1;
        }
        slot = c->ev_used;
        c->ev_class[slot / 2] = cls | ((cls2 + 1) << 8);
        c->ev_used += 2;
        (void)hipEventRecord(c->ev[slot], s);
    }
    ~Scope() {
        if (slot >= 0) (void)hipEventRecord(c->ev[slot + 1], s);
    }
};

RowMap plain_map(int M, int ld) { return RowMap{0, 0, M > 0 ? M : 1, ld}; }

// LDS padding that limits residency to `occ` workgroups per CU (0 = no limit) for a kernel using `lds` bytes.
int occ_pad(int occ, int lds) {
    if (occ <= 0) return 0;
    const int budget = (160 * 1024 / occ) & ~255;
    return budget > lds ? budget - lds : 0;
}

GemmParams dense(const float* A, int lda, const float* W, const float* bias, const float* R, float* C, int M, int N,
                 int K, int gelu) {
    GemmParams p{};
    p.A = A;
    p.amap = plain_map(M, lda);
    p.kchunk = K;
    p.kstride = 0;
    p.W = W;
    p.ldw = K;
    p.C = C;
    p.cmap = plain_map(M, N);
    p.bias = bias;
    p.R = R;
    p.rmap = plain_map(M, N);
    p.M = M;
    p.N = N;
    p.K = K;
    p.n_valid = N;
    p.gelu = gelu;
    return p;
}

}  // namespace

// ---- across translation units ---------------------------------------------------------------------------------------------------
// nomad_hip.hip: split-K wrappers in front of the fp32 dispatch
NOMAD_INTERNAL int run_gemm(nomad_ctx* c, GemmParams p, int groups, int tile, hipStream_t s, int occ = 0);
// nomad_gemm_f32.hip
NOMAD_INTERNAL int gemm_f32_dispatch(nomad_ctx* c, GemmParams p, int groups, int tile, hipStream_t s, int occ);
NOMAD_INTERNAL hipError_t gemm_f32_n48_split(const GemmParams& q, int groups, hipStream_t s, int S);   // the grouped pos-conv, K in S slices over blockIdx.z
NOMAD_INTERNAL int pick_tile(const nomad_ctx* c, int M, int N, int K);
NOMAD_INTERNAL int mixed_split_rows(const nomad_ctx* c, int M, int N);
// nomad_gemm_bf16.hip
NOMAD_INTERNAL int run_gemm_bf16(nomad_ctx* c, GemmParams p, int groups, hipStream_t s, int tile = -1);
#ifdef NOMAD_DIAG
// g_timeline (dtypes.hip.h) is a device variable per translation unit: each GEMM unit reads its own copy (nomad_diag_timeline takes the newer)
NOMAD_INTERNAL int gemm_f32_timeline_read(unsigned long long* out_host, int n);
NOMAD_INTERNAL int gemm_bf16_timeline_read(unsigned long long* out_host, int n);
#endif
