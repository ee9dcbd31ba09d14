// libnomad_hip.so - engine + C ABI (include/nomad_hip.h).  gfx950 only.
//
// The engine owns: repacked fp32 weights in HBM and a pool of hipEvents for the in-library
// timers.  Everything else (waveforms, outputs, scratch) belongs to the caller.  nomad_embed
// enqueues the whole wav2vec 2.0 BASE forward + head on the caller's stream:
//
//   wav_stats -> gn_fold -> conv0+GN+GELU            (frontend.hip.h)
//   conv1..6 as implicit GEMM + GELU                 (gemm_f32.hip.h, A = time-major activations)
//   LayerNorm(512) -> post_extract_proj GEMM (+bias) writing into the zero-padded pos-conv buffer
//   grouped pos-conv GEMM (+bias, GELU, +x) -> LayerNorm(768)
//   12 x { QKV GEMM -> attention -> out_proj GEMM (+bias,+x) -> LN -> fc1 GEMM (+bias,GELU)
//          -> fc2 GEMM (+bias,+x) -> LN }
//   head (mean_t, ReLU, Linear 768->256, L2 normalise)
#include "nomad_ctx.hip.h"

// The device pass of this file is compiled WITHOUT the packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 / v_pk_mov_b32: nomad_amd/build.py passes -target-feature -packed-fp32-ops).  DESIGN.md, "The packed-FP32
// hazard": on MI355X a v_pk_fma_f32 whose low half reads the HIGH register of a source pair (op_sel) can lose that product
// in lanes 48-63 while waves of a bf16 32x32x16-MFMA GEMM from ANOTHER kernel share its SIMD - the cause of the round-2
// "bf16 nondeterminism".  Those instructions are the compiler's choice (SLP vectorisation of scalar source), so the
// target feature is off for every kernel rather than relying on kernels never being co-scheduled.
#include "attention.hip.h"
#include "attention_bf16_v2.hip.h"
#include "attention_bf16_v3.hip.h"
#include "posconv_bf16_slab.hip.h"
#include "attention_f32_v2.hip.h"
#include "attention_bwd.hip.h"
#include "backward.hip.h"
#include "frontend.hip.h"
#include "gemm_bf16.hip.h"
#include "gemm_bf16_8phase.hip.h"
#include "gemm_bf16_p9.hip.h"
#include "gemm_bf16x3.hip.h"
#include "gemm_f32.hip.h"
#ifdef NOMAD_DIAG  // libnomad_diag.so only: experiments kept for A/B measurements (tools/, tests of the experimental tiles)
#endif
#include "pairwise.hip.h"
#include "rowops.hip.h"
#include "train.hip.h"
#include "wav_reader.h"

namespace {

struct Shapes {
    int B, N, L[7], T, M;
};

bool make_shapes(int B, int N, Shapes* s) {
    s->B = B;
    s->N = N;
    int len = N;
    for (int i = 0; i < 7; ++i) {
        if (len < kConvK[i]) return false;
        len = (len - kConvK[i]) / kConvS[i] + 1;
        s->L[i] = len;
    }
    s->T = len;
    s->M = B * len;
    return len >= 1;
}

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

// Workspace carve.  With keep=false the conv stack ping-pongs between two buffers.
// doubles in a forward's GroupNorm-statistics region: [B][65] final sums + [B][chunks][65] per-chunk partial sums
size_t stats_doubles(int B, int max_l0) {
    return (size_t)kStatsPerClip * B * (1 + stats_chunk_slots(max_l0));
}
// the two launches that fill it (frontend.hip.h); L0: conv-0 frames per clip (0 with lens), max_l0: the longest clip's
void launch_wav_stats(const float* wav, int ld, int L0, int max_l0, int B, double* stats, const int* lens, hipStream_t s) {
    const int nchunk = stats_chunk_slots(max_l0);
    double* part = stats + (size_t)kStatsPerClip * B;
    hipLaunchKernelGGL(wav_stats_kernel, dim3(nchunk, B), dim3(256), 0, s, wav, ld, L0, part, lens);
    hipLaunchKernelGGL(wav_stats_fold_kernel, dim3(B), dim3(128), 0, s, part, nchunk, L0, stats, lens);
}

struct Layout {
    size_t stats, scale, shift, conv[7], featln, xpad, x, x2, y, qkv, ctxb, h, splitk, total;
};

Layout make_layout(const Shapes& s, bool keep) {
    Layout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    l.stats = take(sizeof(double) * stats_doubles(s.B, s.L[0]));
    l.scale = take(sizeof(float) * 512 * s.B);
    l.shift = take(sizeof(float) * 512 * s.B);
    auto conv_bytes = [&](int i) { return sizeof(float) * 512 * (size_t)s.B * s.L[i]; };
    if (keep) {
        for (int i = 0; i < 7; ++i) l.conv[i] = take(conv_bytes(i));
        l.featln = take(conv_bytes(6));
    } else {
        const size_t a = take(conv_bytes(0)), b = take(conv_bytes(1));
        for (int i = 0; i < 7; ++i) l.conv[i] = (i % 2 == 0) ? a : b;
        l.featln = b;  // conv6 lands in a; LN(512) writes to b
    }
    const size_t act = sizeof(float) * 768 * (size_t)s.M;
    l.xpad = take(sizeof(float) * 768 * (size_t)s.B * (s.T + 128));
    l.x = take(act);
    l.x2 = take(act);
    l.y = take(act);
    l.qkv = take(act * 3);
    l.ctxb = take(act);
    l.h = take(act * 4);
    // partial products of the split-K GEMMs of a small layer-output forward (forward_impl): part of the CALL's workspace, so that
    // whether such a forward splits depends on its shape alone - not on which streams other calls are running on
    // Sized from the shape: S x M x N floats for the largest problem splitk_applies / posconv_splitk_applies let through at this M
    // (S x N <= 6144: fc1 2 x 3072, qkv 2 x 2304, fc2 / pos-conv 4 x 768), never more than the fixed cap those checks use.
    l.splitk = s.M < kSplitKLayersMaxM ? take(std::min(kSplitKPartFloats, (size_t)6144 * (size_t)s.M) * sizeof(float)) : 0;
    l.total = off;
    return l;
}

// What a training-mode forward keeps for the backward pass (all fp32, carved from the caller's `saved` block).
struct SavedLayer {
    float *qkv, *ctx, *lse, *y1, *u, *y2;
};
struct Saved {
    float *gn_scale, *gn_shift, *gn_mean, *gn_rstd;
    float* u[7];   // pre-GELU conv outputs, layers 1..6, compact [B][L_i][512]
    float* c6;     // conv6 post-GELU output = LayerNorm(512) input
    float *y0, *upc;  // encoder LayerNorm input, pos-conv pre-GELU
    SavedLayer L[NOMAD_NUM_LAYERS];
    size_t total;
};

Saved make_saved(const Shapes& s, void* base) {
    Saved v{};
    size_t off = 0;
    char* b = static_cast<char*>(base);
    auto take = [&](size_t floats) {
        float* ptr = reinterpret_cast<float*>(b + off);
        off += align_up(floats * sizeof(float));
        return ptr;
    };
    const size_t M = s.M;
    v.gn_scale = take(512 * (size_t)s.B);
    v.gn_shift = take(512 * (size_t)s.B);
    v.gn_mean = take(512 * (size_t)s.B);
    v.gn_rstd = take(512 * (size_t)s.B);
    for (int i = 1; i < 7; ++i) v.u[i] = take(512 * (size_t)s.B * s.L[i]);
    v.c6 = take(512 * M);
    v.y0 = take(768 * M);
    v.upc = take(768 * M);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        v.L[l].qkv = take(2304 * M);
        v.L[l].ctx = take(768 * M);
        v.L[l].lse = take(12 * M);
        v.L[l].y1 = take(768 * M);
        v.L[l].u = take(3072 * M);
        v.L[l].y2 = take(768 * M);
    }
    v.total = off;
    return v;
}

// Scratch of nomad_embed_backward.  train = true adds what the parameter gradients need (nomad_train_backward):
// two transposed operand buffers [3072][Mp], split-K partial products, the recomputed pos-conv input.
struct BwdLayout {
    size_t gx, dya, dyb, dh, dqkv, dug, f1, f2, bufa, bufb, partial, gnfold, cwtab, attnd, total;
    int nchunks, nstat;   // frame chunks of the conv0 parameter-gradient pass / of the GroupNorm-backward statistics pass
    size_t ta, tb, kpart, xg, dwe, lnpart, headp, headdz, dmask;
    int Mp, pos_split, ln_blocks;
    size_t h0, convtmp, c0part;  // trainable conv feature extractor: recomputed conv0 output, one layer's dW, conv0 partials
    long long conv_cols;         // columns of the conv layers' transposed operands: B * ceil512(L_1)
};

constexpr size_t kSplitPartFloats = (size_t)16 * 2304 * 768;  // >= S * Nout * Kin for every split chosen below

inline long long conv_lp(int L) { return ((long long)L + 511) / 512 * 512; }  // a clip's columns in the conv dW operands

BwdLayout make_bwd_layout(const Shapes& s, bool train = false, bool train_conv = false) {
    BwdLayout l{};
    size_t off = 0;
    auto take = [&](size_t floats) {
        size_t o = off;
        off += align_up(floats * sizeof(float));
        return o;
    };
    const size_t M = s.M;
    l.gx = take(768 * M);
    l.dya = take(768 * M);
    l.dyb = take(768 * M);
    l.dh = take(3072 * M);
    l.dqkv = take(2304 * M);
    l.dug = take(768 * (size_t)s.B * (s.T + 128));
    l.f1 = take(512 * M);
    l.f2 = take(512 * M);
    l.bufa = take(512 * (size_t)s.B * (s.L[0] + 2));
    l.bufb = take(512 * (size_t)s.B * (s.L[1] + 2));
    l.nchunks = (s.L[0] + kGnChunk - 1) / kGnChunk;
    l.nstat = (s.L[0] + kGnStatsChunk - 1) / kGnStatsChunk;
    l.partial = take(1024 * (size_t)s.B * l.nstat);
    l.gnfold = take(1024 * (size_t)s.B);
    l.cwtab = take((size_t)512 * 16 * s.B);
    l.attnd = take(12 * M);
    if (train) {
        l.Mp = (s.M + 511) / 512 * 512;  // contraction length of the dW GEMMs: any split S | 16 keeps K % 32 == 0
        l.pos_split = s.B < 4 ? s.B : 4;
        l.ln_blocks = (s.M + kLnRows - 1) / kLnRows;
        // conv dW GEMMs (freeze_convnet: False): dU^T [512][cols] and the transposed im2col [taps * 512][cols]
        l.conv_cols = train_conv ? (long long)s.B * conv_lp(s.L[1]) : 0;
        l.ta = take(std::max((size_t)3072 * l.Mp, (size_t)(512 * l.conv_cols)));
        l.tb = take(std::max((size_t)3072 * l.Mp, (size_t)(1536 * l.conv_cols)));
        const size_t pos_part = (size_t)l.pos_split * 16 * 128 * 2304;
        l.kpart = take(pos_part > kSplitPartFloats ? pos_part : kSplitPartFloats);
        l.xg = take(768 * (size_t)s.B * (s.T + 128));
        l.dwe = take((size_t)768 * 6144);
        l.lnpart = take(std::max((size_t)l.ln_blocks * 2 * 768, (size_t)16 * kPbChunks * 48));  // also the pos-conv bias partials
        l.headp = take((size_t)s.B * 768);
        l.headdz = take((size_t)s.B * 256);
        l.dmask = take(768 * M);
        if (train_conv) {
            l.h0 = take(512 * (size_t)s.B * s.L[0]);
            l.convtmp = take((size_t)512 * 1536);
            l.c0part = take((size_t)5120 * s.B * l.nchunks);
        }
    }
    l.total = off;
    return l;
}

// Trainable parameters (everything after the frozen conv feature extractor) as ONE flat fp32 vector; gradients and
// the Adam moments use the same offsets.  q/k/v of a layer are stored fused [q | k | v] in checkpoint scale.
struct LayerOffsets {
    size_t qkv_w, qkv_b, o_w, o_b, ln1_w, ln1_b, fc1_w, fc1_b, fc2_w, fc2_b, ln2_w, ln2_b;
};
struct ParamOffsets {
    size_t fln_w, fln_b, proj_w, proj_b, pos_g, pos_v, pos_b, eln_w, eln_b;
    LayerOffsets L[NOMAD_NUM_LAYERS];
    size_t conv0_w, gn_w, gn_b, conv_w[7];  // the conv feature extractor (checkpoint layout [co][ci][tap]); frozen unless
                                            // nomad_train_set_convnet says otherwise
    size_t emb_w, emb_b, total;
};

ParamOffsets make_param_offsets() {
    ParamOffsets o{};
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t r = off;
        off += n;  // every segment is a multiple of 4 floats: 16-byte alignment is preserved
        return r;
    };
    o.fln_w = take(512); o.fln_b = take(512);
    o.proj_w = take(768 * 512); o.proj_b = take(768);
    o.pos_g = take(128); o.pos_v = take((size_t)768 * 48 * 128); o.pos_b = take(768);
    o.eln_w = take(768); o.eln_b = take(768);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        LayerOffsets& q = o.L[l];
        q.qkv_w = take((size_t)2304 * 768); q.qkv_b = take(2304);
        q.o_w = take(768 * 768); q.o_b = take(768);
        q.ln1_w = take(768); q.ln1_b = take(768);
        q.fc1_w = take((size_t)3072 * 768); q.fc1_b = take(3072);
        q.fc2_w = take((size_t)768 * 3072); q.fc2_b = take(768);
        q.ln2_w = take(768); q.ln2_b = take(768);
    }
    o.conv0_w = take(512 * 10); o.gn_w = take(512); o.gn_b = take(512);
    for (int i = 1; i < 7; ++i) o.conv_w[i] = take((size_t)512 * 512 * kConvK[i]);
    o.emb_w = take(256 * 768); o.emb_b = take(256);
    o.total = off;
    return o;
}

}  // namespace

namespace {

int upload(nomad_ctx* c, const float* host, size_t n, float** out) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, n * sizeof(float)));
    c->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    *out = static_cast<float*>(d);
    return 0;
}


// Small-M dense GEMMs of the loss forward / backward (config C4: M = 1600): N = 768 is 300 tiles of 64 x 64 on 256 CUs,
// 44 CUs get two tiles and the launch takes two tiles' time (fc2: 82 us where a balanced split would take 48).  The
// contraction is cut into S fixed slices - S x as many, shorter workgroups - whose partial products are added in slice
// order by splitk_epilogue_kernel together with bias / GELU / residual: deterministic, no atomics.
struct SplitKScope {  // raises nomad_ctx::splitk_ok for the lifetime of one forward / backward call
    nomad_ctx* c;
    bool prev;
    SplitKScope(nomad_ctx* c_, bool on) : c(c_), prev(c_->splitk_ok) { c->splitk_ok = on; }
    ~SplitKScope() { c->splitk_ok = prev; }
};

static bool splitk_applies(const nomad_ctx* c, const GemmParams& p, int groups, int tile, int* S) {
    if (!c->splitk_ok || !c->splitk_cur || groups != 1 || tile != 37) return false;
    if (p.DG || p.Upre || p.c_colblk || p.kchunk != p.K || p.n_valid != p.N || p.N % 64) return false;
    const bool c_plain = p.cmap.clip_rows >= p.M && p.cmap.off == 0 && p.cmap.ld == p.N;
    const bool r_plain = !p.R || (p.rmap.clip_rows >= p.M && p.rmap.off == 0 && p.rmap.ld == p.N);
    if (!c_plain || !r_plain || p.amap.clip_rows < p.M) return false;
    const long long tiles = (long long)((p.M + 63) / 64) * (p.N / 64);
    if (tiles >= 512 || p.K < 768) return false;
    *S = p.K >= 2304 ? 4 : 2;
    return p.K % (*S * 32) == 0 && (size_t)*S * p.N <= 6144 && (size_t)*S * p.M * p.N <= kSplitKPartFloats;   // (make_layout sizes the block by these two)
}

static int run_gemm_splitk(nomad_ctx* c, const GemmParams& p, int S, hipStream_t s) {
    GemmParams q = p;
    q.K = p.K / S;
    q.kchunk = q.K;
    q.a_goff = q.K;
    q.w_goff = q.K;
    q.C = c->splitk_cur;
    q.cmap = plain_map(p.M, p.N);
    q.c_goff = (long long)p.M * p.N;
    q.bias = nullptr;
    q.R = nullptr;
    q.gelu = 0;
    const bool keep = c->splitk_ok;
    c->splitk_ok = false;  // the slices themselves are ordinary launches
    const int rc = run_gemm(c, q, S, 37, s);
    c->splitk_ok = keep;
    if (rc) return rc;
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    if (c->pending_lnb.armed && !c->pending_lnb.done && p.N == 768 && !p.gelu && !p.bias && c->tune.splitk_lnb_fuse) {
        hipLaunchKernelGGL((layernorm_bwd_kernel<3, true>), dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, s, c->pending_lnb.x, c->splitk_cur,
                           c->pending_lnb.g2, c->pending_lnb.gamma, c->pending_lnb.out, p.M, S, p.R);
        HIP_TRY(hipGetLastError());
        c->pending_lnb.done = true;
        return 0;
    }
    if (c->pending_ln.armed && !c->pending_ln.done && p.N == 768 && !p.gelu) {
        hipLaunchKernelGGL(splitk_epilogue_ln_kernel, dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, s, c->splitk_cur, S, p.M, p.bias, p.R, p.C,
                           c->pending_ln.gamma, c->pending_ln.beta, c->pending_ln.out, c->pending_ln.out2);
        HIP_TRY(hipGetLastError());
        c->pending_ln.done = true;
        return 0;
    }
    const long long count4 = (long long)p.M * p.N / 4;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(c->splitk_cur), S, count4, p.N / 4, reinterpret_cast<const float4*>(p.bias),
                       reinterpret_cast<const float4*>(p.R), reinterpret_cast<float4*>(p.C), p.gelu);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The grouped pos-conv (16 groups x [M x 48 x 6144]) at the loss path's M = 1600 is 7 row tiles x 16 groups = 112 workgroups with a
// serial K loop of 384 tiles: 331 us per launch on fewer than half of the CUs (three launches per configs[3] step).  K in 4 fixed
// slices over blockIdx.z (448 workgroups), partial products added in slice order by posconv_splitk_epilogue_kernel.  Where the
// dense GEMMs of the same call split (splitk_applies): never on a scoring entry point.  Tuning::splitk_posconv = 0 switches it off.
static bool posconv_splitk_applies(const nomad_ctx* c, const GemmParams& p, int groups, int tile) {
    if (!c->tune.splitk_posconv || !c->splitk_ok || !c->splitk_cur || groups != 16 || tile != 48) return false;
    if (p.DG || p.K != 6144 || p.kchunk != p.K || p.n_valid != 48 || p.c_goff != 48) return false;
    const bool c_plain = p.cmap.clip_rows >= p.M && p.cmap.off == 0 && p.cmap.ld == 768;
    return c_plain && (long long)((p.M + 255) / 256) * 16 < 256 && (size_t)4 * p.M * 768 <= kSplitKPartFloats;
}

static int run_posconv_splitk(nomad_ctx* c, const GemmParams& p, hipStream_t s) {
    constexpr int S = 4;
    GemmParams q = p;
    q.K = p.K / S;
    q.kchunk = q.K;
    q.a_soff = q.K;
    q.w_soff = q.K;
    q.C = c->splitk_cur;
    q.cmap = plain_map(p.M, 768);
    q.c_soff = (long long)p.M * 768;
    q.bias = nullptr;
    q.R = nullptr;
    q.Upre = nullptr;
    q.gelu = 0;
    {
        Scope sc(c, s, NOMAD_K_GEMM, 2.0 * p.M * 48.0 * p.K * 16, NOMAD_K_GEMM_FINE);
        HIP_TRY(gemm_f32_n48_split(q, 16, s, S));
    }
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    hipLaunchKernelGGL(posconv_splitk_epilogue_kernel, dim3(p.M), dim3(192), 0, s, c->splitk_cur, S, p.M, p.bias, p.R, p.rmap, p.r_goff, p.Upre, p.C,
                       p.gelu);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

// split-K wrappers in front of the fp32 GEMM dispatch (nomad_gemm_f32.hip)
int run_gemm(nomad_ctx* c, GemmParams p, int groups, int tile, hipStream_t s, int occ) {
    {
        int S = 0;
        if (splitk_applies(c, p, groups, tile, &S)) return run_gemm_splitk(c, p, S, s);
        if (posconv_splitk_applies(c, p, groups, tile)) return run_posconv_splitk(c, p, s);
    }
    return gemm_f32_dispatch(c, p, groups, tile, s, occ);
}

namespace {


int run_layernorm(nomad_ctx* c, const float* in, const float* g, const float* b, float* out, float* out2, int M, int N,
                  hipStream_t s) {
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    const int blocks = (M + 3) / 4;
    if (N == 768) hipLaunchKernelGGL((layernorm_kernel<3, float, float>), dim3(blocks), dim3(256), 0, s, in, g, b, out, out2, M);
    else if (N == 512) hipLaunchKernelGGL((layernorm_kernel<2, float, float>), dim3(blocks), dim3(256), 0, s, in, g, b, out, out2, M);
    else return fail(NOMAD_ERR_INVALID, "layernorm: N=%d unsupported", N);
    HIP_TRY(hipGetLastError());
    return 0;
}

// mean/ReLU/Linear/normalise head in two stages (rowops.hip.h).  T: frames per clip (the longest clip's with tpref);
// pool: scratch of at least (M / 64 + B) * 768 floats (M = total frames) - every caller passes its FFN hidden buffer.
template <typename TIn>
int run_head(nomad_ctx* c, const TIn* x, int B, int T, const float* w, const float* b, float* emb, const int* tpref,
             float* pool, hipStream_t s) {
    Scope sc(c, s, NOMAD_K_ROW, 2.0 * B * 768 * 256);
    hipLaunchKernelGGL(head_pool_kernel<TIn>, dim3((T + kHeadChunk - 1) / kHeadChunk, B), dim3(256), 0, s, x, tpref ? 0 : T, pool, tpref);
    hipLaunchKernelGGL(head_kernel, dim3(B), dim3(1024), 0, s, pool, tpref ? 0 : T, w, b, emb, tpref);
    HIP_TRY(hipGetLastError());
    return 0;
}

DropCfg make_drop(const nomad_ctx* c, float p) {
    DropCfg d{};
    d.seed_lo = (uint32_t)c->drop_seed;
    d.seed_hi = (uint32_t)(c->drop_seed >> 32);
    const double t = (double)p * 4294967296.0;
    d.threshold = p <= 0.f ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)(t + 0.5));
    d.scale = 1.0f / (1.0f - p);
    return d;
}


int run_attention(nomad_ctx* c, const float* qkv, float* out, float* lse, int B, int T, hipStream_t s,
                  const DropCfg* dc = nullptr, uint32_t site = 0, int bh0 = 0) {
    const double flops = 4.0 * B * 12.0 * (double)T * T * 64;
    Scope sc(c, s, NOMAD_K_ATTN, flops);
    const dim3 grid((T + 63) / 64, B * 12);
    if (dc && dc->threshold)
        hipLaunchKernelGGL((attention_f32_kernel<float, true>), grid, dim3(256), 0, s, qkv, out, lse, T, kNoInts, *dc, site, bh0);
    else if (T >= kAttnV2MinT)  // by the clip's length only: the same clip takes the same kernel in every batch
    {
#ifdef NOMAD_DIAG
        if (c->tune.f32_attn_struct_loads) HIP_TRY(launch_attention_f32_v2<false>(qkv, out, lse, B, T, kNoInts, s));
        else if (!c->tune.f32_attn_vt4) HIP_TRY((launch_attention_f32_v2<true, false>(qkv, out, lse, B, T, kNoInts, s)));
        else
#endif
            HIP_TRY(launch_attention_f32_v2(qkv, out, lse, B, T, kNoInts, s));
    }
    else
        hipLaunchKernelGGL((attention_f32_kernel<float, false>), grid, dim3(256), 0, s, qkv, out, lse, T, kNoInts, DropCfg{}, 0u, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

// y = (resid ? resid : 0) + dropout(x) over n floats of an [M][768] tensor (x == y allowed)
int run_dropout(nomad_ctx* c, const float* x, const float* resid, float* y, long long n, const DropCfg& d, uint32_t site,
                hipStream_t s, unsigned long long idx0 = 0) {
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(resid),
                       reinterpret_cast<float4*>(y), n / 4, d, site, idx0);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

const char* nomad_last_error(void) { return g_err; }
const char* nomad_version(void) {
#ifdef NOMAD_DIAG
    return "nomad_hip 0.3 (gfx950) + experimental kernel instantiations (libnomad_diag.so)";
#else
    return "nomad_hip 0.3 (gfx950)";
#endif
}
int nomad_abi_version(void) { return NOMAD_ABI_VERSION; }

int nomad_wav_probe(const char* const* paths, int n, nomad_wav_info* info, int* status, int threads) {
    if (n < 0 || (n > 0 && (!paths || !info || !status))) return fail(NOMAD_ERR_INVALID, "nomad_wav_probe: null argument");
    try {
        wav::parallel_for(n, threads, [&](int i, int) { status[i] = paths[i] ? wav::probe_one(paths[i], &info[i]) : NOMAD_ERR_INVALID; });
    } catch (const std::exception& e) {
        return fail(NOMAD_ERR_IO, "nomad_wav_probe: %s", e.what());
    }
    return NOMAD_OK;
}

int nomad_wav_frames_at(const nomad_wav_info* info, int target_rate, long long* frames) {
    if (!info || !frames || target_rate <= 0 || info->sample_rate <= 0)
        return fail(NOMAD_ERR_INVALID, "nomad_wav_frames_at: bad argument");
    *frames = wav::resampled_frames(info->frames, info->sample_rate, target_rate);
    return NOMAD_OK;
}

int nomad_wav_read_rows(const char* const* paths, const nomad_wav_info* info, int n, const int* row, float* dst_host,
                        long long stride, int target_rate, int* status, int threads) {
    if (n < 0 || (n > 0 && (!paths || !info || !dst_host || !status)) || stride < 0 || target_rate <= 0)
        return fail(NOMAD_ERR_INVALID, "nomad_wav_read_rows: bad argument");
    for (int i = 0; i < n; ++i)
        if (!paths[i] || info[i].sample_rate <= 0 || wav::resampled_frames(info[i].frames, info[i].sample_rate, target_rate) > stride ||
            (row && row[i] < 0))
            return fail(NOMAD_ERR_INVALID, "nomad_wav_read_rows: file %d: %lld frames at %d Hz do not fit a row of %lld", i, info[i].frames,
                        info[i].sample_rate, stride);
    try {
        const int nt = std::max(1, std::min(threads, n));
        std::vector<std::vector<unsigned char>> raw((size_t)nt);
        std::vector<std::vector<float>> mono((size_t)nt);      // a file at another rate is decoded here first
        std::vector<wav::ResampleKernel> kern((size_t)nt);     // per-thread cache of the last kernel built
        std::vector<int> kern_rate((size_t)nt, 0);
        wav::parallel_for(n, nt, [&](int i, int t) {
            float* dst = dst_host + (size_t)(row ? row[i] : i) * (size_t)stride;
            if (info[i].sample_rate == target_rate) {
                status[i] = wav::read_one(paths[i], info[i], dst, raw[(size_t)t]);
                return;
            }
            std::vector<float>& m = mono[(size_t)t];
            if ((long long)m.size() < info[i].frames) m.resize((size_t)info[i].frames);
            status[i] = wav::read_one(paths[i], info[i], m.data(), raw[(size_t)t]);
            if (status[i] != NOMAD_OK) return;
            if (kern_rate[(size_t)t] != info[i].sample_rate) {
                kern[(size_t)t] = wav::make_resample_kernel(info[i].sample_rate, target_rate);
                kern_rate[(size_t)t] = info[i].sample_rate;
            }
            wav::resample_into(m.data(), info[i].frames, kern[(size_t)t], dst,
                               wav::resampled_frames(info[i].frames, info[i].sample_rate, target_rate));
        });
    } catch (const std::exception& e) {
        return fail(NOMAD_ERR_IO, "nomad_wav_read_rows: %s", e.what());
    }
    for (int i = 0; i < n; ++i)
        if (status[i] != NOMAD_OK) return fail(status[i], "nomad_wav_read_rows: %s: %s", paths[i], status[i] == NOMAD_ERR_IO ? "read failed" : "unsupported encoding");
    return NOMAD_OK;
}

int nomad_num_frames(int n_samples) {
    Shapes s;
    return make_shapes(1, n_samples, &s) ? s.T : 0;
}

int nomad_create(nomad_ctx** out, int device, const nomad_weights* w) {
    if (!out || !w) return fail(NOMAD_ERR_INVALID, "nomad_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return fail(NOMAD_ERR_NO_DEVICE, "nomad_create: no HIP device %d (count %d); this engine has no CPU path",
                    device, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NOMAD_ERR_NO_DEVICE, "nomad_create: device %d is %s, kernels are built for gfx950 only", device,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    nomad_ctx* c = new nomad_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#ifdef NOMAD_DIAG
    tuning_from_env(c->tune);
#endif
    int rc = 0;
    auto up = [&](const float* h, size_t n, float** d) {
        if (rc == 0) rc = upload(c, h, n, d);
    };
    up(w->conv_w[0], 512 * 10, &c->conv0_w);
    for (int i = 1; i < 7 && rc == 0; ++i) {
        const int k = kConvK[i];
        std::vector<float> r((size_t)512 * k * 512);
        for (int n = 0; n < 512; ++n)
            for (int ci = 0; ci < 512; ++ci)
                for (int t = 0; t < k; ++t) r[(size_t)n * k * 512 + t * 512 + ci] = w->conv_w[i][((size_t)n * 512 + ci) * k + t];
        up(r.data(), r.size(), &c->conv_w[i]);
    }
    up(w->gn_w, 512, &c->gn_w);
    up(w->gn_b, 512, &c->gn_b);
    up(w->feat_ln_w, 512, &c->fln_w);
    up(w->feat_ln_b, 512, &c->fln_b);
    up(w->proj_w, 768 * 512, &c->proj_w);
    up(w->proj_b, 768, &c->proj_b);
    if (rc == 0) {  // fold weight_norm(dim=2): w = v * g[k] / ||v[:, :, k]||, then [group][64][tap*48 + cin]
        std::vector<double> nrm(128, 0.0);
        for (size_t o = 0; o < 768; ++o)
            for (size_t ci = 0; ci < 48; ++ci)
                for (size_t t = 0; t < 128; ++t) {
                    const double v = w->pos_v[(o * 48 + ci) * 128 + t];
                    nrm[t] += v * v;
                }
        std::vector<float> sc(128);
        for (int t = 0; t < 128; ++t) sc[t] = (float)((double)w->pos_g[t] / std::sqrt(nrm[t]));
        std::vector<float> r((size_t)16 * 64 * 6144, 0.f);
        for (size_t g = 0; g < 16; ++g)
            for (size_t n = 0; n < 48; ++n)
                for (size_t ci = 0; ci < 48; ++ci)
                    for (size_t t = 0; t < 128; ++t)
                        r[(g * 64 + n) * 6144 + t * 48 + ci] = w->pos_v[((g * 48 + n) * 48 + ci) * 128 + t] * sc[t];
        up(r.data(), r.size(), &c->pos_w);
    }
    up(w->pos_b, 768, &c->pos_b);
    up(w->enc_ln_w, 768, &c->eln_w);
    up(w->enc_ln_b, 768, &c->eln_b);
    for (int l = 0; l < NOMAD_NUM_LAYERS && rc == 0; ++l) {
        const nomad_layer_weights& lw = w->layers[l];
        LayerDev& d = c->layers[l];
        std::vector<float> qkv((size_t)2304 * 768), qb(2304);
        for (size_t i = 0; i < (size_t)768 * 768; ++i) {
            qkv[i] = lw.q_w[i] * 0.125f;  // fairseq MultiheadAttention: q *= head_dim^-0.5 (exact in fp32)
            qkv[(size_t)768 * 768 + i] = lw.k_w[i];
            qkv[(size_t)2 * 768 * 768 + i] = lw.v_w[i];
        }
        for (int i = 0; i < 768; ++i) {
            qb[i] = lw.q_b[i] * 0.125f;
            qb[768 + i] = lw.k_b[i];
            qb[1536 + i] = lw.v_b[i];
        }
        up(qkv.data(), qkv.size(), &d.qkv_w);
        up(qb.data(), qb.size(), &d.qkv_b);
        up(lw.o_w, 768 * 768, &d.o_w);
        up(lw.o_b, 768, &d.o_b);
        up(lw.ln1_w, 768, &d.ln1_w);
        up(lw.ln1_b, 768, &d.ln1_b);
        up(lw.fc1_w, (size_t)3072 * 768, &d.fc1_w);
        up(lw.fc1_b, 3072, &d.fc1_b);
        up(lw.fc2_w, (size_t)768 * 3072, &d.fc2_w);
        up(lw.fc2_b, 768, &d.fc2_b);
        up(lw.ln2_w, 768, &d.ln2_w);
        up(lw.ln2_b, 768, &d.ln2_b);
    }
    up(w->emb_w, 256 * 768, &c->emb_w);
    up(w->emb_b, 256, &c->emb_b);
    if (rc == 0) {
        void* d = nullptr;
        const hipError_t e = hipMalloc(&d, sizeof(double) * kPairScratchDoubles);
        if (e != hipSuccess) rc = fail(NOMAD_ERR_HIP, "nomad_create: pairwise scratch: %s", hipGetErrorString(e));
        else {
            c->allocs.push_back(d);
            c->pair_scratch.emplace_back(kNoStream, static_cast<double*>(d));
        }
    }
    if (rc != 0) {
        nomad_destroy(c);
        return rc;
    }
    *out = c;
    return 0;
}

void nomad_destroy(nomad_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void* p : c->allocs) (void)hipFree(p);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    delete c;
}

int nomad_diag_keep_intermediates(nomad_ctx* c, int on) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    c->keep = on != 0;
    return 0;
}

int nomad_workspace_bytes(const nomad_ctx* c, int B, int n_samples, size_t* bytes) {
    Shapes s;
    if (!c || !bytes || B <= 0 || !make_shapes(B, n_samples, &s))
        return fail(NOMAD_ERR_INVALID, "nomad_workspace_bytes: bad shape B=%d N=%d", B, n_samples);
    *bytes = make_layout(s, c->keep).total;
    return 0;
}

int nomad_diag_workspace_region(const nomad_ctx* c, int B, int n_samples, const char* name, size_t* offset,
                                size_t* bytes) {
    Shapes s;
    if (!c || !name || !offset || !bytes || !make_shapes(B, n_samples, &s))
        return fail(NOMAD_ERR_INVALID, "nomad_diag_workspace_region: bad argument");
    const Layout l = make_layout(s, c->keep);
    if (strncmp(name, "conv", 4) == 0 && name[4] >= '0' && name[4] <= '6' && name[5] == 0) {
        const int i = name[4] - '0';
        *offset = l.conv[i];
        *bytes = sizeof(float) * 512 * (size_t)B * s.L[i];
    } else if (strcmp(name, "featln") == 0) {
        *offset = l.featln;
        *bytes = sizeof(float) * 512 * (size_t)s.M;
    } else if (strcmp(name, "encin") == 0) {
        *offset = l.x;
        *bytes = sizeof(float) * 768 * (size_t)s.M;
    } else if (strcmp(name, "xpad") == 0) {  // group-major [16][B][T+128][48]
        *offset = l.xpad;
        *bytes = sizeof(float) * 768 * (size_t)B * (s.T + 128);
    } else {
        return fail(NOMAD_ERR_INVALID, "unknown region %s", name);
    }
    return 0;
}

}  // extern "C"

// The forward pass.  sv == nullptr: scoring mode (intermediates alias inside the workspace).  sv != nullptr:
// training mode - every tensor the backward needs is written to its slot in `sv` instead.
static int forward_impl(nomad_ctx* c, const float* wav, int B, int n_samples, const float* head_w, const float* head_b,
                        float* emb, float* layers_out, void* workspace, size_t workspace_bytes, nomad_stream_t stream,
                        const Saved* sv) {
    Shapes sh;
    if (!c || !wav || !emb || !workspace || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_embed: bad argument (B=%d, n_samples=%d)", B, n_samples);
    const Layout lay = make_layout(sh, c->keep);
    if (workspace_bytes < lay.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed: workspace %zu < required %zu", workspace_bytes, lay.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int T = sh.T, M = sh.M;
    // the loss forward of Nomad.forward() (saving activations, not fine-tuning): small-M GEMMs may split K.  Round 4: so may the
    // other branch of that loss - a forward that returns the 12 layer outputs (LossNetLayers: `clean`, or `estimate` under
    // no_grad) on fewer than 4096 frames.  The split is a function of the GEMM shape only (fixed slices, ordered fold): the
    // results are deterministic, but they are the loss path's bits, not the scoring path's - nomad_embed WITHOUT layer outputs
    // (TripletModel / predict) never splits, whatever the batch.  The partial sums live in the call's own workspace (round 5; they
    // were two context-wide blocks handed to launch streams by hipEventQuery: whether a third stream's forward split depended on
    // timing).  Tuning::splitk_layers = 0 switches the case off (A/B, diag library).
    float* const loss_block = (sv == nullptr && layers_out != nullptr && lay.splitk != 0 && c->tune.splitk_layers) ? F(lay.splitk) : nullptr;
    const SplitKScope splitk(c, (sv != nullptr || loss_block != nullptr) && !c->train_ready);
    float* const prev_cur = c->splitk_cur;
    c->splitk_cur = sv != nullptr ? c->splitk_part : loss_block;
    struct CurRestore { nomad_ctx* c; float* v; ~CurRestore() { c->splitk_cur = v; } } cur_restore{c, prev_cur};
    int rc;
    // model.train() regularisation: only in the training-mode forward, only when switched on
    const bool reg = sv != nullptr;
    const DropCfg d_in = make_drop(c, reg ? c->p_input : 0.f), d_res = make_drop(c, reg ? c->p_drop : 0.f),
                  d_att = make_drop(c, reg ? c->p_attn : 0.f);
    // LayerDrop: one mask for the call, or one per branch (equal groups of clips) of a merged batch
    int nbr = 1;
    unsigned bmask[4] = {reg ? c->layer_mask : 0xFFFu, 0xFFFu, 0xFFFu, 0xFFFu};
    if (reg && c->branches > 1) {
        if (B % c->branches) return fail(NOMAD_ERR_INVALID, "nomad_embed_train: B=%d is not %d equal branches", B, c->branches);
        nbr = c->branches;
        for (int i = 0; i < nbr; ++i) bmask[i] = c->branch_mask[i];
    }
    const long long act = (long long)M * 768;

    // ---- front end: conv0 + GroupNorm + GELU ------------------------------------------------
    double* stats = reinterpret_cast<double*>(ws + lay.stats);
    float* gn_scale = sv ? sv->gn_scale : F(lay.scale);
    float* gn_shift = sv ? sv->gn_shift : F(lay.shift);
    {
        Scope sc(c, s, NOMAD_K_FRONT, 0.0);
        launch_wav_stats(wav, n_samples, sh.L[0], sh.L[0], B, stats, kNoInts, s);
        hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(512), 0, s, stats, c->conv0_w, c->gn_w, c->gn_b, sh.L[0],
                           gn_scale, gn_shift, sv ? sv->gn_mean : nullptr, sv ? sv->gn_rstd : nullptr, kNoInts);
    }
    {
        Scope sc(c, s, NOMAD_K_FRONT, 2.0 * B * (double)sh.L[0] * 512 * 10);
        hipLaunchKernelGGL(conv0_gn_gelu_kernel<float>, dim3((sh.L[0] + kConv0Frames - 1) / kConv0Frames, B), dim3(256), 0, s,
                           wav, n_samples, sh.L[0], c->conv0_w, gn_scale, gn_shift, F(lay.conv[0]), kNoInts, kNoInts);
    }
    HIP_TRY(hipGetLastError());

    // ---- conv1..6: implicit GEMM over time-major activations --------------------------------
    for (int i = 1; i < 7; ++i) {
        const int k = kConvK[i];
        GemmParams p{};
        p.A = F(lay.conv[i - 1]);
        p.amap = RowMap{0, (long long)sh.L[i - 1] * 512, sh.L[i], kConvS[i] * 512};
        p.K = k * 512;
        p.kchunk = p.K;
        p.W = c->conv_w[i];
        p.ldw = p.K;
        p.C = (sv && i == 6) ? sv->c6 : F(lay.conv[i]);
        p.Upre = sv ? sv->u[i] : nullptr;
        p.M = B * sh.L[i];
        p.N = 512;
        p.n_valid = 512;
        p.cmap = plain_map(p.M, 512);
        p.rmap = p.cmap;
        p.gelu = 1;
        if ((rc = run_gemm(c, p, 1, pick_tile(c, p.M, 512, p.K), s))) return rc;
    }

    // ---- LayerNorm(512) + post_extract_proj into the padded pos-conv buffer ------------------
    if ((rc = run_layernorm(c, sv ? sv->c6 : F(lay.conv[6]), c->fln_w, c->fln_b, F(lay.featln), nullptr, M, 512, s)))
        return rc;
    // group-major pos-conv buffer xg[16][B][T+128][48]; x (post_extract_proj output) sits at frames 64..64+T
    float* xpad = F(lay.xpad);
    const long long grp_stride = (long long)B * (T + 128) * 48;
    const RowMap pad_map{64LL * 48, (long long)(T + 128) * 48, T, 48};
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(zero_pad_rows_kernel<float>, dim3(16 * B), dim3(256), 0, s, xpad, T, kNoInts, kNoInts, B);
    }
    {
        GemmParams p = dense(F(lay.featln), 512, c->proj_w, c->proj_b, nullptr, xpad, M, 768, 512, 0);
        p.cmap = pad_map;
        p.c_colblk = 48;
        p.c_colblk_stride = grp_stride;
        if ((rc = run_gemm(c, p, 1, pick_tile(c, M, 768, 512), s))) return rc;
    }
    if (d_in.threshold) {  // dropout_input: on the features that feed both the pos-conv and its residual
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(dropout_groups_kernel, dim3(M), dim3(192), 0, s, xpad, T, grp_stride, d_in, kSiteInput);
    }
    // ---- pos-conv: 16 groups x (M x 48 x 6144), x + gelu(conv + bias) -------------------------
    {
        GemmParams p{};
        p.A = xpad;
        p.amap = RowMap{0, (long long)(T + 128) * 48, T, 48};  // row (clip, t) starts at frame t: taps are contiguous
        p.a_goff = grp_stride;
        p.K = 6144;
        p.kchunk = 6144;
        p.W = c->pos_w;
        p.ldw = 6144;
        p.w_goff = 64LL * 6144;
        p.bias = c->pos_b;
        p.bias_goff = 48;
        p.C = sv ? sv->y0 : F(lay.y);
        p.Upre = sv ? sv->upc : nullptr;
        p.cmap = plain_map(M, 768);
        p.c_goff = 48;
        p.R = xpad;
        p.rmap = pad_map;
        p.r_goff = grp_stride;
        p.M = M;
        p.N = 64;
        p.n_valid = 48;
        p.gelu = 1;
        if ((rc = run_gemm(c, p, 16, 48, s))) return rc;  // one instantiation for every batch size: same summation order
    }
    float* x = F(lay.x);
    float* x2 = F(lay.x2);
    float* y = F(lay.y);
    if ((rc = run_layernorm(c, sv ? sv->y0 : y, c->eln_w, c->eln_b, x, nullptr, M, 768, s))) return rc;
    if (d_res.threshold && (rc = run_dropout(c, x, nullptr, x, act, d_res, kSiteEncoder, s))) return rc;

    // ---- 12 post-LN transformer layers --------------------------------------------------------
    // One layer over clips [c0, c0 + nc) (rows r0 = c0*T ..): the whole batch, or one branch of it when LayerDrop
    // decided differently for the branches.  Rows are independent, so a sub-range call is the same arithmetic.
    auto run_layer = [&](int l, int c0, int nc) -> int {
        const LayerDev& d = c->layers[l];
        const long long r0 = (long long)c0 * T;
        const int Ms = nc * T;
        const long long acts = (long long)Ms * 768;
        float* xs = x + r0 * 768;
        float* x2s = x2 + r0 * 768;
        float* hs = F(lay.h) + r0 * 3072;
        float* qkv = (sv ? sv->L[l].qkv : F(lay.qkv)) + r0 * 2304;
        float* ctxb = (sv ? sv->L[l].ctx : F(lay.ctxb)) + r0 * 768;
        float* y1 = (sv ? sv->L[l].y1 : y) + r0 * 768;
        float* y2 = (sv ? sv->L[l].y2 : y) + r0 * 768;
        float* lse = sv ? sv->L[l].lse + (long long)c0 * 12 * T : nullptr;
        float* lo = layers_out ? layers_out + (size_t)l * M * 768 + r0 * 768 : nullptr;
        const unsigned long long idx0 = (unsigned long long)r0 * 768;
        int rc;
        if ((rc = run_gemm(c, dense(xs, 768, d.qkv_w, d.qkv_b, nullptr, qkv, Ms, 2304, 768, 0), 1, pick_tile(c, Ms, 2304, 768), s)))
            return rc;
        if ((rc = run_attention(c, qkv, ctxb, lse, nc, T, s, &d_att, site_attn(l), c0 * 12))) return rc;
        // residual dropout: y = x + dropout(W a + b) needs the branch on its own, so the residual add moves out
        // of the GEMM epilogue into the dropout kernel
        // (the LayerNorm behind a residual GEMM: inside the GEMM's split-K epilogue where it has one - configs[3] - else on its own)
        auto arm_ln = [&](const float* g_, const float* b_, float* out_, float* out2_) {
            c->pending_ln = {};
            if (!c->tune.splitk_ln_fuse || d_res.threshold) return;   // (residual dropout moves the residual add out of the GEMM)
            c->pending_ln.gamma = g_;
            c->pending_ln.beta = b_;
            c->pending_ln.out = out_;
            c->pending_ln.out2 = out2_;
            c->pending_ln.armed = true;
        };
        auto ln_after = [&](const float* in_, const float* g_, const float* b_, float* out_, float* out2_) -> int {
            const bool done = c->pending_ln.armed && c->pending_ln.done;
            c->pending_ln = {};
            return done ? 0 : run_layernorm(c, in_, g_, b_, out_, out2_, Ms, 768, s);
        };
        arm_ln(d.ln1_w, d.ln1_b, x2s, nullptr);
        if ((rc = run_gemm(c, dense(ctxb, 768, d.o_w, d.o_b, d_res.threshold ? nullptr : xs, y1, Ms, 768, 768, 0), 1,
                           pick_tile(c, Ms, 768, 768), s))) {
            c->pending_ln = {};
            return rc;
        }
        if (d_res.threshold && (rc = run_dropout(c, y1, xs, y1, acts, d_res, site_proj(l), s, idx0))) return rc;
        if ((rc = ln_after(y1, d.ln1_w, d.ln1_b, x2s, nullptr))) return rc;
        {
            GemmParams p = dense(x2s, 768, d.fc1_w, d.fc1_b, nullptr, hs, Ms, 3072, 768, 1);
            p.Upre = sv ? sv->L[l].u + r0 * 3072 : nullptr;
            if ((rc = run_gemm(c, p, 1, pick_tile(c, Ms, 3072, 768), s))) return rc;
        }
        arm_ln(d.ln2_w, d.ln2_b, xs, lo);
        if ((rc = run_gemm(c, dense(hs, 3072, d.fc2_w, d.fc2_b, d_res.threshold ? nullptr : x2s, y2, Ms, 768, 3072, 0), 1,
                           pick_tile(c, Ms, 768, 3072), s))) {
            c->pending_ln = {};
            return rc;
        }
        if (d_res.threshold && (rc = run_dropout(c, y2, x2s, y2, acts, d_res, site_ffn(l), s, idx0))) return rc;
        return ln_after(y2, d.ln2_w, d.ln2_b, xs, lo);
    };
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        unsigned all = 1u, any = 0u;
        for (int br = 0; br < nbr; ++br) {
            all &= (bmask[br] >> l) & 1u;
            any |= (bmask[br] >> l) & 1u;
        }
        if (all) {
            if ((rc = run_layer(l, 0, B))) return rc;
            continue;
        }
        for (int br = 0; br < nbr; ++br) {  // LayerDrop: identity for a dropped branch, the layer for the others
            const int c0 = br * (B / nbr), nc = B / nbr;
            if ((bmask[br] >> l) & 1u) {
                if ((rc = run_layer(l, c0, nc))) return rc;
            } else if (layers_out) {
                const long long r0 = (long long)c0 * T;
                HIP_TRY(hipMemcpyAsync(layers_out + (size_t)l * M * 768 + r0 * 768, x + r0 * 768,
                                       sizeof(float) * (size_t)nc * T * 768, hipMemcpyDeviceToDevice, s));
            }
        }
        (void)any;
    }

    // ---- head -----------------------------------------------------------------------------------
    // the FFN hidden buffer is dead by now: scratch for the time sums
    return run_head<float>(c, x, B, T, head_w ? head_w : c->emb_w, head_b ? head_b : c->emb_b, emb, kNoInts, F(lay.h), s);
}

// ---- ragged batches: clips of different lengths packed back to back ----------------------------------------
// The reference embeds files one at a time (nomad.py:171-183) because every file has its own length; zero-padding
// a batch would change GroupNorm statistics and the time mean.  Here clips of ANY lengths share one launch
// sequence with no padding: every per-frame tensor is packed [sum_c L_i(c)][channels], the transformer GEMMs and
// LayerNorms see plain packed rows, and only the kernels that care about clip boundaries (front end, conv row maps,
// pos-conv buffer, attention, head) read per-clip prefix sums.  Each row goes through exactly the arithmetic it
// would see at batch 1, so results are bit-identical to per-clip calls.
struct RaggedShapes {
    int B = 0, max_l0 = 0, max_t = 0, min_t = 1 << 30;
    long long rows[7] = {};   // total frames per conv level
    long long P = 0;          // total padded pos-conv frames, sum (T_c + 128)
    long long blocks = 0;     // total pos-conv frame blocks, sum ceil(T_c / kPosBlk) (bf16x3 path)
    std::vector<int> meta;    // [lens(B) | pref_0 (B+1) | ... | pref_6 (B+1) | ppref (B+1) | bpref (B+1)]
    size_t off_lens() const { return 0; }
    size_t off_pref(int i) const { return (size_t)B + (size_t)i * (B + 1); }
    size_t off_ppref() const { return (size_t)B + (size_t)7 * (B + 1); }
    size_t off_bpref() const { return (size_t)B + (size_t)8 * (B + 1); }
};

static bool make_ragged(int B, const int* lens, RaggedShapes* r) {
    r->B = B;
    r->meta.assign((size_t)B + 9 * (size_t)(B + 1), 0);
    for (int c = 0; c < B; ++c) {
        Shapes sh;
        if (!make_shapes(1, lens[c], &sh)) return false;
        r->meta[c] = lens[c];
        for (int i = 0; i < 7; ++i) {
            r->meta[r->off_pref(i) + c + 1] = r->meta[r->off_pref(i) + c] + sh.L[i];
            r->rows[i] += sh.L[i];
        }
        r->meta[r->off_ppref() + c + 1] = r->meta[r->off_ppref() + c] + sh.T + 128;
        r->P += sh.T + 128;
        const int nb = (sh.T + kPosBlk - 1) / kPosBlk;
        r->meta[r->off_bpref() + c + 1] = r->meta[r->off_bpref() + c] + nb;
        r->blocks += nb;
        r->max_l0 = sh.L[0] > r->max_l0 ? sh.L[0] : r->max_l0;
        r->max_t = sh.T > r->max_t ? sh.T : r->max_t;
        r->min_t = sh.T < r->min_t ? sh.T : r->min_t;
    }
    return r->rows[0] < (1LL << 31) / 512 * 256;  // row counts stay well inside int
}

struct RaggedLayout {
    size_t meta, stats, scale, shift, conva, convb, xpad, x, x2, y, qkv, ctxb, h, total;
};

static RaggedLayout make_ragged_layout(const RaggedShapes& r) {
    RaggedLayout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    const size_t M = (size_t)r.rows[6];
    l.meta = take(sizeof(int) * r.meta.size());
    l.stats = take(sizeof(double) * stats_doubles(r.B, r.max_l0));
    l.scale = take(sizeof(float) * 512 * r.B);
    l.shift = take(sizeof(float) * 512 * r.B);
    l.conva = take(sizeof(float) * 512 * (size_t)r.rows[0]);
    l.convb = take(sizeof(float) * 512 * (size_t)r.rows[1]);
    l.xpad = take(sizeof(float) * 768 * (size_t)r.P);
    l.x = take(sizeof(float) * 768 * M);
    l.x2 = take(sizeof(float) * 768 * M);
    l.y = take(sizeof(float) * 768 * M);
    l.qkv = take(sizeof(float) * 2304 * M);
    l.ctxb = take(sizeof(float) * 768 * M);
    l.h = take(sizeof(float) * 3072 * M);
    l.total = off;
    return l;
}

static int forward_ragged(nomad_ctx* c, const float* wav, int B, int stride, const int* lens_host, const float* head_w,
                          const float* head_b, float* emb, void* workspace, size_t workspace_bytes,
                          nomad_stream_t stream) {
    RaggedShapes rs;
    if (!c || !wav || !lens_host || !emb || !workspace || B <= 0 || !make_ragged(B, lens_host, &rs))
        return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged: bad argument (B=%d)", B);
    for (int i = 0; i < B; ++i)
        if (lens_host[i] > stride) return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged: clip %d longer than the row stride", i);
    const RaggedLayout lay = make_ragged_layout(rs);
    if (workspace_bytes < lay.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed_ragged: workspace %zu < required %zu", workspace_bytes, lay.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    int* meta = reinterpret_cast<int*>(ws + lay.meta);
    std::vector<int>& staged = c->ragged_meta_ring[c->ragged_seq++ & 3];  // must outlive the asynchronous copy
    staged = rs.meta;
    HIP_TRY(hipMemcpyAsync(meta, staged.data(), sizeof(int) * rs.meta.size(), hipMemcpyHostToDevice, s));
    const int* lens = meta + rs.off_lens();
    auto pref = [&](int i) { return static_cast<const int*>(meta + rs.off_pref(i)); };
    const int* tpref = pref(6);
    const int* ppref = meta + rs.off_ppref();
    const int M = (int)rs.rows[6];
    int rc;

    double* stats = reinterpret_cast<double*>(ws + lay.stats);
    float *scale = F(lay.scale), *shift = F(lay.shift);
    float* cb[2] = {F(lay.conva), F(lay.convb)};
    {
        Scope sc(c, s, NOMAD_K_FRONT, 0.0);
        launch_wav_stats(wav, stride, 0, rs.max_l0, B, stats, lens, s);
        hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(512), 0, s, stats, c->conv0_w, c->gn_w, c->gn_b, 0, scale, shift,
                           static_cast<float*>(nullptr), static_cast<float*>(nullptr), lens);
    }
    {
        Scope sc(c, s, NOMAD_K_FRONT, 2.0 * (double)rs.rows[0] * 512 * 10);
        hipLaunchKernelGGL(conv0_gn_gelu_kernel<float>, dim3((rs.max_l0 + kConv0Frames - 1) / kConv0Frames, B), dim3(256), 0,
                           s, wav, stride, 0, c->conv0_w, scale, shift, cb[0], lens, pref(0));
    }
    for (int i = 1; i < 7; ++i) {
        GemmParams p{};
        p.A = cb[(i - 1) % 2];
        p.amap = RowMap{0, 0, 0, kConvS[i] * 512, pref(i), pref(i - 1), B, 512};
        p.K = kConvK[i] * 512;
        p.kchunk = p.K;
        p.W = c->conv_w[i];
        p.ldw = p.K;
        p.C = cb[i % 2];
        p.M = (int)rs.rows[i];
        p.N = 512;
        p.n_valid = 512;
        p.cmap = plain_map(p.M, 512);
        p.rmap = p.cmap;
        p.gelu = 1;
        if ((rc = run_gemm(c, p, 1, pick_tile(c, p.M, 512, p.K), s))) return rc;
    }
    float* featln = cb[1];
    if ((rc = run_layernorm(c, cb[0], c->fln_w, c->fln_b, featln, nullptr, M, 512, s))) return rc;
    float* xpad = F(lay.xpad);
    const long long grp_stride = rs.P * 48;
    const RowMap pad_map{64LL * 48, 0, 0, 48, tpref, ppref, B, 48};
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(zero_pad_rows_kernel<float>, dim3(16 * B), dim3(256), 0, s, xpad, 0, tpref, ppref, B);
    }
    {
        GemmParams p = dense(featln, 512, c->proj_w, c->proj_b, nullptr, xpad, M, 768, 512, 0);
        p.cmap = pad_map;
        p.c_colblk = 48;
        p.c_colblk_stride = grp_stride;
        if ((rc = run_gemm(c, p, 1, pick_tile(c, M, 768, 512), s))) return rc;
    }
    float *x = F(lay.x), *x2 = F(lay.x2), *y = F(lay.y), *qkv = F(lay.qkv), *ctxb = F(lay.ctxb), *hbuf = F(lay.h);
    {
        GemmParams p{};
        p.A = xpad;
        p.amap = RowMap{0, 0, 0, 48, tpref, ppref, B, 48};
        p.a_goff = grp_stride;
        p.K = 6144;
        p.kchunk = 6144;
        p.W = c->pos_w;
        p.ldw = 6144;
        p.w_goff = 64LL * 6144;
        p.bias = c->pos_b;
        p.bias_goff = 48;
        p.C = y;
        p.cmap = plain_map(M, 768);
        p.c_goff = 48;
        p.R = xpad;
        p.rmap = pad_map;
        p.r_goff = grp_stride;
        p.M = M;
        p.N = 64;
        p.n_valid = 48;
        p.gelu = 1;
        if ((rc = run_gemm(c, p, 16, 48, s))) return rc;  // one instantiation for every batch size: same summation order
    }
    if ((rc = run_layernorm(c, y, c->eln_w, c->eln_b, x, nullptr, M, 768, s))) return rc;
    double attn_flops = 0.0;
    for (int i = 0; i < B; ++i) {
        const double t = rs.meta[rs.off_pref(6) + i + 1] - rs.meta[rs.off_pref(6) + i];
        attn_flops += 4.0 * 12.0 * t * t * 64;
    }
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = run_gemm(c, dense(x, 768, d.qkv_w, d.qkv_b, nullptr, qkv, M, 2304, 768, 0), 1, pick_tile(c, M, 2304, 768), s)))
            return rc;
        {
            Scope sc(c, s, NOMAD_K_ATTN, attn_flops);
            // clips of kAttnV2MinT frames or more: attention_f32_v2_kernel; shorter ones: attention_f32_kernel (each skips the
            // other's clips) - exactly the kernel the clip would get in a batch of its own
            if (rs.max_t >= kAttnV2MinT) HIP_TRY(launch_attention_f32_v2(qkv, ctxb, nullptr, B, rs.max_t, tpref, s, kAttnV2MinT));
            if (rs.min_t < kAttnV2MinT)
                hipLaunchKernelGGL(attention_f32_kernel<float>, dim3((std::min(rs.max_t, kAttnV2MinT - 1) + 63) / 64, B * 12), dim3(256), 0,
                                   s, qkv, ctxb, static_cast<float*>(nullptr), 0, tpref, DropCfg{}, 0u, 0, 0LL, kAttnV2MinT);
        }
        if ((rc = run_gemm(c, dense(ctxb, 768, d.o_w, d.o_b, x, y, M, 768, 768, 0), 1, pick_tile(c, M, 768, 768), s))) return rc;
        if ((rc = run_layernorm(c, y, d.ln1_w, d.ln1_b, x2, nullptr, M, 768, s))) return rc;
        if ((rc = run_gemm(c, dense(x2, 768, d.fc1_w, d.fc1_b, nullptr, hbuf, M, 3072, 768, 1), 1, pick_tile(c, M, 3072, 768), s)))
            return rc;
        if ((rc = run_gemm(c, dense(hbuf, 3072, d.fc2_w, d.fc2_b, x2, y, M, 768, 3072, 0), 1, pick_tile(c, M, 768, 3072), s)))
            return rc;
        if ((rc = run_layernorm(c, y, d.ln2_w, d.ln2_b, x, nullptr, M, 768, s))) return rc;
    }
    return run_head<float>(c, x, B, rs.max_t, head_w ? head_w : c->emb_w, head_b ? head_b : c->emb_b, emb, tpref, hbuf, s);
}

// One backward GEMM: C[M][N] = A[M][K] * Wt[N][K]^T (Wt = transposed forward weight), optional GELU' and residual.
static int bwd_gemm(nomad_ctx* c, const float* A, const float* Wt, float* C, int M, int N, int K, const float* DG,
                    const float* R, hipStream_t s) {
    GemmParams p = dense(A, K, Wt, nullptr, R, C, M, N, K, 0);
    p.DG = DG;
    p.dgmap = plain_map(M, N);
    return run_gemm(c, p, 1, pick_tile(c, M, N, K), s);
}

static int run_ln_bwd(nomad_ctx* c, const float* x, const float* g, const float* g2, const float* gamma, float* dx, int M,
                      int N, hipStream_t s) {
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    const int blocks = (M + 3) / 4;
    if (N == 768) hipLaunchKernelGGL(layernorm_bwd_kernel<3>, dim3(blocks), dim3(256), 0, s, x, g, g2, gamma, dx, M);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, x, g, g2, gamma, dx, M);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- bf16 path (config C5) -----------------------------------------------------------------------------
// bf16 attention (attention_bf16_v2.hip.h).  T = frames per clip (the longest clip's with tpref).  256-query workgroups
// when that still gives the chip >= 2 rounds of them, 128-query ones for small batches.  log2e: q carries log2(e).
// The forward's (log2e) kernels stage K / V by LDS-DMA, 128-key tiles for the 256-query workgroups: attention 3.33 -> 3.05 ms per
// C5 step, 1637-1646 -> 1676-1679 clips/s, bit-identical (gpurun_out/attndma).  NOMAD_BF16_ATTN_DMA = 0: staging through registers,
// 1: LDS-DMA with 64-key tiles (A/B runs).
static hipError_t run_attention_bf16(const nomad_ctx* c, const bf16_t* qkv, bf16_t* out, int B, int T, const int* tpref, bool log2e, hipStream_t s) {
    const int dma = c->tune.bf16_attn_dma;
    const bool big = (long long)((T + 255) / 256) * B * 12 >= 1024;
    // round 5: the 16-wide matrix shape (attention_bf16_v3.hip.h) for every batch size - a clip's bits do not depend on its batch
    // (the wave composition - 32 consecutive queries - is the same in both workgroup shapes, so the deferred-rescale decisions and
    // every bit are too)
    if (log2e && c->tune.bf16_attn_v3 == 2) {
#ifdef NOMAD_DIAG
        // Round 6 probe, measured slower and kept out of the product (profiles/NOTEBOOK.md "the last round of the bf16 attention"): two
        // 256-query workgroups fit a CU, so the launch runs in rounds of 2 x CUs items (configs[4]: 2304 items = 4.5 rounds of 512); here
        // the items of a last round at most `bf16_attn_tail` / 8 full run as twice as many 128-query workgroups in a second launch (the
        // same 32-query waves and 32-key blocks: every bit the same).  227 -> 240 us: a CU with ONE 256-query workgroup in the last round
        // already runs it faster (2 waves per SIMD), and the 128-query workgroups stage 64-key tiles.
        const int nq = (T + 255) / 256, items = nq * B * 12, slots = 2 * c->num_cus, tail = items % slots;
        if (big && tail != 0 && tail * 8 <= slots * c->tune.bf16_attn_tail) {
            if (hipError_t e = launch_attention_bf16_v3<8, 128, 4>(qkv, out, B, T, tpref, s, 0, items - tail); e != hipSuccess) return e;
            // (128-query items: clip-head bh, query block 2 qb + {0, 1} - item numbers double; a second half past the clip's end returns at once)
            return launch_attention_bf16_v3_items<4, 64, 4>(qkv, out, T, 2 * nq, tpref, s, 2 * (items - tail), 2 * tail);
        }
#endif
        return big ? launch_attention_bf16_v3<8, 128, 4>(qkv, out, B, T, tpref, s) : launch_attention_bf16_v3<4, 64, 4>(qkv, out, B, T, tpref, s);
    }
#ifdef NOMAD_DIAG
    // the V reads through the compiler's builtin (it waits for the next tile's LDS-DMA in front of them: attention_bf16_v3.hip.h): A/B
    if (log2e && c->tune.bf16_attn_v3 == 3)
        return big ? launch_attention_bf16_v3<8, 128, 4, 2, false>(qkv, out, B, T, tpref, s) : launch_attention_bf16_v3<4, 64, 4, 2, false>(qkv, out, B, T, tpref, s);
    // round 5's register use (a second copy of the -m_ref quads, the ones operand and the V addresses re-made per block): A/B
    if (log2e && c->tune.bf16_attn_v3 == 5)
        return big ? launch_attention_bf16_v3<8, 128, 4, 2, true, false>(qkv, out, B, T, tpref, s) : launch_attention_bf16_v3<4, 64, 4, 2, true, false>(qkv, out, B, T, tpref, s);
    // 16 waves per workgroup (512 queries share a staged K / V tile: half the LDS-DMA pieces per wave): A/B
    if (log2e && c->tune.bf16_attn_v3 == 16)
        return big ? launch_attention_bf16_v3<16, 128, 4, 2>(qkv, out, B, T, tpref, s) : launch_attention_bf16_v3<4, 64, 4>(qkv, out, B, T, tpref, s);
    // 64 queries per wave (half the LDS bytes per MFMA, 2 waves per SIMD): measured no faster - 262 / 271 / 310 us best launch at C5's
    // shape for 32 queries, 64 queries x 4 waves, 64 queries x 8 waves (profiles/r05_attention_bf16_variants.txt)
    if (log2e && c->tune.bf16_attn_v3 == 8)
        return big ? launch_attention_bf16_v3<8, 128, 2, 4>(qkv, out, B, T, tpref, s) : launch_attention_bf16_v3<4, 64, 2, 4>(qkv, out, B, T, tpref, s);
    if (log2e && c->tune.bf16_attn_v3 == 4)
        return big ? launch_attention_bf16_v3<4, 128, 2, 4>(qkv, out, B, T, tpref, s) : launch_attention_bf16_v3<4, 64, 2, 4>(qkv, out, B, T, tpref, s);
#endif
    if (log2e && dma == 1)
        return big ? launch_attention_bf16_v2<8, 64, 4, true, true>(qkv, out, B, T, tpref, s)
                   : launch_attention_bf16_v2<4, 64, 4, true, true>(qkv, out, B, T, tpref, s);
    if (log2e && dma == 2)
        return big ? launch_attention_bf16_v2<8, 128, 4, true, true>(qkv, out, B, T, tpref, s)
                   : launch_attention_bf16_v2<4, 64, 4, true, true>(qkv, out, B, T, tpref, s);
    if (log2e) return big ? launch_attention_bf16_v2<8, 64, 4, true>(qkv, out, B, T, tpref, s)
                          : launch_attention_bf16_v2<4, 64, 4, true>(qkv, out, B, T, tpref, s);
    return big ? launch_attention_bf16_v2<8, 64, 4, false>(qkv, out, B, T, tpref, s)
               : launch_attention_bf16_v2<4, 64, 4, false>(qkv, out, B, T, tpref, s);
}

// x + gelu(pos_conv(x) + bias) of the bf16 forward from the padded group-major buffer: y[M][768] (posconv_bf16_slab.hip.h).  max_t: the
// (longest) clip's frames; tpref / ppref: nullptr for a uniform batch.  The frames per workgroup follow the clip length only to keep
// short clips from idling waves and from streaming the weights for a few rows - an output's bits do not depend on it.
static int run_posconv_bf16_slab(nomad_ctx* c, const bf16_t* xpad, bf16_t* y, int max_t, int B, long long M, const int* tpref,
                                 const int* ppref, hipStream_t s) {
    Scope sc(c, s, NOMAD_K_GEMM, 2.0 * (double)M * 768.0 * 6144.0);
    hipError_t e;
    if (max_t > 256) e = launch_posconv_bf16_slab<8, 1>(xpad, c->pos_wfrag16, c->pos_b, y, max_t, B, tpref, ppref, s);        // 512 frames of one clip
    else if (max_t > 128) e = launch_posconv_bf16_slab<8, 2>(xpad, c->pos_wfrag16, c->pos_b, y, max_t, B, tpref, ppref, s);   // 256 frames of two clips
    else e = launch_posconv_bf16_slab<4, 2>(xpad, c->pos_wfrag16, c->pos_b, y, max_t, B, tpref, ppref, s);                    // 128 frames of two clips
    HIP_TRY(e);
    return 0;
}

// wav rows `stride` apart; lens == nullptr: every clip has l0 frames, else ragged (max_l0 = the longest clip's, pref0 = packed rows)
static void launch_conv0_bf16(nomad_ctx* c, const float* wav, int stride, int l0, int max_l0, int B, const float* scale,
                              const float* shift, bf16_t* out, const int* lens, const int* pref0, hipStream_t s) {
#ifdef NOMAD_DIAG
    if (c->tune.bf16_conv0_mfma && c->conv0_wfrag && c->tune.bf16_conv0_gelu_erf) {   // A/B: the erf GELU (what shipped up to round 5)
        hipLaunchKernelGGL((conv0_mfma_gn_gelu_kernel<kConv0MfmaOcc, kConv0MfmaUf, 3>), dim3((max_l0 + kConv0MfmaFrames - 1) / kConv0MfmaFrames, B), dim3(256), 0, s, wav,
                           stride, l0, c->conv0_wfrag, scale, shift, out, lens, pref0);
        return;
    }
#endif
    if (c->tune.bf16_conv0_mfma && c->conv0_wfrag)   // (Tuning::bf16_conv0_mfma = 0: the VALU kernel, A/B runs)
        hipLaunchKernelGGL((conv0_mfma_gn_gelu_kernel<kConv0MfmaOcc, kConv0MfmaUf>), dim3((max_l0 + kConv0MfmaFrames - 1) / kConv0MfmaFrames, B), dim3(256), 0, s, wav,
                           stride, l0, c->conv0_wfrag, scale, shift, out, lens, pref0);
    else
        hipLaunchKernelGGL(conv0_gn_gelu_kernel<bf16_t>, dim3((max_l0 + kConv0Frames - 1) / kConv0Frames, B), dim3(256), 0, s, wav,
                           stride, l0, c->conv0_w, scale, shift, out, lens, pref0, 0LL);
}

// plain C / R matrices AND an A map whose divisions have magic numbers (uniform clips of at least two rows, n-fastest tile walk):

#ifdef NOMAD_DIAG
// Race hunting (tools/race_hunt_bf16.py): order-independent checksum of `nseg` equal byte segments of a buffer -
// out[seg] += sum_i word[i] * ((i & 1023) + 1) mod 2^64 (integer adds commute: the value does not depend on scheduling).
__global__ __launch_bounds__(256) void cksum_kernel(const uint32_t* __restrict__ p, long long words, unsigned long long* __restrict__ out) {
    const uint32_t* s = p + (long long)blockIdx.y * words;
    unsigned long long acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long long)gridDim.x * 256)
        acc += (unsigned long long)s[i] * (unsigned long long)((i & 1023) + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.y, acc);
}
// One stage of a forward: segment `seg` of the buffer goes to slot [stage][seg] of the context's checksum table.
struct CkSum {
    nomad_ctx* c;
    hipStream_t s;
    unsigned long long* tab;
    int stages, segs, stage = 0;
    int snap[4];
    void* sdst[4];
    size_t scap[4];
    CkSum(nomad_ctx* c_, hipStream_t s_) : c(c_), s(s_), tab(c_->cksum), stages(c_->cksum_stages), segs(c_->cksum_segs) {
        for (int k = 0; k < 4; ++k) {
            snap[k] = c->snap_stage[k];
            sdst[k] = c->snap_dst[k];
            scap[k] = c->snap_cap[k];
            c->snap_stage[k] = -1;
        }
        c->cksum = nullptr;  // one forward per nomad_diag_set_cksum / nomad_diag_set_snapshot
        if (tab) (void)hipMemsetAsync(tab, 0, sizeof(unsigned long long) * stages * segs, s);
    }
    void operator()(const void* p, int nseg, size_t seg_bytes) {
        if (!tab) return;
        for (int k = 0; k < 4; ++k)
            if (snap[k] == stage && sdst[k]) {
                const size_t n = std::min(scap[k], (size_t)nseg * seg_bytes);
                (void)hipMemcpyAsync(sdst[k], p, n, hipMemcpyDeviceToDevice, s);
            }
        if (stage < stages && nseg <= segs && seg_bytes % 4 == 0) {
            const long long words = (long long)(seg_bytes / 4);
            const int gx = (int)((words + 256 * 16 - 1) / (256 * 16));
            hipLaunchKernelGGL(cksum_kernel, dim3(gx < 1 ? 1 : (gx > 256 ? 256 : gx), nseg), dim3(256), 0, s,
                               static_cast<const uint32_t*>(p), words, tab + (size_t)stage * segs);
        }
        ++stage;
    }
};
#else
struct CkSum {
    CkSum(nomad_ctx*, hipStream_t) {}
    void operator()(const void*, int, size_t) {}
};
#endif

template <int VPT>
static void launch_ln_bf16(const bf16_t* in, const float* g, const float* b, bf16_t* out, int M, hipStream_t s, int rows) {
    if (rows == 4)
        hipLaunchKernelGGL((layernorm_kernel<VPT, bf16_t, bf16_t, 4>), dim3((M + 15) / 16), dim3(256), 0, s, in, g, b, out,
                           static_cast<float*>(nullptr), M);
    else
        hipLaunchKernelGGL((layernorm_kernel<VPT, bf16_t, bf16_t, 1>), dim3((M + 3) / 4), dim3(256), 0, s, in, g, b, out,
                           static_cast<float*>(nullptr), M);
}

struct Bf16Layout {
    size_t stats, scale, shift, conva, convb, xpad, x, x2, y, qkv, ctxb, h, total;
};

static Bf16Layout make_bf16_layout(const Shapes& s) {
    Bf16Layout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    const size_t e = sizeof(bf16_t), M = s.M;
    l.stats = take(sizeof(double) * stats_doubles(s.B, s.L[0]));
    l.scale = take(sizeof(float) * 512 * s.B);
    l.shift = take(sizeof(float) * 512 * s.B);
    l.conva = take(e * 512 * (size_t)s.B * s.L[0]);
    l.convb = take(e * 512 * (size_t)s.B * s.L[1]);
    l.xpad = take(e * 768 * (size_t)s.B * (s.T + 128));
    l.x = take(e * 768 * M);
    l.x2 = take(e * 768 * M);
    l.y = take(e * 768 * M);
    l.qkv = take(e * 2304 * M);
    l.ctxb = take(e * 768 * M);
    l.h = take(e * 3072 * M);
    l.total = off;
    return l;
}

// Scoring forward with bf16 activations / weights and fp32 accumulation, statistics and softmax.
static int forward_bf16(nomad_ctx* c, const float* wav, int B, int n_samples, float* emb, void* workspace,
                        size_t workspace_bytes, nomad_stream_t stream) {
    Shapes sh;
    if (!c || !wav || !emb || !workspace || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_embed_bf16: bad argument (B=%d, n_samples=%d)", B, n_samples);
    if (!c->bf16_ready) return fail(NOMAD_ERR_INVALID, "nomad_embed_bf16: call nomad_enable_bf16 first");
    const Bf16Layout lay = make_bf16_layout(sh);
    if (workspace_bytes < lay.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed_bf16: workspace %zu < required %zu", workspace_bytes, lay.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto H = [&](size_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
    auto asf = [](const bf16_t* p_) { return reinterpret_cast<const float*>(p_); };  // GemmParams carries typeless pointers
    auto asfm = [](bf16_t* p_) { return reinterpret_cast<float*>(p_); };
    const int T = sh.T, M = sh.M;
    int rc;
    double* stats = reinterpret_cast<double*>(ws + lay.stats);
    float* scale = reinterpret_cast<float*>(ws + lay.scale);
    float* shift = reinterpret_cast<float*>(ws + lay.shift);
    bf16_t* cb[2] = {H(lay.conva), H(lay.convb)};
    CkSum CK(c, s);  // no-op in the product library
    {
        Scope sc(c, s, NOMAD_K_FRONT, 0.0);
        launch_wav_stats(wav, n_samples, sh.L[0], sh.L[0], B, stats, kNoInts, s);
        hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(512), 0, s, stats, c->conv0_w, c->gn_w, c->gn_b, sh.L[0], scale,
                           shift, static_cast<float*>(nullptr), static_cast<float*>(nullptr), kNoInts);
    }
    CK(stats, B, sizeof(double) * kStatsPerClip);
    CK(scale, B, sizeof(float) * 512);
    CK(shift, B, sizeof(float) * 512);
    {
        Scope sc(c, s, NOMAD_K_FRONT, 2.0 * B * (double)sh.L[0] * 512 * 10);
        launch_conv0_bf16(c, wav, n_samples, sh.L[0], sh.L[0], B, scale, shift, cb[0], kNoInts, kNoInts, s);
    }
    CK(cb[0], B, sizeof(bf16_t) * 512 * (size_t)sh.L[0]);
    for (int i = 1; i < 7; ++i) {
        GemmParams p{};
        p.A = asf(cb[(i - 1) % 2]);
        p.amap = RowMap{0, (long long)sh.L[i - 1] * 512, sh.L[i], kConvS[i] * 512};
        p.K = kConvK[i] * 512;
        p.kchunk = p.K;
        p.W = asf(c->conv_w16[i]);
        p.ldw = p.K;
        p.C = asfm(cb[i % 2]);
        p.M = B * sh.L[i];
        p.N = 512;
        p.n_valid = 512;
        p.cmap = plain_map(p.M, 512);
        p.rmap = p.cmap;
        p.gelu = 1;
        if ((rc = run_gemm_bf16(c, p, 1, s))) return rc;
        CK(cb[i % 2], B, sizeof(bf16_t) * 512 * (size_t)sh.L[i]);
    }
    bf16_t* conv6 = cb[0];
    bf16_t* featln = cb[1];
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        launch_ln_bf16<2>(conv6, c->fln_w, c->fln_b, featln, M, s, c->tune.bf16_ln_rows);
    }
    CK(featln, B, sizeof(bf16_t) * 512 * (size_t)T);
    bf16_t* xpad = H(lay.xpad);
    const long long grp_stride = (long long)B * (T + 128) * 48;
    const RowMap pad_map{64LL * 48, (long long)(T + 128) * 48, T, 48};
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(zero_pad_rows_kernel<bf16_t>, dim3(16 * B), dim3(256), 0, s, xpad, T, kNoInts, kNoInts, B);
    }
    {
        GemmParams p = dense(asf(featln), 512, asf(c->proj_w16), c->proj_b, nullptr, asfm(xpad), M, 768, 512, 0);
        p.cmap = pad_map;
        p.c_colblk = 48;
        p.c_colblk_stride = grp_stride;
        if ((rc = run_gemm_bf16(c, p, 1, s))) return rc;
    }
    CK(xpad, 16 * B, sizeof(bf16_t) * 48 * (size_t)(T + 128));
    bf16_t *x = H(lay.x), *x2 = H(lay.x2), *y = H(lay.y), *qkv = H(lay.qkv), *ctxb = H(lay.ctxb), *hb = H(lay.h);
    if (c->tune.bf16_posconv_slab) {
        if ((rc = run_posconv_bf16_slab(c, xpad, y, T, B, M, nullptr, nullptr, s))) return rc;
    } else {
        GemmParams p{};
        p.A = asf(xpad);
        p.amap = RowMap{0, (long long)(T + 128) * 48, T, 48};
        p.a_goff = grp_stride;
        p.K = 6144;
        p.kchunk = 6144;
        p.W = asf(c->pos_w16);
        p.ldw = 6144;
        p.w_goff = 64LL * 6144;
        p.bias = c->pos_b;
        p.bias_goff = 48;
        p.C = asfm(y);
        p.cmap = plain_map(M, 768);
        p.c_goff = 48;
        p.R = asf(xpad);
        p.rmap = pad_map;
        p.r_goff = grp_stride;
        p.M = M;
        p.N = 64;
        p.n_valid = 48;
        p.gelu = 1;
        if ((rc = run_gemm_bf16(c, p, 16, s))) return rc;
    }
    const size_t clip768 = sizeof(bf16_t) * 768 * (size_t)T;
    CK(y, B, clip768);
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        launch_ln_bf16<3>(y, c->eln_w, c->eln_b, x, M, s, c->tune.bf16_ln_rows);
    }
    CK(x, B, clip768);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = run_gemm_bf16(c, dense(asf(x), 768, asf(c->qkv_w16[l]), c->qkv_b16[l], nullptr, asfm(qkv), M, 2304, 768, 0), 1, s)))
            return rc;
        CK(qkv, B, clip768 * 3);
        {
            Scope sc(c, s, NOMAD_K_ATTN, 4.0 * B * 12.0 * (double)T * T * 64);
            HIP_TRY(run_attention_bf16(c, qkv, ctxb, B, T, nullptr, true, s));
        }
        CK(ctxb, B, clip768);
        if ((rc = run_gemm_bf16(c, dense(asf(ctxb), 768, asf(c->o_w16[l]), d.o_b, asf(x), asfm(y), M, 768, 768, 0), 1, s)))
            return rc;
        CK(y, B, clip768);
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            launch_ln_bf16<3>(y, d.ln1_w, d.ln1_b, x2, M, s, c->tune.bf16_ln_rows);
        }
        CK(x2, B, clip768);
        if ((rc = run_gemm_bf16(c, dense(asf(x2), 768, asf(c->fc1_w16[l]), d.fc1_b, nullptr, asfm(hb), M, 3072, 768, 1), 1, s)))
            return rc;
        CK(hb, B, clip768 * 4);
        if ((rc = run_gemm_bf16(c, dense(asf(hb), 3072, asf(c->fc2_w16[l]), d.fc2_b, asf(x2), asfm(y), M, 768, 3072, 0), 1, s)))
            return rc;
        CK(y, B, clip768);
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            launch_ln_bf16<3>(y, d.ln2_w, d.ln2_b, x, M, s, c->tune.bf16_ln_rows);
        }
        CK(x, B, clip768);
    }
    rc = run_head<bf16_t>(c, x, B, T, c->emb_w, c->emb_b, emb, kNoInts, reinterpret_cast<float*>(hb), s);
    CK(emb, B, sizeof(float) * 256);
    return rc;
}


// ---- bf16x3 path: fp32-class results on the bf16 matrix cores ------------------------------------------------
// Every dense / conv GEMM runs as three bf16 MFMA products over split operands (gemm_bf16_8phase.hip.h, X3), the
// activations between them live as split planes (dtypes.hip.h) or fp32 - 4 bytes per element either way:
//   conv0 -> split -> conv1..6 (split) -> LN -> split -> post_extract_proj -> split (group-major, padded)
//   -> grouped pos-conv as a Toeplitz GEMM over blocks of 5 frames (N = 5 x 48 per group) -> fp32 -> LN -> split
//   per layer: QKV -> split -> bf16x3 attention -> split -> out-proj (+ split residual) -> fp32 -> LN -> split
//              -> fc1 + GELU -> split -> fc2 (+ split residual) -> fp32 -> LN -> split (fp32 after the last layer)
// Accumulators, bias / GELU / residual, LayerNorm statistics, softmax and the head are fp32 as in the fp32 path.

// bf16x3 attention: K / V of a head resident in LDS when every clip is short enough, the tiled kernel otherwise.
// waves: 0 = tiled kernel, 4 / 8 = resident kernel with that many waves per (clip, head); -1 = the path's choice.
static int run_attention_x3(nomad_ctx* c, const bf16s_t* qkv, long long in_plane, bf16s_t* out, long long out_plane, int B,
                            int T, int max_t, const int* tpref, double flops, hipStream_t s, int waves = -1) {
    Scope sc(c, s, NOMAD_K_ATTN, flops);
    if (waves < 0) waves = max_t <= kAttnResidentMaxT ? kAttnResidentWaves : 0;
    if (waves > 0) {
        if (max_t > kAttnResidentMaxT) return fail(NOMAD_ERR_INVALID, "bf16x3 resident attention: T = %d > %d", max_t, kAttnResidentMaxT);
        static LdsAttrOnce attr_set;
        HIP_TRY(attr_set.ensure(reinterpret_cast<const void*>(attention_x3_resident_kernel), 160 * 1024));
        hipLaunchKernelGGL(attention_x3_resident_kernel, dim3(B * 12), dim3(waves * 64), attn_x3_resident_lds(max_t), s, qkv,
                           in_plane, out, out_plane, T, attn_x3_resident_rows(max_t), tpref);
    } else {
        hipLaunchKernelGGL(attention_x3_kernel, dim3((max_t + 63) / 64, B * 12), dim3(256), 0, s, qkv, in_plane, out, out_plane, T,
                           tpref);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// One implementation serves equal-length batches and ragged ones: X3Geom carries the row maps of either.
struct X3Geom {
    int B = 0;
    long long rows[7] = {};      // total frames per conv level; rows[6] = M
    long long pad_rows = 0;      // rows of one group of the padded pos-conv buffer
    int max_l0 = 0, max_t = 0;
    int wav_ld = 0;              // samples between clips of the wav buffer
    int L0 = 0, T = 0;           // equal-length batches; 0 when ragged (the kernels read lens / prefixes instead)
    RowMap conv_amap[7];         // im2col rows of conv layer i over the output of layer i - 1
    RowMap pad_map;              // post_extract_proj output rows in the group-major, padded pos-conv buffer
    // pos-conv as a GEMM over blocks of kPosBlk frames: input rows, output rows (in y), residual rows
    long long pos_blocks = 0;
    RowMap pos_amap, pos_cmap, pos_rmap;
    double attn_flops = 0.0;
    const RaggedShapes* ragged = nullptr;
};

struct X3Layout {
    size_t meta, stats, scale, shift, conva, convb, xpad, x, x2, y, qkv, ctxb, h, total;
    long long capa, capb;  // elements per plane of the two conv ping-pong buffers
    long long xpad_plane;  // elements per plane of the padded pos-conv buffer (+ kXpadSlack zeroed elements: the last
                           // frame block of a clip reads up to kPosBlk - 1 frames past the clip's padding)
};
constexpr int kXpadSlack = 256;

static X3Layout make_x3_layout(const X3Geom& g) {
    X3Layout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    const size_t e = 4, M = (size_t)g.rows[6];  // fp32, or two bf16 planes
    l.capa = 512LL * g.rows[0];
    l.capb = 512LL * g.rows[1];
    l.meta = take(g.ragged ? sizeof(int) * g.ragged->meta.size() : 0);
    l.stats = take(sizeof(double) * stats_doubles(g.B, g.max_l0));
    l.scale = take(sizeof(float) * 512 * g.B);
    l.shift = take(sizeof(float) * 512 * g.B);
    l.conva = take(e * l.capa);
    l.convb = take(e * l.capb);
    l.xpad_plane = 768LL * g.pad_rows + kXpadSlack;
    l.xpad = take(e * (size_t)l.xpad_plane);
    l.x = take(e * 768 * M);
    l.x2 = take(e * 768 * M);
    l.y = take(e * 768 * M);
    l.qkv = take(e * 2304 * M);
    l.ctxb = take(e * 768 * M);
    l.h = take(e * 3072 * M);
    l.total = off;
    return l;
}

static X3Geom x3_geom_fixed(const Shapes& sh) {
    X3Geom g;
    g.B = sh.B;
    for (int i = 0; i < 7; ++i) g.rows[i] = (long long)sh.B * sh.L[i];
    g.pad_rows = (long long)sh.B * (sh.T + 128);
    g.max_l0 = sh.L[0];
    g.max_t = sh.T;
    g.wav_ld = sh.N;
    g.L0 = sh.L[0];
    g.T = sh.T;
    for (int i = 1; i < 7; ++i) g.conv_amap[i] = RowMap{0, (long long)sh.L[i - 1] * 512, sh.L[i], kConvS[i] * 512};
    g.pad_map = RowMap{64LL * 48, (long long)(sh.T + 128) * 48, sh.T, 48};
    const int nb = (sh.T + kPosBlk - 1) / kPosBlk;
    g.pos_blocks = (long long)sh.B * nb;
    g.pos_amap = RowMap{0, (long long)(sh.T + 128) * 48, nb, kPosBlk * 48};
    g.pos_rmap = RowMap{64LL * 48, (long long)(sh.T + 128) * 48, nb, kPosBlk * 48};
    g.pos_cmap = RowMap{0, (long long)sh.T * 768, nb, kPosBlk * 768};
    g.attn_flops = 4.0 * sh.B * 12.0 * (double)sh.T * sh.T * 64;
    return g;
}

// meta: the device copy of rs.meta (prefix tables); the row maps point into it
static X3Geom x3_geom_ragged(const RaggedShapes& rs, int stride, const int* meta) {
    X3Geom g;
    g.B = rs.B;
    for (int i = 0; i < 7; ++i) g.rows[i] = rs.rows[i];
    g.pad_rows = rs.P;
    g.max_l0 = rs.max_l0;
    g.max_t = rs.max_t;
    g.wav_ld = stride;
    g.ragged = &rs;
    auto pref = [&](int i) { return meta ? meta + rs.off_pref(i) : nullptr; };
    const int* ppref = meta ? meta + rs.off_ppref() : nullptr;
    for (int i = 1; i < 7; ++i) g.conv_amap[i] = RowMap{0, 0, 0, kConvS[i] * 512, pref(i), pref(i - 1), rs.B, 512};
    const int* bpref = meta ? meta + rs.off_bpref() : nullptr;
    g.pad_map = RowMap{64LL * 48, 0, 0, 48, pref(6), ppref, rs.B, 48};
    g.pos_blocks = rs.blocks;
    g.pos_amap = RowMap{0, 0, 0, kPosBlk * 48, bpref, ppref, rs.B, 48};
    g.pos_rmap = RowMap{64LL * 48, 0, 0, kPosBlk * 48, bpref, ppref, rs.B, 48};
    g.pos_cmap = RowMap{0, 0, 0, kPosBlk * 768, bpref, pref(6), rs.B, 768};
    for (int i = 0; i < rs.B; ++i) {
        const double t = rs.meta[rs.off_pref(6) + i + 1] - rs.meta[rs.off_pref(6) + i];
        g.attn_flops += 4.0 * 12.0 * t * t * 64;
    }
    return g;
}

static GemmParams dense_x3(const bf16s_t* A, long long a_plane, int lda, const bf16s_t* W, const float* bias,
                           const bf16s_t* R, long long r_plane, void* C, long long c_plane, int M, int N, int K, int gelu) {
    GemmParams p = dense(reinterpret_cast<const float*>(A), lda, reinterpret_cast<const float*>(W), bias,
                         reinterpret_cast<const float*>(R), static_cast<float*>(C), M, N, K, gelu);
    p.a_plane = a_plane;
    p.w_plane = (long long)N * K;
    p.r_plane = r_plane;
    p.c_plane = c_plane;
    return p;
}

// meta (ragged only): device prefix tables, already on their way (same stream)
// The GEMM instantiation of the path (run_gemm_bf16 tile ids): the kernel that stages every plane once
// (gemm_bf16x3.hip.h), 2-3 % ahead of the K-concatenated form (20 / 21) in profiles/r01_gemm_bf16x3_shapes.json.
// One instantiation for every shape and batch size: a clip's bits do not depend on the batch it is in.
constexpr int kX3Split = 27, kX3F32 = 28;

static int forward_x3_run(nomad_ctx* c, const float* wav, const X3Geom& g, const X3Layout& lay, const int* meta, float* emb,
                          char* ws, hipStream_t s, const float* head_w = nullptr, const float* head_b = nullptr,
                          float* layers_out = nullptr) {
    auto S = [&](size_t off) { return reinterpret_cast<bf16s_t*>(ws + off); };
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int B = g.B, M = (int)g.rows[6];
    const int* lens = g.ragged ? meta + g.ragged->off_lens() : kNoInts;
    const int* pref0 = g.ragged ? meta + g.ragged->off_pref(0) : kNoInts;
    const int* tpref = g.ragged ? meta + g.ragged->off_pref(6) : kNoInts;
    const int* ppref = g.ragged ? meta + g.ragged->off_ppref() : kNoInts;
    const long long pl768 = 768LL * M, pl3072 = 3072LL * M;
    int rc;
    double* stats = reinterpret_cast<double*>(ws + lay.stats);
    float* scale = F(lay.scale);
    float* shift = F(lay.shift);
    bf16s_t* cb[2] = {S(lay.conva), S(lay.convb)};
    const long long cap[2] = {lay.capa, lay.capb};
    {
        Scope sc(c, s, NOMAD_K_FRONT, 0.0);
        launch_wav_stats(wav, g.wav_ld, g.L0, g.max_l0, B, stats, lens, s);
        hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(512), 0, s, stats, c->conv0_w, c->gn_w, c->gn_b, g.L0, scale,
                           shift, static_cast<float*>(nullptr), static_cast<float*>(nullptr), lens);
    }
    {
        Scope sc(c, s, NOMAD_K_FRONT, 2.0 * (double)g.rows[0] * 512 * 10);
        hipLaunchKernelGGL(conv0_gn_gelu_kernel<bf16s_t>, dim3((g.max_l0 + kConv0Frames - 1) / kConv0Frames, B), dim3(256), 0,
                           s, wav, g.wav_ld, g.L0, c->conv0_w, scale, shift, cb[0], lens, pref0, cap[0]);
    }
    for (int i = 1; i < 7; ++i) {
        GemmParams p{};
        p.A = reinterpret_cast<const float*>(cb[(i - 1) % 2]);
        p.a_plane = cap[(i - 1) % 2];
        p.amap = g.conv_amap[i];
        p.K = kConvK[i] * 512;
        p.kchunk = p.K;
        p.W = reinterpret_cast<const float*>(c->conv_wx[i]);
        p.w_plane = 512LL * p.K;
        p.ldw = p.K;
        p.C = reinterpret_cast<float*>(cb[i % 2]);
        p.c_plane = cap[i % 2];
        p.M = (int)g.rows[i];
        p.N = 512;
        p.n_valid = 512;
        p.cmap = plain_map(p.M, 512);
        p.rmap = p.cmap;
        p.gelu = 1;
        if ((rc = run_gemm_bf16(c, p, 1, s, kX3Split))) return rc;
    }
    bf16s_t* conv6 = cb[0];
    bf16s_t* featln = cb[1];
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL((layernorm_kernel<2, bf16s_t, bf16s_t>), dim3((M + 3) / 4), dim3(256), 0, s, conv6, c->fln_w,
                           c->fln_b, featln, static_cast<float*>(nullptr), M, cap[0], cap[1]);
    }
    bf16s_t* xpad = S(lay.xpad);
    const long long grp_stride = g.pad_rows * 48;
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        for (int pl = 0; pl < 2; ++pl) {  // the padding frames (and the slack behind them) are zero in both planes
            bf16_t* plane = reinterpret_cast<bf16_t*>(xpad) + pl * lay.xpad_plane;
            hipLaunchKernelGGL(zero_pad_rows_kernel<bf16_t>, dim3(16 * B), dim3(256), 0, s, plane, g.T, tpref, ppref, B);
            HIP_TRY(hipMemsetAsync(plane + 768LL * g.pad_rows, 0, kXpadSlack * sizeof(bf16_t), s));
        }
    }
    {
        GemmParams p = dense_x3(featln, cap[1], 512, c->proj_wx, c->proj_b, nullptr, 0, xpad, lay.xpad_plane, M, 768, 512, 0);
        p.cmap = g.pad_map;
        p.c_colblk = 48;
        p.c_colblk_stride = grp_stride;
        if ((rc = run_gemm_bf16(c, p, 1, s, kX3Split))) return rc;
    }
    bf16s_t *x = S(lay.x), *x2 = S(lay.x2), *ctxb = S(lay.ctxb), *hb = S(lay.h);
    float* y = F(lay.y);
    bf16s_t* qkv = S(lay.qkv);
    const long long pl2304 = 2304LL * M;
    {   // grouped pos-conv, x + gelu(conv + bias), as 16 dense GEMMs over blocks of kPosBlk frames:
        // row = (clip, block i): the 132 padded input frames from 5 i on (contiguous in the group-major buffer);
        // column (j, co) = output frame 5 i + j, channel co of the group (Toeplitz weight, posconv_toeplitz_kernel)
        GemmParams p{};
        p.A = reinterpret_cast<const float*>(xpad);
        p.a_plane = lay.xpad_plane;
        p.amap = g.pos_amap;
        p.a_goff = grp_stride;
        p.K = kPosKt;
        p.kchunk = kPosKt;
        p.W = reinterpret_cast<const float*>(c->pos_wx);
        p.w_plane = 16LL * 256 * kPosKt;
        p.ldw = kPosKt;
        p.w_goff = 256LL * kPosKt;
        p.bias = c->pos_bx;
        p.bias_goff = 256;
        p.C = y;
        p.cmap = g.pos_cmap;
        p.c_goff = 48;
        p.c_colblk = 48;
        p.c_colblk_stride = 768;   // column block j = the next frame's row of y
        p.c_blk_step = kPosBlk;
        p.c_clip_frames = g.T;
        p.R = reinterpret_cast<const float*>(xpad);   // residual: frame 64 + 5 i + j of the same buffer = rmap(row) + j * 48 + co
        p.r_plane = lay.xpad_plane;
        p.rmap = g.pos_rmap;
        p.r_goff = grp_stride;
        p.M = (int)g.pos_blocks;
        p.N = 256;
        p.n_valid = kPosBlk * 48;
        p.gelu = 1;
        if ((rc = run_gemm_bf16(c, p, 16, s, kX3F32))) return rc;
    }
    auto ln = [&](const float* in, const float* gm, const float* bt, bf16s_t* out, float* out2 = nullptr) {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL((layernorm_kernel<3, float, bf16s_t>), dim3((M + 3) / 4), dim3(256), 0, s, in, gm, bt, out, out2, M,
                           0LL, pl768);
    };
    ln(y, c->eln_w, c->eln_b, x);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = run_gemm_bf16(c, dense_x3(x, pl768, 768, c->qkv_wx[l], d.qkv_b, nullptr, 0, qkv, pl2304, M, 2304, 768, 0), 1, s, kX3Split)))
            return rc;
        if ((rc = run_attention_x3(c, qkv, pl2304, ctxb, pl768, B, g.T, g.max_t, tpref, g.attn_flops, s))) return rc;
        if ((rc = run_gemm_bf16(c, dense_x3(ctxb, pl768, 768, c->o_wx[l], d.o_b, x, pl768, y, 0, M, 768, 768, 0), 1, s, kX3F32)))
            return rc;
        ln(y, d.ln1_w, d.ln1_b, x2);
        if ((rc = run_gemm_bf16(c, dense_x3(x2, pl768, 768, c->fc1_wx[l], d.fc1_b, nullptr, 0, hb, pl3072, M, 3072, 768, 1), 1, s, kX3Split)))
            return rc;
        if ((rc = run_gemm_bf16(c, dense_x3(hb, pl3072, 3072, c->fc2_wx[l], d.fc2_b, x2, pl768, y, 0, M, 768, 3072, 0), 1, s, kX3F32)))
            return rc;
        // layer_results (nomad.py:250-253): the fp32 LayerNorm output, written next to its split copy
        float* lo = layers_out ? layers_out + (size_t)l * M * 768 : nullptr;
        if (l + 1 < NOMAD_NUM_LAYERS) ln(y, d.ln2_w, d.ln2_b, x, lo);
        else if ((rc = run_layernorm(c, y, d.ln2_w, d.ln2_b, reinterpret_cast<float*>(x), lo, M, 768, s))) return rc;
    }
    return run_head<float>(c, reinterpret_cast<const float*>(x), B, g.max_t, head_w ? head_w : c->emb_w, head_b ? head_b : c->emb_b, emb,
                           tpref, reinterpret_cast<float*>(hb), s);
}

static int forward_x3(nomad_ctx* c, const float* wav, int B, int n_samples, float* emb, void* workspace,
                      size_t workspace_bytes, nomad_stream_t stream, const float* head_w = nullptr, const float* head_b = nullptr,
                      float* layers_out = nullptr) {
    Shapes sh;
    if (!c || !wav || !emb || !workspace || B <= 0 || !make_shapes(B, n_samples, &sh) || (!head_w != !head_b))
        return fail(NOMAD_ERR_INVALID, "nomad_embed_bf16x3: bad argument (B=%d, n_samples=%d)", B, n_samples);
    if (!c->x3_ready) return fail(NOMAD_ERR_INVALID, "nomad_embed_bf16x3: call nomad_enable_bf16x3 first");
    const X3Geom g = x3_geom_fixed(sh);
    const X3Layout lay = make_x3_layout(g);
    if (workspace_bytes < lay.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed_bf16x3: workspace %zu < required %zu", workspace_bytes, lay.total);
    return forward_x3_run(c, wav, g, lay, nullptr, emb, static_cast<char*>(workspace), static_cast<hipStream_t>(stream), head_w, head_b,
                          layers_out);
}

static int forward_ragged_x3(nomad_ctx* c, const float* wav, int B, int stride, const int* lens_host, float* emb,
                             void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
    RaggedShapes rs;
    if (!c || !wav || !lens_host || !emb || !workspace || B <= 0 || !make_ragged(B, lens_host, &rs))
        return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged_bf16x3: bad argument (B=%d)", B);
    if (!c->x3_ready) return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged_bf16x3: call nomad_enable_bf16x3 first");
    for (int i = 0; i < B; ++i)
        if (lens_host[i] > stride) return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged_bf16x3: clip %d longer than the row stride", i);
    const X3Layout lay = make_x3_layout(x3_geom_ragged(rs, stride, nullptr));
    if (workspace_bytes < lay.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed_ragged_bf16x3: workspace %zu < required %zu", workspace_bytes, lay.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    int* meta = reinterpret_cast<int*>(ws + lay.meta);
    std::vector<int>& staged = c->ragged_meta_ring[c->ragged_seq++ & 3];  // must outlive the asynchronous copy
    staged = rs.meta;
    HIP_TRY(hipMemcpyAsync(meta, staged.data(), sizeof(int) * rs.meta.size(), hipMemcpyHostToDevice, s));
    const X3Geom g = x3_geom_ragged(rs, stride, meta);
    return forward_x3_run(c, wav, g, lay, meta, emb, ws, s);
}

// Ragged bf16 forward: forward_ragged with the bf16 kernels of forward_bf16 (mixed-length long-form files, config C5
// through predict).  Every clip sees the arithmetic of its own single-clip bf16 call.
static size_t ragged_bf16_layout(const RaggedShapes& r, RaggedLayout* l) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    const size_t e = sizeof(bf16_t), M = (size_t)r.rows[6];
    l->meta = take(sizeof(int) * r.meta.size());
    l->stats = take(sizeof(double) * stats_doubles(r.B, r.max_l0));
    l->scale = take(sizeof(float) * 512 * r.B);
    l->shift = take(sizeof(float) * 512 * r.B);
    l->conva = take(e * 512 * (size_t)r.rows[0]);
    l->convb = take(e * 512 * (size_t)r.rows[1]);
    l->xpad = take(e * 768 * (size_t)r.P);
    l->x = take(e * 768 * M);
    l->x2 = take(e * 768 * M);
    l->y = take(e * 768 * M);
    l->qkv = take(e * 2304 * M);
    l->ctxb = take(e * 768 * M);
    l->h = take(e * 3072 * M);
    l->total = off;
    return off;
}

static int forward_ragged_bf16(nomad_ctx* c, const float* wav, int B, int stride, const int* lens_host, float* emb,
                               void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
    RaggedShapes rs;
    if (!c || !wav || !lens_host || !emb || !workspace || B <= 0 || !make_ragged(B, lens_host, &rs))
        return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged_bf16: bad argument (B=%d)", B);
    if (!c->bf16_ready) return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged_bf16: call nomad_enable_bf16 first");
    for (int i = 0; i < B; ++i)
        if (lens_host[i] > stride) return fail(NOMAD_ERR_INVALID, "nomad_embed_ragged_bf16: clip %d longer than the row stride", i);
    RaggedLayout lay{};
    ragged_bf16_layout(rs, &lay);
    if (workspace_bytes < lay.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed_ragged_bf16: workspace %zu < required %zu", workspace_bytes, lay.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto H = [&](size_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
    auto asf = [](const bf16_t* p_) { return reinterpret_cast<const float*>(p_); };
    auto asfm = [](bf16_t* p_) { return reinterpret_cast<float*>(p_); };
    int* meta = reinterpret_cast<int*>(ws + lay.meta);
    std::vector<int>& staged = c->ragged_meta_ring[c->ragged_seq++ & 3];  // must outlive the asynchronous copy
    staged = rs.meta;
    HIP_TRY(hipMemcpyAsync(meta, staged.data(), sizeof(int) * rs.meta.size(), hipMemcpyHostToDevice, s));
    const int* lens = meta + rs.off_lens();
    auto pref = [&](int i) { return static_cast<const int*>(meta + rs.off_pref(i)); };
    const int* tpref = pref(6);
    const int* ppref = meta + rs.off_ppref();
    const int M = (int)rs.rows[6];
    int rc;
    double* stats = reinterpret_cast<double*>(ws + lay.stats);
    float* scale = reinterpret_cast<float*>(ws + lay.scale);
    float* shift = reinterpret_cast<float*>(ws + lay.shift);
    bf16_t* cb[2] = {H(lay.conva), H(lay.convb)};
    {
        Scope sc(c, s, NOMAD_K_FRONT, 0.0);
        launch_wav_stats(wav, stride, 0, rs.max_l0, B, stats, lens, s);
        hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(512), 0, s, stats, c->conv0_w, c->gn_w, c->gn_b, 0, scale, shift,
                           static_cast<float*>(nullptr), static_cast<float*>(nullptr), lens);
    }
    {
        Scope sc(c, s, NOMAD_K_FRONT, 2.0 * (double)rs.rows[0] * 512 * 10);
        launch_conv0_bf16(c, wav, stride, 0, rs.max_l0, B, scale, shift, cb[0], lens, pref(0), s);
    }
    for (int i = 1; i < 7; ++i) {
        GemmParams p{};
        p.A = asf(cb[(i - 1) % 2]);
        p.amap = RowMap{0, 0, 0, kConvS[i] * 512, pref(i), pref(i - 1), B, 512};
        p.K = kConvK[i] * 512;
        p.kchunk = p.K;
        p.W = asf(c->conv_w16[i]);
        p.ldw = p.K;
        p.C = asfm(cb[i % 2]);
        p.M = (int)rs.rows[i];
        p.N = 512;
        p.n_valid = 512;
        p.cmap = plain_map(p.M, 512);
        p.rmap = p.cmap;
        p.gelu = 1;
        if ((rc = run_gemm_bf16(c, p, 1, s))) return rc;
    }
    bf16_t* featln = cb[1];
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        launch_ln_bf16<2>(cb[0], c->fln_w, c->fln_b, featln, M, s, c->tune.bf16_ln_rows);
    }
    bf16_t* xpad = H(lay.xpad);
    const long long grp_stride = rs.P * 48;
    const RowMap pad_map{64LL * 48, 0, 0, 48, tpref, ppref, B, 48};
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(zero_pad_rows_kernel<bf16_t>, dim3(16 * B), dim3(256), 0, s, xpad, 0, tpref, ppref, B);
    }
    {
        GemmParams p = dense(asf(featln), 512, asf(c->proj_w16), c->proj_b, nullptr, asfm(xpad), M, 768, 512, 0);
        p.cmap = pad_map;
        p.c_colblk = 48;
        p.c_colblk_stride = grp_stride;
        if ((rc = run_gemm_bf16(c, p, 1, s))) return rc;
    }
    bf16_t *x = H(lay.x), *x2 = H(lay.x2), *y = H(lay.y), *qkv = H(lay.qkv), *ctxb = H(lay.ctxb), *hb = H(lay.h);
    if (c->tune.bf16_posconv_slab) {
        if ((rc = run_posconv_bf16_slab(c, xpad, y, rs.max_t, B, M, tpref, ppref, s))) return rc;
    } else {
        GemmParams p{};
        p.A = asf(xpad);
        p.amap = RowMap{0, 0, 0, 48, tpref, ppref, B, 48};
        p.a_goff = grp_stride;
        p.K = 6144;
        p.kchunk = 6144;
        p.W = asf(c->pos_w16);
        p.ldw = 6144;
        p.w_goff = 64LL * 6144;
        p.bias = c->pos_b;
        p.bias_goff = 48;
        p.C = asfm(y);
        p.cmap = plain_map(M, 768);
        p.c_goff = 48;
        p.R = asf(xpad);
        p.rmap = pad_map;
        p.r_goff = grp_stride;
        p.M = M;
        p.N = 64;
        p.n_valid = 48;
        p.gelu = 1;
        if ((rc = run_gemm_bf16(c, p, 16, s))) return rc;
    }
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        launch_ln_bf16<3>(y, c->eln_w, c->eln_b, x, M, s, c->tune.bf16_ln_rows);
    }
    double attn_flops = 0.0;
    for (int i = 0; i < B; ++i) {
        const double t = rs.meta[rs.off_pref(6) + i + 1] - rs.meta[rs.off_pref(6) + i];
        attn_flops += 4.0 * 12.0 * t * t * 64;
    }
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = run_gemm_bf16(c, dense(asf(x), 768, asf(c->qkv_w16[l]), c->qkv_b16[l], nullptr, asfm(qkv), M, 2304, 768, 0), 1, s)))
            return rc;
        {
            Scope sc(c, s, NOMAD_K_ATTN, attn_flops);
            HIP_TRY(run_attention_bf16(c, qkv, ctxb, B, rs.max_t, tpref, true, s));
        }
        if ((rc = run_gemm_bf16(c, dense(asf(ctxb), 768, asf(c->o_w16[l]), d.o_b, asf(x), asfm(y), M, 768, 768, 0), 1, s)))
            return rc;
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            launch_ln_bf16<3>(y, d.ln1_w, d.ln1_b, x2, M, s, c->tune.bf16_ln_rows);
        }
        if ((rc = run_gemm_bf16(c, dense(asf(x2), 768, asf(c->fc1_w16[l]), d.fc1_b, nullptr, asfm(hb), M, 3072, 768, 1), 1, s)))
            return rc;
        if ((rc = run_gemm_bf16(c, dense(asf(hb), 3072, asf(c->fc2_w16[l]), d.fc2_b, asf(x2), asfm(y), M, 768, 3072, 0), 1, s)))
            return rc;
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            launch_ln_bf16<3>(y, d.ln2_w, d.ln2_b, x, M, s, c->tune.bf16_ln_rows);
        }
    }
    return run_head<bf16_t>(c, x, B, rs.max_t, c->emb_w, c->emb_b, emb, tpref, reinterpret_cast<float*>(hb), s);
}

extern "C" {

int nomad_enable_bf16(nomad_ctx* c) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    if (c->bf16_ready) return 0;
    HIP_TRY(hipSetDevice(c->device));
    // A weight update (nomad_train_adam_step / nomad_train_write) rebuilt the fp32 kernel-layout weights on the CALLER's
    // stream, which may be a non-blocking stream the null stream does not wait for: drain the device before converting.
    HIP_TRY(hipDeviceSynchronize());
    auto conv = [&](const float* src, size_t n, bf16_t** out) -> int {
        if (!*out) {  // re-enabling after a weight update reuses the buffers
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, n * sizeof(bf16_t)));
            c->allocs.push_back(d);
            *out = static_cast<bf16_t*>(d);
        }
        hipLaunchKernelGGL(to_bf16_kernel, dim3(1024), dim3(256), 0, 0, src, *out, (long long)(n / 4));
        return 0;
    };
    int rc;
    if (!c->conv0_wfrag) {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, (size_t)8 * 4 * 64 * 8 * sizeof(bf16_t)));
        c->allocs.push_back(d);
        c->conv0_wfrag = static_cast<bf16_t*>(d);
    }
    hipLaunchKernelGGL(conv0_wfrag_kernel, dim3(32), dim3(64), 0, 0, c->conv0_w, c->conv0_wfrag);
    for (int i = 1; i < 7; ++i)
        if ((rc = conv(c->conv_w[i], (size_t)512 * kConvK[i] * 512, &c->conv_w16[i]))) return rc;
    if ((rc = conv(c->proj_w, (size_t)768 * 512, &c->proj_w16))) return rc;
    if ((rc = conv(c->pos_w, (size_t)16 * 64 * 6144, &c->pos_w16))) return rc;
    if (!c->pos_wfrag16) {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, posconv_wfrag_elems() * sizeof(bf16_t)));
        c->allocs.push_back(d);
        c->pos_wfrag16 = static_cast<bf16_t*>(d);
    }
    hipLaunchKernelGGL(posconv_wfrag_kernel, dim3(16 * 192), dim3(192), 0, 0, c->pos_w, c->pos_wfrag16);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = conv(d.qkv_w, (size_t)2304 * 768, &c->qkv_w16[l]))) return rc;
        hipLaunchKernelGGL(to_bf16_scaled_kernel, dim3(1024), dim3(256), 0, 0, d.qkv_w, c->qkv_w16[l], (long long)2304 * 768 / 4,
                           (long long)768 * 768 / 4, kLog2e);  // q rows: q * 64^-0.5 * log2(e), rounded to bf16 once
        if (!c->qkv_b16[l]) {
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, 2304 * sizeof(float)));
            c->allocs.push_back(p);
            c->qkv_b16[l] = static_cast<float*>(p);
        }
        hipLaunchKernelGGL(scale_head_kernel, dim3(9), dim3(256), 0, 0, d.qkv_b, c->qkv_b16[l], 2304, 768, kLog2e);
        if ((rc = conv(d.o_w, (size_t)768 * 768, &c->o_w16[l]))) return rc;
        if ((rc = conv(d.fc1_w, (size_t)3072 * 768, &c->fc1_w16[l]))) return rc;
        if ((rc = conv(d.fc2_w, (size_t)768 * 3072, &c->fc2_w16[l]))) return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    c->bf16_ready = true;
    return 0;
}

int nomad_workspace_bytes_bf16(const nomad_ctx* c, int B, int n_samples, size_t* bytes) {
    Shapes sh;
    if (!c || !bytes || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_workspace_bytes_bf16: bad shape B=%d N=%d", B, n_samples);
    *bytes = make_bf16_layout(sh).total;
    return 0;
}

int nomad_embed_bf16(nomad_ctx* c, const float* wav, int B, int n_samples, float* emb, void* workspace,
                     size_t workspace_bytes, nomad_stream_t stream) {
    return forward_bf16(c, wav, B, n_samples, emb, workspace, workspace_bytes, stream);
}


int nomad_enable_bf16x3(nomad_ctx* c) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    if (c->x3_ready) return 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());  // see nomad_enable_bf16: the fp32 weights may have been rebuilt on another stream
    auto conv = [&](const float* src, size_t n, bf16s_t** out) -> int {
        if (!*out) {  // re-enabling after a weight update reuses the buffers
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, 2 * n * sizeof(bf16_t)));
            c->allocs.push_back(d);
            *out = static_cast<bf16s_t*>(d);
        }
        hipLaunchKernelGGL(split_bf16_kernel, dim3(1024), dim3(256), 0, 0, src, *out, (long long)n, (long long)(n / 4));
        return 0;
    };
    int rc;
    for (int i = 1; i < 7; ++i)
        if ((rc = conv(c->conv_w[i], (size_t)512 * kConvK[i] * 512, &c->conv_wx[i]))) return rc;
    if ((rc = conv(c->proj_w, (size_t)768 * 512, &c->proj_wx))) return rc;
    {
        const size_t n = (size_t)16 * 256 * kPosKt;
        if (!c->pos_wx) {
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, 2 * n * sizeof(bf16_t)));
            c->allocs.push_back(d);
            c->pos_wx = static_cast<bf16s_t*>(d);
            HIP_TRY(hipMalloc(&d, 16 * 256 * sizeof(float)));
            c->allocs.push_back(d);
            c->pos_bx = static_cast<float*>(d);
        }
        hipLaunchKernelGGL(posconv_toeplitz_kernel, dim3(16 * 256), dim3(256), 0, 0, c->pos_w, c->pos_wx, (long long)n);
        hipLaunchKernelGGL(posconv_toeplitz_bias_kernel, dim3(16), dim3(256), 0, 0, c->pos_b, c->pos_bx);
    }
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = conv(d.qkv_w, (size_t)2304 * 768, &c->qkv_wx[l]))) return rc;
        if ((rc = conv(d.o_w, (size_t)768 * 768, &c->o_wx[l]))) return rc;
        if ((rc = conv(d.fc1_w, (size_t)3072 * 768, &c->fc1_wx[l]))) return rc;
        if ((rc = conv(d.fc2_w, (size_t)768 * 3072, &c->fc2_wx[l]))) return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    c->x3_ready = true;
    return 0;
}

int nomad_workspace_bytes_bf16x3(const nomad_ctx* c, int B, int n_samples, size_t* bytes) {
    Shapes sh;
    if (!c || !bytes || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_workspace_bytes_bf16x3: bad shape B=%d N=%d", B, n_samples);
    *bytes = make_x3_layout(x3_geom_fixed(sh)).total;
    return 0;
}

int nomad_workspace_bytes_ragged_bf16x3(const nomad_ctx* c, int B, const int* lengths_host, size_t* bytes) {
    RaggedShapes rs;
    if (!c || !bytes || !lengths_host || B <= 0 || !make_ragged(B, lengths_host, &rs))
        return fail(NOMAD_ERR_INVALID, "nomad_workspace_bytes_ragged_bf16x3: bad argument");
    *bytes = make_x3_layout(x3_geom_ragged(rs, 0, nullptr)).total;
    return 0;
}

int nomad_embed_ragged_bf16x3(nomad_ctx* c, const float* wav, int B, int stride, const int* lengths_host, float* emb,
                              void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
    return forward_ragged_x3(c, wav, B, stride, lengths_host, emb, workspace, workspace_bytes, stream);
}

int nomad_embed_bf16x3(nomad_ctx* c, const float* wav, int B, int n_samples, float* emb, void* workspace,
                       size_t workspace_bytes, nomad_stream_t stream) {
    return forward_x3(c, wav, B, n_samples, emb, workspace, workspace_bytes, stream);
}

int nomad_embed_layers_bf16x3(nomad_ctx* c, const float* wav, int B, int n_samples, const float* head_w, const float* head_b,
                              float* emb, float* layers_out, void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
    if (!layers_out) return fail(NOMAD_ERR_INVALID, "nomad_embed_layers_bf16x3: layers_out is NULL");
    return forward_x3(c, wav, B, n_samples, emb, workspace, workspace_bytes, stream, head_w, head_b, layers_out);
}

int nomad_diag_split_bf16(nomad_ctx* c, const float* in, void* out, long long plane, long long n, int inverse,
                          nomad_stream_t stream) {
    if (!c || !in || !out || n <= 0 || n % 4 || plane < n) return fail(NOMAD_ERR_INVALID, "nomad_diag_split_bf16: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (inverse)  // `in` is the split buffer, `out` the fp32 one
        hipLaunchKernelGGL(unsplit_bf16_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<const bf16s_t*>(in), plane,
                           static_cast<float*>(out), n / 4);
    else
        hipLaunchKernelGGL(split_bf16_kernel, dim3(1024), dim3(256), 0, s, in, static_cast<bf16s_t*>(out), plane, n / 4);
    HIP_TRY(hipGetLastError());
    return 0;
}

int nomad_diag_attention_bf16x3(nomad_ctx* c, const void* qkv, void* out, int B, int T, int waves, nomad_stream_t stream) {
    if (!c || !qkv || !out || B <= 0 || T <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_attention_bf16x3: bad argument");
    return run_attention_x3(c, static_cast<const bf16s_t*>(qkv), (long long)B * T * 2304, static_cast<bf16s_t*>(out),
                            (long long)B * T * 768, B, T, T, kNoInts, 4.0 * B * 12.0 * (double)T * T * 64,
                            static_cast<hipStream_t>(stream), waves);
}

int nomad_diag_gemm_bf16x3(nomad_ctx* c, const void* A, const void* W, const float* bias, const void* R, void* C, int M,
                           int N, int K, int gelu, int out_f32, nomad_stream_t stream) {
    if (!c || !A || !W || !C || M <= 0 || N % 256 || K % ((out_f32 % 100 == 12 || out_f32 % 100 == 13) ? 192 : out_f32 % 100 >= 7 ? 64 : 128))
        return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm_bf16x3: bad argument (N %% 256, K %% 128; K %% 64 for the staged kernel)");
    const int group_m = out_f32 / 100;  // measurement knob: variant + 100 * group_m (tile_coords)
    out_f32 %= 100;
    GemmParams p = dense_x3(static_cast<const bf16s_t*>(A), (long long)M * K, K, static_cast<const bf16s_t*>(W), bias,
                            static_cast<const bf16s_t*>(R), (long long)M * N, C, (long long)M * N, M, N, K, gelu);
    p.group_m = group_m;
    return run_gemm_bf16(c, p, 1, static_cast<hipStream_t>(stream), 20 + (out_f32 < 0 ? 0 : out_f32 > 13 ? 13 : out_f32));
}

int nomad_workspace_bytes_ragged_bf16(const nomad_ctx* c, int B, const int* lengths_host, size_t* bytes) {
    RaggedShapes rs;
    if (!c || !bytes || !lengths_host || B <= 0 || !make_ragged(B, lengths_host, &rs))
        return fail(NOMAD_ERR_INVALID, "nomad_workspace_bytes_ragged_bf16: bad argument");
    RaggedLayout lay{};
    *bytes = ragged_bf16_layout(rs, &lay);
    return 0;
}

int nomad_embed_ragged_bf16(nomad_ctx* c, const float* wav, int B, int stride, const int* lengths_host, float* emb,
                            void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
    return forward_ragged_bf16(c, wav, B, stride, lengths_host, emb, workspace, workspace_bytes, stream);
}

#ifdef NOMAD_DIAG
/* Race hunting: the NEXT nomad_embed_bf16 call on this context writes table[stage][seg] (stages x segs 64-bit sums, device
 * memory) - stage order: GroupNorm sums, scale, shift, conv0..6, feature LN, padded projection (16 x B segments), pos-conv,
 * encoder LN, then per layer qkv / ctx / out_proj+res / LN1 / fc1 / fc2+res / LN2, last the embeddings; seg = clip. */
int nomad_diag_set_cksum(nomad_ctx* c, unsigned long long* table_dev, int stages, int segs) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    c->cksum = table_dev;
    c->cksum_stages = stages;
    c->cksum_segs = segs;
    return 0;
}
/* ... and copies the buffer of stage `stage` (up to cap bytes) to dst_dev on the forward's stream; slot 0..3. */
int nomad_diag_set_snapshot(nomad_ctx* c, int slot, int stage, void* dst_dev, size_t cap) {
    if (!c || slot < 0 || slot > 3) return fail(NOMAD_ERR_INVALID, "nomad_diag_set_snapshot: bad argument");
    c->snap_stage[slot] = stage;
    c->snap_dst[slot] = dst_dev;
    c->snap_cap[slot] = cap;
    return 0;
}
/* The bf16 forward's front end alone: waveform statistics -> GroupNorm scale / shift -> conv0 + GroupNorm + GELU as bf16
 * out_dev [B][L0][512]; scratch_dev: 8 * 65 * B * (1 + chunks) + 2 * 4 * 512 * B bytes (race hunting: the victim kernel).
 * variant: conv0_gn_gelu_kernel's VAR (0 = the forward's kernel, bit 0 no LDS, bit 1 scalar tap loop). */
int nomad_diag_conv0_bf16(nomad_ctx* c, const float* wav, int B, int n_samples, void* out_dev, void* scratch_dev,
                          nomad_stream_t stream, int variant) {
    Shapes sh;
    if (!c || !wav || !out_dev || !scratch_dev || !make_shapes(B, n_samples, &sh)) return fail(NOMAD_ERR_INVALID, "nomad_diag_conv0_bf16");
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* stats = static_cast<double*>(scratch_dev);
    float* scale = reinterpret_cast<float*>(stats + stats_doubles(B, sh.L[0]));
    float* shift = scale + 512 * (size_t)B;
    launch_wav_stats(wav, n_samples, sh.L[0], sh.L[0], B, stats, kNoInts, s);
    hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(512), 0, s, stats, c->conv0_w, c->gn_w, c->gn_b, sh.L[0], scale, shift,
                       static_cast<float*>(nullptr), static_cast<float*>(nullptr), kNoInts);
    const dim3 grid((sh.L[0] + kConv0Frames - 1) / kConv0Frames, B);
    bf16_t* out = static_cast<bf16_t*>(out_dev);
#define NOMAD_C0(V)                                                                                                             \
    case V:                                                                                                                     \
        hipLaunchKernelGGL((conv0_gn_gelu_kernel<bf16_t, V>), grid, dim3(256), 0, s, wav, n_samples, sh.L[0], c->conv0_w, scale, \
                           shift, out, kNoInts, kNoInts, 0LL);                                                                  \
        break;
    switch (variant) {
        NOMAD_C0(0) NOMAD_C0(1) NOMAD_C0(2) NOMAD_C0(3)
#define NOMAD_C0M(V, OCC, UF, ABL)                                                                                                \
    case V:                                                                                                                     \
        if (!c->conv0_wfrag) return fail(NOMAD_ERR_INVALID, "nomad_diag_conv0_bf16: variant %d needs nomad_enable_bf16", V);    \
        hipLaunchKernelGGL((conv0_mfma_gn_gelu_kernel<OCC, UF, ABL>), dim3((sh.L[0] + kConv0MfmaFrames - 1) / kConv0MfmaFrames, B), \
                           dim3(256), 0, s, wav, n_samples, sh.L[0], c->conv0_wfrag, scale, shift, out, kNoInts, kNoInts);      \
        break;
        NOMAD_C0M(4, kConv0MfmaOcc, kConv0MfmaUf, 0)   // the matrix-core kernel as the bf16 forward launches it (needs nomad_enable_bf16)
        NOMAD_C0M(5, 3, 2, 0) NOMAD_C0M(6, 3, 1, 0) NOMAD_C0M(7, 3, 4, 0)   // other budgets / unrolls (tools/conv0_time.py)
        NOMAD_C0M(9, 4, 1, 2)                                             // timing probe: no GELU
#undef NOMAD_C0M
        default: return fail(NOMAD_ERR_INVALID, "nomad_diag_conv0_bf16: variant %d", variant);
    }
#undef NOMAD_C0
    HIP_TRY(hipGetLastError());
    return 0;
}
#endif

#ifdef NOMAD_DIAG
/* The bf16 forward's positional convolution alone: y_dev [B*T][768] bf16 = x + gelu(pos_conv(x) + bias) from xpad_dev, the padded
 * group-major input [16][B][T + 128][48] bf16 (zero frames 0..63 and T+64.. of every clip).  variant 1: posconv_bf16_slab.hip.h (what
 * the forward launches); 0: the grouped GEMM on 128 x 64 tiles it replaced.  Needs nomad_enable_bf16. */
int nomad_diag_posconv_bf16(nomad_ctx* c, const void* xpad_dev, void* y_dev, int B, int T, nomad_stream_t stream, int variant) {
    if (!c || !xpad_dev || !y_dev || B <= 0 || T <= 0 || !c->pos_w16) return fail(NOMAD_ERR_INVALID, "nomad_diag_posconv_bf16: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* xpad = static_cast<const bf16_t*>(xpad_dev);
    bf16_t* y = static_cast<bf16_t*>(y_dev);
    const long long M = (long long)B * T;
    if (variant == 1) return run_posconv_bf16_slab(c, xpad, y, T, B, M, nullptr, nullptr, s);
    if (variant != 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_posconv_bf16: variant %d", variant);
    const long long grp_stride = (long long)B * (T + 128) * 48;
    auto asf = [](const bf16_t* p_) { return reinterpret_cast<const float*>(p_); };  // GemmParams carries typeless pointers
    auto asfm = [](bf16_t* p_) { return reinterpret_cast<float*>(p_); };
    GemmParams p{};
    p.A = asf(xpad);
    p.amap = RowMap{0, (long long)(T + 128) * 48, T, 48};
    p.a_goff = grp_stride;
    p.K = 6144;
    p.kchunk = 6144;
    p.W = asf(c->pos_w16);
    p.ldw = 6144;
    p.w_goff = 64LL * 6144;
    p.bias = c->pos_b;
    p.bias_goff = 48;
    p.C = asfm(y);
    p.cmap = plain_map(M, 768);
    p.c_goff = 48;
    p.R = asf(xpad);
    p.rmap = RowMap{64LL * 48, (long long)(T + 128) * 48, T, 48};
    p.r_goff = grp_stride;
    p.M = (int)M;
    p.N = 64;
    p.n_valid = 48;
    p.gelu = 1;
    return run_gemm_bf16(c, p, 16, s);
}

// timeline of the last probe GEMM: out_host[6 * n] = per workgroup {entry, loop start, loop end, stores done, HW_ID, XCC_ID}.  The probe kernels
// live in two translation units, each with its own copy of the device buffer: the copy with the newer stamps is the last probe's.
int nomad_diag_timeline(unsigned long long* out_host, int n) {
    if (!out_host || n <= 0 || n > kTimelineSlots) return fail(NOMAD_ERR_INVALID, "nomad_diag_timeline: bad argument");
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> a(6 * (size_t)n), b(6 * (size_t)n);
    if (int rc = gemm_f32_timeline_read(a.data(), n)) return rc;
    if (int rc = gemm_bf16_timeline_read(b.data(), n)) return rc;
    unsigned long long ma = 0, mb = 0;
    for (int i = 0; i < n; ++i) { ma = std::max(ma, a[6 * (size_t)i]); mb = std::max(mb, b[6 * (size_t)i]); }
    std::memcpy(out_host, (ma >= mb ? a : b).data(), sizeof(unsigned long long) * 6 * (size_t)n);
    return 0;
}
#endif

int nomad_diag_gemm_bf16(nomad_ctx* c, const void* A, const void* W, const float* bias, const void* R, void* C, int M,
                         int N, int K, int gelu, int tile, nomad_stream_t stream) {
    static const int kBN[] = {128, 128, 64, 256, 64, 128, 128, 128, 128, 256, 256, 128, 128, 128, 128, 256, 256, 256, 256, 256};
    const bool big256 = tile == 36 || (tile >= 42 && tile <= 54) || tile == 57 || tile == 58 || (tile >= 60 && tile <= 68);
    if (tile == 55 || tile == 56) {  // 256 x 192 tiles of the deep-pipelined kernel
        if (!c || !A || !W || !C || M <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm_bf16: bad argument");
        if (N % 192 || K % 128) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm_bf16: N %% 192 or K %% 128 != 0");
        GemmParams p192 = dense(static_cast<const float*>(A), K, static_cast<const float*>(W), bias, static_cast<const float*>(R),
                                static_cast<float*>(C), M, N, K, gelu);
        return run_gemm_bf16(c, p192, 1, static_cast<hipStream_t>(stream), tile);
    }
    if (!c || !A || !W || !C || M <= 0 || tile < 0 || (!big256 && tile >= static_cast<int>(sizeof(kBN) / sizeof(kBN[0])))) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm_bf16: bad argument");
    if (N % (big256 ? 256 : kBN[tile]) || K % 64) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm_bf16: N %% %d or K %% 64 != 0", big256 ? 256 : kBN[tile]);
    GemmParams p = dense(static_cast<const float*>(A), K, static_cast<const float*>(W), bias, static_cast<const float*>(R),
                         static_cast<float*>(C), M, N, K, gelu);
    return run_gemm_bf16(c, p, 1, static_cast<hipStream_t>(stream), tile);
}

int nomad_embed(nomad_ctx* c, const float* wav, int B, int n_samples, const float* head_w, const float* head_b,
                float* emb, float* layers_out, void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
    return forward_impl(c, wav, B, n_samples, head_w, head_b, emb, layers_out, workspace, workspace_bytes, stream, nullptr);
}

int nomad_workspace_bytes_ragged(const nomad_ctx* c, int B, const int* lengths_host, size_t* bytes) {
    RaggedShapes rs;
    if (!c || !bytes || !lengths_host || B <= 0 || !make_ragged(B, lengths_host, &rs))
        return fail(NOMAD_ERR_INVALID, "nomad_workspace_bytes_ragged: bad argument");
    *bytes = make_ragged_layout(rs).total;
    return 0;
}

int nomad_embed_ragged(nomad_ctx* c, const float* wav, int B, int stride, const int* lengths_host, const float* head_w,
                       const float* head_b, float* emb, void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
    return forward_ragged(c, wav, B, stride, lengths_host, head_w, head_b, emb, workspace, workspace_bytes, stream);
}

int nomad_saved_bytes(const nomad_ctx* c, int B, int n_samples, size_t* bytes) {
    Shapes sh;
    if (!c || !bytes || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_saved_bytes: bad shape B=%d N=%d", B, n_samples);
    *bytes = make_saved(sh, nullptr).total;
    return 0;
}

int nomad_backward_workspace_bytes(const nomad_ctx* c, int B, int n_samples, size_t* bytes) {
    Shapes sh;
    if (!c || !bytes || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_backward_workspace_bytes: bad shape B=%d N=%d", B, n_samples);
    *bytes = make_bwd_layout(sh).total;
    return 0;
}

int nomad_embed_train(nomad_ctx* c, const float* wav, int B, int n_samples, const float* head_w, const float* head_b,
                      float* emb, float* layers_out, void* saved, size_t saved_bytes, void* workspace,
                      size_t workspace_bytes, nomad_stream_t stream) {
    Shapes sh;
    if (!c || !saved || !layers_out || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_embed_train: bad argument");
    const Saved sv = make_saved(sh, saved);
    if (saved_bytes < sv.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed_train: saved block %zu < required %zu", saved_bytes, sv.total);
    return forward_impl(c, wav, B, n_samples, head_w, head_b, emb, layers_out, workspace, workspace_bytes, stream, &sv);
}

}  // extern "C"
namespace {
void rebuild_conv_bwd_weights(nomad_ctx* c, hipStream_t s);
}
extern "C" {

int nomad_enable_backward(nomad_ctx* c) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    if (c->bwd_ready) return 0;
    HIP_TRY(hipSetDevice(c->device));
    auto alloc = [&](size_t floats, float** out) -> int {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, floats * sizeof(float)));
        c->allocs.push_back(d);
        *out = static_cast<float*>(d);
        return 0;
    };
    auto transpose = [&](const float* in, int ld_in, float* out, int ld_out, int R, int C) {
        hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, 0, in, ld_in, out, ld_out, R, C);
    };
    int rc;
    for (int i = 1; i <= 4; ++i) {
        if ((rc = alloc(512 * 1024, &c->conv_bw_even[i]))) return rc;
        if ((rc = alloc(512 * 512, &c->conv_bw_odd[i]))) return rc;
    }
    for (int i = 5; i <= 6; ++i)
        if ((rc = alloc(1024 * 512, &c->conv_bw2[i]))) return rc;
    rebuild_conv_bwd_weights(c, 0);
    if ((rc = alloc(512 * 768, &c->proj_wT))) return rc;
    transpose(c->proj_w, 512, c->proj_wT, 768, 768, 512);
    if ((rc = alloc((size_t)16 * 64 * 6144, &c->pos_wb))) return rc;
    hipLaunchKernelGGL(posconv_bwd_weight_kernel, dim3(16 * 64), dim3(256), 0, 0, c->pos_w, c->pos_wb);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = alloc((size_t)768 * 2304, &c->qkv_wT[l]))) return rc;
        if ((rc = alloc((size_t)768 * 768, &c->o_wT[l]))) return rc;
        if ((rc = alloc((size_t)768 * 3072, &c->fc1_wT[l]))) return rc;
        if ((rc = alloc((size_t)3072 * 768, &c->fc2_wT[l]))) return rc;
        transpose(d.qkv_w, 768, c->qkv_wT[l], 2304, 2304, 768);
        transpose(d.o_w, 768, c->o_wT[l], 768, 768, 768);
        transpose(d.fc1_w, 768, c->fc1_wT[l], 3072, 3072, 768);
        transpose(d.fc2_w, 3072, c->fc2_wT[l], 768, 768, 3072);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (!c->splitk_part) {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, kSplitKPartFloats * sizeof(float)));
        c->allocs.push_back(d);
        c->splitk_part = static_cast<float*>(d);
    }

    c->bwd_ready = true;
    return 0;
}

int nomad_l1_loss_backward(nomad_ctx* c, const float* a_layers, const float* b_layers, const float* a_emb,
                           const float* b_emb, int B, int T, const float* upstream, float* dlayers, float* demb,
                           nomad_stream_t stream) {
    if (!c || !a_layers || !b_layers || !a_emb || !b_emb || !upstream || !dlayers || !demb || B <= 0 || T <= 0)
        return fail(NOMAD_ERR_INVALID, "nomad_l1_loss_backward: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long per_layer = (long long)B * T * 768;
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const float4*>(a_layers),
                       reinterpret_cast<const float4*>(b_layers), per_layer * 12 / 4, 1.0f / (float)per_layer, upstream,
                       reinterpret_cast<float4*>(dlayers));
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(64), dim3(256), 0, s, reinterpret_cast<const float4*>(a_emb),
                       reinterpret_cast<const float4*>(b_emb), (long long)B * 64, 1.0f / (float)(B * 256), upstream,
                       reinterpret_cast<float4*>(demb));
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

// dW[Nout][Kin] += scale * dY^T X from the transposed operands TA = dY^T [Nout][Mp], TB = X^T [Kin][Mp]: the forward's
// GEMM kernel with the contraction split over `S` groups, partial products folded in fixed order.
static int dw_gemm(nomad_ctx* c, const float* TA, const float* TB, int Nout, int Kin, int Mp, float* part, float* out,
                   int rows_scaled, float scale, hipStream_t s) {
    // 128x64 tiles; enough splits for ~1500 workgroups (measured on the training shapes: 512 -> 1536 workgroups
    // is 4 % of a step, beyond that nothing)
    // (bf16x3 products, nomad_set_gemm_precision: 128x128 tiles of four 64x64 wave tiles - 12 MFMAs per 4 fragment splits)
    const bool wide = c->gemm_x3 && Kin % 128 == 0;
    const int tile = wide ? 20 : 34;
    const int tiles = (Nout / 128) * (Kin / (wide ? 128 : 64));
    int S = 1;
    while (S < 16 && tiles * S < 1536 && (size_t)(2 * S) * Nout * Kin <= kSplitPartFloats) S *= 2;
    if ((size_t)S * Nout * Kin > kSplitPartFloats) return fail(NOMAD_ERR_INVALID, "dw_gemm: partial buffer too small");
    const int Kc = Mp / S;
    GemmParams p = dense(TA, Mp, TB, nullptr, nullptr, part, Nout, Kin, Kc, 0);
    p.ldw = Mp;
    p.a_goff = Kc;
    p.w_goff = Kc;
    p.c_goff = (long long)Nout * Kin;
    int rc;
    if ((rc = run_gemm(c, p, S, tile, s))) return rc;
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    const long long n4 = (long long)Nout * Kin / 4, n4s = (long long)rows_scaled * Kin / 4;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    float4* o4 = reinterpret_cast<float4*>(out);
    if (n4s > 0)  // leading rows with their own scale (q rows of the fused q/k/v weight); every partial keeps stride n4
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4s + 255) / 256)), dim3(256), 0, s, p4, S, n4, n4s, o4, scale);
    if (n4 > n4s) {
        // rows after the scaled block: same partial stride, shifted start
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 - n4s + 255) / 256)), dim3(256), 0, s, p4 + n4s, S, n4,
                           n4 - n4s, o4 + n4s, 1.0f);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static int backward_impl(nomad_ctx* c, const float* wav, int B, int n_samples, const float* head_w, const float* head_b,
                         const float* layers_out, const void* saved, size_t saved_bytes, const float* dlayers,
                         const float* demb, float* dwav, void* workspace, size_t workspace_bytes,
                         nomad_stream_t stream, bool train) {
    Shapes sh;
    if (!c || !wav || !layers_out || !saved || !demb || (!dwav && !train) || !workspace || B <= 0 ||
        !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_embed_backward: bad argument");
    if (!c->bwd_ready) return fail(NOMAD_ERR_INVALID, "nomad_embed_backward: call nomad_enable_backward first");
    if (train && !c->train_ready) return fail(NOMAD_ERR_INVALID, "nomad_train_backward: call nomad_train_enable first");
    const Saved sv = make_saved(sh, const_cast<void*>(saved));
    const bool conv_pg = train && c->train_convnet;  // parameter gradients of the conv feature extractor too
    const BwdLayout lay = make_bwd_layout(sh, train, conv_pg);
    if (saved_bytes < sv.total || workspace_bytes < lay.total)
        return fail(NOMAD_ERR_WORKSPACE, "nomad_embed_backward: saved %zu/%zu, workspace %zu/%zu", saved_bytes, sv.total,
                    workspace_bytes, lay.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int T = sh.T, M = sh.M;
    const SplitKScope splitk(c, !train && !c->train_ready);  // d loss / d waveform of Nomad.forward(): small-M GEMMs may split K
    float* const prev_cur_b = c->splitk_cur;
    c->splitk_cur = c->splitk_part;
    struct CurRestoreB { nomad_ctx* c; float* v; ~CurRestoreB() { c->splitk_cur = v; } } cur_restore_b{c, prev_cur_b};
    float *gx = F(lay.gx), *dya = F(lay.dya), *dyb = F(lay.dyb), *dh = F(lay.dh), *dqkv = F(lay.dqkv);
    int rc;
    // the regularisation of the forward this backward belongs to (the caller re-sets it: nomad_train_set_stochastic)
    const DropCfg d_in = make_drop(c, train ? c->p_input : 0.f), d_res = make_drop(c, train ? c->p_drop : 0.f),
                  d_att = make_drop(c, train ? c->p_attn : 0.f);
    int nbr = 1;  // LayerDrop masks: one for the call, or one per branch of a merged batch (see forward_impl)
    unsigned bmask[4] = {train ? c->layer_mask : 0xFFFu, 0xFFFu, 0xFFFu, 0xFFFu};
    if (train && c->branches > 1) {
        if (B % c->branches) return fail(NOMAD_ERR_INVALID, "nomad_train_backward: B=%d is not %d equal branches", B, c->branches);
        nbr = c->branches;
        for (int i = 0; i < nbr; ++i) bmask[i] = c->branch_mask[i];
    }
    const long long act = (long long)M * 768;
    float* dmask = train ? F(lay.dmask) : nullptr;

    // ---- parameter-gradient helpers (train only); Ms rows of the (sub-)batch, Mps = Ms rounded up to 512 ------
    const ParamOffsets po = make_param_offsets();
    auto G = [&](size_t off) { return c->grad + off; };
    float *TA = train ? F(lay.ta) : nullptr, *TB = train ? F(lay.tb) : nullptr, *part = train ? F(lay.kpart) : nullptr;
    auto tpose = [&](const float* in, int C, float* out, bool gelu, int Ms, int Mps) {  // out[C][Mps] = f(in[Ms][C])^T, zero padded
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        const dim3 grid(Mps / 64, (C + 63) / 64), blk(256);
        if (gelu) hipLaunchKernelGGL(transpose_pad_kernel<1>, grid, blk, 0, s, in, C, out, Mps, Ms, C);
        else hipLaunchKernelGGL(transpose_pad_kernel<0>, grid, blk, 0, s, in, C, out, Mps, Ms, C);
    };
    auto rowsum = [&](const float* in, int rows, float* out, float scale, int Mps) {  // bias gradient from dY^T
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(rowsum_acc_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in, Mps, rows, out, scale);
    };
    auto ln_params = [&](const float* x, const float* g, const float* g2, int N, float* dgam, float* dbet, int Ms) {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        float* lp = F(lay.lnpart);
        const int nblk = (Ms + kLnRows - 1) / kLnRows;
        if (N == 768) hipLaunchKernelGGL(ln_param_partial_kernel<3>, dim3(nblk), dim3(256), 0, s, x, g, g2, lp, Ms);
        else hipLaunchKernelGGL(ln_param_partial_kernel<2>, dim3(nblk), dim3(256), 0, s, x, g, g2, lp, Ms);
        hipLaunchKernelGGL(ln_param_final_kernel, dim3(2 * N / 64), dim3(256), 0, s, lp, nblk, N, dgam, dbet);
    };

    // parameter gradients of the ENCODER (pos-conv, encoder LayerNorm, the 12 layers): not with freeze_all, where only
    // post_extract_proj, the feature LayerNorm and the head stay trainable (train_triplet.py:76-79)
    const bool pg = train && !c->freeze_encoder;
    // ---- head -> d loss / d x_12 -------------------------------------------------------------------
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(1024), 0, s, layers_out + (size_t)11 * M * 768, T,
                           head_w ? head_w : c->emb_w, head_b ? head_b : c->emb_b, demb, gx,
                           train ? F(lay.headp) : nullptr, train ? F(lay.headdz) : nullptr);
        if (train)
            hipLaunchKernelGGL(head_param_grad_kernel, dim3(256), dim3(256), 0, s, F(lay.headp), F(lay.headdz), B,
                               G(po.emb_w), G(po.emb_b));
    }
    // ---- 12 transformer layers, last to first ---------------------------------------------------------
    // One layer over clips [c0, c0 + nc): the whole batch, or one branch when LayerDrop split the branches.
    bool lnb_prefused = false;   // the LayerNorm backward at the head of the next bwd_layer call (or the encoder's) ran inside the last GEMM's epilogue
    bool whole_batch = true;     // every layer runs for every clip (no LayerDrop branch skips one)
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l)
        for (int br = 0; br < nbr; ++br) whole_batch = whole_batch && ((bmask[br] >> l) & 1u);
    auto bwd_layer = [&](int l, int c0, int nc) -> int {
        const LayerDev& d = c->layers[l];
        const LayerOffsets& lo = po.L[l];
        const long long r0 = (long long)c0 * T;
        const int Ms = nc * T, Mps = (Ms + 511) / 512 * 512;
        const long long acts = (long long)Ms * 768;
        const unsigned long long idx0 = (unsigned long long)r0 * 768;
        const SavedLayer& sl0 = sv.L[l];
        const float *y2 = sl0.y2 + r0 * 768, *y1 = sl0.y1 + r0 * 768, *u = sl0.u + r0 * 3072, *qkv = sl0.qkv + r0 * 2304,
                    *ctx = sl0.ctx + r0 * 768, *lse = sl0.lse + (long long)c0 * 12 * T;
        float *gxs = gx + r0 * 768, *dyas = dya + r0 * 768, *dybs = dyb + r0 * 768, *dhs = dh + r0 * 3072,
              *dqkvs = dqkv + r0 * 2304, *dmasks = dmask ? dmask + r0 * 768 : nullptr;
        const float* dl = dlayers ? dlayers + (size_t)l * M * 768 + r0 * 768 : nullptr;
        int rc;
        if (!lnb_prefused && (rc = run_ln_bwd(c, y2, gxs, dl, d.ln2_w, dyas, Ms, 768, s))) return rc;   // dy2 (prefused: by the layer above's last GEMM)
        lnb_prefused = false;
        // dy2 feeds the residual as is and the fc2 branch through its dropout mask
        const float* dy2b = dyas;
        if (d_res.threshold) {
            if ((rc = run_dropout(c, dyas, nullptr, dmasks, acts, d_res, site_ffn(l), s, idx0))) return rc;
            dy2b = dmasks;
        }
        if (pg) {
            ln_params(y2, gxs, dl, 768, G(lo.ln2_w), G(lo.ln2_b), Ms);
            tpose(dy2b, 768, TA, false, Ms, Mps);
            rowsum(TA, 768, G(lo.fc2_b), 1.0f, Mps);
            tpose(u, 3072, TB, true, Ms, Mps);  // h = gelu(u), recomputed
            if ((rc = dw_gemm(c, TA, TB, 768, 3072, Mps, part, G(lo.fc2_w), 0, 1.0f, s))) return rc;
        }
        if ((rc = bwd_gemm(c, dy2b, c->fc2_wT[l], dhs, Ms, 3072, 768, u, nullptr, s))) return rc;      // du = (dy2 W2) * gelu'(u)
        if (pg) {
            tpose(dhs, 3072, TA, false, Ms, Mps);
            rowsum(TA, 3072, G(lo.fc1_b), 1.0f, Mps);
            // fc1's input = LayerNorm(y1), recomputed into gx (the upstream gradient it held has been consumed)
            if ((rc = run_layernorm(c, y1, d.ln1_w, d.ln1_b, gxs, nullptr, Ms, 768, s))) return rc;
            tpose(gxs, 768, TB, false, Ms, Mps);
            if ((rc = dw_gemm(c, TA, TB, 3072, 768, Mps, part, G(lo.fc1_w), 0, 1.0f, s))) return rc;
        }
        // (dX-only backward: when the GEMM splits K, the LayerNorm backward forms dx1 from the partial products itself - pending_lnb)
        c->pending_lnb = {};
        if (!train) {
            c->pending_lnb.x = y1; c->pending_lnb.gamma = d.ln1_w; c->pending_lnb.out = dyas; c->pending_lnb.armed = true;
        }
        rc = bwd_gemm(c, dhs, c->fc1_wT[l], dybs, Ms, 768, 3072, nullptr, dyas, s);                    // dx1 = du W1 + dy2
        const bool ln1_done = c->pending_lnb.armed && c->pending_lnb.done;
        c->pending_lnb = {};
        if (rc) return rc;
        if (!ln1_done && (rc = run_ln_bwd(c, y1, dybs, nullptr, d.ln1_w, dyas, Ms, 768, s))) return rc;   // dy1
        if (pg) ln_params(y1, dybs, nullptr, 768, G(lo.ln1_w), G(lo.ln1_b), Ms);
        const float* dy1b = dyas;  // dy1 through out_proj's dropout mask
        if (d_res.threshold) {
            if ((rc = run_dropout(c, dyas, nullptr, dmasks, acts, d_res, site_proj(l), s, idx0))) return rc;
            dy1b = dmasks;
        }
        if (pg) {
            tpose(dy1b, 768, TA, false, Ms, Mps);
            rowsum(TA, 768, G(lo.o_b), 1.0f, Mps);
            tpose(ctx, 768, TB, false, Ms, Mps);
            if ((rc = dw_gemm(c, TA, TB, 768, 768, Mps, part, G(lo.o_w), 0, 1.0f, s))) return rc;
        }
        if ((rc = bwd_gemm(c, dy1b, c->o_wT[l], dybs, Ms, 768, 768, nullptr, nullptr, s))) return rc;  // dctx
        {
            Scope sc(c, s, NOMAD_K_ATTN, 14.0 * nc * 12.0 * (double)T * T * 64);  // 7 T x T x 64 products (S, dP twice)
            HIP_TRY(launch_attention_bwd(qkv, ctx, dybs, lse, F(lay.attnd), dqkvs, nc, T, d_att, site_attn(l), s, c0 * 12, !c->tune.attn_bwd_small));
        }
        if (pg) {
            // the forward's fused weight holds q scaled by head_dim^-0.5: d q_proj = 0.125 * d fused rows 0..767
            tpose(dqkvs, 2304, TA, false, Ms, Mps);
            rowsum(TA, 768, G(lo.qkv_b), 0.125f, Mps);
            rowsum(TA + (size_t)768 * Mps, 1536, G(lo.qkv_b) + 768, 1.0f, Mps);
            const float* xin = layers_out + (size_t)(l > 0 ? l - 1 : 0) * M * 768 + r0 * 768;
            if (l == 0) {  // layer 0 reads dropout(LayerNorm(y0)): recomputed into dyb (dctx has been consumed)
                if ((rc = run_layernorm(c, sv.y0 + r0 * 768, c->eln_w, c->eln_b, dybs, nullptr, Ms, 768, s))) return rc;
                if (d_res.threshold && (rc = run_dropout(c, dybs, nullptr, dybs, acts, d_res, kSiteEncoder, s, idx0))) return rc;
                xin = dybs;
            }
            tpose(xin, 768, TB, false, Ms, Mps);
            if ((rc = dw_gemm(c, TA, TB, 2304, 768, Mps, part, G(lo.qkv_w), 768, 0.125f, s))) return rc;
        }
        // dx_in = dqkv Wqkv + dy1; what follows it is the LayerNorm backward of the layer below (its LN2, with that layer's output gradient) or
        // of the encoder's LayerNorm: fused into the split-K epilogue where the whole batch walks the layers together (no LayerDrop)
        c->pending_lnb = {};
        if (!train && whole_batch && !d_res.threshold) {
            if (l > 0) {
                c->pending_lnb.x = sv.L[l - 1].y2; c->pending_lnb.gamma = c->layers[l - 1].ln2_w;
                c->pending_lnb.g2 = dlayers ? dlayers + (size_t)(l - 1) * M * 768 : nullptr;
            } else {
                c->pending_lnb.x = sv.y0; c->pending_lnb.gamma = c->eln_w;
            }
            c->pending_lnb.out = dya; c->pending_lnb.armed = true;
        }
        rc = bwd_gemm(c, dqkvs, c->qkv_wT[l], gxs, Ms, 768, 2304, nullptr, dyas, s);
        lnb_prefused = c->pending_lnb.armed && c->pending_lnb.done;
        c->pending_lnb = {};
        return rc;
    };
    for (int l = NOMAD_NUM_LAYERS - 1; l >= 0; --l) {
        unsigned all = 1u;
        for (int br = 0; br < nbr; ++br) all &= (bmask[br] >> l) & 1u;
        if (all) {
            if ((rc = bwd_layer(l, 0, B))) return rc;
            continue;
        }
        for (int br = 0; br < nbr; ++br)  // LayerDrop: identity in the forward, identity here, per branch
            if ((bmask[br] >> l) & 1u)
                if ((rc = bwd_layer(l, br * (B / nbr), B / nbr))) return rc;
    }
    const int Mp = lay.Mp;
    // ---- encoder input: LayerNorm, x + gelu(pos_conv(x)) --------------------------------------------
    if (d_res.threshold && (rc = run_dropout(c, gx, nullptr, gx, act, d_res, kSiteEncoder, s))) return rc;
    if (!lnb_prefused && (rc = run_ln_bwd(c, sv.y0, gx, nullptr, c->eln_w, dya, M, 768, s))) return rc;   // dy0
    if (pg) ln_params(sv.y0, gx, nullptr, 768, G(po.eln_w), G(po.eln_b), M);
    const long long grp_stride = (long long)B * (T + 128) * 48;
    {
        float* dug = F(lay.dug);
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            hipLaunchKernelGGL(zero_pad_rows_kernel<float>, dim3(16 * B), dim3(256), 0, s, dug, T, kNoInts, kNoInts, B);
            hipLaunchKernelGGL(dgelu_to_groups_kernel, dim3(M), dim3(192), 0, s, dya, sv.upc, dug, T, grp_stride);
        }
        GemmParams p{};
        p.A = dug;
        p.amap = RowMap{48, (long long)(T + 128) * 48, T, 48};  // row (clip, t) starts at buffer frame t + 1
        p.a_goff = grp_stride;
        p.K = 6144;
        p.kchunk = 6144;
        p.W = c->pos_wb;
        p.ldw = 6144;
        p.w_goff = 64LL * 6144;
        p.C = dyb;
        p.cmap = plain_map(M, 768);
        p.c_goff = 48;
        p.R = dya;
        p.rmap = plain_map(M, 768);
        p.r_goff = 48;
        p.M = M;
        p.N = 64;
        p.n_valid = 48;
        if ((rc = run_gemm(c, p, 16, 48, s))) return rc;  // one instantiation for every batch size: same summation order
    }
    if (train) {
        float *dug = F(lay.dug), *xg = F(lay.xg), *dwe = F(lay.dwe), *featln = F(lay.f2);
        // post_extract_proj's input (LayerNorm of the extractor output) is recomputed, not saved
        if ((rc = run_layernorm(c, sv.c6, c->fln_w, c->fln_b, featln, nullptr, M, 512, s))) return rc;
        if (pg) {
        // ---- pos-conv parameters: bias, then weight_g / weight_v through the weight norm --------------------
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            hipLaunchKernelGGL(posconv_bias_partial_kernel, dim3(kPbChunks, 16), dim3(256), 0, s, dug, (long long)B * (T + 128),
                               F(lay.lnpart));
            hipLaunchKernelGGL(posconv_bias_final_kernel, dim3(1), dim3(768), 0, s, F(lay.lnpart), G(po.pos_b));
        }
        // the conv's input (post_extract_proj output, group-major, zero padded) is recomputed, not saved
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            hipLaunchKernelGGL(zero_pad_rows_kernel<float>, dim3(16 * B), dim3(256), 0, s, xg, T, kNoInts, kNoInts, B);
        }
        {
            GemmParams p = dense(featln, 512, c->proj_w, c->proj_b, nullptr, xg, M, 768, 512, 0);
            p.cmap = RowMap{64LL * 48, (long long)(T + 128) * 48, T, 48};
            p.c_colblk = 48;
            p.c_colblk_stride = grp_stride;
            if ((rc = run_gemm(c, p, 1, pick_tile(c, M, 768, 512), s))) return rc;
        }
        if (d_in.threshold) {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            hipLaunchKernelGGL(dropout_groups_kernel, dim3(M), dim3(192), 0, s, xg, T, grp_stride, d_in, kSiteInput);
        }
        {
            const int S = lay.pos_split, cps = (B + S - 1) / S;
            Scope sc(c, s, NOMAD_K_GEMM, 2.0 * M * 768.0 * 48 * 128);
            hipLaunchKernelGGL(posconv_dw_kernel, dim3(128 / kPdwTaps, 16, S), dim3(256), 0, s, dug, xg, part, B, T, cps);
            hipLaunchKernelGGL(posconv_dw_gather_kernel, dim3(768), dim3(256), 0, s, part, S, dwe);
            hipLaunchKernelGGL(tap_dot_partial_kernel, dim3(576), dim3(256), 0, s, dwe, c->theta + po.pos_v, c->tap_partial);
            hipLaunchKernelGGL(tap_sum_final_kernel, dim3(1), dim3(1024), 0, s, c->tap_partial, 576, c->tap_dot);
            hipLaunchKernelGGL(posconv_wn_bwd_kernel, dim3(768 * 48 * 128 / 256), dim3(256), 0, s, dwe, c->theta + po.pos_v,
                               c->theta + po.pos_g, c->pos_nrm2, c->tap_dot, G(po.pos_v), G(po.pos_g));
        }
        }
        // ---- post_extract_proj parameters (its output went through dropout_input) ------------------------
        if (d_in.threshold && (rc = run_dropout(c, dyb, nullptr, dyb, act, d_in, kSiteInput, s))) return rc;
        tpose(dyb, 768, TA, false, M, Mp);
        rowsum(TA, 768, G(po.proj_b), 1.0f, Mp);
        tpose(featln, 512, TB, false, M, Mp);
        if ((rc = dw_gemm(c, TA, TB, 768, 512, Mp, part, G(po.proj_w), 0, 1.0f, s))) return rc;
    }
    // ---- post_extract_proj, LayerNorm(512), GELU of conv6 ----------------------------------------------
    if ((rc = bwd_gemm(c, dyb, c->proj_wT, F(lay.f1), M, 512, 768, nullptr, nullptr, s))) return rc;
    if (train) ln_params(sv.c6, F(lay.f1), nullptr, 512, G(po.fln_w), G(po.fln_b), M);
    HIP_TRY(hipGetLastError());
    if (!dwav && !conv_pg) return 0;  // frozen conv feature extractor (freeze_convnet: True): nothing upstream needs a gradient
    if (c->feature_grad_mult == 0.f) {  // fairseq runs the extractor under no_grad then: no gradient reaches it or the waveform
        if (dwav) HIP_TRY(hipMemsetAsync(dwav, 0, sizeof(float) * (size_t)B * n_samples, s));
        return 0;
    }
    // d W_i of conv layer i (freeze_convnet: False): dU_i^T x im2col(input)^T contracted over all frames, per-clip column
    // blocks of conv_lp(L_i) (zero padded), the same split-K dW GEMM as the encoder's, then back to [co][ci][tap]
    auto conv_dw = [&](int i, const float* dU, const float* in, long long in_clip, bool gelu_in) -> int {
        const int Lout = sh.L[i], taps = kConvK[i], Kin = taps * 512;
        const int Lp = (int)conv_lp(Lout);
        const long long cols = (long long)B * Lp;
        {
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            hipLaunchKernelGGL(transpose_clips_kernel<0>, dim3(Lp / 64, 8, B), dim3(256), 0, s, dU + 512, (long long)(Lout + 2) * 512, 512,
                               TA, cols, Lp, Lout, 512);
            const dim3 grid(Lp / 64, Kin / 64, B);
            if (gelu_in)
                hipLaunchKernelGGL(transpose_clips_kernel<1>, grid, dim3(256), 0, s, in, in_clip, kConvS[i] * 512, TB, cols, Lp, Lout, Kin);
            else
                hipLaunchKernelGGL(transpose_clips_kernel<0>, grid, dim3(256), 0, s, in, in_clip, kConvS[i] * 512, TB, cols, Lp, Lout, Kin);
        }
        float* tmp = F(lay.convtmp);
        HIP_TRY(hipMemsetAsync(tmp, 0, sizeof(float) * 512 * (size_t)Kin, s));
        int rc2;
        if ((rc2 = dw_gemm(c, TA, TB, 512, Kin, (int)cols, part, tmp, 0, 1.0f, s))) return rc2;
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        const long long n = 512LL * Kin;
        hipLaunchKernelGGL(conv_grad_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tmp, G(po.conv_w[i]), taps);
        return 0;
    };
    if ((rc = run_ln_bwd(c, sv.c6, F(lay.f1), nullptr, c->fln_w, F(lay.f2), M, 512, s))) return rc;
    float* bufs[2] = {F(lay.bufa), F(lay.bufb)};  // dU6 -> a, dU5 -> b, ..., dU1 -> b, G0 -> a
    {
        HIP_TRY(hipMemsetAsync(bufs[0], 0, sizeof(float) * 512 * (size_t)B * (T + 2), s));
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        const RowMap om{512, (long long)(T + 2) * 512, T, 512};
        // GradMultiply(features, feature_grad_mult) sits on the extractor's (post-GELU) output: its backward scales the
        // gradient that enters GELU'
        hipLaunchKernelGGL(dgelu_rows512_kernel, dim3(M), dim3(128), 0, s, F(lay.f2), sv.u[6], bufs[0], om, M,
                           c->feature_grad_mult);
    }
    // ---- conv6..conv1: transposed strided convolutions as GEMMs over the padded dU buffers -------------
    for (int i = 6; i >= 1; --i) {
        const float* dU = bufs[i % 2];        // [B][L_i + 2][512], data at rows 1..L_i
        float* out = bufs[(i + 1) % 2];
        const int Lout = sh.L[i], Lin = sh.L[i - 1];
        const bool to_g0 = (i == 1);          // conv0's output gradient is compact and gets no GELU' here
        const long long out_clip = to_g0 ? (long long)Lin * 512 : (long long)(Lin + 2) * 512;
        const long long out_off = to_g0 ? 0 : 512;
        // the two GEMMs below write frames 0 .. covered - 1 of every clip (k = 3: every input frame; k = 2: 2 L_out of them); what they
        // do not write is zeroed here: the two pad rows of a padded buffer and the frames behind the last window (a full memset of
        // the buffer cost 20-40 us per layer at configs[3]'s size)
        if (!to_g0) {
            const int covered = kConvK[i] == 2 ? 2 * Lout : Lin;
            Scope sc(c, s, NOMAD_K_ROW, 0.0);
            hipLaunchKernelGGL(zero_rows512_kernel, dim3(B), dim3(256), 0, s, out, out_clip, 0, 1, covered + 1, Lin + 2);
        } else if (kConvK[i] == 2) {
            HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * out_clip, s));   // (not reached: conv1 has k = 3 and writes every frame of G0)
        }
        if (conv_pg) {  // input of layer i: gelu(u_{i-1}) recomputed in the transpose; for i = 1 conv0's output, recomputed whole
            if (i == 1) {
                Scope sc(c, s, NOMAD_K_FRONT, 0.0);
                hipLaunchKernelGGL(conv0_gn_gelu_kernel<float>, dim3((sh.L[0] + kConv0Frames - 1) / kConv0Frames, B), dim3(256), 0, s, wav,
                                   n_samples, sh.L[0], c->conv0_w, sv.gn_scale, sv.gn_shift, F(lay.h0), kNoInts, kNoInts, 0LL);
            }
            const float* in = i == 1 ? F(lay.h0) : sv.u[i - 1];
            if ((rc = conv_dw(i, dU, in, (long long)Lin * 512, i != 1))) return rc;
        }
        GemmParams p{};
        p.A = dU;
        p.C = out;
        p.DG = to_g0 ? nullptr : sv.u[i - 1];
        p.rmap = plain_map(1, 1);
        if (kConvK[i] == 2) {
            p.amap = RowMap{512, (long long)(Lout + 2) * 512, Lout, 512};
            p.K = 512;
            p.W = c->conv_bw2[i];
            p.N = 1024;
            p.M = B * Lout;
            p.cmap = RowMap{out_off, out_clip, Lout, 1024};
            p.dgmap = RowMap{0, (long long)Lin * 512, Lout, 1024};
            p.kchunk = p.K; p.ldw = p.K; p.n_valid = p.N;
            if ((rc = run_gemm(c, p, 1, pick_tile(c, p.M, p.N, p.K), s))) return rc;
        } else {
            const int E = (Lin + 1) / 2, O = Lin / 2;
            // even input frames 2t': dU[t'-1] W_tap2 + dU[t'] W_tap0
            p.amap = RowMap{0, (long long)(Lout + 2) * 512, E, 512};
            p.K = 1024;
            p.W = c->conv_bw_even[i];
            p.N = 512;
            p.M = B * E;
            p.cmap = RowMap{out_off, out_clip, E, 1024};
            p.dgmap = RowMap{0, (long long)Lin * 512, E, 1024};
            p.kchunk = p.K; p.ldw = p.K; p.n_valid = p.N;
            if ((rc = run_gemm(c, p, 1, pick_tile(c, p.M, p.N, p.K), s))) return rc;
            // odd input frames 2t'+1: dU[t'] W_tap1
            p.amap = RowMap{512, (long long)(Lout + 2) * 512, O, 512};
            p.K = 512;
            p.W = c->conv_bw_odd[i];
            p.M = B * O;
            p.cmap = RowMap{out_off + 512, out_clip, O, 1024};
            p.dgmap = RowMap{512, (long long)Lin * 512, O, 1024};
            p.kchunk = p.K; p.ldw = p.K;
            if (O > 0 && (rc = run_gemm(c, p, 1, pick_tile(c, p.M, p.N, p.K), s))) return rc;
        }
    }
    // ---- conv0 + GroupNorm -> d loss / d waveform -------------------------------------------------------
    {
        const float* G0 = bufs[0];
        float* partial = F(lay.partial);
        if (dwav) HIP_TRY(hipMemsetAsync(dwav, 0, sizeof(float) * (size_t)B * n_samples, s));
        Scope sc(c, s, NOMAD_K_FRONT, 0.0);
        const dim3 grid(lay.nchunks, B);
        float* fold = F(lay.gnfold);
        hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(lay.nstat, B), dim3(256), 0, s, wav, n_samples, sh.L[0], c->conv0_w, sv.gn_scale,
                           sv.gn_shift, sv.gn_mean, sv.gn_rstd, G0, partial);
        hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(B), dim3(512), 0, s, partial, lay.nstat, sh.L[0], c->conv0_w, sv.gn_scale, sv.gn_shift,
                           sv.gn_mean, sv.gn_rstd, fold, F(lay.cwtab));
        if (dwav)
            hipLaunchKernelGGL(conv0_bwd_kernel, dim3((sh.L[0] + kC0Frames - 1) / kC0Frames, B), dim3(256), 0, s, wav, n_samples, sh.L[0],
                               F(lay.cwtab), G0, dwav);
        if (conv_pg) {  // conv0 weight, GroupNorm gamma / beta
            hipLaunchKernelGGL(conv0_param_partial_kernel, grid, dim3(256), 0, s, wav, n_samples, sh.L[0], c->conv0_w, sv.gn_scale,
                               sv.gn_shift, sv.gn_mean, sv.gn_rstd, G0, fold, lay.nchunks, F(lay.c0part));
            hipLaunchKernelGGL(conv0_param_final_kernel, dim3(24), dim3(256), 0, s, F(lay.c0part), fold, B, lay.nchunks,
                               G(po.conv0_w), G(po.gn_w), G(po.gn_b));
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- fine-tuning state ------------------------------------------------------------------------------------
namespace {

struct Segment {
    std::string name;
    size_t offset, count;
};

// Checkpoint key -> slice of the flat parameter vector: every parameter of nomad_best_model.pt that train_triplet.py can
// train (mask_emb gets no gradient with mask=False).  The conv feature extractor's slices keep a zero gradient unless
// nomad_train_set_convnet(ctx, 1) (config freeze_convnet: False).
const std::vector<Segment>& segments() {
    static const std::vector<Segment> segs = [] {
        std::vector<Segment> v;
        const ParamOffsets o = make_param_offsets();
        const std::string p = "ssl_model.";
        v.push_back({p + "layer_norm.weight", o.fln_w, 512});
        v.push_back({p + "layer_norm.bias", o.fln_b, 512});
        v.push_back({p + "post_extract_proj.weight", o.proj_w, 768 * 512});
        v.push_back({p + "post_extract_proj.bias", o.proj_b, 768});
        v.push_back({p + "encoder.pos_conv.0.weight_g", o.pos_g, 128});
        v.push_back({p + "encoder.pos_conv.0.weight_v", o.pos_v, (size_t)768 * 48 * 128});
        v.push_back({p + "encoder.pos_conv.0.bias", o.pos_b, 768});
        v.push_back({p + "encoder.layer_norm.weight", o.eln_w, 768});
        v.push_back({p + "encoder.layer_norm.bias", o.eln_b, 768});
        for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
            const LayerOffsets& q = o.L[l];
            const std::string b = p + "encoder.layers." + std::to_string(l) + ".";
            const size_t ww = 768 * 768;
            v.push_back({b + "self_attn.q_proj.weight", q.qkv_w, ww});
            v.push_back({b + "self_attn.k_proj.weight", q.qkv_w + ww, ww});
            v.push_back({b + "self_attn.v_proj.weight", q.qkv_w + 2 * ww, ww});
            v.push_back({b + "self_attn.q_proj.bias", q.qkv_b, 768});
            v.push_back({b + "self_attn.k_proj.bias", q.qkv_b + 768, 768});
            v.push_back({b + "self_attn.v_proj.bias", q.qkv_b + 1536, 768});
            v.push_back({b + "self_attn.out_proj.weight", q.o_w, ww});
            v.push_back({b + "self_attn.out_proj.bias", q.o_b, 768});
            v.push_back({b + "self_attn_layer_norm.weight", q.ln1_w, 768});
            v.push_back({b + "self_attn_layer_norm.bias", q.ln1_b, 768});
            v.push_back({b + "fc1.weight", q.fc1_w, (size_t)3072 * 768});
            v.push_back({b + "fc1.bias", q.fc1_b, 3072});
            v.push_back({b + "fc2.weight", q.fc2_w, (size_t)768 * 3072});
            v.push_back({b + "fc2.bias", q.fc2_b, 768});
            v.push_back({b + "final_layer_norm.weight", q.ln2_w, 768});
            v.push_back({b + "final_layer_norm.bias", q.ln2_b, 768});
        }
        const std::string fe = p + "feature_extractor.conv_layers.";
        v.push_back({fe + "0.0.weight", o.conv0_w, 512 * 10});
        v.push_back({fe + "0.2.weight", o.gn_w, 512});
        v.push_back({fe + "0.2.bias", o.gn_b, 512});
        for (int i = 1; i < 7; ++i) v.push_back({fe + std::to_string(i) + ".0.weight", o.conv_w[i], (size_t)512 * 512 * kConvK[i]});
        v.push_back({"embedding_layer.1.weight", o.emb_w, 256 * 768});
        v.push_back({"embedding_layer.1.bias", o.emb_b, 256});
        return v;
    }();
    return segs;
}

// The transposed conv weights the dX chain of the backward contracts with (from the kernel-layout forward weights).
void rebuild_conv_bwd_weights(nomad_ctx* c, hipStream_t s) {
    auto transpose = [&](const float* in, int ld_in, float* out, int ld_out, int R, int C) {
        hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, s, in, ld_in, out, ld_out, R, C);
    };
    for (int i = 1; i <= 4; ++i) {  // k = 3: forward repack is [n][tap*512 + c]
        transpose(c->conv_w[i] + 2 * 512, 1536, c->conv_bw_even[i], 1024, 512, 512);        // tap 2 pairs with dU[t'-1]
        transpose(c->conv_w[i], 1536, c->conv_bw_even[i] + 512, 1024, 512, 512);            // tap 0 pairs with dU[t']
        transpose(c->conv_w[i] + 512, 1536, c->conv_bw_odd[i], 512, 512, 512);              // tap 1
    }
    for (int i = 5; i <= 6; ++i)  // k = 2: [n][tap*512 + c] -> [(tap*512 + c)][n]
        transpose(c->conv_w[i], 1024, c->conv_bw2[i], 512, 512, 1024);
}

// Rebuild every kernel-layout weight that is not a plain alias of the master vector: the fused, q-scaled q/k/v
// weights, the weight-normed pos-conv kernel, and the transposed copies the backward contracts with.
int refresh_weights(nomad_ctx* c, hipStream_t s) {
    const ParamOffsets po = make_param_offsets();
    auto transpose = [&](const float* in, int ld_in, float* out, int ld_out, int R, int C) {
        hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, s, in, ld_in, out, ld_out, R, C);
    };
    hipLaunchKernelGGL(tap_dot_partial_kernel, dim3(576), dim3(256), 0, s, c->theta + po.pos_v, (const float*)nullptr,
                       c->tap_partial);
    hipLaunchKernelGGL(tap_sum_final_kernel, dim3(1), dim3(1024), 0, s, c->tap_partial, 576, c->pos_nrm2);
    hipLaunchKernelGGL(posconv_fold_kernel, dim3(768), dim3(256), 0, s, c->theta + po.pos_v, c->theta + po.pos_g,
                       c->pos_nrm2, c->pos_w);
    hipLaunchKernelGGL(posconv_bwd_weight_kernel, dim3(16 * 64), dim3(256), 0, s, c->pos_w, c->pos_wb);
    transpose(c->proj_w, 512, c->proj_wT, 768, 768, 512);
    for (int i = 1; i < 7; ++i) {  // conv feature extractor: kernel layout [co][tap * 512 + ci] and the dX GEMMs' copies
        const long long n = 512LL * 512 * kConvK[i];
        hipLaunchKernelGGL(conv_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->theta + po.conv_w[i], c->conv_w[i],
                           kConvK[i]);
    }
    rebuild_conv_bwd_weights(c, s);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        const LayerOffsets& lo = po.L[l];
        const long long w4 = (long long)2304 * 768 / 4, q4 = (long long)768 * 768 / 4;
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((w4 + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(c->theta + lo.qkv_w), reinterpret_cast<float4*>(d.qkv_w), w4, q4, 0.125f);
        hipLaunchKernelGGL(scale_rows_kernel, dim3(3), dim3(256), 0, s, reinterpret_cast<const float4*>(c->theta + lo.qkv_b),
                           reinterpret_cast<float4*>(d.qkv_b), 576LL, 192LL, 0.125f);
        transpose(d.qkv_w, 768, c->qkv_wT[l], 2304, 2304, 768);
        transpose(d.o_w, 768, c->o_wT[l], 768, 768, 768);
        transpose(d.fc1_w, 768, c->fc1_wT[l], 3072, 3072, 768);
        transpose(d.fc2_w, 3072, c->fc2_wT[l], 768, 768, 3072);
    }
    HIP_TRY(hipGetLastError());
    c->bf16_ready = false;  // the bf16 copies (if any) are stale now; nomad_enable_bf16 rebuilds them
    c->x3_ready = false;    // likewise the split copies (nomad_enable_bf16x3)
    return 0;
}

}  // namespace

extern "C" {

int nomad_set_gemm_precision(nomad_ctx* c, int mode) {
    if (!c || (mode != 0 && mode != 1)) return fail(NOMAD_ERR_INVALID, "nomad_set_gemm_precision: mode must be 0 (fp32 MFMA) or 1 (bf16x3 products)");
    c->gemm_x3 = mode;
    return 0;
}

int nomad_get_gemm_precision(const nomad_ctx* c, int* mode) {
    if (!c || !mode) return fail(NOMAD_ERR_INVALID, "nomad_get_gemm_precision: null argument");
    *mode = c->gemm_x3;
    return 0;
}

int nomad_set_feature_grad_mult(nomad_ctx* c, float mult) {
    if (!c || !(mult >= 0.f)) return fail(NOMAD_ERR_INVALID, "nomad_set_feature_grad_mult: bad argument");
    c->feature_grad_mult = mult;
    return 0;
}

int nomad_get_feature_grad_mult(const nomad_ctx* c, float* mult) {
    if (!c || !mult) return fail(NOMAD_ERR_INVALID, "nomad_get_feature_grad_mult: bad argument");
    *mult = c->feature_grad_mult;
    return 0;
}

int nomad_embed_backward(nomad_ctx* c, const float* wav, int B, int n_samples, const float* head_w, const float* head_b,
                         const float* layers_out, const void* saved, size_t saved_bytes, const float* dlayers,
                         const float* demb, float* dwav, void* workspace, size_t workspace_bytes,
                         nomad_stream_t stream) {
    if (!dwav) return fail(NOMAD_ERR_INVALID, "nomad_embed_backward: bad argument");
    return backward_impl(c, wav, B, n_samples, head_w, head_b, layers_out, saved, saved_bytes, dlayers, demb, dwav,
                         workspace, workspace_bytes, stream, false);
}

int nomad_train_num_segments(void) { return (int)segments().size(); }

int nomad_train_segment(int i, char* name, size_t name_cap, size_t* offset, size_t* count) {
    const auto& v = segments();
    if (i < 0 || i >= (int)v.size() || !name || !offset || !count || name_cap <= v[i].name.size())
        return fail(NOMAD_ERR_INVALID, "nomad_train_segment: bad argument");
    memcpy(name, v[i].name.c_str(), v[i].name.size() + 1);
    *offset = v[i].offset;
    *count = v[i].count;
    return 0;
}

int nomad_train_param_count(size_t* total, size_t* head_begin) {
    if (!total) return fail(NOMAD_ERR_INVALID, "nomad_train_param_count: null argument");
    const ParamOffsets po = make_param_offsets();
    *total = po.total;
    if (head_begin) *head_begin = po.emb_w;
    return 0;
}

int nomad_train_enable(nomad_ctx* c, const nomad_weights* w) {
    if (!c || !w) return fail(NOMAD_ERR_INVALID, "nomad_train_enable: null argument");
    if (c->train_ready) return 0;
    int rc;
    if ((rc = nomad_enable_backward(c))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const ParamOffsets po = make_param_offsets();
    auto alloc = [&](size_t bytes, void** out) -> int {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, bytes));
        c->allocs.push_back(d);
        HIP_TRY(hipMemset(d, 0, bytes));
        *out = d;
        return 0;
    };
    if ((rc = alloc(po.total * sizeof(float), (void**)&c->theta))) return rc;
    if ((rc = alloc(po.total * sizeof(float), (void**)&c->grad))) return rc;
    if ((rc = alloc(po.total * sizeof(float), (void**)&c->adam_m))) return rc;
    if ((rc = alloc(po.total * sizeof(float), (void**)&c->adam_v))) return rc;
    if ((rc = alloc(128 * sizeof(double), (void**)&c->pos_nrm2))) return rc;
    if ((rc = alloc(128 * sizeof(double), (void**)&c->tap_dot))) return rc;
    if ((rc = alloc((size_t)576 * 128 * sizeof(double), (void**)&c->tap_partial))) return rc;
    auto put = [&](size_t off, const float* host, size_t n) {
        if (rc == 0 && hipMemcpy(c->theta + off, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(NOMAD_ERR_HIP, "nomad_train_enable: upload failed");
    };
    put(po.fln_w, w->feat_ln_w, 512); put(po.fln_b, w->feat_ln_b, 512);
    put(po.proj_w, w->proj_w, 768 * 512); put(po.proj_b, w->proj_b, 768);
    put(po.pos_g, w->pos_g, 128); put(po.pos_v, w->pos_v, (size_t)768 * 48 * 128); put(po.pos_b, w->pos_b, 768);
    put(po.eln_w, w->enc_ln_w, 768); put(po.eln_b, w->enc_ln_b, 768);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const nomad_layer_weights& lw = w->layers[l];
        const LayerOffsets& lo = po.L[l];
        const size_t ww = 768 * 768;
        put(lo.qkv_w, lw.q_w, ww); put(lo.qkv_w + ww, lw.k_w, ww); put(lo.qkv_w + 2 * ww, lw.v_w, ww);
        put(lo.qkv_b, lw.q_b, 768); put(lo.qkv_b + 768, lw.k_b, 768); put(lo.qkv_b + 1536, lw.v_b, 768);
        put(lo.o_w, lw.o_w, ww); put(lo.o_b, lw.o_b, 768);
        put(lo.ln1_w, lw.ln1_w, 768); put(lo.ln1_b, lw.ln1_b, 768);
        put(lo.fc1_w, lw.fc1_w, (size_t)3072 * 768); put(lo.fc1_b, lw.fc1_b, 3072);
        put(lo.fc2_w, lw.fc2_w, (size_t)768 * 3072); put(lo.fc2_b, lw.fc2_b, 768);
        put(lo.ln2_w, lw.ln2_w, 768); put(lo.ln2_b, lw.ln2_b, 768);
    }
    put(po.conv0_w, w->conv_w[0], 512 * 10); put(po.gn_w, w->gn_w, 512); put(po.gn_b, w->gn_b, 512);
    for (int i = 1; i < 7; ++i) put(po.conv_w[i], w->conv_w[i], (size_t)512 * 512 * kConvK[i]);
    put(po.emb_w, w->emb_w, 256 * 768); put(po.emb_b, w->emb_b, 256);
    if (rc) return rc;
    // From here on the engine reads these parameters straight from the master vector (same layout as the
    // checkpoint); only q/k/v (fused, q scaled), the pos-conv kernel and the transposes are derived copies.
    float* th = c->theta;
    c->fln_w = th + po.fln_w; c->fln_b = th + po.fln_b;
    c->proj_w = th + po.proj_w; c->proj_b = th + po.proj_b;
    c->pos_b = th + po.pos_b;
    c->eln_w = th + po.eln_w; c->eln_b = th + po.eln_b;
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        LayerDev& d = c->layers[l];
        const LayerOffsets& lo = po.L[l];
        d.o_w = th + lo.o_w; d.o_b = th + lo.o_b;
        d.ln1_w = th + lo.ln1_w; d.ln1_b = th + lo.ln1_b;
        d.fc1_w = th + lo.fc1_w; d.fc1_b = th + lo.fc1_b;
        d.fc2_w = th + lo.fc2_w; d.fc2_b = th + lo.fc2_b;
        d.ln2_w = th + lo.ln2_w; d.ln2_b = th + lo.ln2_b;
    }
    c->emb_w = th + po.emb_w; c->emb_b = th + po.emb_b;
    c->conv0_w = th + po.conv0_w; c->gn_w = th + po.gn_w; c->gn_b = th + po.gn_b;  // conv1..6: derived copies (refresh_weights)
    if ((rc = refresh_weights(c, 0))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    c->adam_t = 0;
    c->train_ready = true;
    return 0;
}

int nomad_train_workspace_bytes(const nomad_ctx* c, int B, int n_samples, size_t* bytes) {
    Shapes sh;
    if (!c || !bytes || B <= 0 || !make_shapes(B, n_samples, &sh))
        return fail(NOMAD_ERR_INVALID, "nomad_train_workspace_bytes: bad shape B=%d N=%d", B, n_samples);
    *bytes = make_bwd_layout(sh, true, c->train_convnet).total;
    return 0;
}

int nomad_train_zero_grad(nomad_ctx* c, nomad_stream_t stream) {
    if (!c || !c->train_ready) return fail(NOMAD_ERR_INVALID, "nomad_train_zero_grad: call nomad_train_enable first");
    HIP_TRY(hipMemsetAsync(c->grad, 0, make_param_offsets().total * sizeof(float), static_cast<hipStream_t>(stream)));
    return 0;
}

int nomad_train_backward(nomad_ctx* c, const float* wav, int B, int n_samples, const float* layers_out, const void* saved,
                         size_t saved_bytes, const float* demb, void* workspace, size_t workspace_bytes,
                         nomad_stream_t stream) {
    return backward_impl(c, wav, B, n_samples, nullptr, nullptr, layers_out, saved, saved_bytes, nullptr, demb, nullptr,
                         workspace, workspace_bytes, stream, true);
}

int nomad_triplet_loss(nomad_ctx* c, const float* a, const float* p, const float* n, int B, float margin, float* loss,
                       float* da, float* dp, float* dn, nomad_stream_t stream) {
    if (!c || !a || !p || !n || !loss || B <= 0 || (da && (!dp || !dn)))
        return fail(NOMAD_ERR_INVALID, "nomad_triplet_loss: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    hipLaunchKernelGGL(triplet_loss_kernel, dim3(1), dim3(256), 0, s, a, p, n, B, margin, loss, da, dp, dn);
    HIP_TRY(hipGetLastError());
    return 0;
}

int nomad_train_adam_step(nomad_ctx* c, float lr_body, float lr_head, float beta1, float beta2, float eps,
                          nomad_stream_t stream) {
    if (!c || !c->train_ready) return fail(NOMAD_ERR_INVALID, "nomad_train_adam_step: call nomad_train_enable first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const ParamOffsets po = make_param_offsets();
    c->adam_t += 1;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)c->adam_t);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)c->adam_t);
    const long long n4 = (long long)po.total / 4;
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<float4*>(c->theta),
                           reinterpret_cast<const float4*>(c->grad), reinterpret_cast<float4*>(c->adam_m),
                           reinterpret_cast<float4*>(c->adam_v), n4, (long long)po.emb_w / 4, lr_body, lr_head, beta1, beta2,
                           eps, (float)bc1, (float)std::sqrt(bc2));
    }
    HIP_TRY(hipGetLastError());
    return refresh_weights(c, s);
}

// what: 0 parameters, 1 gradients, 2 Adam exp_avg, 3 Adam exp_avg_sq.  Device-to-device on the stream.
static float* train_buffer(nomad_ctx* c, int what) {
    switch (what) {
        case 0: return c->theta;
        case 1: return c->grad;
        case 2: return c->adam_m;
        case 3: return c->adam_v;
        default: return nullptr;
    }
}

int nomad_train_read(nomad_ctx* c, int what, float* dst_dev, nomad_stream_t stream) {
    if (!c || !c->train_ready || !dst_dev || !train_buffer(c, what))
        return fail(NOMAD_ERR_INVALID, "nomad_train_read: bad argument");
    HIP_TRY(hipMemcpyAsync(dst_dev, train_buffer(c, what), make_param_offsets().total * sizeof(float),
                           hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return 0;
}

int nomad_train_write(nomad_ctx* c, int what, const float* src_dev, nomad_stream_t stream) {
    if (!c || !c->train_ready || !src_dev || !train_buffer(c, what))
        return fail(NOMAD_ERR_INVALID, "nomad_train_write: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(train_buffer(c, what), src_dev, make_param_offsets().total * sizeof(float),
                           hipMemcpyDeviceToDevice, s));
    return what == 0 ? refresh_weights(c, s) : 0;
}

int nomad_train_set_stochastic(nomad_ctx* c, float dropout, float attention_dropout, float dropout_input,
                               unsigned long long seed, unsigned layer_mask) {
    auto bad = [](float p) { return !(p >= 0.f && p < 1.f); };
    if (!c || bad(dropout) || bad(attention_dropout) || bad(dropout_input))
        return fail(NOMAD_ERR_INVALID, "nomad_train_set_stochastic: probabilities must be in [0, 1)");
    c->p_drop = dropout;
    c->p_attn = attention_dropout;
    c->p_input = dropout_input;
    c->drop_seed = seed;
    c->layer_mask = layer_mask & 0xFFFu;
    return 0;
}

int nomad_train_set_branches(nomad_ctx* c, int branches, const unsigned* layer_masks) {
    if (!c || branches < 1 || branches > 4 || (branches > 1 && !layer_masks))
        return fail(NOMAD_ERR_INVALID, "nomad_train_set_branches: 1..4 branches, one mask each");
    c->branches = branches;
    for (int i = 0; i < branches && layer_masks; ++i) c->branch_mask[i] = layer_masks[i] & 0xFFFu;
    return 0;
}

int nomad_train_set_frozen(nomad_ctx* c, int freeze_encoder) {
    if (!c) return fail(NOMAD_ERR_INVALID, "nomad_train_set_frozen: null ctx");
    c->freeze_encoder = freeze_encoder != 0;
    return 0;
}

int nomad_train_set_convnet(nomad_ctx* c, int trainable) {
    if (!c) return fail(NOMAD_ERR_INVALID, "nomad_train_set_convnet: null ctx");
    c->train_convnet = trainable != 0;
    return 0;
}

int nomad_train_set_step(nomad_ctx* c, long long step) {
    if (!c || !c->train_ready || step < 0) return fail(NOMAD_ERR_INVALID, "nomad_train_set_step: bad argument");
    c->adam_t = step;
    return 0;
}

int nomad_set_concurrent_parts(nomad_ctx* c, int parts) {
    if (!c || parts < 1) return fail(NOMAD_ERR_INVALID, "nomad_set_concurrent_parts: ctx %p, parts = %d", (void*)c, parts);
    c->tune.concurrent_parts = parts;
    return 0;
}

int nomad_build_flags(void) {
    int f = 0;
#ifdef NOMAD_PACKED_FP32_BUILD
    f |= NOMAD_BUILD_PACKED_FP32;
#endif
#ifdef NOMAD_DIAG
    f |= NOMAD_BUILD_DIAG;
#endif
    return f;
}

int nomad_pairwise(nomad_ctx* c, const float* deg, int Nd, const float* ref, int Nr, double* dist, double* mean,
                   nomad_stream_t stream) {
    if (!c || !deg || !ref || !mean || Nd <= 0 || Nr <= 0)
        return fail(NOMAD_ERR_INVALID, "nomad_pairwise: bad argument (Nd=%d, Nr=%d)", Nd, Nr);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // per-(ref tile, deg row) partial sums live in a context-owned scratch, one block per launch stream (calls on the same
    // stream are ordered, calls on different streams must not share it): the block of nomad_create goes to the first
    // stream that calls, any further stream allocates its own on its first call
    double* scratch = nullptr;
    std::unique_lock<std::mutex> pair_lock(c->pair_mu);   // look-up, binding and allocation of a stream's block: one thread at a time
    for (const auto& e : c->pair_scratch)
        if (e.first == s) {
            scratch = e.second;
            break;
        }
    if (!scratch) {
        // a block whose stream slot was released by an earlier recycle (first == kNoStream) is taken before anything new is
        // allocated; at most 64 blocks (1 GB) ever exist, whatever the number of stream handles a long-lived process goes through
        for (auto& e : c->pair_scratch)
            if (e.first == kNoStream) {
                e.first = s;
                scratch = e.second;
                break;
            }
    }
    if (!scratch) {
        if (c->pair_scratch.size() >= 64) {  // every block is bound: drain the device once and release all bindings but this one
            HIP_TRY(hipDeviceSynchronize());
            for (auto& e : c->pair_scratch) e.first = kNoStream;
            c->pair_scratch[0].first = s;
            scratch = c->pair_scratch[0].second;
        } else {
            void* d = nullptr;
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipMalloc(&d, sizeof(double) * kPairScratchDoubles));
            c->allocs.push_back(d);
            c->pair_scratch.emplace_back(s, static_cast<double*>(d));
            scratch = static_cast<double*>(d);
        }
    }
    pair_lock.unlock();
    Scope sc(c, s, NOMAD_K_PAIR, 3.0 * 256 * (double)Nd * Nr);
    // deg rows are processed in slabs that fit the scratch
    const int ntiles = (Nr + kPairTile - 1) / kPairTile;
    const long long slab_max = (kPairScratchDoubles / ntiles) / kPairTile * kPairTile;
    if (slab_max < kPairTile) return fail(NOMAD_ERR_INVALID, "nomad_pairwise: Nr=%d is too large for the scratch", Nr);
    for (long long d0 = 0; d0 < Nd; d0 += slab_max) {
        const int nd = (int)std::min<long long>(slab_max, Nd - d0);
        hipLaunchKernelGGL(pairwise_tile_kernel, dim3(ntiles, (nd + kPairTile - 1) / kPairTile), dim3(256), 0, s, deg + d0 * 256, nd,
                           ref, Nr, dist ? dist + d0 * Nr : nullptr, scratch);
        hipLaunchKernelGGL(pairwise_mean_kernel, dim3((nd + 255) / 256), dim3(256), 0, s, scratch, ntiles, nd, Nr, mean + d0);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t nomad_l1_scratch_bytes(void) { return sizeof(double) * (kL1Blocks + 8); }

int nomad_l1_loss(nomad_ctx* c, const float* a_layers, const float* b_layers, const float* a_emb, const float* b_emb,
                  int B, int T, float* loss, void* scratch, nomad_stream_t stream) {
    if (!c || !a_layers || !b_layers || !a_emb || !b_emb || !loss || !scratch || B <= 0 || T <= 0)
        return fail(NOMAD_ERR_INVALID, "nomad_l1_loss: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long per_layer = (long long)B * T * 768;
    const long long n4 = per_layer * 12 / 4;
    double* partial = static_cast<double*>(scratch);
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(kL1Blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(a_layers),
                       reinterpret_cast<const float4*>(b_layers), n4, a_emb, b_emb, B * 256, partial);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, s, partial, (double)per_layer, (double)B * 256, loss);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- measurement -----------------------------------------------------------------------------
int nomad_profile_enable(nomad_ctx* c, int on) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    if (on && !c->ev_ready) {
        HIP_TRY(hipSetDevice(c->device));
        c->ev.resize(kEventChunk);
        c->ev_class.resize(kEventChunk / 2);
        for (int i = 0; i < kEventChunk; ++i) HIP_TRY(hipEventCreate(&c->ev[i]));
        c->ev_ready = true;
    }
    c->prof = on != 0;
    return 0;
}

static int profile_drain(nomad_ctx* c) {
    for (int i = 0; i + 1 < c->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(c->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        const int code = c->ev_class[i / 2];
        c->p_ms[code & 0xff] += ms;
        if (code >> 8) c->p_ms[(code >> 8) - 1] += ms;
    }
    c->ev_used = 0;
    return 0;
}

int nomad_profile_reset(nomad_ctx* c) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    int rc = profile_drain(c);
    c->prof_overflow = false;
    for (int i = 0; i < NOMAD_K_COUNT; ++i) {
        c->p_ms[i] = 0;
        c->p_n[i] = 0;
        c->p_fl[i] = 0;
    }
    return rc;
}

int nomad_profile_read(nomad_ctx* c, double ms[NOMAD_K_COUNT], long long launches[NOMAD_K_COUNT],
                       double flops[NOMAD_K_COUNT]) {
    if (!c || !ms || !launches || !flops) return fail(NOMAD_ERR_INVALID, "null argument");
    int rc = profile_drain(c);
    if (rc) return rc;
    if (c->prof_overflow) return fail(NOMAD_ERR_HIP, "nomad_profile_read: the event pool could not grow; counters are incomplete");
    for (int i = 0; i < NOMAD_K_COUNT; ++i) {
        ms[i] = c->p_ms[i];
        launches[i] = c->p_n[i];
        flops[i] = c->p_fl[i];
    }
    return 0;
}


// One wave spins for `spin_ticks` ticks of the 100 MHz wall counter and reports the shader-clock cycles that passed:
// launched on a second stream while a kernel under test runs, it reads the clock that kernel actually gets.
__global__ void clock_probe_kernel(unsigned long long spin_ticks, unsigned long long* out) {
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    unsigned long long w1 = w0;
    while (w1 - w0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(32);
        w1 = wall_clock64();
    }
    out[0] = __builtin_readcyclecounter() - c0;
    out[1] = w1 - w0;
}

int nomad_diag_clock_probe(nomad_ctx* c, unsigned long long spin_ticks, unsigned long long* out_dev, nomad_stream_t stream) {
    if (!c || !out_dev) return fail(NOMAD_ERR_INVALID, "nomad_diag_clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), spin_ticks, out_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int nomad_diag_layernorm(nomad_ctx* c, const float* in, const float* g, const float* b, float* out, int M, int N,
                         nomad_stream_t stream) {
    if (!c || !in || !g || !b || !out || M <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_layernorm: bad argument");
    return run_layernorm(c, in, g, b, out, nullptr, M, N, static_cast<hipStream_t>(stream));
}

int nomad_diag_layernorm_bwd(nomad_ctx* c, const float* x, const float* g, const float* gamma, float* dx, int M, int N,
                             nomad_stream_t stream) {
    if (!c || !x || !g || !gamma || !dx || M <= 0 || (N != 512 && N != 768))
        return fail(NOMAD_ERR_INVALID, "nomad_diag_layernorm_bwd: bad argument");
    return run_ln_bwd(c, x, g, nullptr, gamma, dx, M, N, static_cast<hipStream_t>(stream));
}

int nomad_diag_attention_bwd(nomad_ctx* c, const float* qkv, const float* dctx, float* ctx_out, float* lse, float* dqkv,
                             int B, int T, nomad_stream_t stream) {
    if (!c || !qkv || !dctx || !ctx_out || !lse || !dqkv || B <= 0 || T <= 0)
        return fail(NOMAD_ERR_INVALID, "nomad_diag_attention_bwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = run_attention(c, qkv, ctx_out, lse, B, T, s);
    if (rc) return rc;
    float* D = nullptr;  // diagnostics only: the product path carves this from the caller's workspace
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&D), sizeof(float) * 12 * (size_t)B * T));
    const hipError_t e = launch_attention_bwd(qkv, ctx_out, dctx, lse, D, dqkv, B, T, DropCfg{}, 0, s, 0, !c->tune.attn_bwd_small);
    (void)hipStreamSynchronize(s);
    (void)hipFree(D);
    if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "attention backward: %s", hipGetErrorString(e));
    return 0;
}

int nomad_diag_attention_bf16(nomad_ctx* c, const void* qkv, void* out, int B, int T, int q_has_log2e, nomad_stream_t stream) {
    if (!c || !qkv || !out || B <= 0 || T <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_attention_bf16: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(c, s, NOMAD_K_ATTN, 4.0 * B * 12.0 * (double)T * T * 64);
    HIP_TRY(run_attention_bf16(c, static_cast<const bf16_t*>(qkv), static_cast<bf16_t*>(out), B, T, nullptr, q_has_log2e != 0, s));
    return 0;
}

int nomad_diag_attention(nomad_ctx* c, const float* qkv, float* out, int B, int T, nomad_stream_t stream) {
    if (!c || !qkv || !out || B <= 0 || T <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_attention: bad argument");
    return run_attention(c, qkv, out, nullptr, B, T, static_cast<hipStream_t>(stream));
}

}  // extern "C"
