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
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/nomad_hip.h"
#include "attention.hip.h"
#include "frontend.hip.h"
#include "gemm_f32.hip.h"
#include "pairwise.hip.h"
#include "rowops.hip.h"

using namespace nomad;

namespace {

thread_local char g_err[512] = "";

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
constexpr int kMaxEvents = 8192;

struct LayerDev {
    float *qkv_w, *qkv_b, *o_w, *o_b, *ln1_w, *ln1_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ln2_w, *ln2_b;
};

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
struct Layout {
    size_t stats, scale, shift, conv[7], featln, xpad, x, x2, y, qkv, ctxb, h, total;
};

Layout make_layout(const Shapes& s, bool keep) {
    Layout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    l.stats = take(sizeof(double) * kStatsPerClip * s.B);
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
    l.total = off;
    return l;
}

}  // namespace

struct nomad_ctx {
    int device = 0;
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
    std::vector<void*> allocs;
    // profiling
    bool prof = false;
    hipEvent_t ev[kMaxEvents];
    int ev_class[kMaxEvents / 2];
    int ev_used = 0;
    double p_ms[NOMAD_K_COUNT] = {};
    long long p_n[NOMAD_K_COUNT] = {};
    double p_fl[NOMAD_K_COUNT] = {};
    bool ev_ready = false;
};

namespace {

int upload(nomad_ctx* c, const float* host, size_t n, float** out) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, n * sizeof(float)));
    c->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    *out = static_cast<float*>(d);
    return 0;
}

// Brackets one launch with events when profiling is on.
struct Scope {
    nomad_ctx* c;
    hipStream_t s;
    int slot = -1;
    Scope(nomad_ctx* c_, hipStream_t s_, int cls, double flops) : c(c_), s(s_) {
        if (!c->prof) return;
        if (c->ev_used + 2 > kMaxEvents) return;  // pool exhausted: neither counted nor timed
        c->p_fl[cls] += flops;
        c->p_n[cls] += 1;
        slot = c->ev_used;
        c->ev_class[slot / 2] = cls;
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

int run_gemm(nomad_ctx* c, GemmParams p, int groups, int tile, hipStream_t s, int occ = 0) {
    if (tile == 29 && occ == 0) occ = 4;  // measured: 4 workgroups/CU is the best residency for the 128x64x32 kernel
    const double flops = 2.0 * p.M * (double)p.n_valid * p.K * groups;
    Scope sc(c, s, NOMAD_K_GEMM, flops);
    hipError_t e;
    switch (tile) {
        case 0: e = launch_gemm<128, 128, 32, 2, 2>(p, groups, s); break;
        case 1: e = launch_gemm<128, 64, 16, 2, 2>(p, groups, s); break;
        case 2: e = launch_gemm<64, 64, 32, 2, 2>(p, groups, s); break;
        // experimental instantiations (tools/gemm_sweep.py)
        case 3: e = launch_gemm<128, 128, 16, 2, 2>(p, groups, s); break;
        case 4: e = launch_gemm<256, 128, 32, 4, 2>(p, groups, s); break;
        case 5: e = launch_gemm<256, 256, 32, 4, 2>(p, groups, s); break;
        case 6: e = launch_gemm<256, 128, 16, 4, 2>(p, groups, s); break;
        case 7: e = launch_gemm<128, 256, 32, 2, 2>(p, groups, s); break;
        case 8: e = launch_gemm<256, 256, 16, 4, 2>(p, groups, s); break;
        case 9: e = launch_gemm<128, 128, 16, 4, 2>(p, groups, s, 16 * 1024); break;   // 8 waves, forced 2 WG/CU
        case 10: e = launch_gemm<128, 128, 16, 2, 4>(p, groups, s, 16 * 1024); break;
        case 11: e = launch_gemm<128, 128, 32, 4, 2>(p, groups, s); break;
        case 12: e = launch_gemm<128, 128, 32, 2, 4>(p, groups, s); break;
        case 13: e = launch_gemm<128, 128, 16, 4, 2>(p, groups, s); break;               // 3 WG/CU if registers allow
        case 20: e = launch_gemm_glds<128, 128, 32, 2, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES)); break;
        case 21: e = launch_gemm_glds<256, 128, 16, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 128, 16, 4, 2>::LDS_BYTES)); break;
        case 22: e = launch_gemm_glds<128, 128, 16, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 16, 4, 2>::LDS_BYTES)); break;
        case 23: e = launch_gemm_glds<256, 128, 32, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 128, 32, 4, 2>::LDS_BYTES)); break;
        case 24: e = launch_gemm_glds<256, 256, 16, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 256, 16, 4, 2>::LDS_BYTES)); break;
        case 25: e = launch_gemm_glds<256, 256, 32, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 256, 32, 4, 2>::LDS_BYTES)); break;
        case 26: e = launch_gemm_glds<128, 128, 16, 2, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 16, 2, 2>::LDS_BYTES)); break;
        case 27: e = launch_gemm_glds<256, 256, 16, 4, 4>(p, groups, s, occ_pad(occ, GldsCfg<256, 256, 16, 4, 4>::LDS_BYTES)); break;
        case 28: e = launch_gemm_glds<128, 64, 16, 2, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 64, 16, 2, 2>::LDS_BYTES)); break;
        case 29: e = launch_gemm_glds<128, 64, 32, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 64, 32, 4, 2>::LDS_BYTES)); break;
        case 30: e = launch_gemm_glds<128, 64, 32, 2, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 64, 32, 2, 2>::LDS_BYTES)); break;
        case 31: e = launch_gemm_glds<128, 128, 32, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)); break;
        case 14: e = launch_gemm<128, 128, 32, 2, 2, 1>(p, groups, s); break;            // ablations of tile 0
        case 15: e = launch_gemm<128, 128, 32, 2, 2, 2>(p, groups, s); break;
        case 16: e = launch_gemm<128, 128, 32, 2, 2, 3>(p, groups, s); break;
        case 17: e = launch_gemm<256, 128, 16, 4, 2, 1>(p, groups, s); break;            // ablations of tile 6
        case 18: e = launch_gemm<256, 128, 16, 4, 2, 2>(p, groups, s); break;
        case 19: e = launch_gemm<256, 128, 16, 4, 2, 3>(p, groups, s); break;
        default: return fail(NOMAD_ERR_INVALID, "unknown gemm tile id %d", tile);
    }
    if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
    return 0;
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

// Kernel instantiation for a dense problem (measured on MI355X, tools/gemm_sweep.py, profiles/):
//   21 = LDS-DMA 256x128x16, 8 waves: best when the grid is many rounds deep (QKV, fc1, conv1-4)
//   29 = LDS-DMA 128x64x32, 8 waves: finer tiles for N = 768 / 512 problems where a 256x128 grid is only
//        2-3 rounds deep and the last partial round would idle a third of the CUs
//    2 = register-staged 64x64x32 for tiny batches
int pick_tile(int M, int N, int K) {
    if (M < 1024) return 2;
    const long long tiles256 = (long long)((M + 255) / 256) * (N / 128);
    if (N % 128 == 0 && (tiles256 >= 2048 || K >= 2048 || K <= 512)) return 21;
    return 29;
}

int run_layernorm(nomad_ctx* c, const float* in, const float* g, const float* b, float* out, float* out2, int M, int N,
                  hipStream_t s) {
    Scope sc(c, s, NOMAD_K_ROW, 0.0);
    const int blocks = (M + 3) / 4;
    if (N == 768) hipLaunchKernelGGL(layernorm_kernel<3>, dim3(blocks), dim3(256), 0, s, in, g, b, out, out2, M);
    else if (N == 512) hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(256), 0, s, in, g, b, out, out2, M);
    else return fail(NOMAD_ERR_INVALID, "layernorm: N=%d unsupported", N);
    HIP_TRY(hipGetLastError());
    return 0;
}

int run_attention(nomad_ctx* c, const float* qkv, float* out, int B, int T, hipStream_t s) {
    const double flops = 4.0 * B * 12.0 * (double)T * T * 64;
    Scope sc(c, s, NOMAD_K_ATTN, flops);
    hipLaunchKernelGGL(attention_f32_kernel, dim3((T + 63) / 64, B * 12), dim3(256), 0, s, qkv, out, T);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

const char* nomad_last_error(void) { return g_err; }
const char* nomad_version(void) { return "nomad_hip 0.1 (gfx950, fp32 MFMA)"; }

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
    if (c->ev_ready)
        for (int i = 0; i < kMaxEvents; ++i) (void)hipEventDestroy(c->ev[i]);
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

int nomad_embed(nomad_ctx* c, const float* wav, int B, int n_samples, const float* head_w, const float* head_b,
                float* emb, float* layers_out, void* workspace, size_t workspace_bytes, nomad_stream_t stream) {
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
    int rc;

    // ---- front end: conv0 + GroupNorm + GELU ------------------------------------------------
    double* stats = reinterpret_cast<double*>(ws + lay.stats);
    {
        Scope sc(c, s, NOMAD_K_FRONT, 0.0);
        hipLaunchKernelGGL(wav_stats_kernel, dim3(B), dim3(256), 0, s, wav, n_samples, sh.L[0], stats);
        hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(512), 0, s, stats, c->conv0_w, c->gn_w, c->gn_b, sh.L[0],
                           F(lay.scale), F(lay.shift));
    }
    {
        Scope sc(c, s, NOMAD_K_FRONT, 2.0 * B * (double)sh.L[0] * 512 * 10);
        hipLaunchKernelGGL(conv0_gn_gelu_kernel, dim3((sh.L[0] + kConv0Frames - 1) / kConv0Frames, B), dim3(256), 0, s,
                           wav, n_samples, sh.L[0], c->conv0_w, F(lay.scale), F(lay.shift), F(lay.conv[0]));
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
        p.C = F(lay.conv[i]);
        p.M = B * sh.L[i];
        p.N = 512;
        p.n_valid = 512;
        p.cmap = plain_map(p.M, 512);
        p.rmap = p.cmap;
        p.gelu = 1;
        if ((rc = run_gemm(c, p, 1, pick_tile(p.M, 512, p.K), s))) return rc;
    }

    // ---- LayerNorm(512) + post_extract_proj into the padded pos-conv buffer ------------------
    if ((rc = run_layernorm(c, F(lay.conv[6]), c->fln_w, c->fln_b, F(lay.featln), nullptr, M, 512, s))) return rc;
    // group-major pos-conv buffer xg[16][B][T+128][48]; x (post_extract_proj output) sits at frames 64..64+T
    float* xpad = F(lay.xpad);
    const long long grp_stride = (long long)B * (T + 128) * 48;
    const RowMap pad_map{64LL * 48, (long long)(T + 128) * 48, T, 48};
    {
        Scope sc(c, s, NOMAD_K_ROW, 0.0);
        hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(16 * B), dim3(256), 0, s, xpad, T);
    }
    {
        GemmParams p = dense(F(lay.featln), 512, c->proj_w, c->proj_b, nullptr, xpad, M, 768, 512, 0);
        p.cmap = pad_map;
        p.c_colblk = 48;
        p.c_colblk_stride = grp_stride;
        if ((rc = run_gemm(c, p, 1, pick_tile(M, 768, 512), s))) return rc;
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
        p.C = F(lay.y);
        p.cmap = plain_map(M, 768);
        p.c_goff = 48;
        p.R = xpad;
        p.rmap = pad_map;
        p.r_goff = grp_stride;
        p.M = M;
        p.N = 64;
        p.n_valid = 48;
        p.gelu = 1;
        if ((rc = run_gemm(c, p, 16, M >= 1024 ? 29 : 2, s))) return rc;
    }
    float* x = F(lay.x);
    float* x2 = F(lay.x2);
    float* y = F(lay.y);
    if ((rc = run_layernorm(c, y, c->eln_w, c->eln_b, x, nullptr, M, 768, s))) return rc;

    // ---- 12 post-LN transformer layers --------------------------------------------------------
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        const LayerDev& d = c->layers[l];
        if ((rc = run_gemm(c, dense(x, 768, d.qkv_w, d.qkv_b, nullptr, F(lay.qkv), M, 2304, 768, 0), 1,
                           pick_tile(M, 2304, 768), s)))
            return rc;
        if ((rc = run_attention(c, F(lay.qkv), F(lay.ctxb), B, T, s))) return rc;
        if ((rc = run_gemm(c, dense(F(lay.ctxb), 768, d.o_w, d.o_b, x, y, M, 768, 768, 0), 1, pick_tile(M, 768, 768), s)))
            return rc;
        if ((rc = run_layernorm(c, y, d.ln1_w, d.ln1_b, x2, nullptr, M, 768, s))) return rc;
        if ((rc = run_gemm(c, dense(x2, 768, d.fc1_w, d.fc1_b, nullptr, F(lay.h), M, 3072, 768, 1), 1,
                           pick_tile(M, 3072, 768), s)))
            return rc;
        if ((rc = run_gemm(c, dense(F(lay.h), 3072, d.fc2_w, d.fc2_b, x2, y, M, 768, 3072, 0), 1, pick_tile(M, 768, 3072), s)))
            return rc;
        float* lo = layers_out ? layers_out + (size_t)l * M * 768 : nullptr;
        if ((rc = run_layernorm(c, y, d.ln2_w, d.ln2_b, x, lo, M, 768, s))) return rc;
    }

    // ---- head -----------------------------------------------------------------------------------
    {
        Scope sc(c, s, NOMAD_K_ROW, 2.0 * B * 768 * 256);
        hipLaunchKernelGGL(head_kernel, dim3(B), dim3(256), 0, s, x, T, head_w ? head_w : c->emb_w,
                           head_b ? head_b : c->emb_b, emb);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int nomad_pairwise(nomad_ctx* c, const float* deg, int Nd, const float* ref, int Nr, double* dist, double* mean,
                   nomad_stream_t stream) {
    if (!c || !deg || !ref || !mean || Nd <= 0 || Nr <= 0)
        return fail(NOMAD_ERR_INVALID, "nomad_pairwise: bad argument (Nd=%d, Nr=%d)", Nd, Nr);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(c, s, NOMAD_K_PAIR, 3.0 * 256 * (double)Nd * Nr);
    hipLaunchKernelGGL(pairwise_f64_kernel, dim3((Nd + 31) / 32), dim3(256), 0, s, deg, Nd, ref, Nr, dist, mean);
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
        for (int i = 0; i < kMaxEvents; ++i) HIP_TRY(hipEventCreate(&c->ev[i]));
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
        c->p_ms[c->ev_class[i / 2]] += ms;
    }
    c->ev_used = 0;
    return 0;
}

int nomad_profile_reset(nomad_ctx* c) {
    if (!c) return fail(NOMAD_ERR_INVALID, "null ctx");
    int rc = profile_drain(c);
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
    for (int i = 0; i < NOMAD_K_COUNT; ++i) {
        ms[i] = c->p_ms[i];
        launches[i] = c->p_n[i];
        flops[i] = c->p_fl[i];
    }
    return 0;
}

// ---- diagnostics -------------------------------------------------------------------------------
int nomad_diag_gemm(nomad_ctx* c, const float* A, const float* W, const float* bias, const float* R, float* C, int M,
                    int N, int K, int gelu, int tile, nomad_stream_t stream) {
    if (!c || !A || !W || !C || M <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: bad argument");
    // diagnostics: tile id + 100 * group_m (grouped tile order) + 10000 * occ (workgroups per CU limit)
    const int occ = tile / 10000;
    const int group_m = (tile % 10000) / 100;
    tile %= 100;
    static const int kBN[] = {128, 64, 64, 128, 128, 256, 128, 256, 256, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
                              128, 128, 128, 128, 256, 256, 128, 256, 64, 64, 64, 128};
    static const int kBK[] = {32, 16, 32, 16, 32, 32, 16, 32, 16, 16, 16, 32, 32, 16, 32, 32, 32, 16, 16, 16,
                              32, 16, 16, 32, 16, 32, 16, 16, 16, 32, 32, 32};
    if (tile < 0 || tile > 31) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile id %d", tile);
    const int bn = kBN[tile], bk = kBK[tile];
    if (N % bn || K % bk) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: N %% %d or K %% %d != 0", bn, bk);
    GemmParams p = dense(A, K, W, bias, R, C, M, N, K, gelu);
    p.group_m = group_m;
    return run_gemm(c, p, 1, tile, static_cast<hipStream_t>(stream), occ);
}

int nomad_diag_layernorm(nomad_ctx* c, const float* in, const float* g, const float* b, float* out, int M, int N,
                         nomad_stream_t stream) {
    if (!c || !in || !g || !b || !out || M <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_layernorm: bad argument");
    return run_layernorm(c, in, g, b, out, nullptr, M, N, static_cast<hipStream_t>(stream));
}

int nomad_diag_attention(nomad_ctx* c, const float* qkv, float* out, int B, int T, nomad_stream_t stream) {
    if (!c || !qkv || !out || B <= 0 || T <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_attention: bad argument");
    return run_attention(c, qkv, out, B, T, static_cast<hipStream_t>(stream));
}

}  // extern "C"
