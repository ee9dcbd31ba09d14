/* Plain-C client of the C ABI (include/nomad_hip.h): no Python, no torch, no C++.
 * Builds a random parameter set, embeds 3 clips twice (batch of 3, then one by one) and scores them.
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/abi_smoke.c \
 *       -Lnomad_amd -lnomad_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/nomad_amd -o abi_smoke */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nomad_hip.h"

static unsigned long long rng = 88172645463325252ULL;
static float frand(void) { /* xorshift, uniform in [-1, 1) */
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (float)((rng >> 40) / 8388608.0 - 1.0);
}
static float* tensor(size_t n, float scale, float offset) {
    float* p = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) p[i] = offset + scale * frand();
    return p;
}
#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, nomad_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(void) {
    nomad_weights w;
    memset(&w, 0, sizeof(w));
    const int ck[7] = {10, 3, 3, 3, 3, 2, 2};
    for (int i = 0; i < 7; ++i) {
        const size_t cin = i == 0 ? 1 : 512;
        w.conv_w[i] = tensor(512 * cin * ck[i], sqrtf(6.0f / (cin * ck[i])), 0.f);
    }
    w.gn_w = tensor(512, 0.1f, 1.f); w.gn_b = tensor(512, 0.1f, 0.f);
    w.feat_ln_w = tensor(512, 0.1f, 1.f); w.feat_ln_b = tensor(512, 0.1f, 0.f);
    w.proj_w = tensor(768 * 512, 0.03f, 0.f); w.proj_b = tensor(768, 0.02f, 0.f);
    w.pos_v = tensor((size_t)768 * 48 * 128, 0.01f, 0.f); w.pos_g = tensor(128, 0.1f, 1.f); w.pos_b = tensor(768, 0.02f, 0.f);
    w.enc_ln_w = tensor(768, 0.1f, 1.f); w.enc_ln_b = tensor(768, 0.1f, 0.f);
    for (int l = 0; l < NOMAD_NUM_LAYERS; ++l) {
        nomad_layer_weights* lw = &w.layers[l];
        lw->q_w = tensor(768 * 768, 0.03f, 0.f); lw->q_b = tensor(768, 0.02f, 0.f);
        lw->k_w = tensor(768 * 768, 0.03f, 0.f); lw->k_b = tensor(768, 0.02f, 0.f);
        lw->v_w = tensor(768 * 768, 0.03f, 0.f); lw->v_b = tensor(768, 0.02f, 0.f);
        lw->o_w = tensor(768 * 768, 0.03f, 0.f); lw->o_b = tensor(768, 0.02f, 0.f);
        lw->ln1_w = tensor(768, 0.1f, 1.f); lw->ln1_b = tensor(768, 0.1f, 0.f);
        lw->fc1_w = tensor((size_t)3072 * 768, 0.03f, 0.f); lw->fc1_b = tensor(3072, 0.02f, 0.f);
        lw->fc2_w = tensor((size_t)768 * 3072, 0.03f, 0.f); lw->fc2_b = tensor(768, 0.02f, 0.f);
        lw->ln2_w = tensor(768, 0.1f, 1.f); lw->ln2_b = tensor(768, 0.1f, 0.f);
    }
    w.emb_w = tensor(256 * 768, 0.036f, 0.f); w.emb_b = tensor(256, 0.036f, 0.f);

    nomad_ctx* ctx = NULL;
    CHECK(nomad_create(&ctx, 0, &w));
    printf("%s\n", nomad_version());

    const int B = 3, N = 16384;
    float* wav_h = tensor((size_t)B * N, 0.1f, 0.f);
    float *wav_d, *emb_d, *emb1_d;
    double *dist_d, *mean_d;
    void* ws;
    size_t ws_bytes = 0;
    CHECK(nomad_workspace_bytes(ctx, B, N, &ws_bytes));
    HIP(hipMalloc((void**)&wav_d, sizeof(float) * B * N));
    HIP(hipMalloc((void**)&emb_d, sizeof(float) * B * 256));
    HIP(hipMalloc((void**)&emb1_d, sizeof(float) * B * 256));
    HIP(hipMalloc((void**)&dist_d, sizeof(double) * 2));
    HIP(hipMalloc((void**)&mean_d, sizeof(double) * 2));
    HIP(hipMalloc(&ws, ws_bytes));
    HIP(hipMemcpy(wav_d, wav_h, sizeof(float) * B * N, hipMemcpyHostToDevice));
    CHECK(nomad_embed(ctx, wav_d, B, N, NULL, NULL, emb_d, NULL, ws, ws_bytes, NULL));
    for (int b = 0; b < B; ++b)
        CHECK(nomad_embed(ctx, wav_d + (size_t)b * N, 1, N, NULL, NULL, emb1_d + b * 256, NULL, ws, ws_bytes, NULL));
    CHECK(nomad_pairwise(ctx, emb_d, 2, emb_d + 2 * 256, 1, dist_d, mean_d, NULL));
    HIP(hipDeviceSynchronize());

    float emb[3 * 256], emb1[3 * 256];
    double dist[2], mean[2];
    HIP(hipMemcpy(emb, emb_d, sizeof(emb), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(emb1, emb1_d, sizeof(emb1), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(dist, dist_d, sizeof(dist), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(mean, mean_d, sizeof(mean), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        double nrm = 0;
        for (int i = 0; i < 256; ++i) nrm += (double)emb[b * 256 + i] * emb[b * 256 + i];
        if (fabs(sqrt(nrm) - 1.0) > 1e-5) bad |= 1;                      /* unit-norm embeddings */
    }
    if (memcmp(emb, emb1, sizeof(emb)) != 0) bad |= 2;                   /* batch invariance, bit exact */
    for (int d = 0; d < 2; ++d) {
        double s = 0;
        for (int i = 0; i < 256; ++i) { const double e = (double)emb[d * 256 + i] - (double)emb[2 * 256 + i]; s += e * e; }
        if (fabs(sqrt(s) - dist[d]) > 1e-12 || fabs(dist[d] - mean[d]) > 1e-15) bad |= 4;   /* float64 distances */
    }
    printf("emb[0][0..3] = %.6f %.6f %.6f %.6f  dist = %.9f %.9f  status = %d\n", emb[0], emb[1], emb[2], emb[3], dist[0], dist[1], bad);
    nomad_destroy(ctx);
    return bad;
}
