// Standalone A/B harness for the bf16 attention kernels (config C5 shape by default: 32 clips x 12 heads x T = 1499):
// checks both against a float64 reference on sampled query rows (incl. a spiked key that forces a late rescale) and
// times them in interleaved rounds in one process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/attn_bf16 tools/micro/attn_bf16.hip
//   tools/micro/attn_bf16 [B=32] [T=1499] [rounds=5]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../nomad_amd/csrc/attention_bf16_v2.hip.h"
#include "attention_bf16_r1.hip.h"

using namespace nomad;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 1499, rounds = argc > 3 ? atoi(argv[3]) : 5;
    const long long M = (long long)B * T;
    std::vector<uint16_t> qkv((size_t)M * 2304);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : qkv) v = f2bf(0.5f * nd(rng));
    // a spiked key late in clip 0 / head 3: its logit against every query row dwarfs the running maximum
    const int spike_key = T > 700 ? 700 : T - 1;
    for (int d = 0; d < 64; ++d) {
        qkv[(size_t)spike_key * 2304 + 768 + 3 * 64 + d] = f2bf(4.0f * ((d & 1) ? 1.f : -1.f));
        for (int t = 0; t < T; t += 7) qkv[(size_t)t * 2304 + 3 * 64 + d] = f2bf(0.6f * ((d & 1) ? 1.f : -1.f));
    }
    bf16_t *d_qkv, *d_o1, *d_o2, *d_o3;
    CK(hipMalloc(&d_qkv, qkv.size() * 2));
    CK(hipMalloc(&d_o1, (size_t)M * 768 * 2));
    CK(hipMalloc(&d_o2, (size_t)M * 768 * 2));
    CK(hipMalloc(&d_o3, (size_t)M * 768 * 2));
    CK(hipMemcpy(d_qkv, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(d_o1, 0xff, (size_t)M * 768 * 2));
    CK(hipMemset(d_o2, 0xff, (size_t)M * 768 * 2));
    CK(hipMemset(d_o3, 0xff, (size_t)M * 768 * 2));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto run_old = [&]() { hipLaunchKernelGGL(attention_bf16_kernel, dim3((T + 63) / 64, B * 12), dim3(256), 0, s, d_qkv, d_o1, T, (const int*)nullptr); };
    struct Variant { const char* name; hipError_t (*fn)(const bf16_t*, bf16_t*, int, int, const int*, hipStream_t); };
    const Variant variants[] = {
        {"NW8 KT64 occ4", launch_attention_bf16_v2<8, 64, 4, false>},
        {"NW4 KT64 occ4", launch_attention_bf16_v2<4, 64, 4, false>},
        {"NW8 KT64 occ4 LDS-DMA", launch_attention_bf16_v2<8, 64, 4, false, true>},
        {"NW8 KT128 occ4 LDS-DMA", launch_attention_bf16_v2<8, 128, 4, false, true>},
        {"NW4 KT64 occ4 LDS-DMA", launch_attention_bf16_v2<4, 64, 4, false, true>},
    };
    const int nvar = sizeof(variants) / sizeof(variants[0]);
    const int pick = getenv("VARIANT") ? atoi(getenv("VARIANT")) : 0;
    auto run_new = [&]() { CK(variants[pick].fn(d_qkv, d_o2, B, T, nullptr, s)); };
    run_old();
    run_new();
    CK(hipStreamSynchronize(s));
    std::vector<uint16_t> o1((size_t)M * 768), o2((size_t)M * 768);
    CK(hipMemcpy(o1.data(), d_o1, o1.size() * 2, hipMemcpyDeviceToHost));
    // float64 reference on sampled rows
    struct Ref { size_t idx; double o; };
    std::vector<Ref> refs;
    double worst1 = 0, refmax = 0;
    std::vector<double> sc(T);
    const int clips[2] = {0, B - 1};
    for (int ci = 0; ci < (B > 1 ? 2 : 1); ++ci) {
        const int b = clips[ci];
        for (int hd = 0; hd < 12; ++hd) {
            std::vector<int> rows = {0, 1, 31, 32, 63, 255, 256, 257, T / 2, T - 2, T - 1};
            for (int k = 0; k < 12; ++k) rows.push_back((int)(rng() % (unsigned)T));
            for (int q : rows) {
                if (q >= T || q < 0) continue;
                const uint16_t* Q = &qkv[((size_t)b * T + q) * 2304 + hd * 64];
                double mx = -1e300;
                for (int j = 0; j < T; ++j) {
                    const uint16_t* K = &qkv[((size_t)b * T + j) * 2304 + 768 + hd * 64];
                    double a = 0;
                    for (int d = 0; d < 64; ++d) a += (double)bf2f(Q[d]) * (double)bf2f(K[d]);
                    sc[j] = a;
                    mx = std::max(mx, a);
                }
                double l = 0;
                for (int j = 0; j < T; ++j) { sc[j] = std::exp(sc[j] - mx); l += sc[j]; }
                for (int d = 0; d < 64; ++d) {
                    double o = 0;
                    for (int j = 0; j < T; ++j) o += sc[j] * (double)bf2f(qkv[((size_t)b * T + j) * 2304 + 1536 + hd * 64 + d]);
                    o /= l;
                    const size_t idx = ((size_t)b * T + q) * 768 + hd * 64 + d;
                    refmax = std::max(refmax, std::fabs(o));
                    worst1 = std::max(worst1, std::fabs((double)bf2f(o1[idx]) - o));
                    refs.push_back({idx, o});
                }
            }
        }
    }
    printf("B=%d T=%d  ref max|o| %.3f   old kernel max|err| %.3e\n", B, T, refmax, worst1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flops = 4.0 * B * 12.0 * (double)T * T * 64;
    for (int v = 0; v < nvar; ++v) {
        CK(hipMemset(d_o2, 0xff, (size_t)M * 768 * 2));
        CK(variants[v].fn(d_qkv, d_o2, B, T, nullptr, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(o2.data(), d_o2, o2.size() * 2, hipMemcpyDeviceToHost));
        double worst2 = 0;
        int nan2 = 0;
        for (const Ref& rf : refs) {
            const float v2 = bf2f(o2[rf.idx]);
            if (!(v2 == v2)) ++nan2;
            else worst2 = std::max(worst2, std::fabs((double)v2 - rf.o));
        }
        size_t unwritten = 0;  // every output written?
        for (size_t i = 0; i < o2.size(); ++i) unwritten += (o2[i] == 0xffff);
        float best = 1e30f;
        for (int rd = 0; rd < rounds; ++rd) {
            float ms;
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 12; ++i) CK(variants[v].fn(d_qkv, d_o2, B, T, nullptr, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("  [%d] %s  max|err| %.3e nan %d unwritten %zu   12 launches %.3f ms  %.0f TF/s\n", v, variants[v].name, worst2, nan2,
               unwritten, best, 12 * flops / best / 1e9);
    }
    // timing: interleaved rounds
    for (int rd = 0; rd < rounds; ++rd) {
        float ms_old, ms_new;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 12; ++i) run_old();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_old, e0, e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 12; ++i) run_new();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_new, e0, e1));
        printf("round %d: 12 launches  old %.3f ms (%.0f TF/s)   new %.3f ms (%.0f TF/s)\n", rd, ms_old, 12 * flops / ms_old / 1e9,
               ms_new, 12 * flops / ms_new / 1e9);
    }
    return 0;
}
