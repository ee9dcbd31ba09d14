// Standalone A/B harness for the fp32 attention kernels (headline shape by default: 256 clips x 12 heads x T = 199):
// both against a float64 reference on sampled query rows (incl. a spiked key that forces a late rescale), log-sum-exp
// included, and interleaved timing in one process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/attn_f32 tools/micro/attn_f32.hip;  /tmp/attn_f32 [B=256] [T=199] [rounds=4]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../nomad_amd/csrc/attention_f32_v2.hip.h"

using namespace nomad;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 199, rounds = argc > 3 ? atoi(argv[3]) : 4;
    const long long M = (long long)B * T;
    std::vector<float> qkv((size_t)M * 2304);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : qkv) v = 0.5f * nd(rng);
    const int spike_key = T > 150 ? 150 : T - 1;
    for (int d = 0; d < 64; ++d) {
        qkv[(size_t)spike_key * 2304 + 768 + 3 * 64 + d] = 4.0f * ((d & 1) ? 1.f : -1.f);
        for (int t = 0; t < T; t += 7) qkv[(size_t)t * 2304 + 3 * 64 + d] = 0.6f * ((d & 1) ? 1.f : -1.f);
    }
    float *d_qkv, *d_o1, *d_o2, *d_l1, *d_l2;
    CK(hipMalloc(&d_qkv, qkv.size() * 4));
    CK(hipMalloc(&d_o1, (size_t)M * 768 * 4));
    CK(hipMalloc(&d_o2, (size_t)M * 768 * 4));
    CK(hipMalloc(&d_l1, (size_t)B * 12 * T * 4));
    CK(hipMalloc(&d_l2, (size_t)B * 12 * T * 4));
    CK(hipMemcpy(d_qkv, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_o1, 0xff, (size_t)M * 768 * 4));
    CK(hipMemset(d_o2, 0xff, (size_t)M * 768 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_f32_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, attn_f32_v2_lds()));
    auto run_old = [&]() { hipLaunchKernelGGL((attention_f32_kernel<float, false>), dim3((T + 63) / 64, B * 12), dim3(256), 0, s, d_qkv, d_o1, d_l1, T, (const int*)nullptr, DropCfg{}, 0u, 0, 0LL); };
    auto run_new = [&]() { CK(launch_attention_f32_v2(d_qkv, d_o2, d_l2, B, T, nullptr, s)); };
    run_old();
    run_new();
    CK(hipStreamSynchronize(s));
    std::vector<float> o1((size_t)M * 768), o2((size_t)M * 768), l1((size_t)B * 12 * T), l2((size_t)B * 12 * T);
    CK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(l1.data(), d_l1, l1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(l2.data(), d_l2, l2.size() * 4, hipMemcpyDeviceToHost));
    double w1 = 0, w2 = 0, wl1 = 0, wl2 = 0, refmax = 0;
    int nan2 = 0;
    std::vector<double> sc(T);
    const int clips[2] = {0, B - 1};
    for (int ci = 0; ci < (B > 1 ? 2 : 1); ++ci) {
        const int b = clips[ci];
        for (int hd = 0; hd < 12; ++hd) {
            std::vector<int> rows = {0, 1, 7, 31, 32, 63, 64, 127, 128, 129, T / 2, T - 2, T - 1};
            for (int k = 0; k < 10; ++k) rows.push_back((int)(rng() % (unsigned)T));
            for (int q : rows) {
                if (q >= T || q < 0) continue;
                const float* Q = &qkv[((size_t)b * T + q) * 2304 + hd * 64];
                double mx = -1e300;
                for (int j = 0; j < T; ++j) {
                    const float* K = &qkv[((size_t)b * T + j) * 2304 + 768 + hd * 64];
                    double a = 0;
                    for (int d = 0; d < 64; ++d) a += (double)Q[d] * (double)K[d];
                    sc[j] = a;
                    mx = std::max(mx, a);
                }
                double l = 0;
                for (int j = 0; j < T; ++j) { sc[j] = std::exp(sc[j] - mx); l += sc[j]; }
                const double lse = mx + std::log(l);
                const size_t li = ((size_t)b * 12 + hd) * T + q;
                wl1 = std::max(wl1, std::fabs((double)l1[li] - lse));
                wl2 = std::max(wl2, std::fabs((double)l2[li] - lse));
                for (int d = 0; d < 64; ++d) {
                    double o = 0;
                    for (int j = 0; j < T; ++j) o += sc[j] * (double)qkv[((size_t)b * T + j) * 2304 + 1536 + hd * 64 + d];
                    o /= l;
                    const size_t idx = ((size_t)b * T + q) * 768 + hd * 64 + d;
                    refmax = std::max(refmax, std::fabs(o));
                    w1 = std::max(w1, std::fabs((double)o1[idx] - o));
                    if (!(o2[idx] == o2[idx])) ++nan2;
                    else w2 = std::max(w2, std::fabs((double)o2[idx] - o));
                }
            }
        }
    }
    size_t unwritten = 0;
    for (size_t i = 0; i < o2.size(); ++i) { uint32_t u; memcpy(&u, &o2[i], 4); unwritten += (u == 0xffffffffu); }
    printf("B=%d T=%d ref max|o| %.3f | old: max|err| %.3e lse err %.3e | new: max|err| %.3e lse err %.3e nan %d unwritten %zu\n", B, T, refmax,
           w1, wl1, w2, wl2, nan2, unwritten);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flops = 4.0 * B * 12.0 * (double)T * T * 64;
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
        printf("round %d: 12 launches  old %.3f ms (%.1f TF/s)   new %.3f ms (%.1f TF/s)\n", rd, ms_old, 12 * flops / ms_old / 1e9, ms_new,
               12 * flops / ms_new / 1e9);
    }
    return 0;
}
