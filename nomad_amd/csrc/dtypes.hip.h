// Element types of the activation buffers: fp32 (reference precision) or bf16 (config C5 long-form path;
// all statistics, softmax and accumulators stay fp32).  load4/store4 move four consecutive elements.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace nomad {

// per-workgroup timeline of the GEMM timing probes (gemm_bf16_8phase.hip.h ABL 7, gemm_f32.hip.h OPT bit 128; nomad_diag_timeline)
constexpr int kTimelineSlots = 4096;
// (static: this header is part of three translation units; each GEMM unit's probes write, and its reader reads, its own copy)
static __device__ unsigned long long g_timeline[kTimelineSlots * 6];


typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) {
    return *reinterpret_cast<const float4*>(p);
}
template <>
__device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <typename T>
__device__ __forceinline__ void store4(T* p, float4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, float4 v) {
    *reinterpret_cast<float4*>(p) = v;
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;  // v_cvt_pk_bf16_f32, RNE
    *reinterpret_cast<bf16x4*>(p) = o;
}

template <typename T>
__device__ __forceinline__ float load1(const T* p) { return (float)*p; }

// Split storage ("bf16x3" path): an fp32 value x is kept as TWO bf16 planes, hi = bf16(x) and lo = bf16(x - hi),
// `plane` elements apart (hi + lo carries 16 mantissa bits; same bytes per element as fp32).  The GEMMs of that
// path multiply the planes as three bf16 MFMA products, hi*hi + hi*lo + lo*hi, accumulated in fp32
// (gemm_bf16_8phase.hip.h, X3); everything else reads hi + lo.  bf16s_t tags such a buffer (2-byte pointer steps).
struct bf16s_t {
    bf16_t v;
};

// load4p / store4p: load4 / store4 with the plane distance of a split buffer (ignored by the plain types).
template <typename T>
__device__ __forceinline__ float4 load4p(const T* p, long long) {
    return load4<T>(p);
}
template <>
__device__ __forceinline__ float4 load4p<bf16s_t>(const bf16s_t* p, long long plane) {
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(p);
    const bf16x4 l = *reinterpret_cast<const bf16x4*>(p + plane);
    return make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                       (float)h[3] + (float)l[3]);
}
template <typename T>
__device__ __forceinline__ void store4p(T* p, long long, float4 v) {
    store4<T>(p, v);
}
template <>
__device__ __forceinline__ void store4p<bf16s_t>(bf16s_t* p, long long plane, float4 v) {
    bf16x4 h, l;
    h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
    l[0] = (bf16_t)(v.x - (float)h[0]); l[1] = (bf16_t)(v.y - (float)h[1]);
    l[2] = (bf16_t)(v.z - (float)h[2]); l[3] = (bf16_t)(v.w - (float)h[3]);
    *reinterpret_cast<bf16x4*>(p) = h;
    *reinterpret_cast<bf16x4*>(p + plane) = l;
}

// fp32 -> split planes (weights at nomad_enable_bf16x3, test inputs).  out[0..n) = hi, out[plane..plane+n) = lo.
static __global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ in, bf16s_t* __restrict__ out,
                                                         long long plane, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        store4p<bf16s_t>(out + 4 * i, plane, *reinterpret_cast<const float4*>(in + 4 * i));
}
// split planes -> fp32 (tests)
static __global__ __launch_bounds__(256) void unsplit_bf16_kernel(const bf16s_t* __restrict__ in, long long plane,
                                                           float* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        *reinterpret_cast<float4*>(out + 4 * i) = load4p<bf16s_t>(in + 4 * i, plane);
}

// fp32 -> bf16 copy (weights at nomad_enable_bf16).  n % 4 == 0.
static __global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        store4<bf16_t>(out + 4 * i, *reinterpret_cast<const float4*>(in + 4 * i));
}

// the same with the first `n4_scaled` float4 groups multiplied by `scale` (q rows of the fused QKV weight: log2 e)
static __global__ __launch_bounds__(256) void to_bf16_scaled_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long n4,
                                                             long long n4_scaled, float scale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = *reinterpret_cast<const float4*>(in + 4 * i);
        if (i < n4_scaled) v = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
        store4<bf16_t>(out + 4 * i, v);
    }
}
// out[i] = in[i] * (i < n_scaled ? scale : 1)   (fp32 bias of the fused QKV projection, q part)
static __global__ __launch_bounds__(256) void scale_head_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int n_scaled,
                                                         float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] * (i < n_scaled ? scale : 1.0f);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to ONE device: a flag per (kernel instantiation, device), so that a process
// that drives several GPUs (one engine each) configures every kernel on each of them.  Racing host threads at worst set the attribute
// twice (the same value); the flags are relaxed atomics so that this is not a data race in the language's sense either.  These
// flags, the last-error text and the diagnostic library's timeline buffer are the only process-wide state; none affects a result.
// (hipGetDevice is a thread-local read of the runtime, ~20 ns per launch.)
struct LdsAttrOnce {
    static constexpr int kMaxDev = 64;
    std::atomic<bool> done[kMaxDev] = {};
    hipError_t ensure(const void* kern, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < kMaxDev && done[dev].load(std::memory_order_relaxed)) return hipSuccess;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && dev >= 0 && dev < kMaxDev) done[dev].store(true, std::memory_order_relaxed);
        return e;
    }
};

}  // namespace nomad
