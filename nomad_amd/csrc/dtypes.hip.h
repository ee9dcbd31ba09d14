// Element types of the activation buffers: fp32 (reference precision) or bf16 (config C5 long-form path;
// all statistics, softmax and accumulators stay fp32).  load4/store4 move four consecutive elements.
#pragma once
#include <hip/hip_runtime.h>

namespace nomad {

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

// fp32 -> bf16 copy (weights at nomad_enable_bf16).  n % 4 == 0.
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        store4<bf16_t>(out + 4 * i, *reinterpret_cast<const float4*>(in + 4 * i));
}

}  // namespace nomad
