// N_deg x N_ref Euclidean distance matrix + row mean = the NOMAD score (nomad.py:108-111:
// scipy.spatial.distance.cdist on float32 embeddings promoted to float64, then np.mean(axis=1)).
//
// Computed in float64 in the DIFFERENCE form sqrt(sum_k (a_k - b_k)^2), k ascending, exactly the
// loop SciPy runs - the expansion |a|^2+|b|^2-2a.b loses ~4e-5 absolute at d~0.01 in fp32 and is
// never used.  The stage is HBM-write bound (8 B per pair); fp64 VALU throughput is ample.
//
// Workgroup = 32 deg rows x ALL refs (64 at a time through LDS); thread (ty, tx) of a 16x16
// layout owns deg rows {ty, ty+16} and refs {tx + 16j}.  Row sums are accumulated per thread in
// ref order and folded across tx in a fixed order, so the means are run-to-run deterministic
// (no atomics).
#pragma once
#include <hip/hip_runtime.h>

namespace nomad {

constexpr int kPairLD = 65;

__global__ __launch_bounds__(256) void pairwise_f64_kernel(const float* __restrict__ deg, int Nd,
                                                           const float* __restrict__ ref, int Nr,
                                                           double* __restrict__ dist, double* __restrict__ mean) {
    __shared__ float As[32 * kPairLD];
    __shared__ float Bs[64 * kPairLD];
    __shared__ double rs[32][17];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int d0 = blockIdx.x * 32;
    double rowsum[2] = {0.0, 0.0};

    for (int r0 = 0; r0 < Nr; r0 += 64) {
        double acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
        for (int k0 = 0; k0 < 256; k0 += 64) {
            __syncthreads();
            for (int i = tid; i < 32 * 64; i += 256) {
                const int r = i >> 6, k = i & 63;
                const int d = min(d0 + r, Nd - 1);
                As[r * kPairLD + k] = deg[(long long)d * 256 + k0 + k];
            }
            for (int i = tid; i < 64 * 64; i += 256) {
                const int r = i >> 6, k = i & 63;
                const int rr = min(r0 + r, Nr - 1);
                Bs[r * kPairLD + k] = ref[(long long)rr * 256 + k0 + k];
            }
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < 64; ++k) {
                const double a0 = (double)As[ty * kPairLD + k], a1 = (double)As[(ty + 16) * kPairLD + k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double bv = (double)Bs[(tx + 16 * j) * kPairLD + k];
                    const double e0 = a0 - bv, e1 = a1 - bv;
                    acc[0][j] += e0 * e0;
                    acc[1][j] += e1 * e1;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d = d0 + ty + 16 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + tx + 16 * j;
                const double dv = sqrt(acc[i][j]);
                if (d < Nd && r < Nr) {
                    if (dist) dist[(long long)d * Nr + r] = dv;
                    rowsum[i] += dv;
                }
            }
        }
    }
    rs[ty][tx] = rowsum[0];
    rs[ty + 16][tx] = rowsum[1];
    __syncthreads();
    if (tid < 32 && d0 + tid < Nd) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += rs[tid][j];
        mean[d0 + tid] = s / (double)Nr;
    }
}

}  // namespace nomad
