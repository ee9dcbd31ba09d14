// N_deg x N_ref Euclidean distance matrix + row mean = the NOMAD score (nomad.py:108-111:
// scipy.spatial.distance.cdist on float32 embeddings promoted to float64, then np.mean(axis=1)).
//
// Computed in float64 in the DIFFERENCE form sqrt(sum_k (a_k - b_k)^2), k ascending, exactly the
// loop SciPy runs - the expansion |a|^2+|b|^2-2a.b loses ~4e-5 absolute at d~0.01 in fp32 and is
// never used (and has no exact zero on the diagonal).
//
// Bound.  Per pair: 256 subtractions + 256 fused multiply-adds in fp64 and 8 bytes written.  At config C3
// (10 000 x 1 000) that is 5.1e9 fp64 lane-operations against 80 MB: 0.13 ms at the 78.6 TFLOP/s fp64 vector
// peak, 0.016 ms at the HBM write rate - the stage is fp64-VALU bound, not write bound.
//
// pairwise_tile_kernel: workgroup = 64 deg rows x 64 refs, thread (ty, tx) of a 16 x 16 layout owns the 4 x 4 pairs
// deg 4ty.. x ref 4tx.. (32 fp64 operations per k for 8 LDS doubles read).  Both operands are converted to float64
// once, when a 64-deep k-chunk is staged into LDS as [k][row] (two ds_read_b128 per operand and k).  The grid is
// (ref tiles, deg tiles) - 2 512 workgroups at C3 instead of the 313 of a one-dimensional grid - so a deg row's sum
// over the refs is taken per 64-ref tile (refs in order inside the thread, threads folded in tx order) into
// part[ref tile][deg], and pairwise_mean_kernel adds the tiles in order: deterministic, no atomics.
#pragma once
#include <hip/hip_runtime.h>

namespace nomad {

constexpr int kPairTile = 64;

__global__ __launch_bounds__(256) void pairwise_tile_kernel(const float* __restrict__ deg, int Nd,
                                                            const float* __restrict__ ref, int Nr,
                                                            double* __restrict__ dist, double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) double As[64][kPairTile];  // [k][deg row]
    __shared__ __attribute__((aligned(16))) double Bs[64][kPairTile];  // [k][ref]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int r0 = blockIdx.x * kPairTile, d0 = blockIdx.y * kPairTile;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    // staging: lane -> row tid & 63 (consecutive lanes write consecutive doubles of one LDS row: conflict-free), the four
    // waves take k columns 4 * wave + 16 q .. + 3 of the chunk
    const int srow = tid & 63, skq = (tid >> 6) * 4;
    const float* ap = deg + (long long)min(d0 + srow, Nd - 1) * 256 + skq;
    const float* bp = ref + (long long)min(r0 + srow, Nr - 1) * 256 + skq;
    for (int k0 = 0; k0 < 256; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = skq + 16 * q;
            const float4 a = *reinterpret_cast<const float4*>(ap + k0 + 16 * q);
            const float4 b = *reinterpret_cast<const float4*>(bp + k0 + 16 * q);
            As[kk][srow] = (double)a.x; As[kk + 1][srow] = (double)a.y; As[kk + 2][srow] = (double)a.z; As[kk + 3][srow] = (double)a.w;
            Bs[kk][srow] = (double)b.x; Bs[kk + 1][srow] = (double)b.y; Bs[kk + 2][srow] = (double)b.z; Bs[kk + 3][srow] = (double)b.w;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[k][4 * ty + i];
                b[i] = Bs[k][4 * tx + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double e = a[i] - b[j];
                    acc[i][j] = fma(e, e, acc[i][j]);
                }
        }
    }
    __syncthreads();
    double* rs = &As[0][0];  // [64 deg rows][16 tx] row sums of this tile, folded below in tx order
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = d0 + 4 * ty + i;
        double s = 0.0;
        double4 out;
        double* o = &out.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double dv = sqrt(acc[i][j]);
            o[j] = dv;
            if (r0 + 4 * tx + j < Nr) s += dv;
        }
        if (dist && d < Nd) {
            double* dst = dist + (long long)d * Nr + r0 + 4 * tx;
            if (r0 + 4 * tx + 3 < Nr && (Nr & 1) == 0) {  // 16-byte aligned pairs
                *reinterpret_cast<double2*>(dst) = make_double2(out.x, out.y);
                *reinterpret_cast<double2*>(dst + 2) = make_double2(out.z, out.w);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r0 + 4 * tx + j < Nr) dst[j] = o[j];
            }
        }
        rs[(4 * ty + i) * 16 + tx] = s;
    }
    __syncthreads();
    if (tid < kPairTile && d0 + tid < Nd) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += rs[tid * 16 + j];
        part[(long long)blockIdx.x * Nd + d0 + tid] = s;
    }
}

// mean[d] = (sum over the ref tiles, in order) / Nr.  grid: ceil(Nd / 256) blocks of 256 threads.
__global__ __launch_bounds__(256) void pairwise_mean_kernel(const double* __restrict__ part, int ntiles, int Nd, int Nr,
                                                            double* __restrict__ mean) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= Nd) return;
    double s = 0.0;
    for (int t = 0; t < ntiles; ++t) s += part[(long long)t * Nd + d];
    mean[d] = s / (double)Nr;
}

}  // namespace nomad
