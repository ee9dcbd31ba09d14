// Round 6 micro-benchmark: what does the SHAPE of a wave's 16-byte-per-lane store instruction cost?
// The persistent bf16 GEMM's epilogue (gemm_bf16_p9.hip.h) stores an MFMA accumulator as it stands: lane (row = lane & 15, q = lane >> 4)
// writes 16 bytes at row * ld + 16 q, so ONE buffer_store_dwordx4 covers 16 rows x 64 bytes - sixteen half cache lines - and the other half
// of each line comes from a second instruction.  The probes of the round put 4.5-9 us per 128 KB tile on the output stores.
// Variants, same bytes to the same places (a 256 x 256 bf16 tile per workgroup of 8 waves, one workgroup per CU, `tiles` tiles each):
//   0  "half lines":  instruction (i, jh): rows 16 i + (lane & 15), bytes 64 jh + 16 (lane >> 4)       (the GEMM's shape)
//   1  "whole lines": instruction (i, h):  rows 16 i + 8 h + (lane & 7), bytes 16 (lane >> 3)             (8 rows x 128 bytes)
//   2  "row runs":    instruction k:       row 2 k + (lane >> 5), bytes 16 (lane & 31)                  (2 rows x 512 bytes)
// (the wave's 128 x 64 part of the tile: 16 instructions of 1 KB per wave in every variant.)  nt = aux 2 as the GEMM's stores.
// Build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/micro/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512) void k(unsigned short* out, int ld, int tiles, int M) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    u32x4 v = {(unsigned)threadIdx.x, (unsigned)blockIdx.x, 3u, 4u};
    for (int t = 0; t < tiles; ++t) {
        const long long tile = (long long)blockIdx.x * tiles + t;      // tile walk: 3 column tiles per row panel (N = 768)
        const long long m0 = (tile / 3) * 256 + wr * 128, n0 = (tile % 3) * 256 + wc * 64;
        if (m0 >= M) break;
        unsigned short* base = out + m0 * ld + n0;
        __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 1 << 30, 0x00020000);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int off;
            if (V == 0) off = ((i >> 1) * 16 + (lane & 15)) * ld * 2 + (i & 1) * 64 + 16 * (lane >> 4);
            else if (V == 1) off = ((i >> 1) * 16 + (i & 1) * 8 + (lane & 7)) * ld * 2 + 16 * (lane >> 3);
            else off = (8 * i + (lane >> 3)) * ld * 2 + 16 * (lane & 7);   // 8 rows x 128 B, lanes of a row consecutive
            v[2] += i;
            __builtin_amdgcn_raw_buffer_store_b128(v, rc, off, 0, 2);
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

int main() {
    const int M = 47968 + 256, N = 768, tiles = 3, grid = 256;
    unsigned short* out;
    hipMalloc(&out, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)grid * tiles * 256 * 256 * 2;
    for (int rep = 0; rep < 3; ++rep)
        for (int v = 0; v < 3; ++v) {
            float best = 1e9f;
            for (int it = 0; it < 20; ++it) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, N, tiles, M);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, N, tiles, M);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, N, tiles, M);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("variant %d: %.1f us for %.0f MB (%d tiles of 128 KB per CU): %.2f TB/s, %.1f B/clk/CU at 2.1 GHz, %.2f us per tile\n", v, best * 1e3,
                   bytes / 1e6, tiles, bytes / (best * 1e-3) / 1e12, bytes / grid / (best * 1e-3 * 2.1e9), best * 1e3 / tiles);
        }
    return 0;
}
