// Probe: HBM write rate of two store patterns for a [M][2304] fp32 matrix (the q|k|v output of a ViT-B block, 232 MB):
//  A: "row-owner": lane = row, each store instruction writes 16 B of 32 different rows (two lane halves adjacent -> 32 B per row)
//  B: "coalesced": a wave writes 1 KB contiguous per instruction (float4 per lane along a row)
// Both write the same bytes from registers; persistent work-groups walking 256-row tiles like kan_fwd_ws_bf16_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* y, long long M, int ld, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    const int head = blockIdx.x;            // 12 column sets of 192 columns
    for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
        const long long r0 = (long long)tile * 256 + wave * 32;
        f32x4 v = {1.0f * tile, 2.0f, 3.0f, 4.0f};
        if (MODE == 0) {
            const long long r = r0 + l31;
            if (r < M) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<f32x4*>(y + r * ld + (p * 12 + head) * 64 + t * 32 + 8 * q + 4 * hf) = v;
            }
        } else {
            // 32 rows x 3 groups x 64 columns = 32 x 3 x 16 float4; lane -> (row, float4) with 16 lanes per 64-column row segment
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = i * 4 + (lane >> 4), c4 = (lane & 15) * 4;
                    const long long r = r0 + rr;
                    if (r < M) *reinterpret_cast<f32x4*>(y + r * ld + (p * 12 + head) * 64 + c4) = v;
                }
        }
    }
}
int main() {
    const long long M = 25216; const int ld = 2304; const int ntiles = (M + 255) / 256;
    float* y; hipMalloc(&y, M * ld * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int py : {21, 42, 99}) {
            for (int it = 0; it < 3; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(12, py), dim3(512), 0, 0, y, M, ld, ntiles);
                else hipLaunchKernelGGL(k<1>, dim3(12, py), dim3(512), 0, 0, y, M, ld, ntiles);
            }
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(12, py), dim3(512), 0, 0, y, M, ld, ntiles);
                else hipLaunchKernelGGL(k<1>, dim3(12, py), dim3(512), 0, 0, y, M, ld, ntiles);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("mode %d (%s) grid.y %3d: %.1f us  %.2f TB/s\n", mode, mode ? "coalesced" : "row-owner", py, ms * 1e3, M * ld * 4.0 / ms / 1e9);
        }
    return 0;
}
