// Micro-benchmark of the row-block GEMM primitives of csrc/td3_rows.hip: one 256 -> 256 dense layer (mm_nt) and one input-gradient product (mm_nn)
// for B rows, 16 rows per single-wave workgroup.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rows_gemm rows_gemm.hip ; run: ./rows_gemm [B]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define TD3_H 256
#define TD3_S 26
#define TD3_A 18
#define TD3_SA 44
#define TD3_ROW 72
static __device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
static __device__ __forceinline__ float rng_uniform(const uint64_t *rng, uint32_t tag, uint32_t e) { return 0.5f; }
static __device__ __forceinline__ float rng_normal(const uint64_t *rng, uint32_t tag, uint32_t e) { return 0.5f; }
#include "../../include/plentd3.h"
#include "../../plen_ml_walk_amd/csrc/td3_rows.hip"
#define HEAD  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4; const int b0 = blockIdx.x * 16; const int brow = min(b0 + r, B - 1); const RowBlock rb{b0, B, r, g, brow};
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nt(const float* X, const float* W, const float* bias, float* Y, int B, int reps) {
    HEAD
    for (int i = 0; i < reps; i++)
        for (int n0 = 0; n0 < 256; n0 += 128) dense_relu<8, false>(mkrs(X, (size_t)B*1024), rb.aoff(256), 256, mkrs(W, 256*1024), 256, bias, n0, mkrs(Y, (size_t)B*1024), 256, rb);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nn(const float* X, const float* W, float* Y, int B, int reps) {
    HEAD
    for (int i = 0; i < reps; i++)
        for (int j0 = 0; j0 < 256; j0 += 128) {
            floatx4 acc[8];
            for (int t = 0; t < 8; t++) acc[t] = floatx4{0, 0, 0, 0};
            mm_nn<8, false>(mkrs(X, (size_t)B*1024), rb.aoff(256), mkrs(W, 256*1024), 256, j0, 256, acc, r, g);
            for (int t = 0; t < 8; t++) for (int i2 = 0; i2 < 4; i2++) bstore1(acc[t][i2], mkrs(Y, (size_t)B*1024), rb.soff(256), 4u * (i2 * 256 + j0 + 16 * t));
        }
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, reps = 8;
    float *X, *W, *b, *Y;
    hipMalloc(&X, (size_t)B * 1024); hipMalloc(&W, 256 * 1024); hipMalloc(&b, 1024); hipMalloc(&Y, (size_t)B * 1024);
    hipMemset(X, 0, (size_t)B * 1024); hipMemset(W, 0, 256 * 1024); hipMemset(b, 0, 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        for (int it = 0; it < 3; it++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_nt, dim3(B / 16), dim3(64), 0, 0, X, W, b, Y, B, reps);
            else hipLaunchKernelGGL(k_nn, dim3(B / 16), dim3(64), 0, 0, X, W, Y, B, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps, flops = 2.0 * B * 256 * 256;
            if (it == 2) printf("%s B=%d: %.1f us per 256x256 layer = %.1f TFLOP/s (%d waves; MFMA-bound at one wave per CU: %.1f us)\n", which ? "mm_nn" : "mm_nt", B, us, flops / us * 1e-6, B / 16,
                                flops / (B / 16) / 64.0 / 2.4e3);
        }
    }
    return 0;
}
