// micro-benchmark: a solver row whose clamp runs under EXEC = {its lane} and writes the per-pass delta vector in place (no v_writelane commit),
// against the shipped rows.  f32 at 1 / 4 waves per SIMD, f64 at 1 / 2 (the kernels' occupancies).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ROWS(BODY) _Pragma("unroll") for (int k = 0; k < 16; k++) { BODY }
template <int V>
__global__ __launch_bounds__(64) void bench32(float *out, long long *cyc, int reps) {
    float e = threadIdx.x * 0.001f, blo = -1.f, bhi = 1.f, a = 0.01f + threadIdx.x * 1e-4f, dv = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        ROWS(
            float d; int sd;
            if constexpr (V == 0) {          // shipped pipelined motor row: med3 | writelane(prev) | readlane | s_nop 1 | fmac
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\tv_writelane_b32 %[dv], %[sd], 5\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\t"
                             : [d] "=&v"(d), [sd] "+s"(sd), [dv] "+v"(dv), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
            } else if constexpr (V == 1) {   // exec-masked in-place clamp
                asm volatile("s_mov_b64 exec, 32\n\tv_med3_f32 %[dv], -%[e], %[blo], %[bhi]\n\ts_mov_b64 exec, -1\n\tv_readlane_b32 %[sd], %[dv], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\t"
                             : [sd] "=&s"(sd), [dv] "+v"(dv), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
            } else if constexpr (V == 2) {   // the same, the next row's mask set right after the fmac (one s_mov fewer between min and readlane is impossible; this orders them differently)
                asm volatile("v_med3_f32 %[dv], -%[e], %[blo], %[bhi]\n\ts_mov_b64 exec, -1\n\tv_readlane_b32 %[sd], %[dv], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\ts_mov_b64 exec, 32\n\t"
                             : [sd] "=&s"(sd), [dv] "+v"(dv), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
            }
        )
        asm volatile("s_mov_b64 exec, -1");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = e + dv + blo + bhi;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V>
__global__ __launch_bounds__(64) void bench64(double *out, long long *cyc, int reps) {
    double e = threadIdx.x * 0.001, blo = -1., bhi = 1., a = 0.01 + threadIdx.x * 1e-4, dv = 0.;
    int dlo = 0, dhi = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        ROWS(
            if constexpr (V == 0) {          // shipped pipelined motor row (7 VALU)
                asm volatile("v_max_f64 v[0:1], -%[e], %[blo]\n\tv_min_f64 v[0:1], v[0:1], %[bhi]\n\tv_writelane_b32 %[dlo], s4, 5\n\tv_readlane_b32 s39, v1, 5\n\tv_readlane_b32 s38, v0, 5\n\t"
                             "v_writelane_b32 %[dhi], s5, 5\n\ts_nop 0\n\tv_fmac_f64 %[e], s[38:39], %[a]\n\t"
                             "v_max_f64 v[0:1], -%[e], %[blo]\n\tv_min_f64 v[0:1], v[0:1], %[bhi]\n\tv_writelane_b32 %[dlo], s38, 5\n\tv_readlane_b32 s5, v1, 5\n\tv_readlane_b32 s4, v0, 5\n\t"
                             "v_writelane_b32 %[dhi], s39, 5\n\ts_nop 0\n\tv_fmac_f64 %[e], s[4:5], %[a]\n\t"
                             : [e] "+v"(e), [dlo] "+v"(dlo), [dhi] "+v"(dhi) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a) : "v0", "v1", "s4", "s5", "s38", "s39");
            } else if constexpr (V == 1) {   // exec-masked in-place clamp (5 VALU + 2 SALU + s_nop), two rows like V0
                asm volatile("s_mov_b64 exec, 32\n\tv_max_f64 v[0:1], -%[e], %[blo]\n\tv_min_f64 v[2:3], v[0:1], %[bhi]\n\ts_mov_b64 exec, -1\n\t"
                             "v_readlane_b32 s4, v2, 5\n\tv_readlane_b32 s5, v3, 5\n\ts_nop 1\n\tv_fmac_f64 %[e], s[4:5], %[a]\n\t"
                             "s_mov_b64 exec, 32\n\tv_max_f64 v[0:1], -%[e], %[blo]\n\tv_min_f64 v[2:3], v[0:1], %[bhi]\n\ts_mov_b64 exec, -1\n\t"
                             "v_readlane_b32 s4, v2, 5\n\tv_readlane_b32 s5, v3, 5\n\ts_nop 1\n\tv_fmac_f64 %[e], s[4:5], %[a]\n\t"
                             : [e] "+v"(e), "+{v[2:3]}"(dv) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a) : "v0", "v1", "s4", "s5");
            } else if constexpr (V == 2) {   // exec-masked in-place clamp (5 VALU + 2 SALU + s_nop), two rows like V0
                asm volatile("v_max_f64 v[0:1], -%[e], %[blo]\n\tv_min_f64 v[2:3], v[0:1], %[bhi]\n\ts_mov_b64 exec, -1\n\t"
                             "v_readlane_b32 s4, v2, 5\n\tv_readlane_b32 s5, v3, 5\n\ts_nop 1\n\tv_fmac_f64 %[e], s[4:5], %[a]\n\ts_mov_b64 exec, 32\n\t"
                             "v_max_f64 v[0:1], -%[e], %[blo]\n\tv_min_f64 v[2:3], v[0:1], %[bhi]\n\ts_mov_b64 exec, -1\n\t"
                             "v_readlane_b32 s4, v2, 5\n\tv_readlane_b32 s5, v3, 5\n\ts_nop 1\n\tv_fmac_f64 %[e], s[4:5], %[a]\n\ts_mov_b64 exec, 32\n\t"
                             : [e] "+v"(e), "+{v[2:3]}"(dv) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a) : "v0", "v1", "s4", "s5");
            }
        )
        asm volatile("s_mov_b64 exec, -1");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = e + dv + blo + bhi + dlo + dhi;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename T, typename K> void run(K kern, const char *name, int blocks, int rows_per_body, size_t lds) {
    T *out; long long *cyc; hipMalloc(&out, 64 * sizeof(T) * blocks); hipMalloc(&cyc, 8 * blocks);
    int reps = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, cyc, reps);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, cyc, reps);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    long long *h = new long long[blocks]; hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; i++) s += h[i];
    const double rows = (double)reps * 16.0 * rows_per_body;
    printf("%-44s blocks=%5d  %.1f cycles/row per wave, %.2f G rows/s chip-wide\n", name, blocks, s / blocks / rows, blocks * rows / (ms * 1e6)); fflush(stdout);
    hipFree(out); hipFree(cyc); delete[] h;
}
int main() {
    // LDS per block pins the occupancy: 160 KB per CU / (4 SIMDs x waves per SIMD)
    for (int blocks : {1024, 4096, 4096}) {
        const size_t lds = 9936;
        run<float>(bench32<0>, "f32 shipped row (4 VALU)", blocks, 1, lds);
        run<float>(bench32<1>, "f32 exec-masked in-place clamp (3 VALU)", blocks, 1, lds);
        run<float>(bench32<2>, "f32 exec-masked, mask set after the fmac", blocks, 1, lds);
    }
    for (int blocks : {1024, 2048, 2048}) {
        const size_t lds = 19872;
        run<double>(bench64<0>, "f64 shipped row (7 VALU)", blocks, 2, lds);
        run<double>(bench64<1>, "f64 exec-masked in-place clamp (5 VALU)", blocks, 2, lds);
        run<double>(bench64<2>, "f64 exec-masked, mask set after the fmac", blocks, 2, lds);
    }
    return 0;
}
