// micro-benchmark: latency of the PGS row update chain on one wave (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ROWS(BODY) _Pragma("unroll") for (int k = 0; k < 16; k++) { BODY }
template <int V>
__global__ __launch_bounds__(64) void bench(float *out, long long *cyc, int reps) {
    float e = threadIdx.x * 0.001f, blo = -1.f, bhi = 1.f, a = 0.01f + threadIdx.x * 1e-4f, dv = 0.f, dv2 = 0.f;
    unsigned res = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        ROWS(
            float d; int sd;
            if constexpr (V == 0) {   // current deferred-commit row
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_nop 0\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_writelane_b32 %[dv], %[sd], 5\n\tv_fmac_f32 %[e], %[sd], %[a]\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [dv] "+v"(dv), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
                res = max(res, (unsigned)sd & 0x7fffffffu);
            } else if constexpr (V == 1) {   // no writelane, no residual
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_nop 0\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
            } else if constexpr (V == 2) {   // exec-narrowing commit (previous version)
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_lshl_b64 exec, 1, 5\n\tv_readlane_b32 %[sd], %[d], 5\n\tv_sub_f32 %[blo], %[blo], %[d]\n\tv_sub_f32 %[bhi], %[bhi], %[d]\n\ts_mov_b64 exec, -1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [blo] "+v"(blo), [bhi] "+v"(bhi), [e] "+v"(e) : [a] "v"(a));
            } else if constexpr (V == 3) {   // pure dependent VALU chain of 3 (no cross-lane)
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\tv_mul_f32 %[d], %[d], %[a]\n\tv_fmac_f32 %[e], %[d], %[a]\n\t"
                             : [d] "=&v"(d), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a)); sd = 0;
            } else if constexpr (V == 4) {   // fmac before writelane
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_nop 0\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\tv_writelane_b32 %[dv], %[sd], 5\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [dv] "+v"(dv), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
                res = max(res, (unsigned)sd & 0x7fffffffu);
            } else if constexpr (V == 7) {   // two alternating dv accumulators (breaks the writelane RAW chain)
                if (k & 1) asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_nop 0\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\tv_writelane_b32 %[dv], %[sd], 5\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [dv] "+v"(dv), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
                else asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_nop 0\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\tv_writelane_b32 %[dv], %[sd], 5\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [dv] "+v"(dv2), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
                res = max(res, (unsigned)sd & 0x7fffffffu);
            } else if constexpr (V == 8) {   // V0 without the s_nop 0 after med3
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\tv_writelane_b32 %[dv], %[sd], 5\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [dv] "+v"(dv), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
                res = max(res, (unsigned)sd & 0x7fffffffu);
            } else if constexpr (V == 9) {   // exec-free masked commit: v_cmp + 2 v_cndmask... replaced by: blo/bhi kept, u committed with v_cndmask on a mask register
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_nop 0\n\tv_readlane_b32 %[sd], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
                dv = (threadIdx.x == 5) ? d : dv;
                res = max(res, (unsigned)sd & 0x7fffffffu);
            } else if constexpr (V == 5) {   // readfirstlane instead of readlane
                asm volatile("v_med3_f32 %[d], -%[e], %[blo], %[bhi]\n\ts_nop 0\n\tv_readfirstlane_b32 %[sd], %[d]\n\ts_nop 1\n\tv_fmac_f32 %[e], %[sd], %[a]\n\t"
                             : [d] "=&v"(d), [sd] "=&s"(sd), [e] "+v"(e) : [blo] "v"(blo), [bhi] "v"(bhi), [a] "v"(a));
            } else if constexpr (V == 6) {   // DPP-free LDS-free broadcast through ds_bpermute? (reference point)
                d = __builtin_amdgcn_fmed3f(-e, blo, bhi);
                float b = __shfl(d, 5);
                e = fmaf(b, a, e); sd = 0;
            }
        )
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * 64] = e + dv + dv2 + blo + bhi + (float)res;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V> void run(const char *name, int blocks) {
    float *out; long long *cyc; hipMalloc(&out, 64 * 4 * blocks); hipMalloc(&cyc, 8 * blocks);
    int reps = 200; printf("start %s\n", name); fflush(stdout);
    hipLaunchKernelGGL(bench<V>, dim3(blocks), dim3(64), 0, 0, out, cyc, reps);
    hipLaunchKernelGGL(bench<V>, dim3(blocks), dim3(64), 0, 0, out, cyc, reps);
    hipDeviceSynchronize();
    long long h[4096]; hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; i++) s += h[i];
    printf("%-46s blocks=%5d  %.1f cycles/row\n", name, blocks, s / blocks / (reps * 16.0)); fflush(stdout);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int blocks : {1, 1024, 4096}) {
        run<0>("V0 med3,readlane,writelane,fmac (current)", blocks);
        run<1>("V1 med3,readlane,fmac", blocks);
        run<3>("V3 three dependent VALU, no cross-lane", blocks);
        run<4>("V4 fmac before writelane", blocks);
        run<7>("V7 two alternating dv accumulators", blocks);
        run<8>("V8 V4 without s_nop 0", blocks);
        run<9>("V9 commit with v_cndmask on a kept mask", blocks);
        run<5>("V5 readfirstlane variant", blocks);
    }
    return 0;
}
