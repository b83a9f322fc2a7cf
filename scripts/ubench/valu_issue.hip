// Micro-benchmark: wave64 vector-instruction issue rate per SIMD on gfx950 as a function of waves per SIMD.
// Settles the VALU peak that bench.py's roofline.valu_issue is priced against (VERDICT r01 "What's weak" 3).
//
// Every wave runs `reps` x 32 INDEPENDENT instructions of one kind (16 destination registers, no RAW inside a group of
// 16), or a dependent chain (suffix _dep).  Workgroups are 256 threads = 4 waves = one wave per SIMD of a CU; each
// workgroup asks for 160 KB / W of LDS so that exactly W workgroups fit a CU, and the grid is 256 x W workgroups:
// W waves on each of the chip's 1024 SIMDs.  Reported per (op, W):
//   cyc/inst/wave  = s_memtime ticks of a wave / instructions it issued      (what ONE wave sees)
//   cyc/inst/SIMD  = that / W                                                (the SIMD's issue interval)
//   Ginst/s        = all wave-instructions / wall time of the launch (HIP events)
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip ; run: ./valu_issue [out.json]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA = 0, PKFMA, MED3, MUL, DPP, READLANE, WRITELANE, RCP, SIN, FMA64, FMA_DEP, MED3_DEP, ROW, ROWPAIR, NOPS };
static const char *opname[NOPS] = {"v_fma_f32", "v_pk_fma_f32", "v_med3_f32", "v_mul_f32", "v_mov_b32 dpp row_shr:1", "v_readlane_b32",
                                   "v_writelane_b32", "v_rcp_f32", "v_sin_f32", "v_fma_f64", "v_fma_f32 dependent chain",
                                   "v_med3_f32 dependent chain", "PGS row: med3,readlane,writelane,fmac (dependent)",
                                   "two interleaved PGS rows (independent chains)"};
// instructions counted per inner group of the kernel
static const int group_insts[NOPS] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32 * 4, 32 * 4};

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int reps) {
    extern __shared__ float lds[];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    float b = 0.999f, c = 1e-4f, e = threadIdx.x * 1e-3f, e2 = e + 1.f, dv = 0.f, dv2 = 0.f, d, d2;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], pb = {0.999f, 0.998f}, pc = {1e-4f, 2e-4f};
    double q[8], qb = 0.999, qc = 1e-4;
#pragma unroll
    for (int i = 0; i < 8; i++) { p[i] = f2{a[2 * i], a[2 * i + 1]}; q[i] = a[i]; }
    int s0 = 0, s1 = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if constexpr (OP == FMA) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(M)
#undef M
            } else if constexpr (OP == PKFMA) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(pb), "v"(pc));
                R16(M)
#undef M
            } else if constexpr (OP == MED3) {
#define M(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(M)
#undef M
            } else if constexpr (OP == MUL) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                R16(M)
#undef M
            } else if constexpr (OP == DPP) {
#define M(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                R16(M)
#undef M
            } else if constexpr (OP == READLANE) {
#define M(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(a[i])); s1 ^= s0;
                R16(M)
#undef M
            } else if constexpr (OP == WRITELANE) {
#define M(i) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(r));
                R16(M)
#undef M
            } else if constexpr (OP == RCP) {
#define M(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                R16(M)
#undef M
            } else if constexpr (OP == SIN) {
#define M(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
                R16(M)
#undef M
            } else if constexpr (OP == FMA64) {
#define M(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(q[i & 7]) : "v"(qb), "v"(qc));
                R16(M)
#undef M
            } else if constexpr (OP == FMA_DEP) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                R16(M)
#undef M
            } else if constexpr (OP == MED3_DEP) {
#define M(i) asm volatile("v_med3_f32 %0, -%0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                R16(M)
#undef M
            } else if constexpr (OP == ROW) {
                // the production solver row (plenvec.hip row asm): 4 vector instructions, dependent through e
#define M(i) asm volatile("v_med3_f32 %[d], -%[e], %[lo], %[hi]\n\ts_nop 0\n\tv_readlane_b32 %[s], %[d], 5\n\ts_nop 1\n\tv_fmac_f32 %[e], %[s], %[a]\n\tv_writelane_b32 %[dv], %[s], 5" \
                          : [d] "=&v"(d), [s] "=&s"(s0), [dv] "+v"(dv), [e] "+v"(e) : [lo] "v"(-b), [hi] "v"(b), [a] "v"(c));
                R16(M)
#undef M
            } else if constexpr (OP == ROWPAIR) {
                // two such chains interleaved (what two envs per wave would look like): 8 instructions per M, counted as 2 rows
#define M(i) if (i < 8) asm volatile("v_med3_f32 %[d], -%[e], %[lo], %[hi]\n\tv_med3_f32 %[d2], -%[e2], %[lo], %[hi]\n\tv_readlane_b32 %[s], %[d], 5\n\tv_readlane_b32 %[t], %[d2], 37\n\tv_fmac_f32 %[e], %[s], %[a]\n\tv_fmac_f32 %[e2], %[t], %[a]\n\tv_writelane_b32 %[dv], %[s], 5\n\tv_writelane_b32 %[dv2], %[t], 37" \
                          : [d] "=&v"(d), [d2] "=&v"(d2), [s] "=&s"(s0), [t] "=&s"(s1), [dv] "+v"(dv), [dv2] "+v"(dv2), [e] "+v"(e), [e2] "+v"(e2) : [lo] "v"(-b), [hi] "v"(b), [a] "v"(c));
                R16(M)
#undef M
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float acc = e + e2 + dv + dv2 + (float)s1;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) acc += p[i].x + p[i].y + (float)q[i];
    if (lds == nullptr) lds[threadIdx.x] = acc;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Res { int op, w; double cyc_wave, cyc_simd, ginst, ms; };

template <int OP> Res run(int W, int ncu, float *out, long long *cyc) {
    const int reps = (OP == RCP || OP == SIN || OP == FMA64) ? 2000 : 4000;
    size_t lds = (160 * 1024) / W - 512;             // exactly W workgroups per CU
    if (lds > 64 * 1024) CHK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks = ncu * W;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, out, cyc, reps);   // warm
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, out, cyc, reps);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 4);
    CHK(hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    double insts = (double)reps * 2 * group_insts[OP] / 2;   // group_insts is per h-pair of 16-instruction halves
    insts = (double)reps * group_insts[OP];
    Res r; r.op = OP; r.w = W; r.ms = ms;
    r.cyc_wave = med / insts; r.cyc_simd = r.cyc_wave / W;
    r.ginst = insts * blocks * 4 / (ms * 1e-3) / 1e9;
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return r;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
    float *out; long long *cyc;
    CHK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4)); CHK(hipMalloc(&cyc, (size_t)ncu * 8 * 4 * 8));
    std::vector<Res> all;
    const int Ws[4] = {1, 2, 4, 8};
    for (int wi = 0; wi < 4; wi++) {
        int W = Ws[wi];
        all.push_back(run<FMA>(W, ncu, out, cyc));      all.push_back(run<PKFMA>(W, ncu, out, cyc));
        all.push_back(run<MED3>(W, ncu, out, cyc));     all.push_back(run<MUL>(W, ncu, out, cyc));
        all.push_back(run<DPP>(W, ncu, out, cyc));      all.push_back(run<READLANE>(W, ncu, out, cyc));
        all.push_back(run<WRITELANE>(W, ncu, out, cyc)); all.push_back(run<RCP>(W, ncu, out, cyc));
        all.push_back(run<SIN>(W, ncu, out, cyc));      all.push_back(run<FMA64>(W, ncu, out, cyc));
        all.push_back(run<FMA_DEP>(W, ncu, out, cyc));  all.push_back(run<MED3_DEP>(W, ncu, out, cyc));
        all.push_back(run<ROW>(W, ncu, out, cyc));      all.push_back(run<ROWPAIR>(W, ncu, out, cyc));
    }
    printf("%-52s %3s %14s %14s %12s %9s\n", "op", "W", "cyc/inst/wave", "cyc/inst/SIMD", "Ginst/s", "ms");
    for (auto &r : all) printf("%-52s %3d %14.3f %14.3f %12.1f %9.3f\n", opname[r.op], r.w, r.cyc_wave, r.cyc_simd, r.ginst, r.ms);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"rows\": [\n", prop.name, ncu, prop.clockRate);
        for (size_t i = 0; i < all.size(); i++)
            fprintf(f, "  {\"op\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_inst_wave\": %.4f, \"cyc_per_inst_simd\": %.4f, \"ginst_per_s\": %.2f, \"ms\": %.4f}%s\n",
                    opname[all[i].op], all[i].w, all[i].cyc_wave, all[i].cyc_simd, all[i].ginst, all[i].ms, i + 1 < all.size() ? "," : "");
        fprintf(f, "]}\n"); fclose(f);
    }
    return 0;
}
