// Accuracy of the f64 reciprocal square roots the env kernel uses (plenvec.hip rsqrt_, pgs_cone) against 1 / sqrt(x) evaluated in long double on the host:
// v_rsq_f64 alone, + two Newton steps (rounds 2-4), + one third-order step (round 5), and the cone projection's fused  lm * x^-1/2.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/rsqrt_acc scripts/ubench/rsqrt_acc.hip     run: scripts/ubench/rsqrt_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double *x, const double *lm, double *o0, double *o1, double *o2, double *o3, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double y0 = __builtin_amdgcn_rsq(v);
    o0[i] = y0;
    {
        const double h = 0.5 * v;
        double y = y0;
        y = y * __builtin_fma(-(h * y), y, 1.5);
        o1[i] = y * __builtin_fma(-(h * y), y, 1.5);
    }
    {
        const double r = __builtin_fma(-(v * y0), y0, 1.0);
        o2[i] = __builtin_fma(y0 * r, __builtin_fma(r, 0.375, 0.5), y0);
    }
    {
        const double a_ = v * y0, ly0 = lm[i] * y0;
        const double r_ = __builtin_fma(-a_, y0, 1.0);
        o3[i] = __builtin_fma(ly0 * r_, __builtin_fma(r_, 0.375, 0.5), ly0);
    }
}
static double ulps(double got, long double want) {
    const double w = (double)want;
    const double u = std::nextafter(std::fabs(w), INFINITY) - std::fabs(w);
    return (double)(fabsl((long double)got - want) / u);
}
int main() {
    const int n = 1 << 20;
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> e(-30.0, 30.0), m(0.0, 4.0);
    std::vector<double> x(n), lm(n), o[4];
    for (int i = 0; i < n; i++) { x[i] = std::exp(e(g)); lm[i] = m(g); }
    double *dx, *dl, *d[4];
    hipMalloc(&dx, n * 8); hipMalloc(&dl, n * 8);
    for (auto &p : d) hipMalloc(&p, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dl, lm.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dl, d[0], d[1], d[2], d[3], n);
    for (int j = 0; j < 4; j++) { o[j].resize(n); hipMemcpy(o[j].data(), d[j], n * 8, hipMemcpyDeviceToHost); }
    const char *name[4] = {"v_rsq_f64", "+ two Newton steps", "+ one third-order step", "lm * x^-1/2, fused third-order step"};
    for (int j = 0; j < 4; j++) {
        double worst = 0, sum = 0;
        for (int i = 0; i < n; i++) {
            const long double want = (j == 3 ? (long double)lm[i] : 1.0L) / sqrtl((long double)x[i]);
            const double u = ulps(o[j][i], want);
            worst = std::max(worst, u); sum += u;
        }
        printf("%-40s max %.3g ulp, mean %.3g ulp over %d arguments in e^[-30, 30]\n", name[j], worst, sum / n, n);
    }
    return 0;
}
