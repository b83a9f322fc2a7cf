// accuracy of candidate f32 sin/cos evaluations on [-3.2, 3.2] against double (standalone: hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ inline void sincos_poly(float x, float &s, float &c) {
    // |x| <= pi: fold to [-pi/2, pi/2] by sin(pi - x) = sin x, cos(pi - x) = -cos x, then odd/even minimax-like Taylor-Horner
    const float PI = 3.14159265358979f;
    float sign_c = 1.0f;
    if (x > 1.5707963267949f) { x = PI - x; sign_c = -1.0f; }
    else if (x < -1.5707963267949f) { x = -PI - x; sign_c = -1.0f; }
    const float z = x * x;
    float ps = -2.5052108e-8f; ps = fmaf(ps, z, 2.7557319e-6f); ps = fmaf(ps, z, -1.9841270e-4f); ps = fmaf(ps, z, 8.3333333e-3f);
    ps = fmaf(ps, z, -1.6666667e-1f); s = fmaf(ps * z, x, x);
    float pc = 2.0876757e-9f; pc = fmaf(pc, z, -2.7557319e-7f); pc = fmaf(pc, z, 2.4801587e-5f); pc = fmaf(pc, z, -1.3888889e-3f);
    pc = fmaf(pc, z, 4.1666667e-2f); pc = fmaf(pc, z, -0.5f); c = sign_c * fmaf(pc, z, 1.0f);
}
__global__ void k(int n, const float *x, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    float v = x[i];
    out[6 * i] = sinf(v); out[6 * i + 1] = cosf(v);
    out[6 * i + 2] = __sinf(v); out[6 * i + 3] = __cosf(v);
    float s, c; sincos_poly(v, s, c); out[6 * i + 4] = s; out[6 * i + 5] = c;
}
int main() {
    const int n = 1 << 20; std::vector<float> hx(n), ho(6 * n);
    for (int i = 0; i < n; i++) hx[i] = -3.2f + 6.4f * (float)i / (float)(n - 1);
    float *dx, *dout; hipMalloc(&dx, n * 4); hipMalloc(&dout, 6 * n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(n, dx, dout); hipMemcpy(ho.data(), dout, 6 * n * 4, hipMemcpyDeviceToHost);
    double e[6] = {0};
    for (int i = 0; i < n; i++) { double s = sin((double)hx[i]), c = cos((double)hx[i]);
        if (fabs(hx[i]) > 3.1415f) continue;
        for (int j = 0; j < 6; j++) { double r = (j & 1) ? c : s; e[j] = fmax(e[j], fabs((double)ho[6 * i + j] - r)); } }
    printf("max abs err on [-pi,pi]: sinf %.3g cosf %.3g | __sinf %.3g __cosf %.3g | poly sin %.3g cos %.3g\n", e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}
