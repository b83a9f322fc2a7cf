/* A plain C caller of libplenvec.so through include/plenvec.h only (no Python, no ctypes): create -> reset -> step x N -> step2 -> destroy,
 * and the same through plenvec_create_from_model with the default model.  Device buffers come from the HIP runtime's C API.
 * Built and run by tests/test_cabi_gpu.py (gcc -std=c99 tests/cabi_driver.c -Iinclude -L... -lplenvec -lamdhip64); exit code 0 = every check passed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <hip/hip_runtime_api.h>
#include "plenvec.h"

#define CK(x) do { int rc_ = (x); if (rc_ != PLENVEC_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, plenvec_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
    enum { N = 8, STEPS = 12 };
    PlenCfg cfg; PlenModel model;
    CK(plenvec_default_cfg(&cfg, 0));
    cfg.dtype = PLENVEC_DTYPE_F64;
    CK(plenvec_default_model(&model));
    plenvec_t *h[2] = {0, 0};
    CK(plenvec_create(&cfg, N, 0, &h[0]));
    CK(plenvec_create_from_model(&model, &cfg, N, 0, &h[1]));
    if (plenvec_num_envs(h[0]) != N || plenvec_dtype(h[1]) != PLENVEC_DTYPE_F64) return 3;
    float *act; double *obs[2], *rew[2], *cur[2]; uint8_t *done[2], *trunc;
    HK(hipMalloc((void **)&act, N * PLENVEC_ACT * sizeof(float)));
    HK(hipMalloc((void **)&trunc, N));
    for (int k = 0; k < 2; k++) {
        HK(hipMalloc((void **)&obs[k], N * PLENVEC_OBS * sizeof(double))); HK(hipMalloc((void **)&cur[k], N * PLENVEC_OBS * sizeof(double)));
        HK(hipMalloc((void **)&rew[k], N * sizeof(double))); HK(hipMalloc((void **)&done[k], N));
    }
    float a_host[N * PLENVEC_ACT];
    double o0[N * PLENVEC_OBS], o1[N * PLENVEC_OBS], r0[N], r1[N];
    uint8_t d0[N], d1[N], t1[N];
    unsigned s = 12345u;
    for (int k = 0; k < 2; k++) CK(plenvec_reset(h[k], NULL, cur[k], NULL));
    for (int t = 0; t < STEPS; t++) {
        for (int i = 0; i < N * PLENVEC_ACT; i++) { s = s * 1664525u + 1013904223u; a_host[i] = (float)((s >> 8) * (2.0 / 16777216.0) - 1.0); }
        HK(hipMemcpy(act, a_host, sizeof a_host, hipMemcpyHostToDevice));
        CK(plenvec_step(h[0], act, obs[0], rew[0], done[0], cur[0], NULL));
        CK(plenvec_step2(h[1], act, obs[1], rew[1], done[1], trunc, cur[1], NULL));
        HK(hipDeviceSynchronize());
        HK(hipMemcpy(o0, obs[0], sizeof o0, hipMemcpyDeviceToHost)); HK(hipMemcpy(o1, obs[1], sizeof o1, hipMemcpyDeviceToHost));
        HK(hipMemcpy(r0, rew[0], sizeof r0, hipMemcpyDeviceToHost)); HK(hipMemcpy(r1, rew[1], sizeof r1, hipMemcpyDeviceToHost));
        HK(hipMemcpy(d0, done[0], N, hipMemcpyDeviceToHost)); HK(hipMemcpy(d1, done[1], N, hipMemcpyDeviceToHost)); HK(hipMemcpy(t1, trunc, N, hipMemcpyDeviceToHost));
        if (memcmp(o0, o1, sizeof o0) || memcmp(r0, r1, sizeof r0)) { fprintf(stderr, "step %d: create_from_model(default) differs from create\n", t); return 4; }
        for (int e = 0; e < N; e++) {
            const int term = (d0[e] & PLENVEC_DONE_TERMINAL) && !(d0[e] & PLENVEC_DONE_TIMELIMIT), tl = (d0[e] & PLENVEC_DONE_TIMELIMIT) != 0;
            if (d1[e] != term || t1[e] != tl) { fprintf(stderr, "step %d env %d: step2 pair (%d, %d) vs bits %d\n", t, e, d1[e], t1[e], d0[e]); return 5; }
            for (int c = 0; c < PLENVEC_OBS; c++) if (!isfinite(o0[e * PLENVEC_OBS + c])) return 6;
        }
    }
    /* a model that is not the PLEN tree is refused */
    PlenModel bad = model; bad.parent[7] = 6; plenvec_t *hb = 0;
    if (plenvec_create_from_model(&bad, &cfg, N, 0, &hb) != PLENVEC_E_INVAL || hb != NULL) return 7;
    printf("cabi_driver ok: %d envs x %d steps, torso z of env 0 = %.6f, version %s\n", N, STEPS, o0[18], plenvec_version());
    CK(plenvec_destroy(h[0])); CK(plenvec_destroy(h[1]));
    return 0;
}
