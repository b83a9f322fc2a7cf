// td3_kernels.hip -- hand-written HIP kernels (gfx950) for the non-GEMM work of one TD3 iteration.
//
// The reference's TD3Agent.train (plen_ros/src/plen_ros_helpers/td3.py:259-356) is ~170 small library kernels per iteration when
// written with autograd (elementwise clamps, adds, fills, reductions, index ops, one launch each, ~5 us apiece on an MI355X: more GPU
// time than the GEMMs).  plen_ml_walk_amd/td3_fused.py runs the same arithmetic with a hand-derived backward pass: the dense layers stay
// library GEMMs (rocBLAS / hipBLASLt through torch.mm / addmm), everything between them is fused into the kernels below, each one
// coalesced pass over its operands.  C ABI (include/plentd3.h): raw device pointers, sizes and a hipStream_t; no torch types.
//
// Layout conventions: activations are row-major [B][n]; a "strided" operand has an explicit row stride (it is a column slice of a wider
// row-major matrix, e.g. the action columns 26..43 of a [B][44] state-action matrix, or one column of the packed replay rows).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <stdint.h>
#include "../../include/plentd3.h"

#define TD3_H 256          // hidden width of actor and critics (td3.py:34-36, 80-88)

// ---- counter-based random numbers (Philox4x32-10, Salmon et al. 2011) so that the captured graphs need no library RNG call (each costs a
//      draw kernel plus a fill of the generator's offset tensor per graph replay).  rng = device uint64[2] {seed, calls so far}; a kernel reads
//      the call counter at its start and a LATER kernel of the same stream bumps it, so every launch sees one consistent value.
//      Element i of draw `tag` of call c gets the 4 words philox(counter = (i, c_lo, c_hi, tag), key = seed).
struct u4 { uint32_t x, y, z, w; };
static __device__ __forceinline__ u4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u4{c0, c1, c2, c3};
}
static __device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }              // [0, 1)
static __device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); } // (0, 1)
// two independent standard normals from two words (Box-Muller)
static __device__ __forceinline__ void normal2(uint32_t a, uint32_t b, float &n0, float &n1) {
    const float r = sqrtf(-2.0f * logf(u01_open(a))), th = 6.283185307179586f * u01(b);
    n0 = r * cosf(th); n1 = r * sinf(th);
}
// standard normal number e (0..) of draw `tag` of this call
static __device__ __forceinline__ float rng_normal(const uint64_t *rng, uint32_t tag, uint32_t e) {
    const uint64_t seed = rng[0], call = rng[1];
    const u4 w = philox4x32(e >> 2, (uint32_t)call, (uint32_t)(call >> 32), tag, (uint32_t)seed, (uint32_t)(seed >> 32));
    float n0, n1, n2, n3;
    normal2(w.x, w.y, n0, n1); normal2(w.z, w.w, n2, n3);
    const uint32_t k = e & 3u;
    return k == 0 ? n0 : k == 1 ? n1 : k == 2 ? n2 : n3;
}
static __device__ __forceinline__ float rng_uniform(const uint64_t *rng, uint32_t tag, uint32_t e) {
    const uint64_t seed = rng[0], call = rng[1];
    const u4 w = philox4x32(e >> 2, (uint32_t)call, (uint32_t)(call >> 32), tag, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t k = e & 3u;
    return u01(k == 0 ? w.x : k == 1 ? w.y : k == 2 ? w.z : w.w);
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// A workgroup hands three partial sums to whichever workgroup gets here last (called by ONE lane): parked, made visible, counted; true for the last one.
// TD3_LIGHT_HANDOFF (default): the partials leave as write-through (agent-scope) stores and are waited for before the count -- what the last workgroup reads (with
// agent-scope loads, after its acquire) has reached the agent's coherence point.  The textbook form (0: plain stores, __threadfence(), atomicAdd) makes the same three
// words visible by a RELEASE FENCE, which on this chip is `buffer_wbl2 sc1`: a write-back of every dirty line the XCD's L2 holds -- here the kernel's own activations
// and gradients on their way to the weight-gradient kernel, megabytes at batch 4096.  Phase stamps of k_critic_block's workgroup 0: 17.2 k of 132.7 k shader cycles
// in the closing phase with the fence, 5.5 k without (profiles/r06_t_critic_block_variants.txt).
#ifndef TD3_LIGHT_HANDOFF
#define TD3_LIGHT_HANDOFF 1
#endif
static __device__ __forceinline__ bool handoff_last(float *park, float a, float b, float c, int *done_count, int n_workgroups) {
#if TD3_LIGHT_HANDOFF
    __hip_atomic_store(park + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(park + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(park + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(done_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_workgroups - 1;
#else
    park[0] = a; park[1] = b; park[2] = c;
    __threadfence();
    return atomicAdd(done_count, 1) == n_workgroups - 1;
#endif
}
// the last workgroup, before it reads the others' partials (agent-scope loads): an acquire -- an invalidate, not another write-back
static __device__ __forceinline__ void handoff_acquire() {
#if TD3_LIGHT_HANDOFF
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    __threadfence();
#endif
}

// ---- K1: gather sampled replay rows (td3.py:166-193 ReplayBuffer.sample): out[b][:] = data[idx[b]][:] (row = s26 | a18 | s2_26 | r | not_done),
//      and the state part again into the policy pass's state-action matrix sa_pi[b][0:26].  Also zeroes the loss accumulator.
__global__ void k_gather(const float *__restrict__ data, const int64_t *__restrict__ idx, float *__restrict__ out, float *__restrict__ sa_pi, float *loss, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && loss) { loss[0] = 0.f; loss[1] = 0.f; }
    const int b = t / TD3_ROW, c = t % TD3_ROW;
    if (b >= B) return;
    const float v = data[(size_t)idx[b] * TD3_ROW + c];
    out[(size_t)b * TD3_ROW + c] = v;
    if (sa_pi && c < TD3_S) sa_pi[(size_t)b * TD3_SA + c] = v;
}

// ---- K2: target policy smoothing (td3.py:299-304): a2 = clamp(max_a * tanh(pre) + clamp(noise * sigma, +-clip), +-max_a); writes the
//      target critics' input sa2 = [s2 | a2] (s2 taken from the gathered batch rows)
__global__ void k_target_action(const float *__restrict__ pre, const float *__restrict__ noise, const uint64_t *__restrict__ rng, const float *__restrict__ batch,
                                float *__restrict__ sa2, float sigma, float clip, float max_a, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / TD3_SA, c = t % TD3_SA;
    if (b >= B) return;
    float v;
    if (c < TD3_S) v = batch[(size_t)b * TD3_ROW + TD3_S + TD3_A + c];
    else {
        const int j = c - TD3_S;
        const float z = noise ? noise[(size_t)b * TD3_A + j] : rng_normal(rng, 1u, (uint32_t)(b * TD3_A + j));      // torch.randn_like(action), td3.py:300
        const float n = fminf(fmaxf(z * sigma, -clip), clip);
        v = fminf(fmaxf(max_a * tanhf(pre[(size_t)b * TD3_A + j]) + n, -max_a), max_a);
    }
    sa2[(size_t)b * TD3_SA + c] = v;
}

// ---- K3 / K4: the twin critics' last layer (256 -> 1 each) as one wave per batch row over h2 = [h2_a | h2_b] ([B][512], post-ReLU):
//      q_c = h2_c . w3_c + b3_c.
//   mode 0 (targets, td3.py:306-309): y[b] = r + not_done * gamma * min(q_a, q_b)
//   mode 1 (critic loss, td3.py:312-319): dq[b][c] = 2 (q_c - y[b]) / B  (d loss / d q_c for loss = mse(q_a,y) + mse(q_b,y));
//                                         loss[0] += sum_c (q_c - y)^2 / B;  db3_c += dq (bias gradient of the last layer)
__global__ void k_q_heads(const float *__restrict__ h2, const float *__restrict__ w3a, const float *__restrict__ b3a, const float *__restrict__ w3b,
                          const float *__restrict__ b3b, const float *__restrict__ batch, float *__restrict__ y, float *__restrict__ dq, float *loss,
                          float *db3a, float *db3b, uint64_t *rng_bump, float gamma, int B, int mode) {
    if (rng_bump && blockIdx.x == 0 && threadIdx.x == 0) rng_bump[1] += 1;        // this update's draws (sampling, smoothing noise) are done
    // a wave reads a whole row pair [h2_a | h2_b] = 512 floats as two float4 per lane (lanes 0-31: critic a, 32-63: critic b); 4 rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane >> 5, l = lane & 31;
    const float4 wv0 = *reinterpret_cast<const float4 *>((c ? w3b : w3a) + 4 * l), wv1 = *reinterpret_cast<const float4 *>((c ? w3b : w3a) + 128 + 4 * l);
    const float bias = c ? b3b[0] : b3a[0];
    float lsum = 0.f, gsum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int b = (blockIdx.x * 4 + wave) * 4 + r;
        if (b >= B) break;
        const float *row = h2 + (size_t)b * (2 * TD3_H) + c * TD3_H;
        const float4 h0 = *reinterpret_cast<const float4 *>(row + 4 * l), h1 = *reinterpret_cast<const float4 *>(row + 128 + 4 * l);
        float sacc = h0.x * wv0.x + h0.y * wv0.y + h0.z * wv0.z + h0.w * wv0.w + h1.x * wv1.x + h1.y * wv1.y + h1.z * wv1.z + h1.w * wv1.w;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);              // within each 32-lane half
        const float q = sacc + bias;                                               // q_a in lanes 0-31, q_b in lanes 32-63
        const float qo = __shfl_xor(q, 32);
        if (mode == 0) {
            if (lane == 0) y[b] = batch[(size_t)b * TD3_ROW + TD3_ROW - 2] + batch[(size_t)b * TD3_ROW + TD3_ROW - 1] * gamma * fminf(q, qo);
        } else {
            const float e = q - y[b], inv = 1.f / (float)B;
            if (l == 0) { dq[2 * b + c] = 2.f * e * inv; lsum += e * e * inv; gsum += 2.f * e * inv; }
        }
    }
    if (mode == 1) {             // one atomic per workgroup and quantity
        __shared__ float red[4][3];
        if (lane == 0) { red[wave][0] = lsum; red[wave][1] = gsum; }
        __syncthreads();
        if (lane == 32) { red[wave][0] += lsum; red[wave][2] = gsum; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float s0 = 0, s1 = 0, s2 = 0;
            for (int w = 0; w < 4; w++) { s0 += red[w][0]; s1 += red[w][1]; s2 += red[w][2]; }
            atomicAdd(loss, s0); atomicAdd(db3a, s1); atomicAdd(db3b, s2);
        }
    }
}

// ---- K5: back through the last layer and the ReLU before it:  dh2[b][c*256 + j] = dq[b][c] * w3_c[j] * (h2[b][c*256 + j] > 0)
//      dq == nullptr: the policy pass (td3.py:337 actor_loss = -Q1(s, pi(s)).mean()): dq = -1/B for every row, critic a only (ncrit = 1)
__global__ void k_dh2(const float *__restrict__ dq, const float *__restrict__ w3a, const float *__restrict__ w3b, const float *__restrict__ h2,
                      float *__restrict__ dh2, int B, int ncrit, int h2_stride) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = ncrit * TD3_H;
    const int b = t / W, j = t % W;
    if (b >= B) return;
    const int c = j / TD3_H, jj = j % TD3_H;
    const float g = dq ? dq[2 * b + c] : -1.f / (float)B;
    const float h = h2[(size_t)b * h2_stride + j];
    dh2[(size_t)b * W + j] = h > 0.f ? g * (c ? w3b[jj] : w3a[jj]) : 0.f;
}

// ---- K6: g *= (h > 0) in place (ReLU backward); g [B][n] contiguous, h has row stride hs
__global__ void k_relu_mask(float *__restrict__ g, const float *__restrict__ h, int B, int n, int hs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / n, j = t % n;
    if (b >= B) return;
    if (!(h[(size_t)b * hs + j] > 0.f)) g[(size_t)b * n + j] = 0.f;
}

// ---- K7: column sums over the batch (bias gradients):  out[j] += sum_b w[b] * g[b][j]   (w == nullptr: plain sum).  With w = dq column c and
//      g = h2_c this is also the last layer's weight gradient dW3_c = h2_c^T dq_c (td3.py:323 critic_loss.backward()).
//      Workgroup = 64 columns x 4 row-groups; grid.y splits the batch; one atomic per column and workgroup.  out must be zeroed beforehand.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_colsum(const float *__restrict__ g, int gs, const float *__restrict__ w, int ws, float *__restrict__ out, int B, int n) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int rows_per = (B + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(B, r0 + rows_per);
    float s = 0.f;
    if (col < n)
        for (int b = r0 + rg; b < r1; b += NW) s += (w ? w[(size_t)b * ws] : 1.f) * g[(size_t)b * gs + col];
    if constexpr (NW > 1) {
        __shared__ float red[NW][64];
        red[rg][threadIdx.x & 63] = s;
        __syncthreads();
        if (rg == 0 && col < n) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < NW; k++) t += red[k][threadIdx.x];
            atomicAdd(out + col, t);
        }
    } else if (col < n) atomicAdd(out + col, s);            // single-wave workgroup: no LDS, starts in any one free wave slot
}

// ---- K8: the actor's output nonlinearity (td3.py:57): a = max_a * tanh(pre), stored densely and into the action columns of sa_pi
__global__ void k_tanh_out(const float *__restrict__ pre, float *__restrict__ a, float *__restrict__ sa_pi, float max_a, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / TD3_A, j = t % TD3_A;
    if (b >= B) return;
    const float v = max_a * tanhf(pre[t]);
    a[t] = v;
    sa_pi[(size_t)b * TD3_SA + TD3_S + j] = v;
}

// ---- K9: back through it:  dz[b][j] = dsa[b][26 + j] * (max_a - a^2 / max_a)        (d/dx max_a tanh x = max_a (1 - tanh^2 x))
__global__ void k_dtanh(const float *__restrict__ dsa, const float *__restrict__ a, float *__restrict__ dz, float max_a, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / TD3_A, j = t % TD3_A;
    if (b >= B) return;
    const float v = a[t];
    dz[t] = dsa[(size_t)b * TD3_SA + TD3_S + j] * (max_a - v * v / max_a);
}

// ---- K10: h = relu(h + bias) in place (used when the GEMM library offers no fused bias+ReLU epilogue)
__global__ void k_bias_relu(float *__restrict__ h, const float *__restrict__ bias, int B, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n) return;
    h[t] = fmaxf(h[t] + bias[t % n], 0.f);
}

// ---- K11: Polyak averaging of a whole flat parameter buffer (td3.py:348-356):  t = tau * p + (1 - tau) * t
__global__ void k_polyak(float *__restrict__ t, const float *__restrict__ p, float tau, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = tau * p[i] + (1.f - tau) * t[i];
}

// ---- K15: weight gradient of a dense layer on the matrix cores:  dW[n][k] += sum_b dH[b][n] X[b][k],  db[n] += sum_b dH[b][n]
//      (td3.py:323 / :341 .backward() of nn.Linear).  The library GEMM for this shape -- a 4096-long reduction into a 256 x 256 (or smaller)
//      result -- takes 24.5 us whatever the size (scripts/gpu_gemm_probe.py): too few output tiles to fill 256 CUs and no split over the
//      reduction.  Here the batch is split into chunks: one wave per (32 x 32 output tile, chunk), v_mfma_f32_32x32x2_f32 with both operands
//      straight from global memory -- lane l feeds A[i = l%32][kk = l/32] = dH[b + l/32][n0 + l%32] and B[kk][j = l%32] = X[b + l/32][k0 + l%32],
//      two coalesced 128-byte row segments per operand and step, no transpose, no LDS -- and the partial tiles are added into dW (zeroed by
//      the caller) with float atomics.  Waves of the first column tile also sum their dH operand: the bias gradient comes for free.
typedef float floatx16 __attribute__((ext_vector_type(16)));
// where a lane's 16 results of a 32 x 32 tile sit: element v = row n0 + 8 (v / 4) + 4 half + v % 4, column k0 + col (wgrad_tile's epilogue)
struct WgradTileAt { int n0, k0, col, half; };
struct WgradAtomicAdd {        // the plain epilogue: partial tiles are added into the (zeroed) gradient bucket
    __device__ __forceinline__ void tile(float *const (&dst)[16], const float (&v)[16], const WgradTileAt &) const {
#pragma unroll
        for (int k = 0; k < 16; k++) if (dst[k]) atomicAdd(dst[k], v[k]);
    }
    __device__ __forceinline__ void one(float *dst, float v) const { atomicAdd(dst, v); }
};
template <int NW, class Epilogue = WgradAtomicAdd>
static __device__ __forceinline__ void wgrad_tile(const float *__restrict__ dH, int ds, const float *__restrict__ X, int xs, float *__restrict__ dW, int dws,
                                                  float *__restrict__ db, int N, int K, int tile, int b0, int b1, const Epilogue &emit = Epilogue()) {
    // workgroup = one (32 x 32 output tile, batch chunk).  NW = 4: its waves take every 4th pair of batch rows and are summed through LDS, so that a
    // tile costs one set of 1024 atomics per chunk instead of four.  NW = 1: one wave, no LDS -- a workgroup that starts in any single free wave
    // slot (the update beside resident env launches, td3_rows.hip), at the price of shorter chunks' worth of atomics.
    __shared__ float red[NW > 1 ? NW - 1 : 1][NW > 1 ? 64 : 1][NW > 1 ? 17 : 1];
    __shared__ float bred[NW > 1 ? NW - 1 : 1][NW > 1 ? 32 : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kt = (K + 31) / 32;
    const int n0 = (tile / kt) * 32, k0 = (tile % kt) * 32;
    const int col = lane & 31, half = lane >> 5;
    const bool na = n0 + col < N, ka = k0 + col < K;
    floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float bsum = 0.f;
    // wave w: row pairs b0 + 2 (NW u + w), in groups of 8 pairs (16 loads in flight before the first MFMA needs its operands)
    int b = b0 + 2 * w;
#pragma unroll 1
    for (; b + 2 * NW * 7 + 1 < b1; b += 2 * NW * 8) {
        float av[8], bv[8];
        const float *pa = dH + (size_t)(b + half) * ds + n0 + col, *pb = X + (size_t)(b + half) * xs + k0 + col;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            av[u] = na ? pa[(size_t)(2 * NW * u) * ds] : 0.f;
            bv[u] = ka ? pb[(size_t)(2 * NW * u) * xs] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0); bsum += av[u]; }
    }
    if (b < b1) {                                     // ragged tail (also an odd last row): one more group of 8 pairs, rows past the end read as zero --
        float av[8], bv[8];                           // its loads all in flight together (a batch of 100 is 36 tail rows: row by row, each one's trip
#pragma unroll                                        // to memory was exposed and the tail took longer than everything else in the kernel)
        for (int u = 0; u < 8; u++) {
            const int row = b + 2 * NW * u + half;
            const bool ra = row < b1;
            av[u] = (na && ra) ? dH[(size_t)row * ds + n0 + col] : 0.f;
            bv[u] = (ka && ra) ? X[(size_t)row * xs + k0 + col] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0); bsum += av[u]; }
    }
    bsum += __shfl_xor(bsum, 32);
    if constexpr (NW > 1) {
        if (w > 0) {
#pragma unroll
            for (int v = 0; v < 16; v++) red[w - 1][lane][v] = acc[v];
            if (half == 0) bred[w - 1][col] = bsum;
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int k = 0; k < NW - 1; k++) {
#pragma unroll
                for (int v = 0; v < 16; v++) acc[v] += red[k][lane][v];
                bsum += bred[k][col];
            }
        }
    }
    if (w == 0) {
        // result layout of the 32x32 MFMA: lane l holds column j = l % 32 and rows i = 8 * (v / 4) + 4 * (l / 32) + v % 4, v = 0..15
        float *dst[16];
        float val[16];
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int i = 8 * (v / 4) + 4 * half + (v % 4);
            dst[v] = (ka && n0 + i < N) ? dW + (size_t)(n0 + i) * dws + k0 + col : nullptr;
            val[v] = acc[v];
        }
        emit.tile(dst, val, WgradTileAt{n0, k0, col, half});
        if (db && k0 == 0 && half == 0 && na) emit.one(db + n0 + col, bsum);
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_wgrad(const float *__restrict__ dH, int ds, const float *__restrict__ X, int xs, float *__restrict__ dW, int dws,
                                               float *__restrict__ db, int B, int N, int K, int rows_per_chunk) {
    const int tiles = ((N + 31) / 32) * ((K + 31) / 32);
    const int tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles;
    const int b0 = chunk * rows_per_chunk;
    wgrad_tile<NW>(dH, ds, X, xs, dW, dws, db, N, K, tile, b0, min(B, b0 + rows_per_chunk));
}

// ---- K15b: every weight gradient of one backward pass in ONE launch (small batches: the reduction over the batch is a single chunk, so each output
//      tile is written by exactly one workgroup -- the atomics into the zeroed bucket are plain stores in effect, and the result is deterministic).
//      At batch 100 the five launches of a critic update (two head rows, two second layers, the stacked first layers) are 5 x 5 us of launch latency
//      for < 1 us of work each; grouped they are one.  job[j] owns workgroups [tile0[j], tile0[j + 1]).
__global__ __launch_bounds__(256) void k_wgrad_group(PlenTd3WgradGroup G) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < PLENTD3_WGRAD_JOBS; k++) j += (k < G.n_jobs && (int)blockIdx.x >= G.job[k].tile0) ? 1 : 0;
    const PlenTd3WgradJob &J = G.job[j];
    wgrad_tile<4>(J.dH, J.ds, J.X, J.xs, J.dW, J.dws, J.db, J.N, J.K, (int)blockIdx.x - J.tile0, 0, G.B);
}

// ---- K12: replay sampling without a host round trip (td3.py:175 np.random.randint(0, len, B)): u uniform in [0, 1) -> a row of the ring that
//      holds a complete transition, then the gather of K1.  *total = transitions written so far (may exceed the capacity: the ring wraps);
//      guard = rows after position *total that concurrent writers may be filling right now (0 for a synchronous loop): they are excluded
//      once the ring has wrapped onto them.
__global__ void k_sample_gather(const float *__restrict__ data, const float *__restrict__ u, const uint64_t *__restrict__ rng, const int64_t *__restrict__ total,
                                int64_t capacity, int64_t guard, int64_t *__restrict__ idx_out, float *__restrict__ out, float *__restrict__ sa_pi, float *loss, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && loss) { loss[0] = 0.f; loss[1] = 0.f; }
    const int b = t / TD3_ROW, c = t % TD3_ROW;
    if (b >= B) return;
    const int64_t tot = total[0];
    int64_t filled, start;
    if (tot + guard <= capacity) { filled = tot; start = 0; }                               // ring not yet wrapped onto the rows in flight
    else { filled = capacity - guard; start = (tot + guard) % capacity; }                   // the oldest complete row follows the rows in flight
    const float ub = u ? u[b] : rng_uniform(rng, 0u, (uint32_t)b);
    int64_t i = (int64_t)((double)ub * (double)filled);
    i = i < filled - 1 ? i : filled - 1;
    i = i > 0 ? i : 0;
    i = (start + i) % capacity;
    if (c == 0 && idx_out) idx_out[b] = i;
    const float v = data[(size_t)i * TD3_ROW + c];
    out[(size_t)b * TD3_ROW + c] = v;
    if (sa_pi && c < TD3_S) sa_pi[(size_t)b * TD3_SA + c] = v;
}

// ---- K13: exploration action of the collect phase (plen_td3.py:101-104): a = clamp(max_a tanh(pre) + noise * sigma, +-max_a)
__global__ void k_explore(const float *__restrict__ pre, const float *__restrict__ noise, const uint64_t *__restrict__ rng, float *__restrict__ a, float sigma, float max_a, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = fminf(fmaxf(max_a * tanhf(pre[t]) + (noise ? noise[t] : rng_normal(rng, 2u, (uint32_t)t)) * sigma, -max_a), max_a);
}
// uniform random actions of the warm-up phase (plen_td3.py:91-92 env.action_space.sample()): a = U[-1, 1)
__global__ void k_uniform_actions(const uint64_t *__restrict__ rng, float *__restrict__ a, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = 2.0f * rng_uniform(rng, 3u, (uint32_t)t) - 1.0f;
}

// ---- K14: write one vector step into the replay ring (plen_td3.py:109-113): row (total + e) % capacity = s | a | s2 | r | 1 - done_bool,
//      done_bool = compute_done() fired AND the time limit did not (PLENVEC_DONE_TERMINAL = 1, _TIMELIMIT = 2)
__global__ void k_store(float *__restrict__ data, const int64_t *total, int64_t capacity, float *__restrict__ s, const float *__restrict__ a,
                        const float *__restrict__ s2, const float *__restrict__ r, const uint8_t *__restrict__ done, uint64_t *rng_bump,
                        float *__restrict__ ep_ret, double *stats, int n, const float *__restrict__ advance, int64_t *total_step, int64_t step, unsigned *blocks_done) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (rng_bump && t == 0) rng_bump[1] += 1;           // this collect step's action draw is done
    const int e = t / TD3_ROW, c = t % TD3_ROW;
    // plentd3_store_step: *total += step once every block has read it -- the last block to get here does it (each block's first thread counts itself in after the
    // block's reads; the counter is left at zero for the next launch).  Replaces the caller's one-element add kernel on its critical path.
    // (`total` and `total_step` are the SAME word in step mode: neither is restrict-qualified, the read is an atomic load -- not a load the compiler may treat as
    // invariant and sink past the barrier --, and every block's read is ordered before its count, the last block's store after the count it saw.  ADVICE r05.)
    const int64_t total0 = __hip_atomic_load(total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blocks_done) {
#if TD3_LIGHT_HANDOFF
        // what has to be ordered is a READ before the count (every thread's load of `total` has RETURNED before its block counts itself in) and the last block's store
        // after the count it saw (it is conditional on the atomic's result): no release fence -- here an L2 write-back per block, 576 of them per 2048-env step, beside
        // an update whose activations sit dirty in the same L2 (handoff_last above)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && __hip_atomic_fetch_add(blocks_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) { blocks_done[0] = 0; total_step[0] = total0 + step; }
#else
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(blocks_done, 1u) == gridDim.x - 1) { __threadfence(); blocks_done[0] = 0; total_step[0] = total0 + step; }
        }
#endif
    }
    if (e >= n) return;
    // episode bookkeeping at full speed (the reference prints every episode's return, plen_env.py:616-636): per-env running return; when the
    // episode ends (any done bit) its return and length go into stats = {sum of returns, episodes, sum of lengths} and the env starts over.
    // stats are DOUBLES (hardware f64 atomic add): at 6.5 M env-steps/s a float32 episode count saturates at 2^24 after two minutes and the
    // length sum stops absorbing increments within seconds
    if (ep_ret && c == 0) {
        const float ret = ep_ret[2 * e] + r[e], len = ep_ret[2 * e + 1] + 1.f;
        if (done[e]) { atomicAdd(stats, (double)ret); atomicAdd(stats + 1, 1.0); atomicAdd(stats + 2, (double)len); ep_ret[2 * e] = 0.f; ep_ret[2 * e + 1] = 0.f; }
        else { ep_ret[2 * e] = ret; ep_ret[2 * e + 1] = len; }
    }
    const int64_t row = (total0 + e) % capacity;
    float v;
    if (c < TD3_S) {
        v = s[(size_t)e * TD3_S + c];
        // plentd3_store_advance: the state to act on next (the env's current observation: the reset one where the episode ended) replaces the stored one, element by
        // element in the thread that has just read it -- the collector's `state = obs` copy launch, on its critical path every step, folded in
        if (advance) s[(size_t)e * TD3_S + c] = advance[(size_t)e * TD3_S + c];
    }
    else if (c < TD3_SA) v = a[(size_t)e * TD3_A + c - TD3_S];
    else if (c < TD3_SA + TD3_S) v = s2[(size_t)e * TD3_S + c - TD3_SA];
    else if (c == TD3_ROW - 2) v = r[e];
    else v = ((done[e] & 1) && !(done[e] & 2)) ? 0.f : 1.f;
    data[(size_t)row * TD3_ROW + c] = v;
}

// One Adam step of one element, and the Polyak update that may follow it: shared by k_adam and k_wgrad_adam_group, with the multiply-adds spelled out so that
// the two kernels round alike whatever the compiler would have contracted in each context.
// the hyper-parameters arrive as the doubles torch holds them in and are rounded where torch's kernel rounds them: 1 - beta2 = 0.001 computed
// in float would be off by 5e-5 relative, and so would the second moment
struct AdamCoef { float step_size, bc2_sqrt, b2, w1, w2, eps, tau; };
static __device__ __forceinline__ AdamCoef adam_coef(float t, double lr, double b1d, double b2d, float eps, float tau) {
    // beta^t for the integer-valued step count by repeated squaring (<= 24 double multiplies, a few ulp of double: the same float after rounding);
    // the general double-precision pow() cost 1 us per launch, twice, on every workgroup's critical path
    double p1 = 1.0, p2 = 1.0, q1 = b1d, q2 = b2d;
    for (unsigned n = (unsigned)t; n; n >>= 1) {
        if (n & 1u) { p1 *= q1; p2 *= q2; }
        q1 *= q1; q2 *= q2;
    }
    const double bc1 = 1.0 - p1, bc2 = 1.0 - p2;
    return AdamCoef{(float)(lr / bc1), (float)sqrt(bc2), (float)b2d, (float)(1.0 - b1d), (float)(1.0 - b2d), eps, tau};
}
static __device__ __forceinline__ void adam_one(float gi, float &mi, float &vi, float &pi, const AdamCoef &c) {
    mi = __fmaf_rn(gi - mi, c.w1, mi);
    vi = __fmaf_rn(__fmul_rn(c.w2, gi), gi, __fmul_rn(vi, c.b2));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), c.bc2_sqrt), c.eps);
    pi = __fsub_rn(pi, __fdiv_rn(__fmul_rn(c.step_size, mi), denom));
}
static __device__ __forceinline__ float adam_polyak(float pi, float ti, float tau) { return __fmaf_rn(tau, pi, __fmul_rn(1.f - tau, ti)); }

// ---- K17: one Adam step over a network's flat parameter / gradient buffers (torch.optim.Adam as td3.py:236-247 configures it: no weight decay, no
//      amsgrad), optionally with what follows it in the iteration: the gradient bucket zeroed for the next backward pass, the Polyak update of
//      the target network (td3.py:348-356) and a copy of the new parameters (the pipelined trainer's behaviour actor).  torch's fused Adam is two
//      multi-tensor launches of 41 + 5 us for these 12 tensors; this is one pass of single-wave workgroups.
//      step = float32 device scalar (torch's capturable `step`): read by every workgroup, advanced by the last one to finish.
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, float *step, int *done_count, int n,
                                              double lr, double b1d, double b2d, float eps, int zero_grad, float *__restrict__ target, float tau, float *__restrict__ copy_out) {
    // (at most 256 workgroups of 256 lanes, 4 consecutive floats per lane and trip: the done_count atomics of 600 single-wave workgroups, all on one
    // address, were most of this kernel's 20 us on the critic's 154 k parameters)
    const float t = step[0] + 1.f;
    const AdamCoef c = adam_coef(t, lr, b1d, b2d, eps, tau);
    auto one = [&](float gi, float &mi, float &vi, float &pi) { adam_one(gi, mi, vi, pi, c); };
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += gridDim.x * 1024) {
        if (i0 + 3 < n) {                       // four consecutive floats as one 16-byte access per array (the flat buffers are 16-byte aligned)
            const f4 g4 = *reinterpret_cast<const f4 *>(g + i0);
            f4 m4 = *reinterpret_cast<const f4 *>(m + i0), v4 = *reinterpret_cast<const f4 *>(v + i0), p4 = *reinterpret_cast<const f4 *>(p + i0), t4 = {0, 0, 0, 0};
            if (target) t4 = *reinterpret_cast<const f4 *>(target + i0);
#pragma unroll
            for (int j = 0; j < 4; j++) { float mi = m4[j], vi = v4[j], pi = p4[j]; one(g4[j], mi, vi, pi); m4[j] = mi; v4[j] = vi; p4[j] = pi; t4[j] = adam_polyak(pi, t4[j], tau); }
            *reinterpret_cast<f4 *>(m + i0) = m4; *reinterpret_cast<f4 *>(v + i0) = v4; *reinterpret_cast<f4 *>(p + i0) = p4;
            if (zero_grad) *reinterpret_cast<f4 *>(g + i0) = f4{0, 0, 0, 0};
            if (target) *reinterpret_cast<f4 *>(target + i0) = t4;
            if (copy_out) *reinterpret_cast<f4 *>(copy_out + i0) = p4;
        } else {
            for (int i = i0; i < n; i++) {
                float mi = m[i], vi = v[i], pi = p[i];
                one(g[i], mi, vi, pi);
                m[i] = mi; v[i] = vi; p[i] = pi;
                if (zero_grad) g[i] = 0.f;
                if (target) target[i] = adam_polyak(pi, target[i], tau);
                if (copy_out) copy_out[i] = pi;
            }
        }
    }
    __syncthreads();                            // every wave of this workgroup has read the counter (no fence: the next reader of `step` is the next launch)
    if (threadIdx.x == 0 && atomicAdd(done_count, 1) == (int)gridDim.x - 1) { done_count[0] = 0; step[0] = t; }
}

// ---- K15c + K17 in one: the grouped weight gradients of a SMALL batch with the Adam step applied where each gradient element is produced.
//      With the whole batch as one reduction chunk every element of every gradient is owned by exactly one workgroup (its 32 x 32 tile; the bias
//      gradients by the tiles of the first column block), so that workgroup can take the Adam step for it at once: the gradient never travels
//      through the bucket (which therefore stays zero for the next pass), and the iteration loses a launch and two dependent trips to memory.
//      dW / db of the jobs point INTO the flat gradient buffer Ad.g: an element's offset there is its offset in p, m, v and target.
//      Elements whose gradient is already in the bucket (the critics' head biases, summed by k_critic_team) are listed in extra_off: workgroup 0
//      steps them and zeroes them.  Same arithmetic per element as k_adam, same step counter protocol.
struct WgradAdamStep {
    const PlenTd3AdamFused &Ad; const AdamCoef &c;
    float *pack, *pack_t; int pack_ns;          // this job's weight matrix (and its Polyak target) in the small-batch kernels' operand order, or null
    // all of a lane's 16 elements: every load first (one trip to memory for the tile, not one per element), then the arithmetic, then the stores
    __device__ __forceinline__ void tile(float *const (&dst)[16], const float (&gv)[16], const WgradTileAt &at) const {
        float m[16], v[16], p[16], tg[16];
        size_t o[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            o[k] = dst[k] ? (size_t)(dst[k] - Ad.g) : 0;
            m[k] = Ad.m[o[k]]; v[k] = Ad.v[o[k]]; p[k] = Ad.p[o[k]]; tg[k] = Ad.target ? Ad.target[o[k]] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) adam_one(gv[k], m[k], v[k], p[k], c);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (dst[k]) {
                Ad.m[o[k]] = m[k]; Ad.v[o[k]] = v[k]; Ad.p[o[k]] = p[k];
                const float tn = Ad.target ? adam_polyak(p[k], tg[k], c.tau) : 0.f;
                if (Ad.target) Ad.target[o[k]] = tn;
                if (pack) {
                    // element (n, kk) of the matrix in team operand order (PlenTd3PackJob.team): float4 (((n / 32) NS + kk / 64) 8 + (kk % 32) / 4) 64 + 32 ((kk / 32) % 2) + n % 32, word kk % 4
                    // -- written where the parameter is, so that the next pass reads current weights without a packing launch on the chain of updates
                    const int n = at.n0 + 8 * (k / 4) + 4 * at.half + (k % 4), kk = at.k0 + at.col;
                    const size_t po = ((size_t)((n >> 5) * pack_ns + (kk >> 6)) * 8 + ((kk & 31) >> 2)) * 256 + (size_t)(32 * ((kk >> 5) & 1) + (n & 31)) * 4 + (kk & 3);
                    pack[po] = p[k];
                    if (pack_t && Ad.target) pack_t[po] = tn;
                }
            }
        }
    }
    __device__ __forceinline__ void one(float *dst, float gi) const {
        const size_t o = (size_t)(dst - Ad.g);
        float m = Ad.m[o], v = Ad.v[o], p = Ad.p[o];
        adam_one(gi, m, v, p, c);
        Ad.m[o] = m; Ad.v[o] = v; Ad.p[o] = p;
        if (Ad.target) Ad.target[o] = adam_polyak(p, Ad.target[o], c.tau);
    }
};
__global__ __launch_bounds__(256) void k_wgrad_adam_group(PlenTd3WgradGroup G, PlenTd3AdamFused Ad) {
    const float t = Ad.step_advanced ? Ad.step[0] : Ad.step[0] + 1.f;
    int j = 0;
#pragma unroll
    for (int k = 1; k < PLENTD3_WGRAD_JOBS; k++) j += (k < G.n_jobs && (int)blockIdx.x >= G.job[k].tile0) ? 1 : 0;
    const PlenTd3WgradJob &J = G.job[j];
    const AdamCoef c = adam_coef(t, Ad.lr, Ad.beta1, Ad.beta2, Ad.eps, Ad.tau);
    const WgradAdamStep emit{Ad, c, J.pack, J.pack_t, J.pack_ns};
    wgrad_tile<4, WgradAdamStep>(J.dH, J.ds, J.X, J.xs, J.dW, J.dws, J.db, J.N, J.K, (int)blockIdx.x - J.tile0, 0, G.B, emit);
    if (blockIdx.x == 0 && (int)threadIdx.x < Ad.n_extra) {
        float *ge = Ad.g + Ad.extra_off[threadIdx.x];
        emit.one(ge, *ge);
        *ge = 0.f;
    }
    // the step counter: read by every workgroup, so it may only change once all have read it -- the last workgroup to finish writes it (one returning atomic
    // per workgroup on one address: 2.4 us of tail per launch, 5 us with the device-scope fence a first version put in front of it), unless the pass
    // kernel that produced the gradients has advanced it already (step_advanced: the small-batch path)
    if (!Ad.step_advanced) {
        __syncthreads();                        // every wave of this workgroup has read the counter
        if (threadIdx.x == 0 && atomicAdd(Ad.done_count, 1) == (int)gridDim.x - 1) { Ad.done_count[0] = 0; Ad.step[0] = t; }
    }
}

#include "td3_rows.hip"
#include "td3_team.hip"
#include "td3_block.hip"

// ---- K16: timeline probe: slot[0] = the GPU's constant-rate clock (100 MHz) when this one-lane kernel runs.  A node in a captured graph like any
//      other, so the pipelined trainer's schedule can be read without a profiler serialising it.
//      Row (*counter / div) % ring of a [ring][nslots] table, column idx: counter is one of the trainer's device-side step counters, so replays of one
//      graph fill successive rows.
__global__ void k_stamp(uint64_t *table, const int64_t *counter, int64_t div, int ring, int nslots, int idx) {
    const int64_t row = counter ? (counter[0] / div) % ring : 0;
    table[row * nslots + idx] = wall_clock64();
}

#define GRID(n_) dim3(((n_) + 255) / 256), dim3(256), 0, (hipStream_t)stream
#define CHECK() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? 0 : -(int)e_; } while (0)

extern "C" {

int plentd3_gather(const float *data, const int64_t *idx, float *out, float *sa_pi, float *loss, int B, void *stream) {
    hipLaunchKernelGGL(k_gather, GRID(B * TD3_ROW), data, idx, out, sa_pi, loss, B); CHECK();
}
int plentd3_sample_gather(const float *data, const float *u, const uint64_t *rng, const int64_t *total, int64_t capacity, int64_t guard, int64_t *idx_out, float *out, float *sa_pi, float *loss, int B, void *stream) {
    hipLaunchKernelGGL(k_sample_gather, GRID(B * TD3_ROW), data, u, rng, total, capacity, guard, idx_out, out, sa_pi, loss, B); CHECK();
}
int plentd3_explore(const float *pre, const float *noise, const uint64_t *rng, float *a, float sigma, float max_a, int n, void *stream) {
    hipLaunchKernelGGL(k_explore, GRID(n), pre, noise, rng, a, sigma, max_a, n); CHECK();
}
int plentd3_uniform_actions(const uint64_t *rng, float *a, int n, void *stream) {
    hipLaunchKernelGGL(k_uniform_actions, GRID(n), rng, a, n); CHECK();
}
int plentd3_store(float *data, const int64_t *total, int64_t capacity, const float *s, const float *a, const float *s2, const float *r, const uint8_t *done, uint64_t *rng_bump,
                  float *ep_ret, double *stats, int n, void *stream) {
    hipLaunchKernelGGL(k_store, GRID(n * TD3_ROW), data, total, capacity, const_cast<float *>(s), a, s2, r, done, rng_bump, ep_ret, stats, n, (const float *)nullptr, (int64_t *)nullptr, (int64_t)0, (unsigned *)nullptr); CHECK();
}
int plentd3_store_advance(float *data, const int64_t *total, int64_t capacity, float *s, const float *a, const float *s2, const float *r, const uint8_t *done, uint64_t *rng_bump,
                          float *ep_ret, double *stats, int n, const float *next_state, void *stream) {
    if (!next_state || next_state == s) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_store, GRID(n * TD3_ROW), data, total, capacity, s, a, s2, r, done, rng_bump, ep_ret, stats, n, next_state, (int64_t *)nullptr, (int64_t)0, (unsigned *)nullptr); CHECK();
}
int plentd3_store_step(float *data, int64_t *total, int64_t capacity, float *s, const float *a, const float *s2, const float *r, const uint8_t *done, uint64_t *rng_bump,
                       float *ep_ret, double *stats, int n, const float *next_state, int64_t step, unsigned *blocks_done, void *stream) {
    if (!blocks_done || (next_state && next_state == s)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_store, GRID(n * TD3_ROW), data, total, capacity, s, a, s2, r, done, rng_bump, ep_ret, stats, n, next_state, total, step, blocks_done); CHECK();
}
int plentd3_wgrad(const float *dH, int dh_stride, const float *X, int x_stride, float *dW, int dw_stride, float *db, int B, int N, int K, int single_wave, void *stream) {
    const int tiles = ((N + 31) / 32) * ((K + 31) / 32);
    // batch chunks so that about 256-512 workgroups of 4 waves run, at least 256 rows per workgroup (single-wave workgroups: 1024, at least 128 rows)
    const int want = single_wave ? 1024 : 384, min_rows = single_wave ? 128 : 256;
    int chunks = (want + tiles - 1) / tiles;
    if (chunks > (B + min_rows - 1) / min_rows) chunks = (B + min_rows - 1) / min_rows;
    if (chunks < 1) chunks = 1;
    int rows = (B + chunks - 1) / chunks;
    rows = (rows + 63) / 64 * 64;
    chunks = (B + rows - 1) / rows;
    if (single_wave) hipLaunchKernelGGL(k_wgrad<1>, dim3(tiles * chunks), dim3(64), 0, (hipStream_t)stream, dH, dh_stride, X, x_stride, dW, dw_stride, db, B, N, K, rows);
    else hipLaunchKernelGGL(k_wgrad<4>, dim3(tiles * chunks), dim3(256), 0, (hipStream_t)stream, dH, dh_stride, X, x_stride, dW, dw_stride, db, B, N, K, rows);
    CHECK();
}
int plentd3_target_action(const float *pre, const float *noise, const uint64_t *rng, const float *batch, float *sa2, float sigma, float clip, float max_a, int B, void *stream) {
    hipLaunchKernelGGL(k_target_action, GRID(B * TD3_SA), pre, noise, rng, batch, sa2, sigma, clip, max_a, B); CHECK();
}
int plentd3_q_heads(const float *h2, const float *w3a, const float *b3a, const float *w3b, const float *b3b, const float *batch, float *y, float *dq,
                    float *loss, float *db3a, float *db3b, uint64_t *rng_bump, float gamma, int B, int mode, void *stream) {
    hipLaunchKernelGGL(k_q_heads, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, h2, w3a, b3a, w3b, b3b, batch, y, dq, loss, db3a, db3b, rng_bump, gamma, B, mode); CHECK();
}
int plentd3_dh2(const float *dq, const float *w3a, const float *w3b, const float *h2, float *dh2, int B, int ncrit, int h2_stride, void *stream) {
    hipLaunchKernelGGL(k_dh2, GRID(B * ncrit * TD3_H), dq, w3a, w3b, h2, dh2, B, ncrit, h2_stride); CHECK();
}
int plentd3_relu_mask(float *g, const float *h, int B, int n, int h_stride, void *stream) {
    hipLaunchKernelGGL(k_relu_mask, GRID(B * n), g, h, B, n, h_stride); CHECK();
}
int plentd3_colsum(const float *g, int g_stride, const float *w, int w_stride, float *out, int B, int n, int single_wave, void *stream) {
    // enough workgroups to fill the chip (>= 512), at least 16 rows each
    const int cb = (n + 63) / 64;
    int splits = (512 + cb - 1) / cb;
    if (splits > B / 16) splits = B / 16;
    if (splits > 128) splits = 128;
    if (splits < 1) splits = 1;
    if (single_wave) hipLaunchKernelGGL(k_colsum<1>, dim3((n + 63) / 64, splits), dim3(64), 0, (hipStream_t)stream, g, g_stride, w, w_stride, out, B, n);
    else hipLaunchKernelGGL(k_colsum<4>, dim3((n + 63) / 64, splits), dim3(256), 0, (hipStream_t)stream, g, g_stride, w, w_stride, out, B, n);
    CHECK();
}
int plentd3_tanh_out(const float *pre, float *a, float *sa_pi, float max_a, int B, void *stream) {
    hipLaunchKernelGGL(k_tanh_out, GRID(B * TD3_A), pre, a, sa_pi, max_a, B); CHECK();
}
int plentd3_dtanh(const float *dsa, const float *a, float *dz, float max_a, int B, void *stream) {
    hipLaunchKernelGGL(k_dtanh, GRID(B * TD3_A), dsa, a, dz, max_a, B); CHECK();
}
int plentd3_bias_relu(float *h, const float *bias, int B, int n, void *stream) {
    hipLaunchKernelGGL(k_bias_relu, GRID(B * n), h, bias, B, n); CHECK();
}
int plentd3_polyak(float *target, const float *param, float tau, int n, void *stream) {
    hipLaunchKernelGGL(k_polyak, GRID(n), target, param, tau, n); CHECK();
}
int plentd3_adam(float *p, float *g, float *m, float *v, float *step, int *done_count, int n, double lr, double beta1, double beta2, float eps, int zero_grad,
                 float *target, float tau, float *copy_out, void *stream) {
    // k_adam reads and writes four consecutive floats as one 16-byte access: every array must be 16-byte aligned (torch allocations and the flat
    // parameter buffers are; a sub-view at an odd element offset is not -- refused here rather than faulting on the device)
    const uintptr_t bits = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)target | (uintptr_t)copy_out;
    if (!p || !g || !m || !v || !step || !done_count || n < 1 || (bits & 15)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_adam, dim3(std::min((n + 1023) / 1024, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, step, done_count, n, lr, beta1, beta2, eps, zero_grad, target, tau, copy_out); CHECK();
}
int plentd3_critic_rows(const PlenTd3CriticRows *args, void *stream) {
    if (!args || args->B <= 0 || args->idx || args->noise || args->adam_step) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_critic_rows, dim3((args->B + RB - 1) / RB), dim3(64), 0, (hipStream_t)stream, *args); CHECK();
}
int plentd3_policy_rows(const PlenTd3PolicyRows *args, void *stream) {
    if (!args || args->B <= 0 || args->adam_step) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_policy_rows, dim3((args->B + RB - 1) / RB), dim3(64), 0, (hipStream_t)stream, *args); CHECK();
}
int plentd3_critic_team(const PlenTd3CriticRows *args, void *stream) {
    if (!args || args->B <= 0) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_critic_team, dim3((args->B + QB - 1) / QB), dim3(64 * TEAM_NW), 0, (hipStream_t)stream, *args); CHECK();
}
int plentd3_policy_team(const PlenTd3PolicyRows *args, void *stream) {
    if (!args || args->B <= 0 || (args->adam_step && !args->done_count)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_policy_team, dim3((args->B + QB - 1) / QB), dim3(64 * TEAM_NW), 0, (hipStream_t)stream, *args); CHECK();
}
int plentd3_pack(const PlenTd3PackGroup *group, void *stream) {
    if (!group || group->n_jobs < 1 || group->n_jobs > PLENTD3_PACK_JOBS) return -(int)hipErrorInvalidValue;
    PlenTd3PackGroup G = *group;
    int f4 = 0;
    for (int j = 0; j < G.n_jobs; j++) {
        const PlenTd3PackJob &J = G.job[j];
        if (!J.src || !J.dst || J.N < 1 || J.K < 1 || ((uintptr_t)J.dst & 15)) return -(int)hipErrorInvalidValue;
        G.job[j].f4_0 = f4;
        f4 += J.team ? ((J.N + 31) / 32) * ((J.K + 63) / 64) * 512 : ((J.N + 15) / 16) * ((J.K + 15) / 16) * 64;
    }
    hipLaunchKernelGGL(k_pack, dim3((f4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, G); CHECK();
}
int plentd3_critic_block(const PlenTd3CriticBlock *args, void *stream) {
    if (!args || args->rows.B <= 0 || !args->partials || !args->p_at_w1 || !args->p_c_w5t) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_critic_block, dim3((args->rows.B + BLK_R - 1) / BLK_R), dim3(64 * BLK_CRITIC_NW), 0, (hipStream_t)stream, *args); CHECK();
}
int plentd3_policy_block(const PlenTd3PolicyBlock *args, void *stream) {
    if (!args || args->rows.B <= 0 || !args->p_a_w1 || !args->p_a_w2t || (args->rows.adam_step && !args->rows.done_count)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_policy_block, dim3((args->rows.B + BLK_R - 1) / BLK_R), dim3(64 * BLK_NW), 0, (hipStream_t)stream, *args); CHECK();
}
int plentd3_wgrad_big(const PlenTd3WgradBig *group, void *stream) {
    if (!group || group->n_jobs < 1 || group->n_jobs > PLENTD3_WGRAD_BIG_JOBS || group->B <= 0 || group->chunks < 1 || group->rows_per_chunk < 1 || !group->partial ||
        (group->stride & 3) || ((uintptr_t)group->partial & 15) || (long long)group->chunks * group->rows_per_chunk < group->B || (group->rows_per_chunk & 7)) return -(int)hipErrorInvalidValue;
    PlenTd3WgradBig G = *group;
    int wg = 0;
    for (int j = 0; j < G.n_jobs; j++) {
        const PlenTd3WgradBigJob &J = G.job[j];
        if (!J.dH || !J.X || J.N < 1 || J.K < 1 || J.goff < 0 || (size_t)J.goff + (size_t)J.N * J.K > (size_t)G.stride || (J.boff >= 0 && J.boff + J.N > G.stride)) return -(int)hipErrorInvalidValue;
        if (J.kind == 1 && (J.N != 1 || J.K > 256 || (J.K & 3) || (J.xs & 3) || ((uintptr_t)J.X & 15))) return -(int)hipErrorInvalidValue;
        G.job[j].wg0 = wg;
        wg += (J.kind == 1 ? 1 : ((J.N + 31) / 32) * ((J.K + 63) / 64)) * G.chunks;
    }
    hipLaunchKernelGGL(k_wgrad_big, dim3(wg), dim3(64 * WGB_NW), 0, (hipStream_t)stream, G); CHECK();
}
int plentd3_adam_big(float *p, float *g, float *m, float *v, float *step, int *done_count, int n, double lr, double beta1, double beta2, float eps,
                     float *target, float tau, float *copy_out, const float *partial, int chunks, int stride, int reduce_only, void *stream) {
    const uintptr_t bits = (uintptr_t)g | (uintptr_t)partial;
    if (!g || !partial || chunks < 1 || stride < n || (stride & 3) || n < 1 || (bits & 15) || (!reduce_only && (!p || !m || !v || !step || !done_count))) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_adam_big, dim3(std::min((n + 1023) / 1024, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, step, done_count, n, lr, beta1, beta2, eps, target, tau, copy_out,
                       partial, chunks, stride, reduce_only); CHECK();
}
int plentd3_actor_block(const PlenTd3ActorBlock *args, void *stream) {
    if (!args || args->rows.B <= 0 || !args->p_a_w1 || !args->p_a_w2 || !args->p_a_w3 || !args->rows.state || !args->rows.action || !args->rows.rng) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_actor_block, dim3((args->rows.B + BLK_R - 1) / BLK_R), dim3(64 * BLK_ACTOR_NW), 0, (hipStream_t)stream, *args); CHECK();
}
int plentd3_dev_mfma_spin(int workgroups, int iters, float *sink, void *stream) {
    hipLaunchKernelGGL(k_dev_mfma_spin, dim3(workgroups), dim3(64), 0, (hipStream_t)stream, iters, sink); CHECK();
}
int plentd3_wgrad_group(const PlenTd3WgradGroup *group, void *stream) {
    if (!group || group->n_jobs < 1 || group->n_jobs > PLENTD3_WGRAD_JOBS || group->B <= 0) return -(int)hipErrorInvalidValue;
    PlenTd3WgradGroup G = *group;
    int tiles = 0;
    for (int j = 0; j < G.n_jobs; j++) {
        if (!G.job[j].dH || !G.job[j].X || !G.job[j].dW || G.job[j].N < 1 || G.job[j].K < 1) return -(int)hipErrorInvalidValue;
        G.job[j].tile0 = tiles;
        tiles += ((G.job[j].N + 31) / 32) * ((G.job[j].K + 31) / 32);
    }
    hipLaunchKernelGGL(k_wgrad_group, dim3(tiles), dim3(256), 0, (hipStream_t)stream, G); CHECK();
}
int plentd3_wgrad_adam_group(const PlenTd3WgradGroup *group, const PlenTd3AdamFused *adam, void *stream) {
    if (!group || !adam || group->n_jobs < 1 || group->n_jobs > PLENTD3_WGRAD_JOBS || group->B <= 0) return -(int)hipErrorInvalidValue;
    if (!adam->p || !adam->g || !adam->m || !adam->v || !adam->step || (!adam->done_count && !adam->step_advanced) || adam->n < 1 || adam->n_extra < 0 || adam->n_extra > PLENTD3_ADAM_EXTRAS)
        return -(int)hipErrorInvalidValue;
    PlenTd3WgradGroup G = *group;
    int tiles = 0;
    for (int j = 0; j < G.n_jobs; j++) {
        const PlenTd3WgradJob &J = G.job[j];
        if (!J.dH || !J.X || !J.dW || J.N < 1 || J.K < 1 || J.dws != J.K) return -(int)hipErrorInvalidValue;
        // every output element must lie inside the flat gradient buffer the Adam buffers mirror
        if (J.dW < adam->g || J.dW + (size_t)J.N * J.K > adam->g + adam->n || (J.db && (J.db < adam->g || J.db + J.N > adam->g + adam->n))) return -(int)hipErrorInvalidValue;
        G.job[j].tile0 = tiles;
        tiles += ((J.N + 31) / 32) * ((J.K + 31) / 32);
    }
    for (int e = 0; e < adam->n_extra; e++)
        if (adam->extra_off[e] < 0 || adam->extra_off[e] >= adam->n) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_wgrad_adam_group, dim3(tiles), dim3(256), 0, (hipStream_t)stream, G, *adam); CHECK();
}
int plentd3_actor_rows(const PlenTd3ActorRows *args, void *stream) {
    if (!args || args->B <= 0) return -(int)hipErrorInvalidValue;
    // four waves per row block by default (csrc/td3_rows.hip, k_actor_rows4); PLEN_TD3_ACTOR_WAVES=1: the single-wave workgroups
    static const int waves = [] { const char *e = getenv("PLEN_TD3_ACTOR_WAVES"); return e && atoi(e) == 1 ? 1 : 4; }();
    if (waves == 4) hipLaunchKernelGGL(k_actor_rows4, dim3((args->B + RB - 1) / RB), dim3(256), 0, (hipStream_t)stream, *args);
    else hipLaunchKernelGGL(k_actor_rows, dim3((args->B + RB - 1) / RB), dim3(64), 0, (hipStream_t)stream, *args);
    CHECK();
}
int plentd3_stamp(uint64_t *table, const int64_t *counter, int64_t div, int ring, int nslots, int idx, void *stream) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, table, counter, div, ring, nslots, idx); CHECK();
}
const char *plentd3_version(void) { return "plentd3 0.1 (gfx950)"; }

}  // extern "C"
