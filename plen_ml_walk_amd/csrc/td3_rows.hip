// td3_rows.hip -- the row-local part of one TD3 critic update as ONE kernel (included by td3_kernels.hip; C ABI in include/plentd3.h).
//
// Why: beside two collectors whose env launches hold every wave slot of the chip (2 x 2048 single-wave workgroups at 128 VGPRs = 4 per SIMD),
// the update's ~35 library / elementwise kernels each need several free slots on ONE compute unit before a 256-thread workgroup can start,
// and single-wave env workgroups take every slot that frees up first: device-clock timelines (scripts/gpu_td3_timeline.py) show the update's
// first GEMMs waiting 100-450 us whenever both env launches are resident.  Everything between sampling the batch and the weight gradients is
// independent per batch row (td3.py:277-331: target action, twin target critics, twin critics, loss, and the backward pass down to the first
// layer's input gradient), so here one WAVE owns 16 batch rows and carries them through all of it:
//   * single-wave workgroups, <= 128 VGPRs, no LDS: such a workgroup fits the slot ONE retiring env wave frees, and the update stream's
//     dispatch priority hands it that slot before any queued env wave;
//   * one launch: the slot is acquired once per update instead of once per layer;
//   * dense layers on the matrix cores: v_mfma_f32_16x16x4_f32 (full fp32, as the reference's torch fp32 layers), 16 batch rows x 16 outputs
//     per tile, both operands straight from global memory (activations of the wave's own rows; weights, 0.9 MB in all, from L2) as one
//     16-byte load per lane covering four k steps, double-buffered against the MFMAs;
//   * activations go through global memory between layers (each wave reads back only what it wrote itself: a workgroup-scope fence, i.e. a
//     wait on the wave's own stores, is enough); the gradients the weight-gradient kernels need (c1, c2, dh2, dh1, dq) are left there.
// Arithmetic per row is the reference's (td3.py:299-319) with fp32 accumulation in MFMA order instead of the GEMM library's.

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
#define RB 16                         // batch rows per wave
#define FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup")

// Buffer addressing (descriptor in 4 SGPRs + one 32-bit lane offset + a scalar offset): a lane's address costs ONE register whatever the number of
// tiles, and reads / writes outside [base, base + bytes) return 0 / are dropped -- which is all the handling ragged edges need (the actor's 18-wide
// output layer, a batch that is not a multiple of 16).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
static __device__ __forceinline__ rsrc_t mkrs(const float *p, size_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
static __device__ __forceinline__ floatx4 bload4(rsrc_t rs, uint32_t voff, uint32_t soff) { return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0)); }
static __device__ __forceinline__ float bload1(rsrc_t rs, uint32_t voff, uint32_t soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0)); }
static __device__ __forceinline__ void bstore1(float v, rsrc_t rs, uint32_t voff, uint32_t soff) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, voff, soff, 0); }
// Row i (0..3) of a lane's 4-row group, column `col`, of a row-major [B][ld] matrix whose descriptor covers exactly B rows: the ROW part of the
// address goes into voffset, which the hardware range-checks (rows >= B: stores dropped, loads return 0); only the column part rides in soffset,
// which raw buffers exclude from bounds checking (with the row term there, a 4-row group straddling B -- B % 4 != 0 -- could write up to 3
// rows past the end into a neighbouring allocation)
static __device__ __forceinline__ void bstore_row(float v, rsrc_t rs, uint32_t voff, int i, int ld, int col) { bstore1(v, rs, voff + 4u * (uint32_t)(i * ld), 4u * (uint32_t)col); }
static __device__ __forceinline__ float bload_row(rsrc_t rs, uint32_t voff, int i, int ld, int col) { return bload1(rs, voff + 4u * (uint32_t)(i * ld), 4u * (uint32_t)col); }

// acc[t] += X[16 rows][K] * W[n0 + 16 t + (0..15)][K]^T for t < NT.  Lane (r = lane % 16, g = lane / 16) feeds MFMA j of a 16-wide k step with
// A[i = r][k] = X[r][k0 + 4 g + j] and B[k][n = r] = W[n0 + 16 t + r][k0 + 4 g + j] (any assignment of k indices to the instruction's four
// k slots is valid as long as A and B agree), so a lane's four values are one 16-byte load.  xoff = byte offset of this lane's row of X (+ 16 g);
// rows of W beyond the buffer read as 0.  KGUARD: K is not a multiple of 16: A is zeroed at k >= K (B then reads finite values of the next row).
template <int NT, bool KGUARD>
static __device__ __forceinline__ void mm_nt(rsrc_t rx, uint32_t xoff, rsrc_t rw, int ldw, int n0, int K, floatx4 (&acc)[NT], int r, int g) {
    const uint32_t woff = (uint32_t)((n0 + r) * ldw + 4 * g) * 4u;
    auto load = [&](int k0, floatx4 &a, floatx4 (&b)[NT]) {
        a = bload4(rx, xoff, 4u * (uint32_t)k0);
#pragma unroll
        for (int t = 0; t < NT; t++) b[t] = bload4(rw, woff, 4u * (uint32_t)(16 * t * ldw + k0));
    };
    auto guard = [&](int k0, floatx4 &a) {            // at the point of use, not of the load (a select right after the load would wait for it)
        if constexpr (KGUARD) {
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = k0 + 4 * g + j < K ? a[j] : 0.f;
        }
    };
    // two stages in flight: the loads of the next 16 k are issued BEFORE the 32 MFMAs of the current ones (the compiler, left alone, sinks them to
    // their first use and exposes the whole L2 latency every step: hence the scheduling barriers).  Loads past K are harmless (zero-guarded A,
    // in-bounds or zero-returning B) and keep the loop body branch-free.
    floatx4 a0, b0[NT], a1, b1[NT];
    load(0, a0, b0);
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += 32) {
        load(k0 + 16, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        guard(k0, a0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[t][j], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load(k0 + 32, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        guard(k0 + 16, a1);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[t][j], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc[t] += G[16 rows][Kc] * W[Kc][j0 + 16 t + (0..15)] (input gradient of a dense layer: W is the layer's [out = Kc][in] weight).  A as in mm_nt;
// B[k][n = r] = W[n0 + 4 g + j][j0 + 16 t + r]: four 4-byte loads, each 64 contiguous bytes per lane group.
template <int NT, bool KGUARD>
static __device__ __forceinline__ void mm_nn(rsrc_t rg, uint32_t goff, rsrc_t rw, int ldw, int j0, int Kc, floatx4 (&acc)[NT], int r, int g) {
    const uint32_t woff = (uint32_t)(4 * g * ldw + j0 + r) * 4u;
    auto guard = [&](int n0, floatx4 &a) {
        if constexpr (KGUARD) {
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = n0 + 4 * g + j < Kc ? a[j] : 0.f;
        }
    };
    auto load = [&](int n0, floatx4 &a, floatx4 (&b)[NT]) {
        a = bload4(rg, goff, 4u * (uint32_t)n0);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) b[t][j] = bload1(rw, woff, 4u * (uint32_t)((n0 + j) * ldw + 16 * t));
    };
    floatx4 a0, b0[NT], a1, b1[NT];
    load(0, a0, b0);
#pragma unroll 1
    for (int n0 = 0; n0 < Kc; n0 += 32) {
        load(n0 + 16, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        guard(n0, a0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[t][j], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load(n0 + 32, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        guard(n0 + 16, a1);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[t][j], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Result layout of the 16x16 MFMA: lane (r, g) holds rows 4 g + i (i = 0..3) of column 16 t + r in acc[t][i].
// Geometry of the wave's row block.  Every phase builds its own from an opaque lane id (fresh_rows): lane-derived offsets and pointers are then not
// common subexpressions of the whole kernel, computed once at its top and kept alive -- spilled into scratch, in the k loops too -- across all phases.
struct RowBlock {
    int b0, B, r, g, brow, lane;
    // byte offset of element [b0 + 4 g][r] of a row-major [B][ld] matrix (stores), and of [brow][4 g] (A-operand loads)
    __device__ __forceinline__ uint32_t soff(int ld) const { return (uint32_t)((b0 + 4 * g) * ld + r) * 4u; }
    __device__ __forceinline__ uint32_t aoff(int ld, int col0 = 0) const { return (uint32_t)(brow * ld + col0 + 4 * g) * 4u; }
};

static __device__ __forceinline__ RowBlock fresh_rows(int b0, int B) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int r = lane & 15;
    return RowBlock{b0, B, r, lane >> 4, min(b0 + r, B - 1), lane};
}
#define PHASE() const RowBlock rb = fresh_rows(b0, B); const int r = rb.r, g = rb.g, lane = rb.lane, brow = rb.brow; (void)r; (void)g; (void)lane; (void)brow

// Y[b0 + row][n0 + (0..16 NT)] = relu(X W^T + bias): one column block of a hidden layer, stored row-major (rows beyond B fall outside ry: dropped)
template <int NT, bool KGUARD>
static __device__ __forceinline__ void dense_relu(rsrc_t rx, uint32_t xoff, int K, rsrc_t rw, int ldw, const float *bias, int n0, rsrc_t ry, int ldy, const RowBlock &rb) {
    floatx4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = floatx4{0, 0, 0, 0};
    mm_nt<NT, KGUARD>(rx, xoff, rw, ldw, n0, K, acc, rb.r, rb.g);
    const uint32_t yoff = rb.soff(ldy);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float bv = bias[n0 + 16 * t + rb.r];
#pragma unroll
        for (int i = 0; i < 4; i++) bstore_row(fmaxf(acc[t][i] + bv, 0.f), ry, yoff, i, ldy, n0 + 16 * t);
    }
}

// q[i] (rows 4 g + i) = relu(X W2^T + b2) . w3 + b3: second hidden layer of one critic and its scalar head; the hidden activations are also
// stored at column hcol0 of rh (row stride ldh) when STORE.  The sum over the 16 columns a lane group holds is a butterfly over r.
template <bool STORE>
static __device__ __forceinline__ void critic_l2_head(rsrc_t rx, uint32_t xoff, const float *W2, const float *b2, const float *w3, const float *b3, rsrc_t rh, int ldh, int hcol0,
                                                      const RowBlock &rb, float (&q)[4]) {
    const rsrc_t rw = mkrs(W2, (size_t)TD3_H * TD3_H * 4);
    const uint32_t hoff = rb.soff(ldh);
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int n0 = 0; n0 < TD3_H; n0 += 128) {
        floatx4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = floatx4{0, 0, 0, 0};
        mm_nt<8, false>(rx, xoff, rw, TD3_H, n0, TD3_H, acc, rb.r, rb.g);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int c = n0 + 16 * t + rb.r;
            const float bv = b2[c], wv = w3[c];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float v = fmaxf(acc[t][i] + bv, 0.f);
                part[i] += v * wv;
                if constexpr (STORE) bstore_row(v, rh, hoff, i, ldh, hcol0 + n0 + 16 * t);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float v = part[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);      // lanes r = 0..15 of the group
        q[i] = v + b3[0];
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_critic_rows(PlenTd3CriticRows A) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int B = A.B, b0 = blockIdx.x * RB;
    const int brow = min(b0 + r, B - 1);                 // the batch row whose activations this lane feeds to the matrix cores
    // ---- sample 16 rows of the replay ring (td3.py:166-193; same draw as k_sample_gather) and gather them ----
    {
        PHASE();
        int64_t idx = 0;
        if (lane < RB) {
            const int b = min(b0 + lane, B - 1);
            const int64_t tot = A.total[0];
            int64_t filled, start;
            if (tot + A.guard <= A.capacity) { filled = tot; start = 0; }
            else { filled = A.capacity - A.guard; start = (tot + A.guard) % A.capacity; }
            const float ub = rng_uniform(A.rng, 0u, (uint32_t)b);
            int64_t i = (int64_t)((double)ub * (double)filled);
            i = i < filled - 1 ? i : filled - 1;
            i = i > 0 ? i : 0;
            idx = (start + i) % A.capacity;
        }
        const int lo = (int)(idx & 0xffffffff), hi = (int)(idx >> 32);
#pragma unroll 4
        for (int i = 0; i < RB; i++) {
            const int b = b0 + i;
            if (b >= B) break;
            const int64_t id = ((int64_t)__shfl(hi, i) << 32) | (uint32_t)__shfl(lo, i);
            const float *src = A.data + (size_t)id * TD3_ROW;
            const float v0 = src[lane], v1 = lane < TD3_ROW - 64 ? src[64 + lane] : 0.f;
            float *dst = A.batch + (size_t)b * TD3_ROW;
            dst[lane] = v0;
            if (lane < TD3_ROW - 64) dst[64 + lane] = v1;
            if (lane < TD3_S) A.sa_pi[(size_t)b * TD3_SA + lane] = v0;
        }
    }
    FENCE();
    const RowBlock rb{b0, B, r, g, brow, lane};
    const size_t Bz = (size_t)B;
    const rsrc_t r_batch = mkrs(A.batch, Bz * TD3_ROW * 4), r_t0 = mkrs(A.t0, Bz * 2 * TD3_H * 4), r_t1 = mkrs(A.t1, Bz * 2 * TD3_H * 4), r_sa2 = mkrs(A.sa2, Bz * TD3_SA * 4);
    const rsrc_t r_c1 = mkrs(A.c1, Bz * 2 * TD3_H * 4), r_c2 = mkrs(A.c2, Bz * 2 * TD3_H * 4);
    // gathered rows: s 0..25 | a 26..43 | s2 44..69 | r 70 | not_done 71
    // ---- target policy smoothing (td3.py:299-304): actor_target(s2) -> noisy clipped action -> sa2 = [s2 | a2] ----
    // (t0 / t1: scratch rows of 512 floats private to the wave's batch rows -- one row stride for every use, so no two waves ever share a line)
    {
        PHASE();
        const rsrc_t rw = mkrs(A.at_w1, (size_t)TD3_H * TD3_S * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) dense_relu<8, true>(r_batch, rb.aoff(TD3_ROW, TD3_SA), TD3_S, rw, TD3_S, A.at_b1, n0, r_t0, 2 * TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        const rsrc_t rw = mkrs(A.at_w2, (size_t)TD3_H * TD3_H * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) dense_relu<8, false>(r_t0, rb.aoff(2 * TD3_H), TD3_H, rw, TD3_H, A.at_b2, n0, r_t1, 2 * TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        mm_nt<2, false>(r_t1, rb.aoff(2 * TD3_H), mkrs(A.at_w3, (size_t)TD3_A * TD3_H * 4), TD3_H, 0, TD3_H, acc, r, g);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = 16 * t + r;
            if (j < TD3_A) {
                const float bv = A.at_b3[j];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int b = b0 + 4 * g + i;
                    if (b < B) {
                        const float z = rng_normal(A.rng, 1u, (uint32_t)(b * TD3_A + j));                 // torch.randn_like(action), td3.py:300
                        const float n = fminf(fmaxf(z * A.sigma, -A.clip), A.clip);
                        A.sa2[(size_t)b * TD3_SA + TD3_S + j] = fminf(fmaxf(A.max_a * tanhf(acc[t][i] + bv) + n, -A.max_a), A.max_a);
                    }
                }
            }
        }
#pragma unroll 4
        for (int i = 0; i < RB; i++) {
            const int b = b0 + i;
            if (b >= B) break;
            if (lane < TD3_S) A.sa2[(size_t)b * TD3_SA + lane] = A.batch[(size_t)b * TD3_ROW + TD3_SA + lane];
        }
    }
    FENCE();
    // ---- clipped double-Q target (td3.py:306-309): both target critics' first layers stacked (W14 = [fc1.w; fc4.w]) ----
    {
        PHASE();
        const rsrc_t rw = mkrs(A.ct_w14, (size_t)2 * TD3_H * TD3_SA * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < 2 * TD3_H; n0 += 128) dense_relu<8, true>(r_sa2, rb.aoff(TD3_SA), TD3_SA, rw, TD3_SA, A.ct_b14, n0, r_t0, 2 * TD3_H, rb);
    }
    FENCE();
    // (values that would have to survive the next matrix products in registers -- y, dq -- go through memory instead: the products' two load stages
    // and accumulators leave no room, and spills inside the k loops cost more than these few loads)
    {
        PHASE();
        float qa[4], qb[4];
        critic_l2_head<false>(r_t0, rb.aoff(2 * TD3_H), A.ct_w2, A.ct_b2, A.ct_w3, A.ct_b3, r_t0, 0, 0, rb, qa);
#pragma unroll
        for (int i = 0; i < 4; i++) if (r == 0) A.t1[(size_t)min(b0 + 4 * g + i, B - 1) * 2 * TD3_H] = qa[i];            // park q_a (t1 is free by now)
        critic_l2_head<false>(r_t0, rb.aoff(2 * TD3_H, TD3_H), A.ct_w5, A.ct_b5, A.ct_w6, A.ct_b6, r_t0, 0, 0, rb, qb);
        FENCE();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int bb = min(b0 + 4 * g + i, B - 1);
            const float *row = A.batch + (size_t)bb * TD3_ROW;
            const float y = row[TD3_ROW - 2] + row[TD3_ROW - 1] * A.gamma * fminf(A.t1[(size_t)bb * 2 * TD3_H], qb[i]);
            if (r == 0) A.t1[(size_t)bb * 2 * TD3_H + 1] = y;                                                          // y[b] for the loss below
        }
    }
    // ---- critic forward, loss, and its gradient down to the hidden layers (td3.py:312-323) ----
    {
        PHASE();
        const rsrc_t rw = mkrs(A.c_w14, (size_t)2 * TD3_H * TD3_SA * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < 2 * TD3_H; n0 += 128) dense_relu<8, true>(r_batch, rb.aoff(TD3_ROW), TD3_SA, rw, TD3_SA, A.c_b14, n0, r_c1, 2 * TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        const float inv = 1.f / (float)B;
        float lsum = 0.f, gsum[2] = {0.f, 0.f};
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            float q[4];
            critic_l2_head<true>(r_c1, rb.aoff(2 * TD3_H, c * TD3_H), c ? A.c_w5 : A.c_w2, c ? A.c_b5 : A.c_b2, c ? A.c_w6 : A.c_w3, c ? A.c_b6 : A.c_b3, r_c2, 2 * TD3_H, c * TD3_H, rb, q);
            float gs = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = b0 + 4 * g + i;
                if (r == 0 && b < B) {
                    const float e = q[i] - A.t1[(size_t)b * 2 * TD3_H + 1], d = 2.f * e * inv;
                    A.dq[2 * b + c] = d;
                    lsum += e * e * inv; gs += d;
                }
            }
            gsum[c] = gs;
        }
        lsum = wave_sum(lsum);
        const float ga = wave_sum(gsum[0]), gb = wave_sum(gsum[1]);
        if (lane == 0) { atomicAdd(A.loss, lsum); atomicAdd(A.db3a, ga); atomicAdd(A.db3b, gb); }
    }
    FENCE();
    // dh2 = dq (x) w3 where the hidden unit was active: rows of this wave, 4 columns per lane and critic
    {
        PHASE();
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
        const floatx4 wv = *reinterpret_cast<const floatx4 *>((c ? A.c_w6 : A.c_w3) + 4 * lane);
#pragma unroll 4
        for (int i = 0; i < RB; i++) {
            const int b = b0 + i;
            if (b >= B) break;
            const float d = A.dq[2 * b + c];
            const size_t o = (size_t)b * 2 * TD3_H + c * TD3_H + 4 * lane;
            const floatx4 h = *reinterpret_cast<const floatx4 *>(A.c2 + o);
            floatx4 dv;
#pragma unroll
            for (int j = 0; j < 4; j++) dv[j] = h[j] > 0.f ? d * wv[j] : 0.f;
            *reinterpret_cast<floatx4 *>(A.dh2 + o) = dv;
        }
    }
    }
    FENCE();
    // dh1_c = (dh2_c W2_c) where c1_c was active
    {
        PHASE();
        const rsrc_t r_dh2 = mkrs(A.dh2, Bz * 2 * TD3_H * 4), r_dh1 = mkrs(A.dh1, Bz * 2 * TD3_H * 4);
        const uint32_t ooff = rb.soff(2 * TD3_H);
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            const rsrc_t rw = mkrs(c ? A.c_w5 : A.c_w2, (size_t)TD3_H * TD3_H * 4);
#pragma unroll 1
            for (int j0 = 0; j0 < TD3_H; j0 += 128) {
                floatx4 acc[8];
#pragma unroll
                for (int t = 0; t < 8; t++) acc[t] = floatx4{0, 0, 0, 0};
                mm_nn<8, false>(r_dh2, rb.aoff(2 * TD3_H, c * TD3_H), rw, TD3_H, j0, TD3_H, acc, r, g);
#pragma unroll
                for (int t = 0; t < 8; t++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int col = c * TD3_H + j0 + 16 * t;
                        bstore_row(bload_row(r_c1, ooff, i, 2 * TD3_H, col) > 0.f ? acc[t][i] : 0.f, r_dh1, ooff, i, 2 * TD3_H, col);
                    }
            }
        }
    }
    // the last wave to finish advances the random stream's call counter: every wave has read it by then (its draws are long consumed: no release fence, which
    // here would be an L2 write-back per wave -- td3_kernels.hip: handoff_last)
    if (lane == 0) {
#if !TD3_LIGHT_HANDOFF
        __threadfence();
#endif
        if (atomicAdd(A.done_count, 1) == (int)gridDim.x - 1) { A.done_count[0] = 0; if (A.rng_bump) A.rng_bump[1] += 1; }
    }
}

// ---- the delayed policy update's row-local part (td3.py:334-341): actor(s) -> a -> critic.Q1(s, a) -> d(-mean Q1)/d(actor activations), one wave per 16 rows.
//      Left for the weight-gradient kernels: p1, p2 (actor hidden activations), dz (gradient at the actor's output pre-activation), dp2, dp1.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_policy_rows(PlenTd3PolicyRows A) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int B = A.B, b0 = blockIdx.x * RB;
    const RowBlock rb{b0, B, r, g, min(b0 + r, B - 1), lane};
    const size_t Bz = (size_t)B;
    const rsrc_t r_sa = mkrs(A.sa_pi, Bz * TD3_SA * 4), r_p1 = mkrs(A.p1, Bz * TD3_H * 4), r_p2 = mkrs(A.p2, Bz * TD3_H * 4), r_g1 = mkrs(A.g1, Bz * TD3_H * 4);
    const rsrc_t r_dg2 = mkrs(A.dg2, Bz * TD3_H * 4), r_dg1 = mkrs(A.dg1, Bz * TD3_H * 4), r_dp2 = mkrs(A.dp2, Bz * TD3_H * 4), r_dp1 = mkrs(A.dp1, Bz * TD3_H * 4);
    const rsrc_t r_a = mkrs(A.a_pi, Bz * TD3_A * 4), r_dz = mkrs(A.dz, Bz * TD3_A * 4);
    const uint32_t hoff = rb.soff(TD3_H);
    // actor forward: s = state columns of sa_pi (left there by the critic pass)
    {
        PHASE();
        const rsrc_t rw = mkrs(A.a_w1, (size_t)TD3_H * TD3_S * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) dense_relu<8, true>(r_sa, rb.aoff(TD3_SA), TD3_S, rw, TD3_S, A.a_b1, n0, r_p1, TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        const rsrc_t rw = mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) dense_relu<8, false>(r_p1, rb.aoff(TD3_H), TD3_H, rw, TD3_H, A.a_b2, n0, r_p2, TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        mm_nt<2, false>(r_p2, rb.aoff(TD3_H), mkrs(A.a_w3, (size_t)TD3_A * TD3_H * 4), TD3_H, 0, TD3_H, acc, r, g);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = 16 * t + r;
            if (j < TD3_A) {
                const float bv = A.a_b3[j];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int b = b0 + 4 * g + i;
                    if (b < B) {
                        const float a = A.max_a * tanhf(acc[t][i] + bv);                                   // td3.py:57
                        A.a_pi[(size_t)b * TD3_A + j] = a;
                        A.sa_pi[(size_t)b * TD3_SA + TD3_S + j] = a;
                    }
                }
            }
        }
    }
    FENCE();
    // critic.Q1 forward (fc1 = the first 256 rows of W14) and the gradient of -mean Q1 at its second hidden layer: dg2 = -(1/B) w3 (g2 > 0)
    {
        PHASE();
        const rsrc_t rw = mkrs(A.c_w1, (size_t)TD3_H * TD3_SA * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) dense_relu<8, true>(r_sa, rb.aoff(TD3_SA), TD3_SA, rw, TD3_SA, A.c_b1, n0, r_g1, TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        const rsrc_t rw = mkrs(A.c_w2, (size_t)TD3_H * TD3_H * 4);
        const float ginv = -1.f / (float)B;
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) {
            floatx4 acc[8];
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = floatx4{0, 0, 0, 0};
            mm_nt<8, false>(r_g1, rb.aoff(TD3_H), rw, TD3_H, n0, TD3_H, acc, r, g);
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int c = n0 + 16 * t + r;
                const float bv = A.c_b2[c], wv = ginv * A.c_w3[c];
#pragma unroll
                for (int i = 0; i < 4; i++) bstore_row(acc[t][i] + bv > 0.f ? wv : 0.f, r_dg2, hoff, i, TD3_H, n0 + 16 * t);
            }
        }
    }
    FENCE();
    // dg1 = (dg2 W2)(g1 > 0)
    {
        PHASE();
        const rsrc_t rw = mkrs(A.c_w2, (size_t)TD3_H * TD3_H * 4);
#pragma unroll 1
        for (int j0 = 0; j0 < TD3_H; j0 += 128) {
            floatx4 acc[8];
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = floatx4{0, 0, 0, 0};
            mm_nn<8, false>(r_dg2, rb.aoff(TD3_H), rw, TD3_H, j0, TD3_H, acc, r, g);
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bstore_row(bload_row(r_g1, hoff, i, TD3_H, j0 + 16 * t) > 0.f ? acc[t][i] : 0.f, r_dg1, hoff, i, TD3_H, j0 + 16 * t);
                }
        }
    }
    FENCE();
    // d/d action = (dg1 W1)[:, 26:44], through the tanh: dz = that * (max_a - a^2 / max_a)
    {
        PHASE();
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        mm_nn<2, false>(r_dg1, rb.aoff(TD3_H), mkrs(A.c_w1, (size_t)TD3_H * TD3_SA * 4), TD3_SA, TD3_S, TD3_H, acc, r, g);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = 16 * t + r;
            if (j < TD3_A) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int b = b0 + 4 * g + i;
                    if (b < B) {
                        const float a = A.a_pi[(size_t)b * TD3_A + j];
                        A.dz[(size_t)b * TD3_A + j] = acc[t][i] * (A.max_a - a * a / A.max_a);
                    }
                }
            }
        }
    }
    FENCE();
    // back through the actor: dp2 = (dz W3)(p2 > 0), dp1 = (dp2 W2)(p1 > 0)
    {
        PHASE();
        const rsrc_t rw = mkrs(A.a_w3, (size_t)TD3_A * TD3_H * 4);
#pragma unroll 1
        for (int j0 = 0; j0 < TD3_H; j0 += 128) {
            floatx4 acc[8];
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = floatx4{0, 0, 0, 0};
            mm_nn<8, true>(r_dz, rb.aoff(TD3_A), rw, TD3_H, j0, TD3_A, acc, r, g);
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bstore_row(bload_row(r_p2, hoff, i, TD3_H, j0 + 16 * t) > 0.f ? acc[t][i] : 0.f, r_dp2, hoff, i, TD3_H, j0 + 16 * t);
                }
        }
    }
    FENCE();
    {
        PHASE();
        const rsrc_t rw = mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4);
#pragma unroll 1
        for (int j0 = 0; j0 < TD3_H; j0 += 128) {
            floatx4 acc[8];
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = floatx4{0, 0, 0, 0};
            mm_nn<8, false>(r_dp2, rb.aoff(TD3_H), rw, TD3_H, j0, TD3_H, acc, r, g);
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bstore_row(bload_row(r_p1, hoff, i, TD3_H, j0 + 16 * t) > 0.f ? acc[t][i] : 0.f, r_dp1, hoff, i, TD3_H, j0 + 16 * t);
                }
        }
    }
}

// ---- the collect phase's action (plen_td3.py:101-104): a = clamp(actor(state) + N(0, sigma), +-max_a) for 16 envs per wave (as plentd3_explore, noise = NULL)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_actor_rows(PlenTd3ActorRows A) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int B = A.B, b0 = blockIdx.x * RB;
    const RowBlock rb{b0, B, r, g, min(b0 + r, B - 1), lane};
    const size_t Bz = (size_t)B;
    const rsrc_t r_s = mkrs(A.state, Bz * TD3_S * 4), r_p1 = mkrs(A.p1, Bz * TD3_H * 4), r_p2 = mkrs(A.p2, Bz * TD3_H * 4);
    {
        PHASE();
        const rsrc_t rw = mkrs(A.a_w1, (size_t)TD3_H * TD3_S * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) dense_relu<8, true>(r_s, rb.aoff(TD3_S), TD3_S, rw, TD3_S, A.a_b1, n0, r_p1, TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        const rsrc_t rw = mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4);
#pragma unroll 1
        for (int n0 = 0; n0 < TD3_H; n0 += 128) dense_relu<8, false>(r_p1, rb.aoff(TD3_H), TD3_H, rw, TD3_H, A.a_b2, n0, r_p2, TD3_H, rb);
    }
    FENCE();
    {
        PHASE();
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        mm_nt<2, false>(r_p2, rb.aoff(TD3_H), mkrs(A.a_w3, (size_t)TD3_A * TD3_H * 4), TD3_H, 0, TD3_H, acc, r, g);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = 16 * t + r;
            if (j < TD3_A) {
                const float bv = A.a_b3[j];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int b = b0 + 4 * g + i;
                    if (b < B) {
                        const int e = b * TD3_A + j;
                        A.action[e] = fminf(fmaxf(A.max_a * tanhf(acc[t][i] + bv) + rng_normal(A.rng, 2u, (uint32_t)e) * A.sigma, -A.max_a), A.max_a);
                    }
                }
            }
        }
    }
}

// ---- the same, FOUR waves per 16 envs (one 256-thread workgroup): the collect phase's actor forward is on every vector step's critical path (the env kernel of
//      a sub-batch cannot start before it) and as one wave per row block it is a chain of 26 dependent k steps of 32 MFMAs: 50 us for 2048 envs, a sixth of
//      the step.  It runs when its own sub-batch's env launch has retired, so -- unlike the update -- it finds free wave slots on every SIMD and need not be
//      a single-wave workgroup: each wave takes 64 of a hidden layer's 256 columns, the 18-wide output layer is split along k (64 each) and summed through LDS
//      in wave order.  Same draws (noise index = element index), same arithmetic per element up to the order of the output layer's four partial sums.
__global__ __launch_bounds__(256) void k_actor_rows4(PlenTd3ActorRows A) {
    __shared__ float zp[4][2][64][4];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
    const int B = A.B, b0 = blockIdx.x * RB;
    const RowBlock rb{b0, B, r, g, min(b0 + r, B - 1), lane};
    const size_t Bz = (size_t)B;
    const rsrc_t r_s = mkrs(A.state, Bz * TD3_S * 4), r_p1 = mkrs(A.p1, Bz * TD3_H * 4), r_p2 = mkrs(A.p2, Bz * TD3_H * 4);
    dense_relu<4, true>(r_s, rb.aoff(TD3_S), TD3_S, mkrs(A.a_w1, (size_t)TD3_H * TD3_S * 4), TD3_S, A.a_b1, 64 * w, r_p1, TD3_H, rb);
    __syncthreads();
    dense_relu<4, false>(r_p1, rb.aoff(TD3_H), TD3_H, mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, A.a_b2, 64 * w, r_p2, TD3_H, rb);
    __syncthreads();
    {
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        const int kb = 64 * w;
        mm_nt<2, false>(r_p2, rb.aoff(TD3_H, kb), mkrs(A.a_w3 + kb, ((size_t)TD3_A * TD3_H - kb) * 4), TD3_H, 0, 64, acc, r, g);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) zp[w][t][lane][i] = acc[t][i];
    }
    __syncthreads();
    if (w < 2) {
        const int t = w, j = 16 * t + r;
        if (j < TD3_A) {
            const float bv = A.a_b3[j];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = b0 + 4 * g + i;
                if (b < B) {
                    const float z = ((zp[0][t][lane][i] + zp[1][t][lane][i]) + zp[2][t][lane][i]) + zp[3][t][lane][i];
                    const int e = b * TD3_A + j;
                    A.action[e] = fminf(fmaxf(A.max_a * tanhf(z + bv) + rng_normal(A.rng, 2u, (uint32_t)e) * A.sigma, -A.max_a), A.max_a);
                }
            }
        }
    }
}
