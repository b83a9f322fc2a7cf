// td3_team.hip -- the row-local part of a TD3 update for SMALL batches (the reference's own recipe: batch 100, one update per env-step,
// plen_td3.py:28, :119-120; td3.py:259-356), included by td3_kernels.hip after td3_rows.hip whose matrix-core helpers it uses.
//
// Why a second shape of the same arithmetic: with one update per env-step the updates form a dependent chain (update k reads the weights
// update k - 1 wrote), so what counts is the LATENCY of one update, not its throughput.  At batch 100
//   * the layer-by-layer path is ~35 launches of library GEMMs whose 256 x 112 tiles take 31 us whatever the size: 355 us per critic update,
//     240 us per policy update (profiles/r04_td3ref_before_kernel_stats.txt);
//   * td3_rows.hip's one-wave-per-16-rows kernels put the whole chain of 13 dense layers on ONE wave per row block: 7 waves on a 256-CU chip.
// Here a row block of 16 batch rows belongs to a TEAM of 8 waves (one 512-thread workgroup, 2 waves per SIMD): every dense layer's output columns are
// split over the team (a wave owns 32 or 64 columns = NT 2 or 4 tiles of v_mfma_f32_16x16x4_f32), the twin critics run side by side on
// the two halves of the team, and layers are separated by workgroup barriers instead of launches.  Activations still go through global memory
// between layers (they are wanted there by the weight-gradient kernel anyway; the team's waves share one compute unit's vector cache, so a workgroup-scope
// barrier makes them visible).  One compute unit's matrix cores bound the kernel: 16.8 MFLOP per row block / 256 flop per cycle = 27 us.
//
// Per element the arithmetic is td3_rows.hip's (same k order inside every dot product); only the twin critics' scalar heads sum their 256 products
// in a different order (per wave, then over the four waves of a half team).

#define TEAM_NW 8
#ifdef TEAM_STAMPS            // development (scripts/gpu_td3_team_stamps.py): the clock after every barrier, parked in unused columns of workgroup 0's scratch row of t1
#define TEAM_SYNC() do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) { reinterpret_cast<unsigned long long *>(TEAM_STAMP_ROW + TD3_H + 8)[team_stamp_k++] = __builtin_readcyclecounter(); } } while (0)
#else
#define TEAM_SYNC() __syncthreads()
#endif
// ---- dense layers with COALESCED operand loads ----
// td3_rows.hip's mm_nt loads both MFMA operands straight in the instruction's layout: lane (r, g) reads 16 bytes of matrix row r, so the 16 lanes the
// vector cache serves together touch 16 different cache lines and a 1 KB wave load costs 64 tag look-ups instead of 8.  With one wave per compute unit
// (the row kernels) nobody notices; with a team of 8 the vector cache's look-up rate IS the kernel's speed (stamps: a 256 x 256 layer took 28 k cycles
// against 8 k of matrix-core time).  Here a 16-row x 32-k tile is loaded as two wave loads of 8 full 128-byte lines each (lane l: row l / 8, 16-byte
// chunk l % 8, permuted by the row so that the reads below are free of bank conflicts), parked in the wave's private 2 KB of LDS, and read back in
// MFMA layout with ds_read_b128.  Same products, same k order per output element as mm_nt.
#define TEAM_TILE 512                          // floats per staged tile (16 rows x 32 k)
#define TEAM_LDS_PER_WAVE (5 * TEAM_TILE)      // A + up to 4 column tiles of W
static __device__ __forceinline__ int team_sigma(int row) { return ((row >> 1) ^ ((row & 1) << 2)) & 7; }

template <int NT, bool KGUARD>
static __device__ __forceinline__ void mm_nt_c(rsrc_t rx, int ldx, int xcol0, rsrc_t rw, int ldw, int n0, int K, floatx4 (&acc)[NT], const RowBlock &rb, float *lds) {
    const int lane = rb.lane, r = rb.r, g = rb.g, rho = lane >> 3, c8 = lane & 7;
    const int k_lo = c8 ^ team_sigma(rho), k_hi = c8 ^ team_sigma(rho + 8);
    const uint32_t xo0 = (uint32_t)(min(rb.b0 + rho, rb.B - 1) * ldx + xcol0 + 4 * k_lo) * 4u, xo1 = (uint32_t)(min(rb.b0 + rho + 8, rb.B - 1) * ldx + xcol0 + 4 * k_hi) * 4u;
    const uint32_t wo0 = (uint32_t)((n0 + rho) * ldw + 4 * k_lo) * 4u, wo1 = (uint32_t)((n0 + rho + 8) * ldw + 4 * k_hi) * 4u;
    const int rd0 = 4 * (8 * r + (g ^ team_sigma(r))), rd1 = 4 * (8 * r + ((4 + g) ^ team_sigma(r)));
    struct Stage { floatx4 a[2], b[NT][2]; };
    auto load = [&](int k0, Stage &S) {
        S.a[0] = bload4(rx, xo0, 4u * (uint32_t)k0); S.a[1] = bload4(rx, xo1, 4u * (uint32_t)k0);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            S.b[t][0] = bload4(rw, wo0, 4u * (uint32_t)(16 * t * ldw + k0)); S.b[t][1] = bload4(rw, wo1, 4u * (uint32_t)(16 * t * ldw + k0));
        }
    };
    auto park = [&](const Stage &S) {
        *reinterpret_cast<floatx4 *>(lds + 4 * lane) = S.a[0]; *reinterpret_cast<floatx4 *>(lds + 256 + 4 * lane) = S.a[1];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            *reinterpret_cast<floatx4 *>(lds + (t + 1) * TEAM_TILE + 4 * lane) = S.b[t][0]; *reinterpret_cast<floatx4 *>(lds + (t + 1) * TEAM_TILE + 256 + 4 * lane) = S.b[t][1];
        }
    };
    auto compute = [&](int k0) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int rd = s ? rd1 : rd0;
            floatx4 fa = *reinterpret_cast<const floatx4 *>(lds + rd);
            if constexpr (KGUARD) {
#pragma unroll
                for (int j = 0; j < 4; j++) fa[j] = k0 + 16 * s + 4 * g + j < K ? fa[j] : 0.f;
            }
            floatx4 fb[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) fb[t] = *reinterpret_cast<const floatx4 *>(lds + (t + 1) * TEAM_TILE + rd);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[t][j], acc[t], 0, 0, 0);
        }
    };
    // two stages of 32 k: the next one's global loads are issued before the current one's LDS round trip and MFMAs
    Stage S0, S1;
    load(0, S0);
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += 64) {
        park(S0);
        load(k0 + 32, S1);
        __builtin_amdgcn_sched_barrier(0);
        compute(k0);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 32 < K) {
            park(S1);
            load(k0 + 64, S0);
            __builtin_amdgcn_sched_barrier(0);
            compute(k0 + 32);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// acc[t] += G[16 rows][gcol0 + (0..Kc)] * W[(0..Kc)][j0 + 16 t + (0..15)] (input gradient of a dense layer: W is the layer's [out = Kc][in] weight).
// The B operand of td3_rows.hip's mm_nn is coalesced as it stands (16 lanes read 64 contiguous bytes of one row of W), its A operand is not: here A
// goes through LDS as in mm_nt_c, B stays a direct load.  (B through LDS as well -- whole row segments per load, parked row-major -- was measured
// and lost: 32 scalar LDS reads per stage cost more than the look-ups they save.)
template <int NT, bool KGUARD>
static __device__ __forceinline__ void mm_nn_a(rsrc_t rg, int ldg, int gcol0, rsrc_t rw, int ldw, int j0, int Kc, floatx4 (&acc)[NT], const RowBlock &rb, float *lds) {
    const int lane = rb.lane, r = rb.r, g = rb.g, rho = lane >> 3, c8 = lane & 7;
    const int k_lo = c8 ^ team_sigma(rho), k_hi = c8 ^ team_sigma(rho + 8);
    const uint32_t xo0 = (uint32_t)(min(rb.b0 + rho, rb.B - 1) * ldg + gcol0 + 4 * k_lo) * 4u, xo1 = (uint32_t)(min(rb.b0 + rho + 8, rb.B - 1) * ldg + gcol0 + 4 * k_hi) * 4u;
    const int rd0 = 4 * (8 * r + (g ^ team_sigma(r))), rd1 = 4 * (8 * r + ((4 + g) ^ team_sigma(r)));
    const uint32_t woff = (uint32_t)(4 * g * ldw + j0 + r) * 4u;
    struct Stage { floatx4 a[2], b[2][NT]; };
    auto load = [&](int k0, Stage &S) {
        S.a[0] = bload4(rg, xo0, 4u * (uint32_t)k0); S.a[1] = bload4(rg, xo1, 4u * (uint32_t)k0);
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) S.b[s][t][j] = bload1(rw, woff, 4u * (uint32_t)((k0 + 16 * s + j) * ldw + 16 * t));
    };
    auto compute = [&](int k0, const Stage &S) {
        *reinterpret_cast<floatx4 *>(lds + 4 * lane) = S.a[0]; *reinterpret_cast<floatx4 *>(lds + 256 + 4 * lane) = S.a[1];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            floatx4 fa = *reinterpret_cast<const floatx4 *>(lds + (s ? rd1 : rd0));
            if constexpr (KGUARD) {
#pragma unroll
                for (int j = 0; j < 4; j++) fa[j] = k0 + 16 * s + 4 * g + j < Kc ? fa[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], S.b[s][t][j], acc[t], 0, 0, 0);
        }
    };
    Stage S0, S1;
    load(0, S0);
#pragma unroll 1
    for (int k0 = 0; k0 < Kc; k0 += 64) {
        load(k0 + 32, S1);
        __builtin_amdgcn_sched_barrier(0);
        compute(k0, S0);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 32 < Kc) {
            load(k0 + 64, S0);
            __builtin_amdgcn_sched_barrier(0);
            compute(k0 + 32, S1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Y[b0 + row][n0 + (0..16 NT)] = relu(X[:, xcol0 ...] W^T + bias) (dense_relu with the coalesced product; the bias is requested before the product)
template <int NT, bool KGUARD>
static __device__ __forceinline__ void dense_relu_c(rsrc_t rx, int ldx, int xcol0, int K, rsrc_t rw, int ldw, const float *bias, int n0, rsrc_t ry, int ldy, const RowBlock &rb, float *lds) {
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bv[t] = bias[n0 + 16 * t + rb.r];
    floatx4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = floatx4{0, 0, 0, 0};
    mm_nt_c<NT, KGUARD>(rx, ldx, xcol0, rw, ldw, n0, K, acc, rb, lds);
    const uint32_t yoff = rb.soff(ldy);
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) bstore_row(fmaxf(acc[t][i] + bv[t], 0.f), ry, yoff, i, ldy, n0 + 16 * t);
}

// partial head of one critic over the 64 columns [n0, n0 + 64) of its second hidden layer: part[i] (rows 4 g + i) = sum_cols relu(X W2^T + b2) w3
template <bool STORE>
static __device__ __forceinline__ void team_l2_part(rsrc_t rx, int ldx, int xcol0, const float *W2, const float *b2, const float *w3, int n0, rsrc_t rh, int ldh, int hcol0,
                                                    const RowBlock &rb, float *lds, float (&q)[4]) {
    const rsrc_t rw = mkrs(W2, (size_t)TD3_H * TD3_H * 4);
    const uint32_t hoff = rb.soff(ldh);
    float bv[4], wv[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { bv[t] = b2[n0 + 16 * t + rb.r]; wv[t] = w3[n0 + 16 * t + rb.r]; }
    floatx4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = floatx4{0, 0, 0, 0};
    mm_nt_c<4, false>(rx, ldx, xcol0, rw, TD3_H, n0, TD3_H, acc, rb, lds);
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float v = fmaxf(acc[t][i] + bv[t], 0.f);
            part[i] += v * wv[t];
            if constexpr (STORE) bstore_row(v, rh, hoff, i, ldh, hcol0 + n0 + 16 * t);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float v = part[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
        q[i] = v;
    }
}

// ---- td3.py:277-323 for 16 batch rows per workgroup (arguments and outputs exactly as k_critic_rows / plentd3_critic_rows) ----
__global__ __launch_bounds__(64 * TEAM_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_critic_team(PlenTd3CriticRows A) {
    __shared__ float qp[4][4][RB];              // [target a, target b, critic a, critic b][wave of the half team][row]: partial heads
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), wq = w & 3, half = w >> 2;
    const int B = A.B, b0 = blockIdx.x * RB;
#ifdef TEAM_STAMPS
#define TEAM_STAMP_ROW A.t1
    int team_stamp_k = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long *>(A.t1 + TD3_H + 8)[0] = __builtin_readcyclecounter();
#endif
    __shared__ float team_lds[TEAM_NW][TEAM_LDS_PER_WAVE];
    float *lds = team_lds[w];
    // ---- sample 16 rows of the replay ring (as k_critic_rows: every wave derives the same 16 indices) and gather two of them per wave ----
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
#pragma unroll
        for (int i = w; i < RB; i += TEAM_NW) {
            const int b = b0 + i;
            if (b >= B) break;
            const int64_t id = ((int64_t)__shfl(hi, i) << 32) | (uint32_t)__shfl(lo, i);
            const float *src = A.data + (size_t)id * TD3_ROW;
            const float v0 = src[lane], v1 = lane < TD3_ROW - 64 ? src[64 + lane] : 0.f;
            float *dst = A.batch + (size_t)b * TD3_ROW;
            dst[lane] = v0;
            if (lane < TD3_ROW - 64) dst[64 + lane] = v1;
            if (lane < TD3_S) A.sa_pi[(size_t)b * TD3_SA + lane] = v0;
            if (lane >= TD3_SA) A.sa2[(size_t)b * TD3_SA + lane - TD3_SA] = v0;                  // s2 = columns 44..69 of the row: 44..63 here,
            if (lane < TD3_SA + TD3_S - 64) A.sa2[(size_t)b * TD3_SA + 64 - TD3_SA + lane] = v1;  // 64..69 there
        }
    }
    TEAM_SYNC();
    const size_t Bz = (size_t)B;
    const rsrc_t r_batch = mkrs(A.batch, Bz * TD3_ROW * 4), r_t0 = mkrs(A.t0, Bz * 2 * TD3_H * 4), r_t1 = mkrs(A.t1, Bz * 2 * TD3_H * 4), r_sa2 = mkrs(A.sa2, Bz * TD3_SA * 4);
    const rsrc_t r_c1 = mkrs(A.c1, Bz * 2 * TD3_H * 4), r_c2 = mkrs(A.c2, Bz * 2 * TD3_H * 4);
    // ---- target actor's hidden layers (32 columns per wave) and, independent of them, the critics' stacked first layers on (s, a) (64 per wave) ----
    {
        PHASE();
        dense_relu_c<2, true>(r_batch, TD3_ROW, TD3_SA, TD3_S, mkrs(A.at_w1, (size_t)TD3_H * TD3_S * 4), TD3_S, A.at_b1, 32 * w, r_t0, 2 * TD3_H, rb, lds);
        dense_relu_c<4, true>(r_batch, TD3_ROW, 0, TD3_SA, mkrs(A.c_w14, (size_t)2 * TD3_H * TD3_SA * 4), TD3_SA, A.c_b14, 64 * w, r_c1, 2 * TD3_H, rb, lds);
    }
    TEAM_SYNC();
    {
        PHASE();
        dense_relu_c<2, false>(r_t0, 2 * TD3_H, 0, TD3_H, mkrs(A.at_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, A.at_b2, 32 * w, r_t1, 2 * TD3_H, rb, lds);
    }
    TEAM_SYNC();
    // ---- target action (td3.py:299-304): the 18-wide output layer split over the team along k (32 each), summed through LDS in wave order ----
    {
        PHASE();
        __shared__ float zp[TEAM_NW][2][64][4];
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        const int kb = 32 * w;
        // the smoothing noise of this lane's four actions (waves 0, 1: column 16 w + r) and the output bias: drawn / requested before the product
        float noise[4] = {0.f, 0.f, 0.f, 0.f}, bv3 = 0.f;
        if (w < 2 && 16 * w + r < TD3_A) {
            bv3 = A.at_b3[16 * w + r];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = min(b0 + 4 * g + i, B - 1);
                noise[i] = fminf(fmaxf(rng_normal(A.rng, 1u, (uint32_t)(b * TD3_A + 16 * w + r)) * A.sigma, -A.clip), A.clip);
            }
        }
        mm_nt_c<2, false>(r_t1, 2 * TD3_H, kb, mkrs(A.at_w3 + kb, ((size_t)TD3_A * TD3_H - kb) * 4), TD3_H, 0, 32, acc, rb, lds);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) zp[w][t][lane][i] = acc[t][i];
        TEAM_SYNC();
        if (w < 2) {
            const int t = w, j = 16 * t + r;
            floatx4 z = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < TEAM_NW; k++)
#pragma unroll
                for (int i = 0; i < 4; i++) z[i] += zp[k][t][lane][i];
            if (j < TD3_A) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int b = b0 + 4 * g + i;
                    if (b < B) A.sa2[(size_t)b * TD3_SA + TD3_S + j] = fminf(fmaxf(A.max_a * tanhf(z[i] + bv3) + noise[i], -A.max_a), A.max_a);
                }
            }
        }
    }
    TEAM_SYNC();
    // ---- both target critics' first layers stacked (64 columns per wave) ----
    {
        PHASE();
        dense_relu_c<4, true>(r_sa2, TD3_SA, 0, TD3_SA, mkrs(A.ct_w14, (size_t)2 * TD3_H * TD3_SA * 4), TD3_SA, A.ct_b14, 64 * w, r_t0, 2 * TD3_H, rb, lds);
    }
    TEAM_SYNC();
    // ---- second layers + heads: waves 0..3 = critic a, 4..7 = critic b, 64 columns each; target critics, then the critics themselves (c2 stored) ----
    {
        PHASE();
        float q[4];
        team_l2_part<false>(r_t0, 2 * TD3_H, half * TD3_H, half ? A.ct_w5 : A.ct_w2, half ? A.ct_b5 : A.ct_b2, half ? A.ct_w6 : A.ct_w3, 64 * wq, r_t0, 0, 0, rb, lds, q);
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) qp[half][wq][4 * g + i] = q[i];
        }
        team_l2_part<true>(r_c1, 2 * TD3_H, half * TD3_H, half ? A.c_w5 : A.c_w2, half ? A.c_b5 : A.c_b2, half ? A.c_w6 : A.c_w3, 64 * wq, r_c2, 2 * TD3_H, half * TD3_H, rb, lds, q);
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) qp[2 + half][wq][4 * g + i] = q[i];
        }
    }
    TEAM_SYNC();
    // ---- clipped double-Q target, loss and its gradient at the heads (td3.py:306-319): one lane per row ----
    if (w == 0) {
        PHASE();
        float lsum = 0.f, ga = 0.f, gb = 0.f;
        const int b = b0 + lane;
        if (lane < RB && b < B) {
            auto head = [&](int k, const float *b3) { return ((qp[k][0][lane] + qp[k][1][lane]) + (qp[k][2][lane] + qp[k][3][lane])) + b3[0]; };
            const float *row = A.batch + (size_t)b * TD3_ROW;
            const float y = row[TD3_ROW - 2] + row[TD3_ROW - 1] * A.gamma * fminf(head(0, A.ct_b3), head(1, A.ct_b6));
            const float inv = 1.f / (float)B;
            const float ea = head(2, A.c_b3) - y, eb = head(3, A.c_b6) - y;
            ga = 2.f * ea * inv; gb = 2.f * eb * inv;
            A.dq[2 * b] = ga; A.dq[2 * b + 1] = gb;
            lsum = ea * ea * inv + eb * eb * inv;
        }
        lsum = wave_sum(lsum); ga = wave_sum(ga); gb = wave_sum(gb);
        // the loss and the two head biases' gradients: one partial per workgroup, parked in unused columns of the workgroup's own scratch row and summed
        // in order by the last workgroup to finish (below): the sums do not depend on the finishing order (same bits every run), and loss[0] is a
        // plain store that nobody has to zero first
        if (lane == 0) { float *park = A.t1 + (size_t)b0 * 2 * TD3_H + TD3_H; park[0] = lsum; park[1] = ga; park[2] = gb; }
    }
    TEAM_SYNC();
    // ---- dh2 = dq (x) w3 where the hidden unit was active: two rows per wave, 4 columns per lane and critic ----
    {
        PHASE();
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const floatx4 wv = *reinterpret_cast<const floatx4 *>((c ? A.c_w6 : A.c_w3) + 4 * lane);
#pragma unroll
            for (int i = w; i < RB; i += TEAM_NW) {
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
    TEAM_SYNC();
    // ---- dh1_c = (dh2_c W2_c) where c1_c was active: half a team per critic, 64 columns per wave ----
    {
        PHASE();
        const rsrc_t r_dh2 = mkrs(A.dh2, Bz * 2 * TD3_H * 4), r_dh1 = mkrs(A.dh1, Bz * 2 * TD3_H * 4);
        const uint32_t ooff = rb.soff(2 * TD3_H);
        const rsrc_t rw = mkrs(half ? A.c_w5 : A.c_w2, (size_t)TD3_H * TD3_H * 4);
        const int j0 = 64 * wq;
        floatx4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = floatx4{0, 0, 0, 0};
        mm_nn_a<4, false>(r_dh2, 2 * TD3_H, half * TD3_H, rw, TD3_H, j0, TD3_H, acc, rb, lds);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int col = half * TD3_H + j0 + 16 * t;
                bstore_row(bload_row(r_c1, ooff, i, 2 * TD3_H, col) > 0.f ? acc[t][i] : 0.f, r_dh1, ooff, i, 2 * TD3_H, col);
            }
    }
    // the last workgroup to finish advances the random stream's call counter: every wave has read it by then
    TEAM_SYNC();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(A.done_count, 1) == (int)gridDim.x - 1) {
            __threadfence();
            float l = 0.f, ga = 0.f, gb = 0.f;
            for (int k = 0; k < (int)gridDim.x; k++) {
                const float *park = A.t1 + (size_t)k * RB * 2 * TD3_H + TD3_H;
                l += __hip_atomic_load(park, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ga += __hip_atomic_load(park + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gb += __hip_atomic_load(park + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            A.loss[0] = l; A.db3a[0] += ga; A.db3b[0] += gb;
            A.done_count[0] = 0;
            if (A.rng_bump) A.rng_bump[1] += 1;
        }
    }
}

#ifdef TEAM_STAMPS            // (stamps exist in the critic kernel only)
#undef TEAM_SYNC
#define TEAM_SYNC() __syncthreads()
#endif
// ---- td3.py:334-341 for 16 batch rows per workgroup (arguments and outputs exactly as k_policy_rows / plentd3_policy_rows): 32 columns per wave ----
__global__ __launch_bounds__(64 * TEAM_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_policy_team(PlenTd3PolicyRows A) {
    __shared__ float zp[TEAM_NW][2][64][4];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int B = A.B, b0 = blockIdx.x * RB;
    const size_t Bz = (size_t)B;
    const rsrc_t r_sa = mkrs(A.sa_pi, Bz * TD3_SA * 4), r_p1 = mkrs(A.p1, Bz * TD3_H * 4), r_p2 = mkrs(A.p2, Bz * TD3_H * 4), r_g1 = mkrs(A.g1, Bz * TD3_H * 4);
    const rsrc_t r_dg2 = mkrs(A.dg2, Bz * TD3_H * 4), r_dg1 = mkrs(A.dg1, Bz * TD3_H * 4), r_dp2 = mkrs(A.dp2, Bz * TD3_H * 4), r_dp1 = mkrs(A.dp1, Bz * TD3_H * 4);
    const rsrc_t r_dz = mkrs(A.dz, Bz * TD3_A * 4);
    const int n0 = 32 * w;
    __shared__ float team_lds[TEAM_NW][3 * TEAM_TILE];          // A + two column tiles of W
    float *lds = team_lds[w];
    // an 18-wide result of a 256-long product, split over the team along k and summed through LDS in wave order: z[i] valid in waves 0, 1 (tile t = w)
    auto team_reduce = [&](const floatx4 (&acc)[2], int lane, floatx4 &z) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) zp[w][t][lane][i] = acc[t][i];
        TEAM_SYNC();
        z = floatx4{0, 0, 0, 0};
        if (w < 2) {
#pragma unroll
            for (int k = 0; k < TEAM_NW; k++)
#pragma unroll
                for (int i = 0; i < 4; i++) z[i] += zp[k][w][lane][i];
        }
    };
    {
        PHASE();
        dense_relu_c<2, true>(r_sa, TD3_SA, 0, TD3_S, mkrs(A.a_w1, (size_t)TD3_H * TD3_S * 4), TD3_S, A.a_b1, n0, r_p1, TD3_H, rb, lds);
    }
    TEAM_SYNC();
    {
        PHASE();
        dense_relu_c<2, false>(r_p1, TD3_H, 0, TD3_H, mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, A.a_b2, n0, r_p2, TD3_H, rb, lds);
    }
    TEAM_SYNC();
    {
        PHASE();
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}}, z;
        mm_nt_c<2, false>(r_p2, TD3_H, n0, mkrs(A.a_w3 + n0, ((size_t)TD3_A * TD3_H - n0) * 4), TD3_H, 0, 32, acc, rb, lds);
        team_reduce(acc, lane, z);
        const int j = 16 * w + r;
        if (w < 2 && j < TD3_A) {
            const float bv = A.a_b3[j];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = b0 + 4 * g + i;
                if (b < B) {
                    const float a = A.max_a * tanhf(z[i] + bv);                                   // td3.py:57
                    A.a_pi[(size_t)b * TD3_A + j] = a;
                    A.sa_pi[(size_t)b * TD3_SA + TD3_S + j] = a;
                }
            }
        }
    }
    TEAM_SYNC();
    // critic.Q1 forward and the gradient of -mean Q1 at its second hidden layer: dg2 = -(1/B) w3 (g2 > 0)
    {
        PHASE();
        dense_relu_c<2, true>(r_sa, TD3_SA, 0, TD3_SA, mkrs(A.c_w1, (size_t)TD3_H * TD3_SA * 4), TD3_SA, A.c_b1, n0, r_g1, TD3_H, rb, lds);
    }
    TEAM_SYNC();
    {
        PHASE();
        const uint32_t hoff = rb.soff(TD3_H);
        const float ginv = -1.f / (float)B;
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        float bv[2], wv[2];
#pragma unroll
        for (int t = 0; t < 2; t++) { bv[t] = A.c_b2[n0 + 16 * t + r]; wv[t] = ginv * A.c_w3[n0 + 16 * t + r]; }
        mm_nt_c<2, false>(r_g1, TD3_H, 0, mkrs(A.c_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, n0, TD3_H, acc, rb, lds);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) bstore_row(acc[t][i] + bv[t] > 0.f ? wv[t] : 0.f, r_dg2, hoff, i, TD3_H, n0 + 16 * t);
    }
    TEAM_SYNC();
    // dg1 = (dg2 W2)(g1 > 0)
    {
        PHASE();
        const uint32_t hoff = rb.soff(TD3_H);
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        mm_nn_a<2, false>(r_dg2, TD3_H, 0, mkrs(A.c_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, n0, TD3_H, acc, rb, lds);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) bstore_row(bload_row(r_g1, hoff, i, TD3_H, n0 + 16 * t) > 0.f ? acc[t][i] : 0.f, r_dg1, hoff, i, TD3_H, n0 + 16 * t);
    }
    TEAM_SYNC();
    // d/d action = (dg1 W1)[:, 26:44] (rows 32 w .. 32 w + 31 of W1 per wave), through the tanh: dz = that * (max_a - a^2 / max_a)
    {
        PHASE();
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}}, z;
        mm_nn_a<2, false>(r_dg1, TD3_H, n0, mkrs(A.c_w1 + (size_t)n0 * TD3_SA, ((size_t)TD3_H - n0) * TD3_SA * 4), TD3_SA, TD3_S, 32, acc, rb, lds);
        team_reduce(acc, lane, z);
        const int j = 16 * w + r;
        if (w < 2 && j < TD3_A) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = b0 + 4 * g + i;
                if (b < B) {
                    const float a = A.a_pi[(size_t)b * TD3_A + j];
                    A.dz[(size_t)b * TD3_A + j] = z[i] * (A.max_a - a * a / A.max_a);
                }
            }
        }
    }
    TEAM_SYNC();
    // back through the actor: dp2 = (dz W3)(p2 > 0), dp1 = (dp2 W2)(p1 > 0)
    {
        PHASE();
        const uint32_t hoff = rb.soff(TD3_H);
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        mm_nn_a<2, true>(r_dz, TD3_A, 0, mkrs(A.a_w3, (size_t)TD3_A * TD3_H * 4), TD3_H, n0, TD3_A, acc, rb, lds);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) bstore_row(bload_row(r_p2, hoff, i, TD3_H, n0 + 16 * t) > 0.f ? acc[t][i] : 0.f, r_dp2, hoff, i, TD3_H, n0 + 16 * t);
    }
    TEAM_SYNC();
    {
        PHASE();
        const uint32_t hoff = rb.soff(TD3_H);
        floatx4 acc[2] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
        mm_nn_a<2, false>(r_dp2, TD3_H, 0, mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, n0, TD3_H, acc, rb, lds);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) bstore_row(bload_row(r_p1, hoff, i, TD3_H, n0 + 16 * t) > 0.f ? acc[t][i] : 0.f, r_dp1, hoff, i, TD3_H, n0 + 16 * t);
    }
}
