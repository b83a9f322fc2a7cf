// td3_block.hip -- the row-local part of a TD3 update for LARGE batches (BASELINE.json configs[2]: batch 4096 beside 4096 envs; td3.py:259-356),
// included by td3_kernels.hip after td3_team.hip (buffer helpers of td3_rows.hip, the LDS-only barrier of td3_team.hip).  C ABI: include/plentd3.h.
//
// Why a third shape of the same arithmetic.  td3_rows.hip gives a block of 16 batch rows to ONE wave: 256 waves at batch 4096, i.e. one of the four
// SIMDs of every compute unit, a chain of 13 dense layers per wave with activations round-tripping through global memory and weights fetched as 16
// scattered 64-byte pieces per 16 lanes: 272 us = 0.10 of the fp32 matrix peak (profiles/r05_td3leg_before_*).  td3_team.hip is built for latency at
// batch 100 (4 rows per workgroup: every workgroup streams all 1.5 MB of weights for 4 rows of work).  Here:
//   * a block of 16 batch rows belongs to a workgroup of FOUR waves, one per SIMD: 256 workgroups at batch 4096 = one per compute unit, all 1024
//     matrix pipes busy.  Every dense layer's OUTPUT FEATURES are split over the four waves (64 or 128 each);
//   * the product is formed TRANSPOSED, Y^T = W X^T, with v_mfma_f32_16x16x4_f32: A = a 16-feature x 4-k piece of W, B = X^T (4 k x 16 batch rows),
//     D = 16 features x 16 batch rows.  A lane of D then holds FOUR CONSECUTIVE FEATURES of ONE batch row: the epilogue (bias, ReLU, mask) writes them
//     as one 16-byte store into the block's row-major activations in LDS, which is exactly what the next layer reads back as its B operand
//     (one ds_read_b128 per lane and 16 k).  Activations never leave LDS between layers; what the weight-gradient kernels need is copied to global
//     memory from LDS as whole 1-KB row segments, asynchronously;
//   * the weights come PRE-PACKED in MFMA operand order (plentd3_pack, run once after each optimiser step: 3 us): the A operand of (feature tile t,
//     k step s) is 64 lanes x 16 bytes = ONE contiguous kilobyte, so a wave's load is a single fully coalesced request instead of 16 lines per quarter
//     wave -- the tag look-up rate of the vector cache, which bounds td3_team.hip's weight stream, is out of the picture, and each weight element
//     is fetched exactly once per compute unit and pass (1.5 MB per workgroup from L2: 32 B per clock and compute unit at matrix-pipe speed);
//   * LDS activations are swizzled (the 16-byte chunk index is XOR-ed with the batch row) so that the B reads -- 16 rows x 4 k-groups per
//     instruction -- and the epilogue's stores are free of bank conflicts with a row stride that is a multiple of 64 dwords.
// Sums: a dot product adds its k values in ascending order per (k % 4) class inside the matrix pipe (the MFMA's own order, one accumulator per tile);
// the critics' heads add per-lane partials over the wave's tiles, then the four k-groups, then the waves in wave order; loss and head-bias gradients add
// per-workgroup partials in workgroup order (last workgroup to arrive).  Nothing depends on timing: same bits every run.

#define BLK_R 16                                // batch rows per workgroup
#define BLK_NW 4                                // waves per workgroup
// LDS activation buffers (floats per row; every stride a multiple of 64 so that the XOR swizzle stays inside one 64-float group)
#define BLK_LD_ROW 128                          // the gathered replay row (72 used, 72..79 zero)
#define BLK_LD_SA2 64                           // [s2 | target action] (44 used, 44..47 zero)
#define BLK_LD_W 512                            // the two wide buffers U and V

struct Blk { int lane, r, g, w; };
static __device__ __forceinline__ Blk blk_ids() {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return Blk{lane, lane & 15, lane >> 4, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)};
}
// address (in floats) of 16-byte chunk q of row r of an LDS activation buffer with row stride ld
static __device__ __forceinline__ int blk_at(int ld, int r, int q) { return r * ld + 4 * (q ^ r); }
static __device__ __forceinline__ int blk_at1(int ld, int r, int c) { return r * ld + 4 * ((c >> 2) ^ r) + (c & 3); }

// acc[t] (features 16 (tile0 + t) + 4 g + v, batch row r) += sum_k W[feature][k] X[r][k] over KS steps of 16 k.
// wp: the layer's packed weights (plentd3_pack: float4 index ((tile KS + s) 64 + lane)); x: row 0 of the LDS input, chunk q0 = its first column / 4.
template <int NT, int KS>
static __device__ __forceinline__ void blk_mm(rsrc_t wp, int tile0, const float *x, int ld, int q0, floatx4 (&acc)[NT], const Blk &k) {
    const uint32_t voff = (uint32_t)k.lane * 16u;
    const float *xr = x + k.r * ld;
    auto loadA = [&](int s, floatx4 (&a)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; t++) a[t] = bload4(wp, voff, (uint32_t)(((tile0 + t) * KS + s) * 1024));
    };
    auto loadB = [&](int s) { return *reinterpret_cast<const floatx4 *>(xr + 4 * ((q0 + 4 * s + k.g) ^ k.r)); };
    // three stages in flight: the loads of step s + 2 are issued before the MFMAs of step s (an L2 hit is ~500 cycles under load, a step of NT = 4 tiles
    // is 512 cycles of matrix pipe); left alone the compiler sinks the loads to their first use
    floatx4 a[3][NT], b[3];
    loadA(0, a[0]); b[0] = loadB(0);
    if (KS > 1) { loadA(1, a[1]); b[1] = loadB(1); }
#pragma unroll
    for (int s = 0; s < KS; s++) {
        if (s + 2 < KS) { loadA(s + 2, a[(s + 2) % 3]); b[(s + 2) % 3] = loadB(s + 2); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s % 3][t][v], b[s % 3][v], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT>
static __device__ __forceinline__ void blk_zero(floatx4 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = floatx4{0, 0, 0, 0};
}

// NT tiles of a hidden layer: relu(W x + bias) into the LDS buffer y (row stride ldy, first chunk qy0 = column of feature 0 / 4)
template <int NT, int KS>
static __device__ __forceinline__ void blk_dense_relu(rsrc_t wp, int tile0, const float *x, int ldx, int qx0, const float *bias, float *y, int ldy, int qy0, const Blk &k) {
    floatx4 acc[NT];
    blk_zero(acc);
    blk_mm<NT, KS>(wp, tile0, x, ldx, qx0, acc, k);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int f = 16 * (tile0 + t) + 4 * k.g;
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = fmaxf(acc[t][i] + bias[f + i], 0.f);
        *reinterpret_cast<floatx4 *>(y + blk_at(ldy, k.r, qy0 + 4 * (tile0 + t) + k.g)) = v;
    }
}

// NT tiles of a critic's second layer and their share of its scalar head: part += sum over the lane's features of relu(W2 x + b2) w3; the activations go
// to LDS (y) when STORE
template <int NT, bool STORE>
static __device__ __forceinline__ float blk_l2_head(rsrc_t wp, int tile0, const float *x, int ldx, int qx0, const float *b2, const float *w3, float *y, int ldy, int qy0, const Blk &k) {
    floatx4 acc[NT];
    blk_zero(acc);
    blk_mm<NT, 16>(wp, tile0, x, ldx, qx0, acc, k);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int f = 16 * (tile0 + t) + 4 * k.g;
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = fmaxf(acc[t][i] + b2[f + i], 0.f); part += v[i] * w3[f + i]; }
        if constexpr (STORE) *reinterpret_cast<floatx4 *>(y + blk_at(ldy, k.r, qy0 + 4 * (tile0 + t) + k.g)) = v;
    }
    return part;
}
// the four k-groups' partials of batch row r (lanes r, r + 16, r + 32, r + 48), in group order
static __device__ __forceinline__ float blk_rowsum(float part) {
    const float p1 = __shfl_xor(part, 16);
    const float s01 = part + p1;                 // (g, g ^ 1)
    return s01 + __shfl_xor(s01, 32);
}

// copy nq 16-byte chunks per row (from chunk q0) of an LDS buffer to the row-major global matrix Y [B][ldy] at column col0: whole row segments per wave
static __device__ __forceinline__ void blk_flush(const float *x, int ld, int q0, int nq, float *Y, int ldy, int col0, int b0, int B) {
    for (int i = threadIdx.x; i < BLK_R * nq; i += 64 * BLK_NW) {
        const int row = i / nq, q = i - row * nq;
        if (b0 + row < B) *reinterpret_cast<floatx4 *>(Y + (size_t)(b0 + row) * ldy + col0 + 4 * q) = *reinterpret_cast<const floatx4 *>(x + blk_at(ld, row, q0 + q));
    }
}

// ---- weight packing: M (N x K; element (i, k) at src[i rs + k cs]) -> MFMA A-operand order, zero-padded to 16-row tiles and 16-k steps:
//      dst float4 ((t KS + s) 64 + lane) = M[16 t + (lane % 16)][16 s + 4 (lane / 16) + (0..3)]
__global__ __launch_bounds__(256) void k_pack(PlenTd3PackGroup G) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < PLENTD3_PACK_JOBS; k++) j += (k < G.n_jobs && e >= G.job[k].f4_0) ? 1 : 0;
    const PlenTd3PackJob &J = G.job[j];
    const int KS = (J.K + 15) / 16, T = (J.N + 15) / 16;
    const int el = e - J.f4_0;
    if (el >= T * KS * 64) return;
    const int lane = el & 63, s = (el >> 6) % KS, t = (el >> 6) / KS;
    const int i = 16 * t + (lane & 15), k0 = 16 * s + 4 * (lane >> 4);
    floatx4 v;
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = (i < J.N && k0 + u < J.K) ? J.src[(size_t)i * J.rs + (size_t)(k0 + u) * J.cs] : 0.f;
    reinterpret_cast<floatx4 *>(J.dst)[el] = v;
}

// ---- td3.py:277-323 for 16 batch rows per workgroup (arguments and outputs as k_critic_rows; t0, t1, sa2 are not written) ----
__global__ __launch_bounds__(64 * BLK_NW) void k_critic_block(PlenTd3CriticBlock P) {
    const PlenTd3CriticRows &A = P.rows;
    __shared__ __attribute__((aligned(16))) float Rb[BLK_R * BLK_LD_ROW];
    __shared__ __attribute__((aligned(16))) float Sb[BLK_R * BLK_LD_SA2];
    __shared__ __attribute__((aligned(16))) float Ub[BLK_R * BLK_LD_W];
    __shared__ __attribute__((aligned(16))) float Vb[BLK_R * BLK_LD_W];
    __shared__ float qp[4][2][BLK_R];            // [target a, target b, critic a, critic b][wave of the pair][row]: partial heads
    __shared__ float dql[BLK_R][2];
    const int B = A.B, b0 = blockIdx.x * BLK_R, n_blk = (B + BLK_R - 1) / BLK_R;
    // ---- sample the block's 16 rows of the replay ring (td3.py:166-193; same draw as k_sample_gather / k_critic_rows) and gather them: 4 rows per wave ----
    {
        const Blk k = blk_ids();
        int64_t id = 0;
        if (k.lane < 4) {
            const int b = min(b0 + 4 * k.w + k.lane, B - 1);
            if (A.idx) id = A.idx[b];
            else {
                const int64_t tot = A.total[0];
                int64_t filled, start;
                if (tot + A.guard <= A.capacity) { filled = tot; start = 0; }
                else { filled = A.capacity - A.guard; start = (tot + A.guard) % A.capacity; }
                const float ub = rng_uniform(A.rng, 0u, (uint32_t)b);
                int64_t i = (int64_t)((double)ub * (double)filled);
                i = i < filled - 1 ? i : filled - 1;
                i = i > 0 ? i : 0;
                id = (start + i) % A.capacity;
            }
        }
        const int lo = (int)(id & 0xffffffff), hi = (int)(id >> 32);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = 4 * k.w + i, b = b0 + row;
            const int64_t rid = ((int64_t)__shfl(hi, i) << 32) | (uint32_t)__shfl(lo, i);
            const float *src = A.data + (size_t)rid * TD3_ROW;
            const float v0 = src[k.lane], v1 = k.lane < TD3_ROW - 64 ? src[64 + k.lane] : 0.f;
            Rb[blk_at1(BLK_LD_ROW, row, k.lane)] = v0;
            if (k.lane < 16) Rb[blk_at1(BLK_LD_ROW, row, 64 + k.lane)] = v1;                     // 64..71 data, 72..79 zero (the padded k of the layers that read s2)
            if (k.lane >= TD3_SA) Sb[blk_at1(BLK_LD_SA2, row, k.lane - TD3_SA)] = v0;             // s2 = columns 44..69: 44..63 here,
            if (k.lane < TD3_SA + TD3_S - 64) Sb[blk_at1(BLK_LD_SA2, row, 64 - TD3_SA + k.lane)] = v1;   // 64..69 there
            if (k.lane >= TD3_SA && k.lane < TD3_SA + 4) Sb[blk_at1(BLK_LD_SA2, row, k.lane)] = 0.f;   // columns 44..47: padded k
            if (b < B) {
                float *dst = A.batch + (size_t)b * TD3_ROW;
                dst[k.lane] = v0;
                if (k.lane < TD3_ROW - 64) dst[64 + k.lane] = v1;
                if (k.lane < TD3_S) A.sa_pi[(size_t)b * TD3_SA + k.lane] = v0;
            }
        }
    }
    TEAM_LDS_BARRIER();
    // gathered rows: s 0..25 | a 26..43 | s2 44..69 | r 70 | not_done 71
    // ---- target actor (td3.py:299): 64 features per wave and layer ----
    {
        const Blk k = blk_ids();
        blk_dense_relu<4, 2>(mkrs(P.p_at_w1, (size_t)16 * 2 * 1024), 4 * k.w, Rb, BLK_LD_ROW, TD3_SA / 4, A.at_b1, Ub, BLK_LD_W, 0, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids();
        blk_dense_relu<4, 16>(mkrs(P.p_at_w2, (size_t)16 * 16 * 1024), 4 * k.w, Ub, BLK_LD_W, 0, A.at_b2, Vb, BLK_LD_W, 0, k);
    }
    TEAM_LDS_BARRIER();
    // ---- target action (td3.py:299-304): the 18-wide output layer is two tiles: wave 0.  Beside it, on the other three waves, the critics' stacked first
    //      layers on (s, a) -- 32 tiles that depend on the gathered rows only (c1 -> U: the target actor's first layer is dead) ----
    {
        const Blk k = blk_ids();
        if (k.w == 0) {
            floatx4 acc[2];
            blk_zero(acc);
            blk_mm<2, 16>(mkrs(P.p_at_w3, (size_t)2 * 16 * 1024), 0, Vb, BLK_LD_W, 0, acc, k);
            const int b = min(b0 + k.r, B - 1);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = 16 * t + 4 * k.g + i;
                    if (j < TD3_A) {
                        const int e = b * TD3_A + j;
                        const float z = A.noise ? A.noise[e] : rng_normal(A.rng, 1u, (uint32_t)e);                 // torch.randn_like(action), td3.py:300
                        const float n = fminf(fmaxf(z * A.sigma, -A.clip), A.clip);
                        Sb[blk_at1(BLK_LD_SA2, k.r, TD3_S + j)] = fminf(fmaxf(A.max_a * tanhf(acc[t][i] + A.at_b3[j]) + n, -A.max_a), A.max_a);
                    }
                }
        } else {
            const rsrc_t wp = mkrs(P.p_c_w14, (size_t)32 * 3 * 1024);
            // 32 tiles over three waves: 11, 11, 10 (as 4 + 4 + 3 / 4 + 4 + 2)
            const int t0 = 11 * (k.w - 1), nt = k.w == 3 ? 10 : 11;
            blk_dense_relu<4, 3>(wp, t0, Rb, BLK_LD_ROW, 0, A.c_b14, Ub, BLK_LD_W, 0, k);
            blk_dense_relu<4, 3>(wp, t0 + 4, Rb, BLK_LD_ROW, 0, A.c_b14, Ub, BLK_LD_W, 0, k);
            if (nt == 11) blk_dense_relu<3, 3>(wp, t0 + 8, Rb, BLK_LD_ROW, 0, A.c_b14, Ub, BLK_LD_W, 0, k);
            else blk_dense_relu<2, 3>(wp, t0 + 8, Rb, BLK_LD_ROW, 0, A.c_b14, Ub, BLK_LD_W, 0, k);
        }
    }
    TEAM_LDS_BARRIER();
    blk_flush(Ub, BLK_LD_W, 0, 2 * TD3_H / 4, A.c1, 2 * TD3_H, 0, b0, B);          // c1 for the weight gradients (asynchronous: nothing below reads it back)
    // ---- both target critics' first layers stacked (W14 = [fc1.w; fc4.w]) on (s2, a2): 128 features per wave -> V (the target actor's second layer is dead) ----
    {
        const Blk k = blk_ids();
        const rsrc_t wp = mkrs(P.p_ct_w14, (size_t)32 * 3 * 1024);
        blk_dense_relu<4, 3>(wp, 8 * k.w, Sb, BLK_LD_SA2, 0, A.ct_b14, Vb, BLK_LD_W, 0, k);
        blk_dense_relu<4, 3>(wp, 8 * k.w + 4, Sb, BLK_LD_SA2, 0, A.ct_b14, Vb, BLK_LD_W, 0, k);
    }
    TEAM_LDS_BARRIER();
    // ---- target critics' second layers + heads (td3.py:306-309): waves 0, 1 = critic a, 2, 3 = critic b, 128 features each ----
    {
        const Blk k = blk_ids();
        const int c = k.w >> 1, h = k.w & 1;
        const rsrc_t wp = mkrs(c ? P.p_ct_w5 : P.p_ct_w2, (size_t)16 * 16 * 1024);
        const float *b2 = c ? A.ct_b5 : A.ct_b2, *w3 = c ? A.ct_w6 : A.ct_w3;
        float part = blk_l2_head<4, false>(wp, 8 * h, Vb, BLK_LD_W, 64 * c, b2, w3, nullptr, 0, 0, k);
        part += blk_l2_head<4, false>(wp, 8 * h + 4, Vb, BLK_LD_W, 64 * c, b2, w3, nullptr, 0, 0, k);
        part = blk_rowsum(part);
        if (k.lane < BLK_R) qp[c][h][k.lane] = part;
    }
    TEAM_LDS_BARRIER();
    // ---- the critics' second layers + heads (td3.py:312): c1 in U -> c2 in V (the target critics' first layers are dead) ----
    {
        const Blk k = blk_ids();
        const int c = k.w >> 1, h = k.w & 1;
        const rsrc_t wp = mkrs(c ? P.p_c_w5 : P.p_c_w2, (size_t)16 * 16 * 1024);
        const float *b2 = c ? A.c_b5 : A.c_b2, *w3 = c ? A.c_w6 : A.c_w3;
        float part = blk_l2_head<4, true>(wp, 8 * h, Ub, BLK_LD_W, 64 * c, b2, w3, Vb, BLK_LD_W, 64 * c, k);
        part += blk_l2_head<4, true>(wp, 8 * h + 4, Ub, BLK_LD_W, 64 * c, b2, w3, Vb, BLK_LD_W, 64 * c, k);
        part = blk_rowsum(part);
        if (k.lane < BLK_R) qp[2 + c][h][k.lane] = part;
    }
    TEAM_LDS_BARRIER();
    // ---- clipped double-Q target, loss and its gradient at the heads (td3.py:306-319): one lane per row ----
    {
        const Blk k = blk_ids();
        if (k.w == 0) {
            float lsum = 0.f, ga = 0.f, gb = 0.f;
            const int b = b0 + k.lane;
            if (k.lane < BLK_R && b < B) {
                const int row = k.lane;
                const float r = Rb[blk_at1(BLK_LD_ROW, row, TD3_ROW - 2)], nd = Rb[blk_at1(BLK_LD_ROW, row, TD3_ROW - 1)];
                const float y = r + nd * A.gamma * fminf((qp[0][0][row] + qp[0][1][row]) + A.ct_b3[0], (qp[1][0][row] + qp[1][1][row]) + A.ct_b6[0]);
                const float inv = 1.f / (float)B;
                const float ea = ((qp[2][0][row] + qp[2][1][row]) + A.c_b3[0]) - y, eb = ((qp[3][0][row] + qp[3][1][row]) + A.c_b6[0]) - y;
                ga = 2.f * ea * inv; gb = 2.f * eb * inv;
                A.dq[2 * b] = ga; A.dq[2 * b + 1] = gb;
                dql[row][0] = ga; dql[row][1] = gb;
                lsum = ea * ea * inv + eb * eb * inv;
            } else if (k.lane < BLK_R) { dql[k.lane][0] = 0.f; dql[k.lane][1] = 0.f; }
            lsum = wave_sum(lsum); ga = wave_sum(ga); gb = wave_sum(gb);
            // loss and the head biases' gradients: one partial per workgroup; the last workgroup to get here adds them in workgroup order (as k_critic_team)
            int last = 0;
            if (k.lane == 0) {
                float *park = P.partials + 4 * blockIdx.x;
                park[0] = lsum; park[1] = ga; park[2] = gb;
                __threadfence();
                last = atomicAdd(A.done_count, 1) == n_blk - 1;
            }
            if (__builtin_amdgcn_readfirstlane(last)) {
                __threadfence();
                float l = 0.f, sa = 0.f, sb = 0.f;
                for (int c0 = 0; c0 < n_blk; c0 += 64) {
                    const int j = c0 + k.lane;
                    float pl = 0.f, pa = 0.f, pb = 0.f;
                    if (j < n_blk) {
                        const float *park = P.partials + 4 * j;
                        pl = __hip_atomic_load(park, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pa = __hip_atomic_load(park + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb = __hip_atomic_load(park + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int m = min(64, n_blk - c0);
                    for (int i = 0; i < m; i++) {
                        l += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pl), i));
                        sa += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pa), i));
                        sb += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pb), i));
                    }
                }
                if (k.lane == 0) {
                    A.loss[0] = l; A.db3a[0] += sa; A.db3b[0] += sb;
                    A.done_count[0] = 0;
                    if (A.rng_bump) A.rng_bump[1] += 1;
                    if (A.adam_step) A.adam_step[0] += 1.f;
                }
            }
        }
    }
    TEAM_LDS_BARRIER();
    // ---- c2 to global memory, and in its place dh2 = dq (x) w3 where the hidden unit was active (also to global memory) ----
    for (int i = threadIdx.x; i < BLK_R * (2 * TD3_H / 4); i += 64 * BLK_NW) {
        const int row = i >> 7, q = i & 127, c = q >> 6, f = 4 * (q & 63);
        const float *w3 = c ? A.c_w6 : A.c_w3;
        float *pv = Vb + blk_at(BLK_LD_W, row, q);
        const floatx4 h = *reinterpret_cast<const floatx4 *>(pv);
        const float d = dql[row][c];
        floatx4 dv;
#pragma unroll
        for (int j = 0; j < 4; j++) dv[j] = h[j] > 0.f ? d * w3[f + j] : 0.f;
        *reinterpret_cast<floatx4 *>(pv) = dv;
        if (b0 + row < B) {
            const size_t o = (size_t)(b0 + row) * 2 * TD3_H + 4 * q;
            *reinterpret_cast<floatx4 *>(A.c2 + o) = h;
            *reinterpret_cast<floatx4 *>(A.dh2 + o) = dv;
        }
    }
    TEAM_LDS_BARRIER();
    // ---- dh1_c = (W2_c^T dh2_c) where c1_c was active, written over c1 in U: waves 0, 1 = critic a, 2, 3 = critic b, 128 input features each ----
    {
        const Blk k = blk_ids();
        const int c = k.w >> 1, h = k.w & 1;
        const rsrc_t wp = mkrs(c ? P.p_c_w5t : P.p_c_w2t, (size_t)16 * 16 * 1024);
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int tile0 = 8 * h + 4 * half;
            floatx4 acc[4];
            blk_zero(acc);
            blk_mm<4, 16>(wp, tile0, Vb, BLK_LD_W, 64 * c, acc, k);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float *pu = Ub + blk_at(BLK_LD_W, k.r, 64 * c + 4 * (tile0 + t) + k.g);
                const floatx4 m = *reinterpret_cast<const floatx4 *>(pu);
                floatx4 v;
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = m[i] > 0.f ? acc[t][i] : 0.f;
                *reinterpret_cast<floatx4 *>(pu) = v;
            }
        }
    }
    TEAM_LDS_BARRIER();
    blk_flush(Ub, BLK_LD_W, 0, 2 * TD3_H / 4, A.dh1, 2 * TD3_H, 0, b0, B);
}

// ---- td3.py:334-341 for 16 batch rows per workgroup (arguments and outputs as k_policy_rows; g1, dg2, dg1 are not written: they never leave LDS) ----
#define PB_LD_SA 64                             // [s | a] (44 used, 44..47 zero)
#define PB_LD_DZ 64                             // dz (18 used, 18..31 zero)
__global__ __launch_bounds__(64 * BLK_NW) void k_policy_block(PlenTd3PolicyBlock P) {
    const PlenTd3PolicyRows &A = P.rows;
    __shared__ __attribute__((aligned(16))) float Sb[BLK_R * PB_LD_SA];
    __shared__ __attribute__((aligned(16))) float Zb[BLK_R * PB_LD_DZ];
    __shared__ __attribute__((aligned(16))) float P1[BLK_R * TD3_H];      // p1, later dp1
    __shared__ __attribute__((aligned(16))) float P2[BLK_R * TD3_H];      // p2, later dp2
    __shared__ __attribute__((aligned(16))) float G1[BLK_R * TD3_H];      // g1, later dg1
    __shared__ __attribute__((aligned(16))) float G2[BLK_R * TD3_H];      // dg2
    const int B = A.B, b0 = blockIdx.x * BLK_R;
    // the block's states: the state columns of sa_pi (left there by the critic pass: an earlier launch); rows past the batch repeat the last one
    for (int i = threadIdx.x; i < BLK_R * 64; i += 64 * BLK_NW) {
        const int row = i >> 6, c = i & 63;
        Sb[blk_at1(PB_LD_SA, row, c)] = c < TD3_S ? A.sa_pi[(size_t)min(b0 + row, B - 1) * TD3_SA + c] : 0.f;
        Zb[blk_at1(PB_LD_DZ, row, c)] = 0.f;
    }
    TEAM_LDS_BARRIER();
    // actor forward (td3.py:335)
    {
        const Blk k = blk_ids();
        blk_dense_relu<4, 2>(mkrs(P.p_a_w1, (size_t)16 * 2 * 1024), 4 * k.w, Sb, PB_LD_SA, 0, A.a_b1, P1, TD3_H, 0, k);
    }
    TEAM_LDS_BARRIER();
    blk_flush(P1, TD3_H, 0, TD3_H / 4, A.p1, TD3_H, 0, b0, B);
    {
        const Blk k = blk_ids();
        blk_dense_relu<4, 16>(mkrs(P.p_a_w2, (size_t)16 * 16 * 1024), 4 * k.w, P1, TD3_H, 0, A.a_b2, P2, TD3_H, 0, k);
    }
    TEAM_LDS_BARRIER();
    blk_flush(P2, TD3_H, 0, TD3_H / 4, A.p2, TD3_H, 0, b0, B);
    // the 18-wide output layer: its 256 k split over the four waves (64 each), the partial sums added through LDS (G2 is free) in wave order
    {
        const Blk k = blk_ids();
        floatx4 acc[2];
        blk_zero(acc);
        {
            // (the packed matrix's k steps 4 w .. 4 w + 3: a sub-range of each tile's 16 steps)
            const rsrc_t wp = mkrs(P.p_a_w3, (size_t)2 * 16 * 1024);
            const uint32_t voff = (uint32_t)k.lane * 16u;
            floatx4 a[4][2], b[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
#pragma unroll
                for (int t = 0; t < 2; t++) a[s][t] = bload4(wp, voff, (uint32_t)((t * 16 + 4 * k.w + s) * 1024));
                b[s] = *reinterpret_cast<const floatx4 *>(P2 + blk_at(TD3_H, k.r, 16 * k.w + 4 * s + k.g));
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int v = 0; v < 4; v++)
#pragma unroll
                    for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][t][v], b[s][v], acc[t], 0, 0, 0);
        }
        // partials: G2 as [wave][tile][lane] float4
#pragma unroll
        for (int t = 0; t < 2; t++) *reinterpret_cast<floatx4 *>(G2 + 4 * ((k.w * 2 + t) * 64 + k.lane)) = acc[t];
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids();
        if (k.w < 2) {
            const int t = k.w;
            floatx4 z = *reinterpret_cast<const floatx4 *>(G2 + 4 * ((0 * 2 + t) * 64 + k.lane));
#pragma unroll
            for (int w2 = 1; w2 < 4; w2++) z += *reinterpret_cast<const floatx4 *>(G2 + 4 * ((w2 * 2 + t) * 64 + k.lane));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 16 * t + 4 * k.g + i;
                if (j < TD3_A) {
                    const float a = A.max_a * tanhf(z[i] + A.a_b3[j]);                                   // td3.py:57
                    Sb[blk_at1(PB_LD_SA, k.r, TD3_S + j)] = a;
                    if (b0 + k.r < B) {
                        A.a_pi[(size_t)(b0 + k.r) * TD3_A + j] = a;
                        A.sa_pi[(size_t)(b0 + k.r) * TD3_SA + TD3_S + j] = a;
                    }
                }
            }
        }
    }
    TEAM_LDS_BARRIER();
    // critic.Q1 forward (fc1 = the first 16 tiles of the packed W14) and the gradient of -mean Q1 at its second hidden layer: dg2 = -(1/B) w3 (g2 > 0)
    {
        const Blk k = blk_ids();
        blk_dense_relu<4, 3>(mkrs(P.p_c_w14, (size_t)32 * 3 * 1024), 4 * k.w, Sb, PB_LD_SA, 0, A.c_b1, G1, TD3_H, 0, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids();
        floatx4 acc[4];
        blk_zero(acc);
        blk_mm<4, 16>(mkrs(P.p_c_w2, (size_t)16 * 16 * 1024), 4 * k.w, G1, TD3_H, 0, acc, k);
        const float ginv = -1.f / (float)B;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int f = 16 * (4 * k.w + t) + 4 * k.g;
            floatx4 v;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = acc[t][i] + A.c_b2[f + i] > 0.f ? ginv * A.c_w3[f + i] : 0.f;
            *reinterpret_cast<floatx4 *>(G2 + blk_at(TD3_H, k.r, 4 * (4 * k.w + t) + k.g)) = v;
        }
    }
    TEAM_LDS_BARRIER();
    // masked input gradient of a 256 x 256 layer: y = (W^T x)(mask > 0), written over the mask
    auto back = [&](const float *pk, const float *x, float *m, const Blk &k) {
        floatx4 acc[4];
        blk_zero(acc);
        blk_mm<4, 16>(mkrs(pk, (size_t)16 * 16 * 1024), 4 * k.w, x, TD3_H, 0, acc, k);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float *pm = m + blk_at(TD3_H, k.r, 4 * (4 * k.w + t) + k.g);
            const floatx4 mv = *reinterpret_cast<const floatx4 *>(pm);
            floatx4 v;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = mv[i] > 0.f ? acc[t][i] : 0.f;
            *reinterpret_cast<floatx4 *>(pm) = v;
        }
    };
    // dg1 = (W2^T dg2)(g1 > 0)
    {
        const Blk k = blk_ids();
        back(P.p_c_w2t, G2, G1, k);
    }
    TEAM_LDS_BARRIER();
    // d/d action = (W1^T dg1)[26:44], through the tanh: dz = that * (max_a - a^2 / max_a): two tiles, k split over the waves as above (partials through G2)
    {
        const Blk k = blk_ids();
        floatx4 acc[2];
        blk_zero(acc);
        {
            const rsrc_t wp = mkrs(P.p_c_w1ta, (size_t)2 * 16 * 1024);
            const uint32_t voff = (uint32_t)k.lane * 16u;
            floatx4 a[4][2], b[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
#pragma unroll
                for (int t = 0; t < 2; t++) a[s][t] = bload4(wp, voff, (uint32_t)((t * 16 + 4 * k.w + s) * 1024));
                b[s] = *reinterpret_cast<const floatx4 *>(G1 + blk_at(TD3_H, k.r, 16 * k.w + 4 * s + k.g));
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int v = 0; v < 4; v++)
#pragma unroll
                    for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][t][v], b[s][v], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; t++) *reinterpret_cast<floatx4 *>(G2 + 4 * ((k.w * 2 + t) * 64 + k.lane)) = acc[t];
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids();
        if (k.w < 2) {
            const int t = k.w;
            floatx4 z = *reinterpret_cast<const floatx4 *>(G2 + 4 * ((0 * 2 + t) * 64 + k.lane));
#pragma unroll
            for (int w2 = 1; w2 < 4; w2++) z += *reinterpret_cast<const floatx4 *>(G2 + 4 * ((w2 * 2 + t) * 64 + k.lane));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 16 * t + 4 * k.g + i;
                if (j < TD3_A) {
                    const float a = Sb[blk_at1(PB_LD_SA, k.r, TD3_S + j)];
                    const float dz = z[i] * (A.max_a - a * a / A.max_a);
                    Zb[blk_at1(PB_LD_DZ, k.r, j)] = dz;
                    if (b0 + k.r < B) A.dz[(size_t)(b0 + k.r) * TD3_A + j] = dz;
                }
            }
        }
    }
    TEAM_LDS_BARRIER();
    // back through the actor: dp2 = (W3^T dz)(p2 > 0), dp1 = (W2^T dp2)(p1 > 0)
    {
        const Blk k = blk_ids();
        floatx4 acc[4];
        blk_zero(acc);
        blk_mm<4, 2>(mkrs(P.p_a_w3t, (size_t)16 * 2 * 1024), 4 * k.w, Zb, PB_LD_DZ, 0, acc, k);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float *pm = P2 + blk_at(TD3_H, k.r, 4 * (4 * k.w + t) + k.g);
            const floatx4 mv = *reinterpret_cast<const floatx4 *>(pm);
            floatx4 v;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = mv[i] > 0.f ? acc[t][i] : 0.f;
            *reinterpret_cast<floatx4 *>(pm) = v;
        }
    }
    TEAM_LDS_BARRIER();
    blk_flush(P2, TD3_H, 0, TD3_H / 4, A.dp2, TD3_H, 0, b0, B);
    {
        const Blk k = blk_ids();
        back(P.p_a_w2t, P2, P1, k);
    }
    TEAM_LDS_BARRIER();
    blk_flush(P1, TD3_H, 0, TD3_H / 4, A.dp1, TD3_H, 0, b0, B);
    if (A.adam_step && threadIdx.x == 0) {
        if (atomicAdd(A.done_count, 1) == (int)((B + BLK_R - 1) / BLK_R) - 1) { A.done_count[0] = 0; A.adam_step[0] += 1.f; }
    }
}
