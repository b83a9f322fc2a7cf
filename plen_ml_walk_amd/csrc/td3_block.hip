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
#define BLK_NW 4                                // waves per workgroup (policy and actor passes)
// k_critic_block's own wave count, 8 (4: round 5's form, an A/B build): eight waves of 32 features per layer put TWO waves on every SIMD inside the register
// footprint of one (2 x 64 against 1 x 128 registers per SIMD -- the slots one retired f32 env wave per SIMD leaves; LDS 40 960 B either way).  At batch 4096 there
// is one workgroup per compute unit, so the second wave is the only thing that can issue while the first waits: in-phase matrix-pipe occupancy 0.79 -> 0.88 of the
// 256 x 256 layers, the thin phases (split 18-wide layer, gather) at half the per-wave work; 132 k -> 108 k shader cycles for workgroup 0 together with the hand-off
// below, 75.6 -> 65.5 us per launch alone (profiles/r06_w_td3_block_stamps.json, r06_w_td3_block_kernel_stats.csv)
#ifndef BLK_CRITIC_NW
#define BLK_CRITIC_NW 8
#endif
// LDS activation buffers (floats per row; every stride a multiple of 64 so that the XOR swizzle stays inside one 64-float group)
// the gathered replay row, not swizzled, re-arranged so that both critic inputs are contiguous: s 0..25 | a 26..43 | s2 44..69 | target action 70..87 | r 88 | not_done 89 | 0 0
#define BLK_LD_ROW 92
#define BLK_C_S2 44
#define BLK_C_A2 70
#define BLK_C_R 88
#define BLK_LD_W 256                            // the two wide buffers U and V (swizzled)

// development (scripts/gpu_td3_block_stamps.py, -DBLK_STAMPS): the shader clock at the phase boundaries of workgroup 0, behind the per-workgroup partials
#ifdef BLK_STAMPS
#define BLK_STAMP(P_, n_blk_) do { if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long *>((P_).partials + 4 * (n_blk_))[blk_stamp_k++] = __builtin_readcyclecounter(); } while (0)
#define BLK_STAMP_INIT() int blk_stamp_k = 0
#else
#define BLK_STAMP(P_, n_blk_) do { } while (0)
#define BLK_STAMP_INIT() do { } while (0)
#endif
struct Blk { int lane, r, g, w; };
// (the wave's number is read from threadIdx once, at the kernel's start, into a scalar register; the lane comes from the execution mask: no vector register
//  carries the thread id through the kernel)
static __device__ __forceinline__ Blk blk_ids(int wv) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return Blk{lane, lane & 15, lane >> 4, wv};
}
#define BLK_WAVE() const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)
#define BLK_TID(k_) (64 * (k_).w + (k_).lane)
// address (in floats) of 16-byte chunk q of row r of a SWIZZLED LDS activation buffer with row stride ld (a multiple of 64), and of column c of row r of a
// plain one (the narrow inputs: few k steps, conflicts do not matter)
static __device__ __forceinline__ int blk_at(int ld, int r, int q) { return r * ld + 4 * (q ^ r); }
static __device__ __forceinline__ int blk_lin(int ld, int r, int c) { return r * ld + c; }

// The first two k steps' A operands of a product, requested ahead of time: a phase ends with the request for the next phase's first weights, so that the trip to L2
// overlaps the barrier and the next phase starts on the matrix pipe (stamps: ~1000 of the ~3000 cycles a phase cost beyond its MFMAs).
template <int NT> struct BlkPre { floatx4 a0[NT], a1[NT]; };
template <int NT, int KS>
static __device__ __forceinline__ BlkPre<NT> blk_pre(rsrc_t wp, int tile0, const Blk &k) {
    BlkPre<NT> p;
    const uint32_t voff = (uint32_t)k.lane * 16u;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        p.a0[t] = bload4(wp, voff, (uint32_t)(((tile0 + t) * KS + 0) * 1024));
        p.a1[t] = KS > 1 ? bload4(wp, voff, (uint32_t)(((tile0 + t) * KS + 1) * 1024)) : floatx4{0, 0, 0, 0};
    }
    return p;
}
// four consecutive floats of a parameter vector that need not be 16-byte aligned (views of a flat buffer)
static __device__ __forceinline__ floatx4 blk_vec4(const float *p) { return floatx4{p[0], p[1], p[2], p[3]}; }

// acc[t] (features 16 (tile0 + t) + 4 g + v, batch row r) += sum_k W[feature][k] X[r][k] over KS steps of 16 k.
// wp: the layer's packed weights (plentd3_pack: float4 index ((tile KS + s) 64 + lane)); x: row 0 of the LDS input, chunk q0 = its first column / 4;
// pre: blk_pre<NT, KS>(wp, tile0) (steps 0 and 1).
struct BlkNoTail { __device__ __forceinline__ void operator()() const {} };
// tail: called once, when the last A loads have been issued (two k steps = up to 1024 matrix-pipe cycles before the product ends): the place to request what
// the epilogue needs (biases, head weights) -- earlier they would cost registers through the whole product, later their trip to memory would be exposed
template <int NT, int KS, bool SWZ = true, class Tail = BlkNoTail>
static __device__ __forceinline__ void blk_mm(const BlkPre<NT> &pre, rsrc_t wp, int tile0, const float *x, int ld, int q0, floatx4 (&acc)[NT], const Blk &k, Tail &&tail = Tail()) {
    const uint32_t voff = (uint32_t)k.lane * 16u;
    const float *xr = x + k.r * ld;
    const int sw = SWZ ? k.r : 0;
    auto loadA = [&](int s, floatx4 (&a)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; t++) a[t] = bload4(wp, voff, (uint32_t)(((tile0 + t) * KS + s) * 1024));
    };
    auto loadB = [&](int s) { return *reinterpret_cast<const floatx4 *>(xr + 4 * ((q0 + 4 * s + k.g) ^ sw)); };
    // three stages in flight: the loads of step s + 2 are issued before the MFMAs of step s (an L2 hit is ~500 cycles under load, a step of NT = 4 tiles
    // is 512 cycles of matrix pipe); left alone the compiler sinks the loads to their first use
    // (the B operand comes from LDS, ~100 cycles away: one step ahead is enough)
    floatx4 a[3][NT], b[3];
#pragma unroll
    for (int t = 0; t < NT; t++) { a[0][t] = pre.a0[t]; a[1][t] = pre.a1[t]; }
    b[0] = loadB(0);
    auto step = [&](int s, int i, bool more2, bool more1) {          // k step s in ring slot i (compile-time at every call site)
        if (more2) loadA(s + 2, a[(i + 2) % 3]);
        if (more1) b[(i + 1) % 3] = loadB(s + 1);
        if (s == (KS > 2 ? KS - 2 : 0)) tail();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][t][v], b[i][v], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (KS >= 8) {
        // a real loop over groups of three k steps (the ring's period) instead of KS unrolled copies: the unrolled passes were ~55 KB of straight-line code
        // streaming through the 64 KB instruction cache a pair of compute units shares WITH THE ENV WAVES beside them (scripts/gpu_clock_probe.py: the envs'
        // longest waves took 25 % more cycles beside the update, none beside matrix-core instructions alone)
        constexpr int G = (KS - 2) / 3;                    // whole groups whose loads all exist (s + 2 < KS throughout)
        int s0 = 0;
#pragma unroll 1
        for (int gi = 0; gi < G; gi++, s0 += 3) { step(s0, 0, true, true); step(s0 + 1, 1, true, true); step(s0 + 2, 2, true, true); }
#pragma unroll
        for (int s = 3 * G; s < KS; s++) step(s, s % 3, s + 2 < KS, s + 1 < KS);
    } else {
#pragma unroll
        for (int s = 0; s < KS; s++) step(s, s % 3, s + 2 < KS, s + 1 < KS);
    }
}

template <int NT>
static __device__ __forceinline__ void blk_zero(floatx4 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = floatx4{0, 0, 0, 0};
}

// NT tiles of a hidden layer: relu(W x + bias) into the swizzled LDS buffer y; the output buffer's chunk 4 (ytile0 + t) + g receives tile tile0 + t (a 256-wide
// buffer holding one critic's half of a stacked layer).  The bias is requested before the product (its trip to memory hides behind the MFMAs).
template <int NT, int KS, bool SWZ = true>
static __device__ __forceinline__ void blk_dense_relu(const BlkPre<NT> &pre, rsrc_t wp, int tile0, const float *x, int ldx, int qx0, const float *bias, float *y, int ldy, int ytile0, const Blk &k) {
    floatx4 acc[NT], bz[NT];
    blk_zero(acc);
    blk_mm<NT, KS, SWZ>(pre, wp, tile0, x, ldx, qx0, acc, k, [&]() {
#pragma unroll
        for (int t = 0; t < NT; t++) bz[t] = blk_vec4(bias + 16 * (tile0 + t) + 4 * k.g);
    });
#pragma unroll
    for (int t = 0; t < NT; t++) {
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = fmaxf(acc[t][i] + bz[t][i], 0.f);
        *reinterpret_cast<floatx4 *>(y + blk_at(ldy, k.r, 4 * (ytile0 + t) + k.g)) = v;
    }
}

// NT tiles of a critic's second layer and their share of its scalar head: part += sum over the lane's features of relu(W2 x + b2) w3; the activations go
// to LDS (y) when STORE
template <int NT, bool STORE>
static __device__ __forceinline__ float blk_l2_head(const BlkPre<NT> &pre, rsrc_t wp, int tile0, const float *x, int ldx, const float *b2, const float *w3, float *y, int ldy, const Blk &k) {
    floatx4 acc[NT], bz[NT], wz[NT];
    blk_zero(acc);
    blk_mm<NT, 16>(pre, wp, tile0, x, ldx, 0, acc, k, [&]() {
#pragma unroll
        for (int t = 0; t < NT; t++) { bz[t] = blk_vec4(b2 + 16 * (tile0 + t) + 4 * k.g); wz[t] = blk_vec4(w3 + 16 * (tile0 + t) + 4 * k.g); }
    });
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = fmaxf(acc[t][i] + bz[t][i], 0.f); part += v[i] * wz[t][i]; }
        if constexpr (STORE) *reinterpret_cast<floatx4 *>(y + blk_at(ldy, k.r, 4 * (tile0 + t) + k.g)) = v;
    }
    return part;
}
// lane ^ m's value (as __shfl_xor, without its width test: the bound `lane + 64` it keeps in a register across the whole kernel cost the eight-wave build a scratch slot)
static __device__ __forceinline__ float blk_xor(float v, int lane, int m) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, __builtin_bit_cast(int, v)));
}
static __device__ __forceinline__ float blk_wave_sum(float v, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += blk_xor(v, lane, o);
    return v;
}
// the four k-groups' partials of batch row r (lanes r, r + 16, r + 32, r + 48), in group order
static __device__ __forceinline__ float blk_rowsum(float part, int lane) {
    const float p1 = blk_xor(part, lane, 16);
    const float s01 = part + p1;                 // (g, g ^ 1)
    return s01 + blk_xor(s01, lane, 32);
}

// copy a 256-wide swizzled LDS buffer to columns col0.. of the row-major global matrix Y [B][ldy]: whole 1-KB row segments per wave
template <int NW = BLK_NW>
static __device__ __forceinline__ void blk_flush(const float *x, float *Y, int ldy, int col0, int b0, int B, int wv) {
    const Blk k = blk_ids(wv);
#pragma unroll
    for (int j = 0; j < BLK_R * 64 / (64 * NW); j++) {
        const int i = BLK_TID(k) + 64 * NW * j, row = i >> 6, q = i & 63;
        if (b0 + row < B) *reinterpret_cast<floatx4 *>(Y + (size_t)(b0 + row) * ldy + col0 + 4 * q) = *reinterpret_cast<const floatx4 *>(x + blk_at(BLK_LD_W, row, q));
    }
}
// an 18-wide layer (two 16-feature tiles) on a 256-wide swizzled input: its 256 k split over the NW waves (64 each of four), the partial sums parked in a free LDS
// buffer as [wave][tile][lane] float4 and added in wave order by blk_split_sum after a barrier
template <int NW = BLK_NW>
static __device__ __forceinline__ void blk_split_k(rsrc_t wp, const float *x, float *park, const Blk &k) {
    constexpr int S = 16 / NW;                         // k steps of 16 per wave
    const uint32_t voff = (uint32_t)k.lane * 16u;
    floatx4 acc[2], a[S][2], b[S];
    blk_zero(acc);
#pragma unroll
    for (int s = 0; s < S; s++) {
#pragma unroll
        for (int t = 0; t < 2; t++) a[s][t] = bload4(wp, voff, (uint32_t)((t * 16 + S * k.w + s) * 1024));
        b[s] = *reinterpret_cast<const floatx4 *>(x + blk_at(BLK_LD_W, k.r, 4 * S * k.w + 4 * s + k.g));
    }
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][t][v], b[s][v], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) *reinterpret_cast<floatx4 *>(park + 4 * ((k.w * 2 + t) * 64 + k.lane)) = acc[t];
}
template <int NW = BLK_NW>
static __device__ __forceinline__ floatx4 blk_split_sum(const float *park, int t, const Blk &k) {
    floatx4 z = *reinterpret_cast<const floatx4 *>(park + 4 * ((0 * 2 + t) * 64 + k.lane));
#pragma unroll
    for (int w2 = 1; w2 < NW; w2++) z += *reinterpret_cast<const floatx4 *>(park + 4 * ((w2 * 2 + t) * 64 + k.lane));
    return z;
}

// ---- weight packing: M (N x K; element (i, k) at src[i rs + k cs]) -> MFMA A-operand order, zero-padded to 16-row tiles and 16-k steps:
//      dst float4 ((t KS + s) 64 + lane) = M[16 t + (lane % 16)][16 s + 4 (lane / 16) + (0..3)]
// Issue priority of the update's waves (s_setprio 0..3; env waves run at 0): an experiment knob, -DBLK_PRIO=n.  The update shares every SIMD with resident env waves.
#ifndef BLK_PRIO
#define BLK_PRIO 0
#endif
#define BLK_SETPRIO() do { if (BLK_PRIO) __builtin_amdgcn_s_setprio(BLK_PRIO); } while (0)
__global__ __launch_bounds__(256) void k_pack(PlenTd3PackGroup G) {
    BLK_SETPRIO();
    const int e = blockIdx.x * 256 + threadIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < PLENTD3_PACK_JOBS; k++) j += (k < G.n_jobs && e >= G.job[k].f4_0) ? 1 : 0;
    const PlenTd3PackJob &J = G.job[j];
    const int el = e - J.f4_0;
    int i, k0;
    if (J.team) {          // the small-batch kernels' order (td3_team.hip quad_nt): 32-column tiles, stages of 64 k, eight 4-k pieces per stage and k half
        const int NS = (J.K + 63) / 64, T = (J.N + 31) / 32;
        if (el >= T * NS * 512) return;
        const int lane = el & 63, c = (el >> 6) & 7, s = (el >> 9) % NS, t = (el >> 9) / NS;
        i = 32 * t + (lane & 31); k0 = 64 * s + 32 * (lane >> 5) + 4 * c;
    } else {
        const int KS = (J.K + 15) / 16, T = (J.N + 15) / 16;
        if (el >= T * KS * 64) return;
        const int lane = el & 63, s = (el >> 6) % KS, t = (el >> 6) / KS;
        i = 16 * t + (lane & 15); k0 = 16 * s + 4 * (lane >> 4);
    }
    floatx4 v;
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = (i < J.N && k0 + u < J.K) ? J.src[(size_t)i * J.rs + (size_t)(k0 + u) * J.cs] : 0.f;
    reinterpret_cast<floatx4 *>(J.dst)[el] = v;
}

// ---- td3.py:277-323 for 16 batch rows per workgroup (arguments and outputs as k_critic_rows; t0, t1, sa2 are not written).
//      Resources are kept to what FOUR retiring env waves leave behind on a compute unit (one wave slot of 128 registers per SIMD, 41 KB of LDS): the update runs
//      beside env launches that hold every wave slot of the chip (train_vec.PipelinedVecTD3Trainer), and a workgroup that needs more waits for more to retire.
//      Hence two 256-wide activation buffers only: the twin critics go through them one after the other.
// The workgroup's loss / head-bias-gradient partials (wave 0): parked, made visible, counted (td3_kernels.hip: handoff_last); nonzero in lane 0 of the LAST workgroup
// to get here.  (Counting BEFORE the kernel's last flush, so that the counter's round trip runs beside it: measured, 1.4 k of 111 k cycles, nothing in the leg; not kept.)
static __device__ __forceinline__ int blk_park_sums(const float (*lt)[4], float *park, int *done_count, int n_blk, int lane) {
    const float l = blk_wave_sum(lane < BLK_R ? lt[lane][0] + lt[lane][2] : 0.f, lane), ga = blk_wave_sum(lane < BLK_R ? lt[lane][1] : 0.f, lane), gb = blk_wave_sum(lane < BLK_R ? lt[lane][3] : 0.f, lane);
    int last = 0;
    if (lane == 0) last = handoff_last(park, l, ga, gb, done_count, n_blk);
    return last;
}
__global__ __launch_bounds__(64 * BLK_CRITIC_NW) __attribute__((amdgpu_waves_per_eu(BLK_CRITIC_NW, BLK_CRITIC_NW))) void k_critic_block(PlenTd3CriticBlock P) {
    BLK_SETPRIO();
    constexpr int NW = BLK_CRITIC_NW, NT = 16 / NW, RW = BLK_R / NW;          // waves; 16-feature tiles per wave and 256-wide layer; gathered rows per wave
    static_assert(NW == 4 || NW == 8, "four waves of 64 features or eight of 32");
    const PlenTd3CriticRows &A = P.rows;
    __shared__ __attribute__((aligned(16))) float Rb[BLK_R * BLK_LD_ROW];
    __shared__ __attribute__((aligned(16))) float Ub[BLK_R * BLK_LD_W];
    __shared__ __attribute__((aligned(16))) float Vb[BLK_R * BLK_LD_W];
    // [target a, target b, critic a, critic b][wave][row]: partial heads.  The target action's clipped smoothing noise [row][TD3_A] lives in the same words: it is
    // consumed (target-action phase) two barriers before the first partial head is parked
    constexpr int QPN = 4 * NW * BLK_R > BLK_R * TD3_A ? 4 * NW * BLK_R : BLK_R * TD3_A;
    __shared__ float qpn[QPN];
    __shared__ float lt[BLK_R][4];               // the rows' loss terms and head-bias gradients [critic a: e^2 / B, dq | critic b: ...]
    float (*const qp)[NW][BLK_R] = reinterpret_cast<float (*)[NW][BLK_R]>(qpn);
    float (*const nz)[TD3_A] = reinterpret_cast<float (*)[TD3_A]>(qpn);
    static_assert(sizeof(float) * (BLK_R * BLK_LD_ROW + 2 * BLK_R * BLK_LD_W + QPN + BLK_R * 4) <= 40960, "four workgroups' worth of LDS per compute unit: the register cap below binds only then");
    BlkPre<NT> pre;
    const int B = A.B, b0 = blockIdx.x * BLK_R, n_blk = (B + BLK_R - 1) / BLK_R;
    BLK_WAVE();
    BLK_STAMP_INIT();
    BLK_STAMP(P, n_blk);
    // ---- sample the block's 16 rows of the replay ring (td3.py:166-193; same draw as k_sample_gather / k_critic_rows) and gather them: 4 rows per wave ----
    {
        const Blk k = blk_ids(wv);
        int64_t id = 0;
        if (k.lane < RW) {
            const int b = min(b0 + RW * k.w + k.lane, B - 1);
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
        for (int i = 0; i < RW; i++) {
            const int row = RW * k.w + i, b = b0 + row;
            const int64_t rid = ((int64_t)__shfl(hi, i) << 32) | (uint32_t)__shfl(lo, i);
            const float *src = A.data + (size_t)rid * TD3_ROW;
            const float v0 = src[k.lane], v1 = k.lane < TD3_ROW - 64 ? src[64 + k.lane] : 0.f;
            Rb[blk_lin(BLK_LD_ROW, row, k.lane)] = v0;
            // source columns 64..69 (the tail of s2) stay where they are, 70 / 71 (r, not_done) move behind the target action's columns, which start as zeros
            // (the target actor's first layer reads them as padded k: finite values against zero weights)
            if (k.lane < BLK_LD_ROW - 64) Rb[blk_lin(BLK_LD_ROW, row, 64 + k.lane)] = k.lane < 6 ? v1 : 0.f;
            if (k.lane == 6 || k.lane == 7) Rb[blk_lin(BLK_LD_ROW, row, BLK_C_R + k.lane - 6)] = v1;
            if (b < B) {
                float *dst = A.batch + (size_t)b * TD3_ROW;
                dst[k.lane] = v0;
                if (k.lane < TD3_ROW - 64) dst[64 + k.lane] = v1;
                if (k.lane < TD3_S) A.sa_pi[(size_t)b * TD3_SA + k.lane] = v0;
            }
        }
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    // ---- target actor (td3.py:299): 64 features per wave and layer; the 18-wide output layer with its k split over the waves.
    //      (every phase requests the first weights of the NEXT one before its closing barrier: `pre`) ----
    {
        const Blk k = blk_ids(wv);
        const rsrc_t w1 = mkrs(P.p_at_w1, (size_t)16 * 2 * 1024);
        blk_dense_relu<NT, 2, false>(blk_pre<NT, 2>(w1, NT * k.w, k), w1, NT * k.w, Rb, BLK_LD_ROW, BLK_C_S2 / 4, A.at_b1, Ub, BLK_LD_W, NT * k.w, k);
        pre = blk_pre<NT, 16>(mkrs(P.p_at_w2, (size_t)16 * 16 * 1024), NT * k.w, k);
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    {
        const Blk k = blk_ids(wv);
        blk_dense_relu<NT, 16>(pre, mkrs(P.p_at_w2, (size_t)16 * 16 * 1024), NT * k.w, Ub, BLK_LD_W, 0, A.at_b2, Vb, BLK_LD_W, NT * k.w, k);
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    {
        // the target action's clipped smoothing noise (td3.py:300-301), one element per thread, beside the split product (Philox + Box-Muller: 8 k cycles when
        // two waves drew it after the sum with the other two idle)
        const Blk k = blk_ids(wv);
        for (int e = BLK_TID(k); e < BLK_R * TD3_A; e += 64 * NW) {
            const int row = e / TD3_A, j = e - row * TD3_A, ge = min(b0 + row, B - 1) * TD3_A + j;
            const float zn = A.noise ? A.noise[ge] : rng_normal(A.rng, 1u, (uint32_t)ge);                 // torch.randn_like(action)
            nz[row][j] = fminf(fmaxf(zn * A.sigma, -A.clip), A.clip);
        }
        blk_split_k<NW>(mkrs(P.p_at_w3, (size_t)2 * 16 * 1024), Vb, Ub, k);
        pre = blk_pre<NT, 3>(mkrs(P.p_ct_w14, (size_t)32 * 3 * 1024), NT * k.w, k);
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    // ---- target action (td3.py:299-304) by waves 0, 1 (one tile each); the first target critic's first layer on (s2, a2) has to wait for it ----
    {
        const Blk k = blk_ids(wv);
        if (k.w < 2) {
            const floatx4 z = blk_split_sum<NW>(Ub, k.w, k);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 16 * k.w + 4 * k.g + i;
                if (j < TD3_A) Rb[blk_lin(BLK_LD_ROW, k.r, BLK_C_A2 + j)] = fminf(fmaxf(A.max_a * tanhf(z[i] + A.at_b3[j]) + nz[k.r][j], -A.max_a), A.max_a);
            }
        }
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    // ---- clipped double-Q target (td3.py:306-309): target critic a's first layer -> V, then its second layer + head from V beside target critic b's first
    //      layer -> U, then b's second layer + head from U beside the critic's own first layer (a) -> V: every phase has a full matrix load on every wave ----
    {
        const Blk k = blk_ids(wv);
        blk_dense_relu<NT, 3, false>(pre, mkrs(P.p_ct_w14, (size_t)32 * 3 * 1024), NT * k.w, Rb, BLK_LD_ROW, BLK_C_S2 / 4, A.ct_b14, Vb, BLK_LD_W, NT * k.w, k);
        pre = blk_pre<NT, 16>(mkrs(P.p_ct_w2, (size_t)16 * 16 * 1024), NT * k.w, k);
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    {
        const Blk k = blk_ids(wv);
        const rsrc_t w14 = mkrs(P.p_ct_w14, (size_t)32 * 3 * 1024);
        const float part = blk_rowsum(blk_l2_head<NT, false>(pre, mkrs(P.p_ct_w2, (size_t)16 * 16 * 1024), NT * k.w, Vb, BLK_LD_W, A.ct_b2, A.ct_w3, nullptr, 0, k), k.lane);
        if (k.lane < BLK_R) qp[0][k.w][k.lane] = part;
        blk_dense_relu<NT, 3, false>(blk_pre<NT, 3>(w14, 16 + NT * k.w, k), w14, 16 + NT * k.w, Rb, BLK_LD_ROW, BLK_C_S2 / 4, A.ct_b14, Ub, BLK_LD_W, NT * k.w, k);
        pre = blk_pre<NT, 16>(mkrs(P.p_ct_w5, (size_t)16 * 16 * 1024), NT * k.w, k);
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    {
        const Blk k = blk_ids(wv);
        const rsrc_t w14 = mkrs(P.p_c_w14, (size_t)32 * 3 * 1024);
        const float part = blk_rowsum(blk_l2_head<NT, false>(pre, mkrs(P.p_ct_w5, (size_t)16 * 16 * 1024), NT * k.w, Ub, BLK_LD_W, A.ct_b5, A.ct_w6, nullptr, 0, k), k.lane);
        if (k.lane < BLK_R) qp[1][k.w][k.lane] = part;
        blk_dense_relu<NT, 3, false>(blk_pre<NT, 3>(w14, NT * k.w, k), w14, NT * k.w, Rb, BLK_LD_ROW, 0, A.c_b14, Vb, BLK_LD_W, NT * k.w, k);
        pre = blk_pre<NT, 16>(mkrs(P.p_c_w2, (size_t)16 * 16 * 1024), NT * k.w, k);
    }
    TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
    // ---- the critics (td3.py:312-323), one after the other: c1 in X, c2 -> Y (+ head), dh2 over c2, dh1 over c1, out; X / Y = V / U for critic a, U / V for b ----
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float *X = c ? Ub : Vb, *Y = c ? Vb : Ub;
        {
            const Blk k = blk_ids(wv);
            const float part = blk_rowsum(blk_l2_head<NT, true>(pre, mkrs(c ? P.p_c_w5 : P.p_c_w2, (size_t)16 * 16 * 1024), NT * k.w, X, BLK_LD_W, c ? A.c_b5 : A.c_b2, c ? A.c_w6 : A.c_w3, Y, BLK_LD_W, k), k.lane);
            if (k.lane < BLK_R) qp[2 + c][k.w][k.lane] = part;
        }
        blk_flush<NW>(X, A.c1, 2 * TD3_H, c * TD3_H, b0, B, wv);                 // c1 for the weight gradients (asynchronous: nothing below reads it back)
        TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
        // the loss gradient at this critic's head, per row (every thread derives the rows it needs from the parked partial heads: no phase of its own), c2 to global
        // memory, and in its place dh2 = dq (x) w3 where the hidden unit was active (also to global memory)
        {
            const float *w3 = c ? A.c_w6 : A.c_w3;
            const float b3 = (c ? A.c_b6 : A.c_b3)[0], tb3a = A.ct_b3[0], tb3b = A.ct_b6[0], inv = 1.f / (float)B;
            const Blk k = blk_ids(wv);
            const floatx4 wh = blk_vec4(w3 + 4 * k.lane);            // (a thread's chunk q = its lane: the same four head weights for its four rows)
#pragma unroll
            for (int j = 0; j < BLK_R / NW; j++) {
                const int row = k.w + NW * j, q = k.lane;
                auto head = [&](int h, float b3_) {
                    float q4 = (qp[h][0][row] + qp[h][1][row]) + (qp[h][2][row] + qp[h][3][row]);
                    if constexpr (NW == 8) q4 += (qp[h][4][row] + qp[h][5][row]) + (qp[h][6][row] + qp[h][7][row]);
                    return q4 + b3_;
                };
                const float y = Rb[blk_lin(BLK_LD_ROW, row, BLK_C_R)] + Rb[blk_lin(BLK_LD_ROW, row, BLK_C_R + 1)] * A.gamma * fminf(head(0, tb3a), head(1, tb3b));
                const float e = head(2 + c, b3) - y;
                const bool in = b0 + row < B;
                const float d = in ? 2.f * e * inv : 0.f;
                float *pv = Y + blk_at(BLK_LD_W, row, q);
                const floatx4 h = *reinterpret_cast<const floatx4 *>(pv);
                floatx4 dv;
#pragma unroll
                for (int u = 0; u < 4; u++) dv[u] = h[u] > 0.f ? d * wh[u] : 0.f;
                *reinterpret_cast<floatx4 *>(pv) = dv;
                if (in) {
                    const size_t o = (size_t)(b0 + row) * 2 * TD3_H + c * TD3_H + 4 * q;
                    *reinterpret_cast<floatx4 *>(A.c2 + o) = h;
                    *reinterpret_cast<floatx4 *>(A.dh2 + o) = dv;
                    if (q == 0) A.dq[2 * (b0 + row) + c] = d;
                }
                // the row's loss term and head-bias gradient, parked per row for the sums at the end
                if (q == 0) { lt[row][2 * c] = in ? e * e * inv : 0.f; lt[row][2 * c + 1] = d; }
            }
        }
        {
            const Blk k = blk_ids(wv);
            pre = blk_pre<NT, 16>(mkrs(c ? P.p_c_w5t : P.p_c_w2t, (size_t)16 * 16 * 1024), NT * k.w, k);
        }
        TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
        // dh1 = (W2^T dh2) where c1 was active, written over c1
        {
            const Blk k = blk_ids(wv);
            floatx4 acc[NT];
            blk_zero(acc);
            blk_mm<NT, 16>(pre, mkrs(c ? P.p_c_w5t : P.p_c_w2t, (size_t)16 * 16 * 1024), NT * k.w, Y, BLK_LD_W, 0, acc, k);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                float *pu = X + blk_at(BLK_LD_W, k.r, 4 * (NT * k.w + t) + k.g);
                const floatx4 m = *reinterpret_cast<const floatx4 *>(pu);
                floatx4 v;
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = m[i] > 0.f ? acc[t][i] : 0.f;
                *reinterpret_cast<floatx4 *>(pu) = v;
            }
            if (c == 0) pre = blk_pre<NT, 3>(mkrs(P.p_c_w14, (size_t)32 * 3 * 1024), 16 + NT * k.w, k);
        }
        TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
        if (c == 0) {        // critic b's first layer -> U (= Y of critic a: its dh2 is dead), beside the flush of dh1_a from V
            blk_flush<NW>(X, A.dh1, 2 * TD3_H, c * TD3_H, b0, B, wv);
            const Blk k = blk_ids(wv);
            blk_dense_relu<NT, 3, false>(pre, mkrs(P.p_c_w14, (size_t)32 * 3 * 1024), 16 + NT * k.w, Rb, BLK_LD_ROW, 0, A.c_b14, Ub, BLK_LD_W, NT * k.w, k);
            pre = blk_pre<NT, 16>(mkrs(P.p_c_w5, (size_t)16 * 16 * 1024), NT * k.w, k);
            TEAM_LDS_BARRIER(); BLK_STAMP(P, n_blk);
        }
    }
    // ---- loss and the head biases' gradients: per-row terms live in the threads with q == 0 (lanes 0 of ... every wave holds 4 rows' worth): workgroup sums
    //      through LDS in thread order, one partial per workgroup; the last workgroup to get here adds them in workgroup order (as k_critic_team) ----
    //      (the per-row terms were parked before the last two barriers; critic b's dh1 leaves from U)
    blk_flush<NW>(Ub, A.dh1, 2 * TD3_H, TD3_H, b0, B, wv);
    if (wv == 0) {
        const int lane = blk_ids(wv).lane;
        const int last = blk_park_sums(lt, P.partials + 4 * blockIdx.x, A.done_count, n_blk, lane);
        if (__builtin_amdgcn_readfirstlane(last)) {          // one workgroup's partials per lane, then added in lane order (a single lane fetching 256 x 3 partials
            handoff_acquire();                               // one after the other was 9 k cycles at the end of the kernel)
            float sl = 0.f, sa = 0.f, sb = 0.f;
            for (int c0 = 0; c0 < n_blk; c0 += 64) {
                const int j = c0 + lane;
                float pl = 0.f, pa = 0.f, pb = 0.f;
                if (j < n_blk) {
                    const float *pk = P.partials + 4 * j;
                    pl = __hip_atomic_load(pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pa = __hip_atomic_load(pk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pb = __hip_atomic_load(pk + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int mm = min(64, n_blk - c0);
                for (int i = 0; i < mm; i++) {
                    sl += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pl), i));
                    sa += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pa), i));
                    sb += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pb), i));
                }
            }
            if (lane == 0) {
                A.loss[0] = sl; A.db3a[0] += sa; A.db3b[0] += sb;
                A.done_count[0] = 0;
                if (A.rng_bump) A.rng_bump[1] += 1;
                if (A.adam_step) A.adam_step[0] += 1.f;
            }
        }
    }
    BLK_STAMP(P, n_blk);
}

// ---- td3.py:334-341 for 16 batch rows per workgroup (arguments and outputs as k_policy_rows; g1, dg2, dg1 are not written: they never leave LDS).
//      Same resource budget as k_critic_block: two 256-wide LDS buffers X, Y.  A layer's ReLU mask is needed again when the gradient comes back through it,
//      by the SAME lane (same feature split forward and backward): it is kept as 16 bits in a register instead of keeping the activations in LDS.
#define PB_LD_SA 52                             // [s | a] (44 used, 44..47 zero; not swizzled)
#define PB_LD_DZ 36                             // dz (18 used, 18..31 zero; not swizzled)
// NT = 4 tiles of relu(W x + bias) into y, returning the mask of active units (bit 4 t + i)
template <int KS, bool SWZ>
static __device__ __forceinline__ uint32_t blk_dense_relu_mask(const BlkPre<4> &pre, rsrc_t wp, int tile0, const float *x, int ldx, const float *bias, float *y, const Blk &k) {
    floatx4 acc[4], bz[4];
    blk_zero(acc);
    blk_mm<4, KS, SWZ>(pre, wp, tile0, x, ldx, 0, acc, k, [&]() {
#pragma unroll
        for (int t = 0; t < 4; t++) bz[t] = blk_vec4(bias + 16 * (tile0 + t) + 4 * k.g);
    });
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = fmaxf(acc[t][i] + bz[t][i], 0.f); m |= v[i] > 0.f ? 1u << (4 * t + i) : 0u; }
        *reinterpret_cast<floatx4 *>(y + blk_at(BLK_LD_W, k.r, 4 * (tile0 + t) + k.g)) = v;
    }
    asm volatile("; relu mask packed" : "+v"(m));        // opaque: otherwise the compiler keeps the 16 compared values alive (and spills them) instead of the 16 bits
    return m;
}
// y = (W^T x) where the forward activation was active (mask m), NT = 4 tiles
template <int KS, bool SWZ>
static __device__ __forceinline__ void blk_back_mask(const BlkPre<4> &pre, rsrc_t wp, int tile0, const float *x, int ldx, uint32_t m, float *y, const Blk &k) {
    floatx4 acc[4];
    blk_zero(acc);
    blk_mm<4, KS, SWZ>(pre, wp, tile0, x, ldx, 0, acc, k);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = (m >> (4 * t + i)) & 1u ? acc[t][i] : 0.f;
        *reinterpret_cast<floatx4 *>(y + blk_at(BLK_LD_W, k.r, 4 * (tile0 + t) + k.g)) = v;
    }
}
__global__ __launch_bounds__(64 * BLK_NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_policy_block(PlenTd3PolicyBlock P) {
    BLK_SETPRIO();
    const PlenTd3PolicyRows &A = P.rows;
    __shared__ __attribute__((aligned(16))) float Sb[BLK_R * PB_LD_SA];
    __shared__ __attribute__((aligned(16))) float Zb[BLK_R * PB_LD_DZ];
    __shared__ __attribute__((aligned(16))) float Xb[BLK_R * BLK_LD_W];
    __shared__ __attribute__((aligned(16))) float Yb[BLK_R * BLK_LD_W];
    const int B = A.B, b0 = blockIdx.x * BLK_R;
    BLK_WAVE();
    BlkPre<4> pre;
    uint32_t m1, m2, mg;
    // the block's states: the state columns of sa_pi (left there by the critic pass: an earlier launch); rows past the batch repeat the last one
    {
        const Blk k = blk_ids(wv);
        pre = blk_pre<4, 2>(mkrs(P.p_a_w1, (size_t)16 * 2 * 1024), 4 * k.w, k);
    }
    {
        const Blk k = blk_ids(wv);
        for (int i = BLK_TID(k); i < BLK_R * PB_LD_SA; i += 64 * BLK_NW) {
            const int row = i / PB_LD_SA, c = i - row * PB_LD_SA;
            Sb[i] = c < TD3_S ? A.sa_pi[(size_t)min(b0 + row, B - 1) * TD3_SA + c] : 0.f;
        }
        for (int i = BLK_TID(k); i < BLK_R * PB_LD_DZ; i += 64 * BLK_NW) Zb[i] = 0.f;
    }
    TEAM_LDS_BARRIER();
    // actor forward (td3.py:335): p1 -> X, p2 -> Y
    {
        const Blk k = blk_ids(wv);
        m1 = blk_dense_relu_mask<2, false>(pre, mkrs(P.p_a_w1, (size_t)16 * 2 * 1024), 4 * k.w, Sb, PB_LD_SA, A.a_b1, Xb, k);
        pre = blk_pre<4, 16>(mkrs(P.p_a_w2, (size_t)16 * 16 * 1024), 4 * k.w, k);
    }
    TEAM_LDS_BARRIER();
    blk_flush(Xb, A.p1, TD3_H, 0, b0, B, wv);
    {
        const Blk k = blk_ids(wv);
        m2 = blk_dense_relu_mask<16, true>(pre, mkrs(P.p_a_w2, (size_t)16 * 16 * 1024), 4 * k.w, Xb, BLK_LD_W, A.a_b2, Yb, k);
    }
    TEAM_LDS_BARRIER();
    blk_flush(Yb, A.p2, TD3_H, 0, b0, B, wv);
    {
        const Blk k = blk_ids(wv);
        blk_split_k(mkrs(P.p_a_w3, (size_t)2 * 16 * 1024), Yb, Xb, k);                  // (p1 is out of LDS: flushed, its mask kept)
        pre = blk_pre<4, 3>(mkrs(P.p_c_w14, (size_t)32 * 3 * 1024), 4 * k.w, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids(wv);
        if (k.w < 2) {
            const floatx4 z = blk_split_sum(Xb, k.w, k);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 16 * k.w + 4 * k.g + i;
                if (j < TD3_A) {
                    const float a = A.max_a * tanhf(z[i] + A.a_b3[j]);                                   // td3.py:57
                    Sb[blk_lin(PB_LD_SA, k.r, TD3_S + j)] = a;
                    if (b0 + k.r < B) {
                        A.a_pi[(size_t)(b0 + k.r) * TD3_A + j] = a;
                        A.sa_pi[(size_t)(b0 + k.r) * TD3_SA + TD3_S + j] = a;
                    }
                }
            }
        }
    }
    TEAM_LDS_BARRIER();
    // critic.Q1 forward (fc1 = the first 16 tiles of the packed W14): g1 -> X; the gradient of -mean Q1 at its second hidden layer, dg2 = -(1/B) w3 (g2 > 0) -> Y
    {
        const Blk k = blk_ids(wv);
        mg = blk_dense_relu_mask<3, false>(pre, mkrs(P.p_c_w14, (size_t)32 * 3 * 1024), 4 * k.w, Sb, PB_LD_SA, A.c_b1, Xb, k);
        pre = blk_pre<4, 16>(mkrs(P.p_c_w2, (size_t)16 * 16 * 1024), 4 * k.w, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids(wv);
        floatx4 acc[4], bz[4], wz[4];
        blk_zero(acc);
        blk_mm<4, 16>(pre, mkrs(P.p_c_w2, (size_t)16 * 16 * 1024), 4 * k.w, Xb, BLK_LD_W, 0, acc, k, [&]() {
#pragma unroll
            for (int t = 0; t < 4; t++) { bz[t] = blk_vec4(A.c_b2 + 16 * (4 * k.w + t) + 4 * k.g); wz[t] = blk_vec4(A.c_w3 + 16 * (4 * k.w + t) + 4 * k.g); }
        });
        const float ginv = -1.f / (float)B;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            floatx4 v;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = acc[t][i] + bz[t][i] > 0.f ? ginv * wz[t][i] : 0.f;
            *reinterpret_cast<floatx4 *>(Yb + blk_at(BLK_LD_W, k.r, 4 * (4 * k.w + t) + k.g)) = v;
        }
        pre = blk_pre<4, 16>(mkrs(P.p_c_w2t, (size_t)16 * 16 * 1024), 4 * k.w, k);
    }
    TEAM_LDS_BARRIER();
    // dg1 = (W2^T dg2)(g1 > 0) -> X (g1 is dead: dg2 is complete)
    {
        const Blk k = blk_ids(wv);
        blk_back_mask<16, true>(pre, mkrs(P.p_c_w2t, (size_t)16 * 16 * 1024), 4 * k.w, Yb, BLK_LD_W, mg, Xb, k);
    }
    TEAM_LDS_BARRIER();
    // d/d action = (W1^T dg1)[26:44], through the tanh: dz = that * (max_a - a^2 / max_a): two tiles, k split over the waves (partials through Y: dg2 is dead)
    {
        const Blk k = blk_ids(wv);
        blk_split_k(mkrs(P.p_c_w1ta, (size_t)2 * 16 * 1024), Xb, Yb, k);
        pre = blk_pre<4, 2>(mkrs(P.p_a_w3t, (size_t)16 * 2 * 1024), 4 * k.w, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids(wv);
        if (k.w < 2) {
            const floatx4 z = blk_split_sum(Yb, k.w, k);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 16 * k.w + 4 * k.g + i;
                if (j < TD3_A) {
                    const float a = Sb[blk_lin(PB_LD_SA, k.r, TD3_S + j)];
                    const float dz = z[i] * (A.max_a - a * a / A.max_a);
                    Zb[blk_lin(PB_LD_DZ, k.r, j)] = dz;
                    if (b0 + k.r < B) A.dz[(size_t)(b0 + k.r) * TD3_A + j] = dz;
                }
            }
        }
    }
    TEAM_LDS_BARRIER();
    // back through the actor: dp2 = (W3^T dz)(p2 > 0) -> Y, dp1 = (W2^T dp2)(p1 > 0) -> X
    {
        const Blk k = blk_ids(wv);
        blk_back_mask<2, false>(pre, mkrs(P.p_a_w3t, (size_t)16 * 2 * 1024), 4 * k.w, Zb, PB_LD_DZ, m2, Yb, k);
        pre = blk_pre<4, 16>(mkrs(P.p_a_w2t, (size_t)16 * 16 * 1024), 4 * k.w, k);
    }
    TEAM_LDS_BARRIER();
    blk_flush(Yb, A.dp2, TD3_H, 0, b0, B, wv);
    {
        const Blk k = blk_ids(wv);
        blk_back_mask<16, true>(pre, mkrs(P.p_a_w2t, (size_t)16 * 16 * 1024), 4 * k.w, Yb, BLK_LD_W, m1, Xb, k);
    }
    TEAM_LDS_BARRIER();
    blk_flush(Xb, A.dp1, TD3_H, 0, b0, B, wv);
    if (A.adam_step && wv == 0 && blk_ids(wv).lane == 0) {
        if (atomicAdd(A.done_count, 1) == (int)((B + BLK_R - 1) / BLK_R) - 1) { A.done_count[0] = 0; A.adam_step[0] += 1.f; }
    }
}

// ---- weight gradients of a LARGE batch: every job of a pass in ONE launch of single-wave workgroups, reduced in two deterministic stages.
//      dW[n][k] = sum_b dH[b][n] X[b][k] (td3.py:323 / :341 .backward() of nn.Linear) is a reduction over the batch with few outputs (<= 256 x 256): the batch is
//      cut into `chunks`; workgroup (job, 32 x 64 output tile, chunk) forms its partial tile with v_mfma_f32_32x32x2_f32 -- both operands straight from the
//      row-major activations, 128 contiguous bytes per half wave and row, no transpose, no LDS -- and STORES it into partial[chunk][...] (same layout as the
//      flat gradient bucket); plentd3_adam_big then adds the chunks in chunk order where it takes the optimiser step.  No atomics: same bits every run, and the
//      gradient never makes a round trip through the bucket.  Four waves per workgroup, each a quarter of the chunk's rows, added through LDS in wave order.
//      kind 1 = a critic's head row (N = 1, K <= 256: dW3[k] = sum_b dq[b] c2[b][k]): plain multiply-adds, 4 columns per lane.
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define WGB_NW 4
__global__ __launch_bounds__(64 * WGB_NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wgrad_big(PlenTd3WgradBig G) {
    BLK_SETPRIO();
    __shared__ float red[WGB_NW - 1][64][33];             // the partial tiles of waves 1..3 (33: the lanes' rows fall on different banks)
    int j = 0;
#pragma unroll
    for (int q = 1; q < PLENTD3_WGRAD_BIG_JOBS; q++) j += (q < G.n_jobs && (int)blockIdx.x >= G.job[q].wg0) ? 1 : 0;
    const PlenTd3WgradBigJob &J = G.job[j];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), local = (int)blockIdx.x - J.wg0;
    const int chunk = local % G.chunks, tile = local / G.chunks;
    // the chunk's rows in four consecutive quarters, one per wave
    const int quarter = G.rows_per_chunk / WGB_NW;
    const int r0 = min(G.B, chunk * G.rows_per_chunk + w * quarter), r1 = min(G.B, r0 + quarter);
    float *out = G.partial + (size_t)chunk * G.stride;
    if (J.kind == 1) {
        floatx4 acc = {0, 0, 0, 0};
        const bool in = 4 * lane < J.K;
        const rsrc_t rd = mkrs(J.dH, ((size_t)(G.B - 1) * J.ds + 1) * 4), rx = mkrs(J.X, ((size_t)(G.B - 1) * J.xs + J.K) * 4);
        const uint32_t cx = in ? (uint32_t)lane * 16u : 0x7fffff00u;        // (offsets beyond the matrices read as zero: rows past the quarter, columns past K)
#pragma unroll 1
        for (int b = r0; b < r1; b += 8) {
            float d[8]; floatx4 xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t orow = b + u < r1 ? (uint32_t)(b + u) : 0x1fffffu;
                d[u] = bload1(rd, orow * (uint32_t)J.ds * 4u, 0);
                xv[u] = bload4(rx, orow * (uint32_t)J.xs * 4u + cx, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc += d[u] * xv[u];
        }
        if (w > 0) {
#pragma unroll
            for (int u = 0; u < 4; u++) red[w - 1][lane][u] = acc[u];
        }
        __syncthreads();
        if (w == 0 && in) {
#pragma unroll
            for (int q = 0; q < WGB_NW - 1; q++)
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] += red[q][lane][u];
#pragma unroll
            for (int u = 0; u < 4; u++) out[J.goff + 4 * lane + u] = acc[u];
        }
        return;
    }
    const int tk = (J.K + 63) / 64;
    const int n0 = (tile / tk) * 32, k0 = (tile % tk) * 64;
    const int col = lane & 31, half = lane >> 5;
    const bool na = n0 + col < J.N, ka0 = k0 + col < J.K, ka1 = k0 + 32 + col < J.K;
    floatx16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    float bsum = 0.f;
    // buffer addressing: one 32-bit offset per load; the descriptors END at the wave's last row, so rows past its quarter read as zero without a test, and a
    // column past the layer gets an offset beyond everything.  Offsets are a running base (first row pair of the set) + a scalar step per pair: a table of
    // per-pair offsets, which the compiler builds when it can, was spilled to scratch and reloaded in the loop behind s_waitcnt vmcnt(0)
    const rsrc_t ra_ = mkrs(J.dH, (size_t)r1 * J.ds * 4), rb_ = mkrs(J.X, (size_t)r1 * J.xs * 4);
    const uint32_t oob = 0x7fffff00u;
    const uint32_t ca = na ? (uint32_t)(n0 + col) * 4u : oob, cb0 = ka0 ? (uint32_t)(k0 + col) * 4u : oob, cb1 = ka1 ? (uint32_t)(k0 + 32 + col) * 4u : oob;
    const uint32_t sa = (uint32_t)J.ds * 8u, sb = (uint32_t)J.xs * 8u;             // bytes per row pair
    constexpr int NP = 4;                             // row pairs per operand set
    struct Set { float av[NP], x0[NP], x1[NP]; };
    auto load = [&](int b, Set &S) {
        const uint32_t row = (uint32_t)(b + half);
        uint32_t oa = row * (uint32_t)J.ds * 4u + ca, ob0 = row * (uint32_t)J.xs * 4u + cb0, ob1 = row * (uint32_t)J.xs * 4u + cb1;
        asm volatile("" : "+v"(oa), "+v"(ob0), "+v"(ob1));
#pragma unroll
        for (int u = 0; u < NP; u++) {
            S.av[u] = bload1(ra_, oa + (uint32_t)u * sa, 0);
            S.x0[u] = bload1(rb_, ob0 + (uint32_t)u * sb, 0);
            S.x1[u] = bload1(rb_, ob1 + (uint32_t)u * sb, 0);
        }
    };
    auto mma = [&](const Set &S) {
#pragma unroll
        for (int u = 0; u < NP; u++) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(S.av[u], S.x0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(S.av[u], S.x1[u], acc1, 0, 0, 0);
            bsum += S.av[u];
        }
    };
    // three sets of operands in flight: the rows two sets ahead are requested before the MFMAs of the current set (one wave per SIMD and workgroup: nobody
    // else hides the trip to memory; a set is 8 MFMAs = 512 matrix-pipe cycles)
    Set S0, S1, S2;
    load(r0, S0); load(r0 + 2 * NP, S1);
#pragma unroll 1
    for (int b = r0; b < r1; b += 6 * NP) {
        load(b + 4 * NP, S2);
        __builtin_amdgcn_sched_barrier(0);
        mma(S0);
        __builtin_amdgcn_sched_barrier(0);
        load(b + 6 * NP, S0);
        __builtin_amdgcn_sched_barrier(0);
        mma(S1);
        __builtin_amdgcn_sched_barrier(0);
        load(b + 8 * NP, S1);
        __builtin_amdgcn_sched_barrier(0);
        mma(S2);
        __builtin_amdgcn_sched_barrier(0);
    }
    bsum += __shfl_xor(bsum, 32);
    if (w > 0) {
#pragma unroll
        for (int v = 0; v < 16; v++) { red[w - 1][lane][v] = acc0[v]; red[w - 1][lane][16 + v] = acc1[v]; }
        if (half == 0) red[w - 1][lane][32] = bsum;
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int q = 0; q < WGB_NW - 1; q++) {
#pragma unroll
            for (int v = 0; v < 16; v++) { acc0[v] += red[q][lane][v]; acc1[v] += red[q][lane][16 + v]; }
            bsum += red[q][col][32];
        }
        // result layout of the 32x32 MFMA: lane l holds column j = l % 32 and rows i = 8 * (v / 4) + 4 * (l / 32) + v % 4, v = 0..15.  Buffer stores: an offset
        // beyond the layer's N x K block (rows >= N, columns >= K) is dropped by the hardware
        const rsrc_t ro = mkrs(out + J.goff, (size_t)J.N * J.K * 4);
        const uint32_t c0 = ka0 ? (uint32_t)(k0 + col) * 4u : oob, c1 = ka1 ? (uint32_t)(k0 + 32 + col) * 4u : oob;
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const uint32_t orow = (uint32_t)(n0 + 8 * (v / 4) + 4 * half + (v % 4)) * (uint32_t)J.K * 4u;
            bstore1(acc0[v], ro, orow + c0, 0);
            bstore1(acc1[v], ro, orow + c1, 0);
        }
        if (J.boff >= 0 && k0 == 0 && half == 0 && na) out[J.boff + n0 + col] = bsum;
    }
}

// ---- plentd3_adam on the partial gradients of k_wgrad_big: g[i] = bucket[i] + sum over chunks of partial[chunk][i] (chunk order), then the step of k_adam
//      (same arithmetic per element), the Polyak update and the parameter copy; the bucket is left zero.  reduce_only: bucket[i] = that sum and nothing else
//      (several ranks: the bucket is all-reduced before plentd3_adam takes the step).
__global__ __launch_bounds__(256) void k_adam_big(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, float *step, int *done_count, int n,
                                                  double lr, double b1d, double b2d, float eps, float *__restrict__ target, float tau, float *__restrict__ copy_out,
                                                  const float *__restrict__ partial, int chunks, int stride, int reduce_only) {
    BLK_SETPRIO();
    const float t = reduce_only ? 1.f : step[0] + 1.f;
    const AdamCoef c = adam_coef(t, lr, b1d, b2d, eps, tau);
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += gridDim.x * 1024) {
        const int cnt = min(4, n - i0);
        f4 g4 = {0, 0, 0, 0};
        if (cnt == 4) {
            g4 = *reinterpret_cast<const f4 *>(g + i0);
            for (int k = 0; k < chunks; k++) g4 += *reinterpret_cast<const f4 *>(partial + (size_t)k * stride + i0);
        } else {
            for (int u = 0; u < cnt; u++) { g4[u] = g[i0 + u]; for (int k = 0; k < chunks; k++) g4[u] += partial[(size_t)k * stride + i0 + u]; }
        }
        if (reduce_only) {
            for (int u = 0; u < cnt; u++) g[i0 + u] = g4[u];
            continue;
        }
        for (int u = 0; u < cnt; u++) {
            const int i = i0 + u;
            float mi = m[i], vi = v[i], pi = p[i];
            adam_one(g4[u], mi, vi, pi, c);
            m[i] = mi; v[i] = vi; p[i] = pi; g[i] = 0.f;
            if (target) target[i] = adam_polyak(pi, target[i], tau);
            if (copy_out) copy_out[i] = pi;
        }
    }
    if (reduce_only) return;
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(done_count, 1) == (int)gridDim.x - 1) { done_count[0] = 0; step[0] = t; }
}

// ---- the collect phase's action (plen_td3.py:101-104): a = clamp(actor(state) + N(0, sigma), +-max_a), 16 envs per workgroup, in the shape of the passes above
//      (packed weights, transposed products, activations in LDS).  It is on every vector step's critical path -- a sub-batch's env launch cannot start before it --
//      and k_actor_rows4 (unpacked weights, activations through global memory) took 20 us per 2048 envs alone, 30-67 us beside the other collector's env launch.
//      Same draws as k_actor_rows (noise index = element index); p1 / p2 are not written.
// (BLK_ACTOR_NW waves, 8 as in k_critic_block: two waves per SIMD in the registers of one; 4 = round 5's form.  The exploration noise -- Philox + Box-Muller, ~8 k cycles
//  when the two waves that own the output tiles drew it after the last sum -- is drawn by ALL threads at the kernel's start, beside the state rows' trip from memory,
//  and parked in LDS: same counters, same values.)
#ifndef BLK_ACTOR_NW
#define BLK_ACTOR_NW 8
#endif
__global__ __launch_bounds__(64 * BLK_ACTOR_NW) __attribute__((amdgpu_waves_per_eu(BLK_ACTOR_NW, BLK_ACTOR_NW))) void k_actor_block(PlenTd3ActorBlock P) {
    constexpr int NW = BLK_ACTOR_NW, NT = 16 / NW;
    static_assert(NW == 4 || NW == 8, "four waves of 64 features or eight of 32");
    const PlenTd3ActorRows &A = P.rows;
    __shared__ __attribute__((aligned(16))) float Sb[BLK_R * PB_LD_SA];
    __shared__ __attribute__((aligned(16))) float Xb[BLK_R * BLK_LD_W];
    __shared__ __attribute__((aligned(16))) float Yb[BLK_R * BLK_LD_W];
    __shared__ float nz[BLK_R * TD3_A];          // noise * sigma of the block's 16 x 18 actions
    static_assert(sizeof(float) * (BLK_R * PB_LD_SA + 2 * BLK_R * BLK_LD_W + BLK_R * TD3_A) <= 40960, "four workgroups' worth of LDS per compute unit");
    const int B = A.B, b0 = blockIdx.x * BLK_R;
    BLK_WAVE();
    BlkPre<NT> pre;
    {
        const Blk k = blk_ids(wv);
        float sv[(BLK_R * PB_LD_SA + 64 * NW - 1) / (64 * NW)];
#pragma unroll
        for (int u = 0; u < (BLK_R * PB_LD_SA + 64 * NW - 1) / (64 * NW); u++) {
            const int i = BLK_TID(k) + 64 * NW * u, row = i / PB_LD_SA, c = i - row * PB_LD_SA;
            sv[u] = (i < BLK_R * PB_LD_SA && c < TD3_S) ? A.state[(size_t)min(b0 + row, B - 1) * TD3_S + c] : 0.f;
        }
        for (int e = BLK_TID(k); e < BLK_R * TD3_A; e += 64 * NW) {          // (beside the loads above)
            const int row = e / TD3_A, j = e - row * TD3_A;
            nz[e] = rng_normal(A.rng, 2u, (uint32_t)(min(b0 + row, B - 1) * TD3_A + j)) * A.sigma;
        }
#pragma unroll
        for (int u = 0; u < (BLK_R * PB_LD_SA + 64 * NW - 1) / (64 * NW); u++) {
            const int i = BLK_TID(k) + 64 * NW * u;
            if (i < BLK_R * PB_LD_SA) Sb[i] = sv[u];
        }
        pre = blk_pre<NT, 2>(mkrs(P.p_a_w1, (size_t)16 * 2 * 1024), NT * k.w, k);          // (after the draw: its registers would not fit beside Philox's at 64)
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids(wv);
        blk_dense_relu<NT, 2, false>(pre, mkrs(P.p_a_w1, (size_t)16 * 2 * 1024), NT * k.w, Sb, PB_LD_SA, 0, A.a_b1, Xb, BLK_LD_W, NT * k.w, k);
        pre = blk_pre<NT, 16>(mkrs(P.p_a_w2, (size_t)16 * 16 * 1024), NT * k.w, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids(wv);
        blk_dense_relu<NT, 16>(pre, mkrs(P.p_a_w2, (size_t)16 * 16 * 1024), NT * k.w, Xb, BLK_LD_W, 0, A.a_b2, Yb, BLK_LD_W, NT * k.w, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids(wv);
        blk_split_k<NW>(mkrs(P.p_a_w3, (size_t)2 * 16 * 1024), Yb, Xb, k);
    }
    TEAM_LDS_BARRIER();
    {
        const Blk k = blk_ids(wv);
        if (k.w < 2) {
            const floatx4 z = blk_split_sum<NW>(Xb, k.w, k);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 16 * k.w + 4 * k.g + i, b = b0 + k.r;
                if (j < TD3_A && b < B) A.action[b * TD3_A + j] = fminf(fmaxf(A.max_a * tanhf(z[i] + A.a_b3[j]) + nz[k.r * TD3_A + j], -A.max_a), A.max_a);
            }
        }
    }
}

// ---- development (scripts/gpu_clock_probe.py): a workload of matrix-core instructions ONLY -- no memory, no LDS, 20 registers, one wave per workgroup -- to see what
//      the matrix pipe by itself costs env waves that share its SIMD (issue slots: an MFMA holds the vector issue port for 8 of its 32 cycles)
__global__ __launch_bounds__(64) void k_dev_mfma_spin(int iters, float *sink) {
    floatx4 acc[4] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
    const float a = (float)threadIdx.x * 1e-3f, b = 1.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    if (sink) sink[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
