// td3_team.hip -- the row-local part of a TD3 update for SMALL batches (the reference's own recipe: batch 100, one update per env-step,
// plen_td3.py:28, :119-120; td3.py:259-356), included by td3_kernels.hip after td3_rows.hip (buffer-addressing helpers, struct layouts).
//
// Why a second shape of the same arithmetic: with one update per env-step the updates form a dependent chain (update k reads the weights
// update k - 1 wrote), so what counts is the LATENCY of one update, not its throughput.  At batch 100
//   * the layer-by-layer path is ~35 launches of library GEMMs whose 256 x 112 tiles take 31 us whatever the size: 355 us per critic update,
//     240 us per policy update (profiles/r04_td3ref_before_kernel_stats.txt);
//   * td3_rows.hip's one-wave-per-16-rows kernels put the whole chain of 13 dense layers on ONE wave per row block: 7 waves on a 256-CU chip.
// Here a block of FOUR batch rows belongs to a TEAM of 8 waves (one 512-thread workgroup, 2 waves per SIMD): 25 workgroups at batch 100.  Every dense
// layer's output columns are split over the team -- a wave owns 32 (or 2 x 32) columns --, the twin critics run side by side on the two halves of
// the team, and layers are separated by workgroup barriers instead of launches.  The block's activations live in LDS (4 rows x ~2400 floats) from
// the gathered replay rows to the last gradient: a layer reads its input there and writes its output there, so a barrier between layers has to
// wait for LDS only (fence on the local address space + s_barrier: no drain of the stores to global memory, no trip to L2 for the next layer's
// input).  What the weight-gradient launch needs is ALSO stored to global memory, asynchronously.
//
// The matrix-core shape is v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 outer products per instruction.  Lane l feeds block l / 4 with
// A[i = l % 4] and B[j = l % 4] and receives D[i = 0..3][j = l % 4].  The blocks are used as 8 column groups x 2 halves of the reduction index:
// lane l owns output column (l % 32) of the wave's tile, all four batch rows, and the k values of half l / 32; the two halves are added at the end
// (one cross-lane exchange).  Four rows instead of sixteen per block quarters a workgroup's matrix-core time (same 64 flop per cycle and SIMD);
// a first version with 16-row blocks and 16 x 16 x 4 tiles ran at 70 % of one compute unit's matrix-core peak and could go no further on 7 compute
// units (58 us per critic pass).
//
// Operand loads are COALESCED: a lane that loads what its MFMA needs reads 16 bytes of "its" matrix row, so the 16 lanes the vector cache serves
// together touch 16 different cache lines and a 1 KB wave load costs 64 tag look-ups instead of 8 -- with 8 waves per compute unit the look-up rate
// WAS the kernel's speed (stamps: 28 k cycles for a 256 x 256 layer against 8 k of matrix-core time).  So a stage's 32 rows x 64 k of W are loaded as
// whole 256-byte row segments per 16 lanes, parked in the wave's private LDS (chunks permuted by the row, so that the reads are free of bank
// conflicts) and read back as the 16 bytes per lane the MFMAs consume.  The four activation rows need no staging or replication: the MFMA's block
// broadcast (CBSZ / ABID) hands one block's A values to all blocks of a half, so ONE 16-byte read per lane (from the block's LDS activations) feeds a whole stage.
//
// Sums: every dot product adds its k values in ascending order per (k % 4, half) class, then the four classes, then the two halves; the critics'
// scalar heads add per-wave partial sums in wave order; loss and head-bias gradients add per-workgroup partials in workgroup order.  Nothing depends on timing: same bits every run.

#define TEAM_NW 8
#define QB 4                                   // batch rows per workgroup
#define TEAM_B_F (32 * 64)                     // floats of a staged W tile: 32 rows x 64 k
#define TEAM_LDS_PER_WAVE TEAM_B_F
#ifdef TEAM_STAMPS            // development (scripts/gpu_td3_team_stamps.py): the clock after every barrier, parked in unused columns of workgroup 0's scratch row of t1
#define TEAM_SYNC() do { TEAM_LDS_BARRIER(); if (threadIdx.x == 0 && blockIdx.x == 0) { reinterpret_cast<unsigned long long *>(A.t1 + TD3_H + 8)[team_stamp_k++] = __builtin_readcyclecounter(); } } while (0)
#else
#define TEAM_SYNC() TEAM_LDS_BARRIER()
#endif
// a workgroup barrier that orders LDS accesses only: stores to global memory stay in flight across it (nothing in these kernels reads back from
// global memory what another wave of the same launch wrote, except where noted)
#define TEAM_LDS_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

// Geometry of a wave's tile.  Built per phase from an opaque lane id (as td3_rows.hip's fresh_rows: lane-derived offsets are then not common
// subexpressions of the whole kernel, kept alive -- and spilled -- across all phases).
struct Quad {
    int b0, B, lane, col, half, row4;
};
static __device__ __forceinline__ Quad fresh_quad(int b0, int B) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return Quad{b0, B, lane, lane & 31, lane >> 5, lane & 3};
}
#define QPHASE() const Quad q = fresh_quad(b0, B); const int lane = q.lane, col = q.col; (void)lane; (void)col

// The A operand needs no staging at all: with CBSZ = 3 an MFMA broadcasts the A values of ONE block (ABID = 0..7) to the 8 blocks of its group -- here
// the two groups are the two halves of the reduction index.  So lane l reads 16 bytes of row l % 4 at k = 32 (l / 32) + 4 ((l % 32) / 4)
// of the stage, and MFMA (c, j) of the stage broadcasts element j of the lanes of blocks c and 8 + c: k = 32 half + 4 c + j, the k its B operand holds.
// (the activations are in LDS: xa = the block's row 0 of the layer's input, row stride ld floats; 16-byte aligned)
static __device__ __forceinline__ const float *quad_a_ptr(const Quad &q, const float *xa, int ld) { return xa + q.row4 * ld + 32 * q.half + 4 * (q.col >> 2); }
template <bool KGUARD>
static __device__ __forceinline__ floatx4 quad_a_guard(floatx4 a, int k0, int K, const Quad &q) {
    if constexpr (KGUARD) {
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = k0 + 32 * q.half + 4 * (q.col >> 2) + j < K ? a[j] : 0.f;
    }
    return a;
}
// (four accumulators, one per k % 4: back-to-back MFMAs on ONE accumulator wait for each other's result -- 128 of them per 256-long product)
#define QUAD_MFMA_STAGE(ACC, FA, FB_OF_C_J)                                                            \
    do {                                                                                               \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(0, 0), ACC[0], 3, 0, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(0, 1), ACC[1], 3, 0, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(0, 2), ACC[2], 3, 0, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(0, 3), ACC[3], 3, 0, 0); \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(1, 0), ACC[0], 3, 1, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(1, 1), ACC[1], 3, 1, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(1, 2), ACC[2], 3, 1, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(1, 3), ACC[3], 3, 1, 0); \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(2, 0), ACC[0], 3, 2, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(2, 1), ACC[1], 3, 2, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(2, 2), ACC[2], 3, 2, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(2, 3), ACC[3], 3, 2, 0); \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(3, 0), ACC[0], 3, 3, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(3, 1), ACC[1], 3, 3, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(3, 2), ACC[2], 3, 3, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(3, 3), ACC[3], 3, 3, 0); \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(4, 0), ACC[0], 3, 4, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(4, 1), ACC[1], 3, 4, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(4, 2), ACC[2], 3, 4, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(4, 3), ACC[3], 3, 4, 0); \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(5, 0), ACC[0], 3, 5, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(5, 1), ACC[1], 3, 5, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(5, 2), ACC[2], 3, 5, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(5, 3), ACC[3], 3, 5, 0); \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(6, 0), ACC[0], 3, 6, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(6, 1), ACC[1], 3, 6, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(6, 2), ACC[2], 3, 6, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(6, 3), ACC[3], 3, 6, 0); \
        ACC[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[0], FB_OF_C_J(7, 0), ACC[0], 3, 7, 0); ACC[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[1], FB_OF_C_J(7, 1), ACC[1], 3, 7, 0); ACC[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[2], FB_OF_C_J(7, 2), ACC[2], 3, 7, 0); ACC[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(FA[3], FB_OF_C_J(7, 3), ACC[3], 3, 7, 0); \
    } while (0)
static __device__ __forceinline__ floatx4 quad_combine(const floatx4 (&a)[4]) {
    floatx4 r = (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] += __shfl_xor(r[i], 32);           // the two halves of the reduction index
    return r;
}

// acc[i] (rows b0 + i, column n0 + col; valid in every lane) = sum_{k < K} X[b0 + i][xcol0 + k] W[n0 + col][k]      (W: nn.Linear's [out][in], row stride ldw)
// Rows of W at or beyond the end of rw read as zero (the 18-wide output layers); k beyond K must be harmless: KGUARD zeroes A there (B then holds
// finite values of the next row, or zeros).
// wp != null (round 6): W in the team's OPERAND order (plentd3_pack with team = 1, kept current by plentd3_wgrad_adam_group): the lane's eight B pieces of a stage are eight
// float4 at ((tile NS + stage) 8 + c) 64 + lane -- every wave load one contiguous kilobyte, nothing parked, nothing read back from LDS.  Same values into the same
// matrix instructions in the same order as the parked path: bit-identical results (tests), 9 us less per batch-100 update (profiles/r06_h_batch100_floor.json).
template <bool KGUARD, int NS>          // NS = stages of 64 k: K <= 64 NS
static __device__ __forceinline__ floatx4 quad_nt(const float *xa, int ldx, rsrc_t rw, int ldw, int n0, int K, const Quad &q, float *lds, const float *wp = nullptr) {
    const int lane = q.lane, brow = lane >> 4, bch = lane & 15;
    const float *xp = quad_a_ptr(q, xa, ldx);
    struct Stage { floatx4 b[8]; };
    if (wp) {          // (wave-uniform)
        const rsrc_t rp = mkrs(wp + (size_t)(n0 >> 5) * NS * 2048, (size_t)NS * 2048 * 4);
        Stage S[NS];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int c = 0; c < 8; c++) S[s].b[c] = bload4(rp, (uint32_t)(((s * 8 + c) * 64 + lane) * 16), 0u);
        __builtin_amdgcn_sched_barrier(0);
        floatx4 acc[4] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const floatx4 fa = quad_a_guard<KGUARD>(*reinterpret_cast<const floatx4 *>(xp + 64 * s), 64 * s, K, q);
#define QUAD_FB(c, j) S[s].b[c][j]
            QUAD_MFMA_STAGE(acc, fa, QUAD_FB);
#undef QUAD_FB
        }
        return quad_combine(acc);
    }
    auto load = [&](int k0, Stage &S) {
#pragma unroll
        for (int i = 0; i < 8; i++) {                      // rows brow + 4 i of the tile; the chunk this lane fetches is the one whose parking slot is (row, bch)
            const int row = brow + 4 * i, ch = bch ^ (row & 15);
#ifdef TEAM_EXPERIMENT_NO_LOAD                             /* development: what the phases cost without their trips to memory (results are wrong) */
            S.b[i] = floatx4{(float)(row + k0), 1.f, 2.f, (float)ch};
#else
            S.b[i] = bload4(rw, (uint32_t)((n0 + row) * ldw + 4 * ch) * 4u, 4u * (uint32_t)k0);
#endif
        }
    };
    auto park = [&](const Stage &S) {
#ifndef TEAM_EXPERIMENT_NO_PARK
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<floatx4 *>(lds + 4 * (64 * i + lane)) = S.b[i];
#else
        if (lane > 64) *reinterpret_cast<floatx4 *>(lds) = S.b[0] + S.b[1] + S.b[2] + S.b[3] + S.b[4] + S.b[5] + S.b[6] + S.b[7];
#endif
    };
    floatx4 acc[4] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
    auto compute = [&](int k0, const Stage &) {
        const floatx4 fa = quad_a_guard<KGUARD>(*reinterpret_cast<const floatx4 *>(xp + k0), k0, K, q);
        floatx4 fb[8];
#pragma unroll
        for (int c = 0; c < 8; c++) fb[c] = *reinterpret_cast<const floatx4 *>(lds + 4 * (16 * q.col + ((8 * q.half + c) ^ (q.col & 15))));
#define QUAD_FB(c, j) fb[c][j]
        QUAD_MFMA_STAGE(acc, fa, QUAD_FB);
#undef QUAD_FB
    };
    // every stage of 64 k is requested before the first one is used (NS = 4: K = 256 -- 36 x 16 bytes per lane in flight): with four batch rows per block
    // a stage is 256 cycles of MFMAs, nothing to hide a trip to memory behind, so the layer pays for one trip instead of one per stage
    Stage S[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) load(64 * s, S[s]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; s++) {
        park(S[s]);
        compute(64 * s, S[s]);
    }
    return quad_combine(acc);
}

// acc[i] (rows b0 + i, column j0 + col) = sum_{n < Kc} G[b0 + i][gcol0 + n] W[n][j0 + col]    (input gradient of a dense layer: W is the layer's [out = Kc][in])
// B needs no staging: the 32 lanes of a half read 128 contiguous bytes of one row of W per load.  Rows n >= Kc must lie outside rw (they read as zero);
// KGUARD also zeroes A there (a G whose rows are shorter than a stage: the next row's values are not wanted).
template <bool KGUARD, int NS>
static __device__ __forceinline__ floatx4 quad_nn(const float *ga, int ldg, rsrc_t rw, int ldw, int j0, int Kc, const Quad &q) {
    const float *gp = quad_a_ptr(q, ga, ldg);
    const uint32_t wo = (uint32_t)(32 * q.half * ldw + j0 + q.col) * 4u;
    struct Stage { float b[32]; };
    auto load = [&](int k0, Stage &S) {
#pragma unroll
        for (int m = 0; m < 32; m++) S.b[m] = bload1(rw, wo + 4u * (uint32_t)((k0 + m) * ldw), 0);          // (the row term in voffset: range-checked)
    };
    floatx4 acc[4] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
    auto compute = [&](int k0, const Stage &S) {
        const floatx4 fa = quad_a_guard<KGUARD>(*reinterpret_cast<const floatx4 *>(gp + k0), k0, Kc, q);
#define QUAD_FB(c, j) S.b[4 * (c) + (j)]
        QUAD_MFMA_STAGE(acc, fa, QUAD_FB);
#undef QUAD_FB
    };
    // two stages in registers (measured: all four up front, as in quad_nt, is slower here -- 33 separate loads per stage crowd the address path)
    Stage S0, S1;
    load(0, S0);
#pragma unroll
    for (int s = 0; s < NS; s += 2) {
        if (s + 1 < NS) load(64 * (s + 1), S1);
        __builtin_amdgcn_sched_barrier(0);
        compute(64 * s, S0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < NS) {
            if (s + 2 < NS) load(64 * (s + 2), S0);
            __builtin_amdgcn_sched_barrier(0);
            compute(64 * (s + 1), S1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return quad_combine(acc);
}

// v[i] (rows b0 + i, column n0 + col) into the block's LDS activations (L: row 0 of the destination, row stride ldl; may be null) and / or into
// global memory (Y: [B][ldy], rows inside the batch only; may be null): one lane per column (the lower half of the wave), 128 contiguous bytes per row
static __device__ __forceinline__ void quad_store(float *L, int ldl, float *Y, int ldy, int n0, const floatx4 &v, const Quad &q) {
    if (q.lane < 32) {
#pragma unroll
        for (int i = 0; i < QB; i++) {
            if (L) L[i * ldl + n0 + q.col] = v[i];
            if (Y && q.b0 + i < q.B) Y[(size_t)(q.b0 + i) * ldy + n0 + q.col] = v[i];
        }
    }
}

// one 32-column tile of a hidden layer: relu(X W^T + bias)   (the bias is requested before the product)
template <bool KGUARD, int NS>
static __device__ __forceinline__ void quad_dense_relu(const float *xa, int ldx, int K, rsrc_t rw, int ldw, const float *bias, int n0, float *L, int ldl, float *Y, int ldy, const Quad &q, float *lds, const float *wp = nullptr) {
    const float bv = bias[n0 + q.col];
    floatx4 acc = quad_nt<KGUARD, NS>(xa, ldx, rw, ldw, n0, K, q, lds, wp);
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = fmaxf(acc[i] + bv, 0.f);
    quad_store(L, ldl, Y, ldy, n0, acc, q);
}

// sum over the wave's 32 columns (every lane of the lower half holds one; the upper half holds copies)
static __device__ __forceinline__ float quad_colsum(float v, const Quad &q) { return wave_sum(q.lane < 32 ? v : 0.f); }

// LDS activations of a critic block, per row: the gathered replay row | [s2 | target action] | t0 (512) | t1 (256) | c1 (512) | c2 (512) | dh2 (512).
// Row stride = 16 mod 64 dwords: the four rows' 16-byte A reads of one ds_read_b128 lane group fall on 16 different bank quads.
#define CA_BATCH 0
#define CA_SA2 (CA_BATCH + TD3_ROW)
#define CA_T0 (CA_SA2 + TD3_SA)
#define CA_T1 (CA_T0 + 2 * TD3_H)
#define CA_C1 (CA_T1 + TD3_H)
#define CA_C2 (CA_C1 + 2 * TD3_H)
#define CA_DH2 (CA_C2 + 2 * TD3_H)
#define CA_LD 2448
static_assert(CA_DH2 + 2 * TD3_H <= CA_LD && CA_LD % 64 == 16 && TD3_ROW % 4 == 0 && TD3_SA % 4 == 0, "critic block layout");

// ---- td3.py:277-323 for 4 batch rows per workgroup (arguments and outputs as k_critic_rows / plentd3_critic_rows; t0, sa2 and all of t1 but the
//      workgroup's three parked partial sums are not written: those activations never leave LDS) ----
__global__ __launch_bounds__(64 * TEAM_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_critic_team(PlenTd3CriticRows A) {
    __shared__ float qp[4][4][QB];              // [target a, target b, critic a, critic b][wave of the half team][row]: partial heads
    __shared__ float dql[QB][2];                // the loss gradient at the two heads, per row
    __shared__ float noise_l[QB][32];           // the target action's clipped smoothing noise, per row (drawn beside the gather by the waves it leaves idle)
    __shared__ float team_lds[TEAM_NW][TEAM_LDS_PER_WAVE];
    __shared__ __attribute__((aligned(16))) float act[QB * CA_LD];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), wq = w & 3, half = w >> 2;
    const int B = A.B, b0 = blockIdx.x * QB, n_team = (B + QB - 1) / QB;
    float *lds = team_lds[w];
#ifdef TEAM_STAMPS
    int team_stamp_k = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long *>(A.t1 + TD3_H + 8)[0] = __builtin_readcyclecounter();
#endif
    // ---- sample the block's 4 rows of the replay ring (td3.py:166-193; same draw as k_sample_gather) and gather them: one row per wave 0..3
    //      (a row past the end of the batch repeats the last one in LDS -- finite inputs for the block's unused rows -- and is not stored) ----
    if (w < QB) {
        QPHASE();
        const int b = min(b0 + w, B - 1);
        const bool inside = b0 + w < B;
        int64_t id;
        if (A.idx) {
            id = A.idx[b];                       // the caller's rows (golden iterations)
        } else {
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
        const float *src = A.data + (size_t)id * TD3_ROW;
        const float v0 = src[lane], v1 = lane < TD3_ROW - 64 ? src[64 + lane] : 0.f;
        float *al = act + w * CA_LD;
        al[CA_BATCH + lane] = v0;
        if (lane < TD3_ROW - 64) al[CA_BATCH + 64 + lane] = v1;
        if (lane >= TD3_SA) al[CA_SA2 + lane - TD3_SA] = v0;                  // s2 = columns 44..69 of the row: 44..63 here,
        if (lane < TD3_SA + TD3_S - 64) al[CA_SA2 + 64 - TD3_SA + lane] = v1;  // 64..69 there
        if (inside) {
            float *dst = A.batch + (size_t)b * TD3_ROW;
            dst[lane] = v0;
            if (lane < TD3_ROW - 64) dst[64 + lane] = v1;
            if (lane < TD3_S) A.sa_pi[(size_t)b * TD3_SA + lane] = v0;
        }
    } else {
        QPHASE();
        const int i = w - QB;
        if (lane < TD3_A) {
            const int e = min(b0 + i, B - 1) * TD3_A + lane;
            const float z = A.noise ? A.noise[e] : rng_normal(A.rng, 1u, (uint32_t)e);                 // torch.randn_like(action), td3.py:300
            noise_l[i][lane] = fminf(fmaxf(z * A.sigma, -A.clip), A.clip);
        }
    }
    TEAM_SYNC();
    // gathered rows: s 0..25 | a 26..43 | s2 44..69 | r 70 | not_done 71
    // ---- target actor's first layer on s2 (32 columns per wave) ----
    {
        QPHASE();
        quad_dense_relu<true, 1>(act + CA_BATCH + TD3_SA, CA_LD, TD3_S, mkrs(A.at_w1, (size_t)TD3_H * TD3_S * 4), TD3_S, A.at_b1, 32 * w, act + CA_T0, CA_LD, nullptr, 0, q, lds, A.tp_at_w1);
    }
    TEAM_SYNC();
    {
        QPHASE();
        quad_dense_relu<false, 4>(act + CA_T0, CA_LD, TD3_H, mkrs(A.at_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, A.at_b2, 32 * w, act + CA_T1, CA_LD, nullptr, 0, q, lds, A.tp_at_w2);
    }
    TEAM_SYNC();
    // ---- target action (td3.py:299-304): the 18-wide output layer is one tile: wave 0.  Beside it, on the other seven waves, the critics' stacked
    //      first layers on (s, a) -- 16 tiles that depend on the gathered rows only and are not needed before the critics' second layers ----
    if (w != 0) {
        QPHASE();
        const rsrc_t rw = mkrs(A.c_w14, (size_t)2 * TD3_H * TD3_SA * 4);
#pragma unroll 1
        for (int t = w - 1; t < 16; t += TEAM_NW - 1) quad_dense_relu<true, 1>(act + CA_BATCH, CA_LD, TD3_SA, rw, TD3_SA, A.c_b14, 32 * t, act + CA_C1, CA_LD, A.c1, 2 * TD3_H, q, lds, A.tp_c_w14);
    } else {
        QPHASE();
        const float bv3 = col < TD3_A ? A.at_b3[col] : 0.f;
        const floatx4 z = quad_nt<false, 4>(act + CA_T1, CA_LD, mkrs(A.at_w3, (size_t)TD3_A * TD3_H * 4), TD3_H, 0, TD3_H, q, lds, A.tp_at_w3);
        if (lane < TD3_A) {
#pragma unroll
            for (int i = 0; i < 4; i++) act[i * CA_LD + CA_SA2 + TD3_S + col] = fminf(fmaxf(A.max_a * tanhf(z[i] + bv3) + noise_l[i][col], -A.max_a), A.max_a);
        }
    }
    TEAM_SYNC();
    // ---- both target critics' first layers stacked (W14 = [fc1.w; fc4.w]): 64 columns per wave ----
    {
        QPHASE();
        const rsrc_t rw = mkrs(A.ct_w14, (size_t)2 * TD3_H * TD3_SA * 4);
#pragma unroll 1
        for (int t = 0; t < 2; t++) quad_dense_relu<true, 1>(act + CA_SA2, CA_LD, TD3_SA, rw, TD3_SA, A.ct_b14, 64 * w + 32 * t, act + CA_T0, CA_LD, nullptr, 0, q, lds, A.tp_ct_w14);
    }
    TEAM_SYNC();
    // ---- second layers + heads: waves 0..3 = critic a, 4..7 = critic b, 64 columns each; target critics, then the critics themselves (c2 kept) ----
    {
        QPHASE();
#pragma unroll 1
        for (int k = 0; k < 2; k++) {                       // 0: target critics on t0, 1: critics on c1
            const float *W2 = k ? (half ? A.c_w5 : A.c_w2) : (half ? A.ct_w5 : A.ct_w2), *b2 = k ? (half ? A.c_b5 : A.c_b2) : (half ? A.ct_b5 : A.ct_b2);
            const float *w3 = k ? (half ? A.c_w6 : A.c_w3) : (half ? A.ct_w6 : A.ct_w3);
            const float *W2p = k ? (half ? A.tp_c_w5 : A.tp_c_w2) : (half ? A.tp_ct_w5 : A.tp_ct_w2);
            const rsrc_t rw = mkrs(W2, (size_t)TD3_H * TD3_H * 4);
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int t = 0; t < 2; t++) {
                const int n0 = 64 * wq + 32 * t;
                const float bv = b2[n0 + col], wv = w3[n0 + col];
                floatx4 acc = quad_nt<false, 4>(act + (k ? CA_C1 : CA_T0) + half * TD3_H, CA_LD, rw, TD3_H, n0, TD3_H, q, lds, W2p);
#pragma unroll
                for (int i = 0; i < 4; i++) { acc[i] = fmaxf(acc[i] + bv, 0.f); part[i] += acc[i] * wv; }
                if (k) quad_store(act + CA_C2, CA_LD, A.c2, 2 * TD3_H, half * TD3_H + n0, acc, q);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float s = quad_colsum(part[i], q);
                if (lane == 0) qp[2 * k + half][wq][i] = s;
            }
        }
    }
    TEAM_SYNC();
    // ---- clipped double-Q target, loss and its gradient at the heads (td3.py:306-319): one lane per row ----
    if (w == 0) {
        QPHASE();
        float lsum = 0.f, ga = 0.f, gb = 0.f;
        const int b = b0 + lane;
        if (lane < QB && b < B) {
            auto head = [&](int k, const float *b3) { return ((qp[k][0][lane] + qp[k][1][lane]) + (qp[k][2][lane] + qp[k][3][lane])) + b3[0]; };
            const float *row = act + lane * CA_LD + CA_BATCH;
            const float y = row[TD3_ROW - 2] + row[TD3_ROW - 1] * A.gamma * fminf(head(0, A.ct_b3), head(1, A.ct_b6));
            const float inv = 1.f / (float)B;
            const float ea = head(2, A.c_b3) - y, eb = head(3, A.c_b6) - y;
            ga = 2.f * ea * inv; gb = 2.f * eb * inv;
            A.dq[2 * b] = ga; A.dq[2 * b + 1] = gb;
            dql[lane][0] = ga; dql[lane][1] = gb;
            lsum = ea * ea * inv + eb * eb * inv;
        }
        lsum = wave_sum(lsum); ga = wave_sum(ga); gb = wave_sum(gb);
        // the loss and the two head biases' gradients: one partial per workgroup, parked in unused columns of the workgroup's own scratch row; the last
        // workgroup to get HERE sums them in workgroup order (the sums do not depend on the arrival order: same bits every run; loss[0] is a plain store
        // that nobody has to zero first) and advances the counters -- every workgroup is past its draws from the random stream by now.  (At the END of the
        // kernel, behind a barrier and a fence that waited for the last layer's stores, and with the partials fetched one after the other, this
        // protocol was 8 us of every update.)
        int last = 0;
        if (lane == 0) last = handoff_last(A.t1 + (size_t)b0 * 2 * TD3_H + TD3_H, lsum, ga, gb, A.done_count, n_team);
        if (__builtin_amdgcn_readfirstlane(last)) {
            handoff_acquire();
            float l = 0.f, sa = 0.f, sb = 0.f;
            for (int c0 = 0; c0 < n_team; c0 += 64) {            // one workgroup's partials per lane, then added in lane order
                const int k = c0 + lane;
                float pl = 0.f, pa = 0.f, pb = 0.f;
                if (k < n_team) {
                    const float *park = A.t1 + (size_t)k * QB * 2 * TD3_H + TD3_H;
                    pl = __hip_atomic_load(park, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pa = __hip_atomic_load(park + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pb = __hip_atomic_load(park + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int m = min(64, n_team - c0);
                for (int i = 0; i < m; i++) {
                    l += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pl), i));
                    sa += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pa), i));
                    sb += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pb), i));
                }
            }
            if (lane == 0) {
                A.loss[0] = l; A.db3a[0] += sa; A.db3b[0] += sb;
                A.done_count[0] = 0;
                if (A.rng_bump) A.rng_bump[1] += 1;
                if (A.adam_step) A.adam_step[0] += 1.f;          // the optimiser step the weight-gradient launch is about to take (PlenTd3AdamFused.step_advanced)
            }
        }
    }
    TEAM_SYNC();
    // ---- dh2 = dq (x) w3 where the hidden unit was active: wave = (row, critic), 4 columns per lane ----
    {
        QPHASE();
        const int c = half;
        const floatx4 wv = *reinterpret_cast<const floatx4 *>((c ? A.c_w6 : A.c_w3) + 4 * lane);
        const float d = b0 + wq < B ? dql[wq][c] : 0.f;
        float *al = act + wq * CA_LD;
        const floatx4 h = *reinterpret_cast<const floatx4 *>(al + CA_C2 + c * TD3_H + 4 * lane);
        floatx4 dv;
#pragma unroll
        for (int j = 0; j < 4; j++) dv[j] = h[j] > 0.f ? d * wv[j] : 0.f;
        *reinterpret_cast<floatx4 *>(al + CA_DH2 + c * TD3_H + 4 * lane) = dv;
        if (b0 + wq < B) *reinterpret_cast<floatx4 *>(A.dh2 + (size_t)(b0 + wq) * 2 * TD3_H + c * TD3_H + 4 * lane) = dv;
    }
    TEAM_SYNC();
    // ---- dh1_c = (dh2_c W2_c) where c1_c was active: half a team per critic, 64 columns per wave ----
    {
        QPHASE();
        const rsrc_t rw = mkrs(half ? A.c_w5 : A.c_w2, (size_t)TD3_H * TD3_H * 4);
#pragma unroll 1
        for (int t = 0; t < 2; t++) {
            const int j0 = 64 * wq + 32 * t, oc = half * TD3_H + j0;
            floatx4 acc = quad_nn<false, 4>(act + CA_DH2 + half * TD3_H, CA_LD, rw, TD3_H, j0, TD3_H, q);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = act[i * CA_LD + CA_C1 + oc + col] > 0.f ? acc[i] : 0.f;
            quad_store(nullptr, 0, A.dh1, 2 * TD3_H, oc, acc, q);
        }
    }
}

#ifdef TEAM_STAMPS            // (stamps exist in the critic kernel only)
#undef TEAM_SYNC
#define TEAM_SYNC() TEAM_LDS_BARRIER()
#endif
// LDS activations of a policy block, per row: [s | a] | p1 | p2 | g1 | dg2 | dg1 | dz (18 + 2) | dp2
#define PA_SA 0
#define PA_P1 (PA_SA + TD3_SA)
#define PA_P2 (PA_P1 + TD3_H)
#define PA_G1 (PA_P2 + TD3_H)
#define PA_DG2 (PA_G1 + TD3_H)
#define PA_DG1 (PA_DG2 + TD3_H)
#define PA_DZ (PA_DG1 + TD3_H)
#define PA_DP2 (PA_DZ + 20)
#define PA_LD 1616
static_assert(PA_DP2 + TD3_H <= PA_LD && PA_LD % 64 == 16, "policy block layout");

// ---- td3.py:334-341 for 4 batch rows per workgroup (arguments and outputs as k_policy_rows / plentd3_policy_rows; g1, dg2, dg1 are not written:
//      they never leave LDS): 32 columns per wave ----
__global__ __launch_bounds__(64 * TEAM_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_policy_team(PlenTd3PolicyRows A) {
    __shared__ float team_lds[TEAM_NW][TEAM_LDS_PER_WAVE];
    __shared__ __attribute__((aligned(16))) float act[QB * PA_LD];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int B = A.B, b0 = blockIdx.x * QB;
    float *lds = team_lds[w];
    const int n0 = 32 * w;
    // masked by the forward activation (> 0, at LDS offset hoff of each row) and stored (LDS offset loff, or none: -1) / to global Y
    auto store_masked = [&](floatx4 acc, int hoff, int loff, float *Y, const Quad &q) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = act[i * PA_LD + hoff + n0 + q.col] > 0.f ? acc[i] : 0.f;
        quad_store(loff >= 0 ? act + loff : nullptr, PA_LD, Y, TD3_H, n0, acc, q);
    };
    // the block's states: the state columns of sa_pi (left there by the critic pass: an earlier launch); rows past the batch repeat the last one
    if (w < QB) {
        QPHASE();
        if (lane < TD3_S) act[w * PA_LD + PA_SA + lane] = A.sa_pi[(size_t)min(b0 + w, B - 1) * TD3_SA + lane];
    }
    TEAM_SYNC();
    // actor forward
    {
        QPHASE();
        quad_dense_relu<true, 1>(act + PA_SA, PA_LD, TD3_S, mkrs(A.a_w1, (size_t)TD3_H * TD3_S * 4), TD3_S, A.a_b1, n0, act + PA_P1, PA_LD, A.p1, TD3_H, q, lds, A.tp_a_w1);
    }
    TEAM_SYNC();
    {
        QPHASE();
        quad_dense_relu<false, 4>(act + PA_P1, PA_LD, TD3_H, mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, A.a_b2, n0, act + PA_P2, PA_LD, A.p2, TD3_H, q, lds, A.tp_a_w2);
    }
    TEAM_SYNC();
    if (w == 0) {
        QPHASE();
        const float bv = col < TD3_A ? A.a_b3[col] : 0.f;
        const floatx4 z = quad_nt<false, 4>(act + PA_P2, PA_LD, mkrs(A.a_w3, (size_t)TD3_A * TD3_H * 4), TD3_H, 0, TD3_H, q, lds, A.tp_a_w3);
        if (lane < TD3_A) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float a = A.max_a * tanhf(z[i] + bv);                                   // td3.py:57
                act[i * PA_LD + PA_SA + TD3_S + col] = a;
                if (b0 + i < B) {
                    A.a_pi[(size_t)(b0 + i) * TD3_A + col] = a;
                    A.sa_pi[(size_t)(b0 + i) * TD3_SA + TD3_S + col] = a;
                }
            }
        }
    }
    TEAM_SYNC();
    // critic.Q1 forward (fc1 = the first 256 rows of W14) and the gradient of -mean Q1 at its second hidden layer: dg2 = -(1/B) w3 (g2 > 0)
    {
        QPHASE();
        quad_dense_relu<true, 1>(act + PA_SA, PA_LD, TD3_SA, mkrs(A.c_w1, (size_t)TD3_H * TD3_SA * 4), TD3_SA, A.c_b1, n0, act + PA_G1, PA_LD, nullptr, 0, q, lds, A.tp_c_w14);
    }
    TEAM_SYNC();
    {
        QPHASE();
        const float bv = A.c_b2[n0 + col], wv = (-1.f / (float)B) * A.c_w3[n0 + col];
        floatx4 acc = quad_nt<false, 4>(act + PA_G1, PA_LD, mkrs(A.c_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, n0, TD3_H, q, lds, A.tp_c_w2);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = acc[i] + bv > 0.f ? wv : 0.f;
        quad_store(act + PA_DG2, PA_LD, nullptr, 0, n0, acc, q);
    }
    TEAM_SYNC();
    // dg1 = (dg2 W2)(g1 > 0)
    {
        QPHASE();
        store_masked(quad_nn<false, 4>(act + PA_DG2, PA_LD, mkrs(A.c_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, n0, TD3_H, q), PA_G1, PA_DG1, nullptr, q);
    }
    TEAM_SYNC();
    // d/d action = (dg1 W1)[:, 26:44], through the tanh: dz = that * (max_a - a^2 / max_a): one tile, wave 0
    if (w == 0) {
        QPHASE();
        const floatx4 z = quad_nn<false, 4>(act + PA_DG1, PA_LD, mkrs(A.c_w1, (size_t)TD3_H * TD3_SA * 4), TD3_SA, TD3_S, TD3_H, q);
        if (lane < TD3_A) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float a = act[i * PA_LD + PA_SA + TD3_S + col];
                const float dz = z[i] * (A.max_a - a * a / A.max_a);
                act[i * PA_LD + PA_DZ + col] = dz;
                if (b0 + i < B) A.dz[(size_t)(b0 + i) * TD3_A + col] = dz;
            }
        }
    }
    TEAM_SYNC();
    // back through the actor: dp2 = (dz W3)(p2 > 0), dp1 = (dp2 W2)(p1 > 0)
    {
        QPHASE();
        store_masked(quad_nn<true, 1>(act + PA_DZ, PA_LD, mkrs(A.a_w3, (size_t)TD3_A * TD3_H * 4), TD3_H, n0, TD3_A, q), PA_P2, PA_DP2, A.dp2, q);
    }
    TEAM_SYNC();
    {
        QPHASE();
        store_masked(quad_nn<false, 4>(act + PA_DP2, PA_LD, mkrs(A.a_w2, (size_t)TD3_H * TD3_H * 4), TD3_H, n0, TD3_H, q), PA_P1, -1, A.dp1, q);
    }
    // the last workgroup to finish advances the actor optimiser's step counter for the weight-gradient launch that follows (as k_critic_team)
    if (A.adam_step && threadIdx.x == 0) {
        if (atomicAdd(A.done_count, 1) == (int)((B + QB - 1) / QB) - 1) { A.done_count[0] = 0; A.adam_step[0] += 1.f; }
    }
}
