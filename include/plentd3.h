/*
 * plentd3.h -- C ABI of libplentd3.so: the hand-written HIP kernels (gfx950) for the non-GEMM work of one TD3 iteration.
 *
 * The reference's learner is Python/torch (plen_ros/src/plen_ros_helpers/td3.py); it has no FFI of its own.  These entry points are
 * what plen_ml_walk_amd/td3_fused.py binds with ctypes; each cites the reference lines whose arithmetic it carries out.  The dense
 * layers between them stay library GEMMs (torch.mm / addmm -> rocBLAS / hipBLASLt).
 *
 * Conventions: every pointer is a caller-owned DEVICE pointer to float32 (int64 for idx); work is enqueued on the hipStream_t passed
 * as `stream` (void* so the header needs no HIP) and is capturable in a hipGraph; return 0 or a negative HIP error code.
 * Row layouts: packed replay / batch row = s[26] | a[18] | s2[26] | r | not_done (TD3_ROW = 72 floats, td3.py:136-147 tuple order
 * state, action, next_state, reward, with not_done = 1 - done as td3.py:187-191 hands it to train); state-action matrix = s | a (44).
 */
#ifndef PLENTD3_H
#define PLENTD3_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Random numbers: functions that draw take EITHER an explicit array (noise / u: golden tests feed the reference's draws) OR, with that pointer
 * NULL, `rng` = device uint64[2] {seed, number of calls so far}: counter-based Philox4x32-10, so a captured graph needs no library RNG call.
 * The call counter is bumped by a later kernel of the same launch sequence (`rng_bump` arguments, NULL = leave it). */
#define TD3_S 26
#define TD3_A 18
#define TD3_SA 44
#define TD3_ROW 72

/* td3.py:166-193 ReplayBuffer.sample: out[b] = data[idx[b]]; also sa_pi[b][0:26] = state (policy pass input), loss[0:2] = 0 */
int plentd3_gather(const float *data, const int64_t *idx, float *out, float *sa_pi, float *loss, int B, void *stream);
/* td3.py:175 sampling on the device + the gather above: u in [0,1) -> a uniformly drawn COMPLETE row of the ring.  *total = transitions written so
 * far (the ring wraps when it exceeds capacity); guard = rows from position *total on that concurrent writers may be filling (0 for a synchronous
 * loop), excluded once the ring has wrapped onto them; idx_out may be NULL */
int plentd3_sample_gather(const float *data, const float *u, const uint64_t *rng, const int64_t *total, int64_t capacity, int64_t guard, int64_t *idx_out, float *out, float *sa_pi, float *loss, int B, void *stream);
/* plen_td3.py:101-104 exploration: a = clamp(max_a tanh(pre) + noise sigma, +-max_a) over n = B*18 elements */
int plentd3_explore(const float *pre, const float *noise, const uint64_t *rng, float *a, float sigma, float max_a, int n, void *stream);
/* plen_td3.py:91-92 warm-up actions: a = U[-1, 1) over n elements */
int plentd3_uniform_actions(const uint64_t *rng, float *a, int n, void *stream);
/* plen_td3.py:109-113 replay_buffer.add for a whole vector step: ring rows (*total + e) % capacity = s | a | s2 | r | 1 - done_bool,
 * done_bool = terminal and not time-limit (done = PLENVEC_DONE_* bits of plenvec_step); real arrays are float32.  ep_ret [n][2] / stats [3]
 * (both or neither NULL): per-env running return and length; finished episodes are added to stats = {sum of returns, episodes, sum of lengths} (float64: exact counts to 2^53)
 * (the reference prints each episode's return and a moving average, plen_env.py:616-636) */
int plentd3_store(float *data, const int64_t *total, int64_t capacity, const float *s, const float *a, const float *s2, const float *r, const uint8_t *done, uint64_t *rng_bump,
                  float *ep_ret, double *stats, int n, void *stream);
/* plentd3_store, and then s = next_state (n x 26 floats, a different buffer: the env's current observation, i.e. the reset observation where the episode ended --
 * plen_td3.py:115 `state = next_state` / :133 `state = env.reset()`) in the same launch */
int plentd3_store_advance(float *data, const int64_t *total, int64_t capacity, float *s, const float *a, const float *s2, const float *r, const uint8_t *done, uint64_t *rng_bump,
                          float *ep_ret, double *stats, int n, const float *next_state, void *stream);
/* plentd3_store_advance (next_state may be NULL: plain plentd3_store), and then *total += step in the same launch (the caller's ring position for its next vector step:
 * train_vec.py's collectors advance by the rows of ALL collectors).  blocks_done: one zero-initialised unsigned per caller, left at zero by every launch. */
int plentd3_store_step(float *data, int64_t *total, int64_t capacity, float *s, const float *a, const float *s2, const float *r, const uint8_t *done, uint64_t *rng_bump,
                       float *ep_ret, double *stats, int n, const float *next_state, int64_t step, unsigned *blocks_done, void *stream);
/* td3.py:299-304: sa2 = [s2 | clamp(max_a tanh(pre) + clamp(noise sigma, +-clip), +-max_a)], pre = actor_target's last pre-activation */
int plentd3_target_action(const float *pre, const float *noise, const uint64_t *rng, const float *batch, float *sa2, float sigma, float clip, float max_a, int B, void *stream);
/* twin last layers on h2 = [h2_a | h2_b] ([B][512]).  mode 0, td3.py:306-309: y = r + not_done gamma min(q_a, q_b).
 * mode 1, td3.py:312-319: dq[b][c] = 2 (q_c - y) / B, loss[0] += sum (q_c - y)^2 / B, db3_c += sum_b dq[b][c] */
int plentd3_q_heads(const float *h2, const float *w3a, const float *b3a, const float *w3b, const float *b3b, const float *batch, float *y, float *dq,
                    float *loss, float *db3a, float *db3b, uint64_t *rng_bump, float gamma, int B, int mode, void *stream);
/* back through last layer + ReLU: dh2 = dq w3 (h2 > 0); dq NULL = policy pass (td3.py:337), dq = -1/B, critic a only */
int plentd3_dh2(const float *dq, const float *w3a, const float *w3b, const float *h2, float *dh2, int B, int ncrit, int h2_stride, void *stream);
/* ReLU backward in place: g *= (h > 0) */
int plentd3_relu_mask(float *g, const float *h, int B, int n, int h_stride, void *stream);
/* out[j] += sum_b w[b] g[b][j] (w NULL: bias gradient; w = dq column, g = h2_c: the last layer's weight gradient); out zeroed by the caller */
int plentd3_colsum(const float *g, int g_stride, const float *w, int w_stride, float *out, int B, int n, int single_wave, void *stream);
/* nn.Linear backward (td3.py:323, :341): dW[n][k] += sum_b dH[b][n] X[b][k] and (db non-NULL) db[n] += sum_b dH[b][n]; split over the batch on
 * the matrix cores (v_mfma_f32_32x32x2_f32), partial tiles added with float atomics: dW / db must be zeroed by the caller.  Strides in floats.
 * single_wave (here and in plentd3_colsum): 64-thread workgroups without LDS instead of 256-thread ones -- each can start in any single free wave
 * slot, which matters when the update runs beside env launches that hold every slot of the chip (same sums, other summation order). */
int plentd3_wgrad(const float *dH, int dh_stride, const float *X, int x_stride, float *dW, int dw_stride, float *db, int B, int N, int K, int single_wave, void *stream);
/* td3.py:57: a = max_a tanh(pre), also written into sa_pi[:, 26:44] */
int plentd3_tanh_out(const float *pre, float *a, float *sa_pi, float max_a, int B, void *stream);
/* its backward: dz = dsa[:, 26:44] (max_a - a^2 / max_a) */
int plentd3_dtanh(const float *dsa, const float *a, float *dz, float max_a, int B, void *stream);
/* h = relu(h + bias) in place */
int plentd3_bias_relu(float *h, const float *bias, int B, int n, void *stream);
/* td3.py:348-356: target = tau param + (1 - tau) target over a flat parameter buffer */
int plentd3_polyak(float *target, const float *param, float tau, int n, void *stream);
/* td3.py:236-247 / :326-331 / :343-345 optimizer.step() of torch.optim.Adam (no weight decay, no amsgrad) over a network's flat parameter, gradient and
 * moment buffers of n floats: step[0] += 1 (float32 device scalar, torch's capturable `step`); m, v, p updated with torch's bias corrections.
 * Optional extras of the same pass: zero_grad (g = 0 for the next backward), target != NULL (td3.py:348-356: target = tau p + (1 - tau) target),
 * copy_out != NULL (a copy of the new parameters).  done_count: device int, zero before the first call.  p, g, m, v, target and copy_out must be 16-byte
 * aligned (the kernel moves four floats per access); a misaligned pointer is refused with -hipErrorInvalidValue. */
int plentd3_adam(float *p, float *g, float *m, float *v, float *step, int *done_count, int n, double lr, double beta1, double beta2, float eps, int zero_grad,
                 float *target, float tau, float *copy_out, void *stream);

/* td3.py:277-331 without the weight gradients, as ONE launch of single-wave workgroups (16 batch rows each, dense layers on the matrix cores,
 * csrc/td3_rows.hip): sample + gather (as plentd3_sample_gather, u = NULL), target action (as plentd3_target_action, noise = NULL), twin target
 * critics -> y, twin critics -> q, loss, dq, and back to dh2 = dq w3 (h2 > 0), dh1 = (dh2 W2)(h1 > 0).  Row-major outputs left for the weight-gradient kernels
 * (plentd3_colsum, plentd3_wgrad): batch [B][72], sa_pi [B][44] (state columns), c1 / c2 / dh2 / dh1 [B][512] (critic a | critic b), dq [B][2];
 * loss[0] += critic loss, db3a / db3b += last-layer bias gradients (zeroed by the caller); t0, t1 [B][512] and sa2 [B][44] are scratch;
 * done_count = device int, zero before the first call (the last workgroup to finish bumps rng_bump[1] and resets it).
 * Weights are nn.Linear layouts [out][in]; *_w14 = [fc1.weight; fc4.weight] (512 x 44), *_b14 likewise; every matrix 16-byte aligned. */
typedef struct PlenTd3CriticRows {
    const float *data; const uint64_t *rng; const int64_t *total; int64_t capacity, guard;
    const float *at_w1, *at_b1, *at_w2, *at_b2, *at_w3, *at_b3;
    const float *ct_w14, *ct_b14, *ct_w2, *ct_b2, *ct_w5, *ct_b5, *ct_w3, *ct_b3, *ct_w6, *ct_b6;
    const float *c_w14, *c_b14, *c_w2, *c_b2, *c_w5, *c_b5, *c_w3, *c_b3, *c_w6, *c_b6;
    float *batch, *sa_pi, *t0, *t1, *sa2, *c1, *c2, *dh2, *dh1, *dq;
    float *loss, *db3a, *db3b;
    int *done_count; uint64_t *rng_bump;
    float sigma, clip, max_a, gamma;
    int B;
    /* plentd3_critic_team only (NULL otherwise; plentd3_critic_rows refuses them): the replay rows to use instead of drawing them (int64 [B]; total is
     * then not read) and the raw target-smoothing noise z ~ N(0, 1) to use instead of drawing it ([B][18], scaled and clipped as td3.py:300-301) -- what
     * lets the reference's recorded iterations (indices and noise captured from td3.py) be replayed through these kernels */
    const int64_t *idx; const float *noise;
    /* plentd3_critic_team only (NULL otherwise): the step counter of the optimiser whose step plentd3_wgrad_adam_group will take on this pass's gradients
     * (float32 device scalar, torch's capturable `step`): advanced by one by the last workgroup to finish, so that the weight-gradient launch finds it
     * advanced (PlenTd3AdamFused.step_advanced = 1) and needs no counter protocol of its own */
    float *adam_step;
    /* plentd3_critic_team only (NULL: the row-major matrices above, parked through LDS): the forward products' weight matrices in the team kernels' OPERAND order
     * (plentd3_pack with PlenTd3PackJob.team = 1; kept current by plentd3_wgrad_adam_group: PlenTd3WgradJob.pack / pack_t) -- target actor, target critic, critic */
    const float *tp_at_w1, *tp_at_w2, *tp_at_w3, *tp_ct_w14, *tp_ct_w2, *tp_ct_w5, *tp_c_w14, *tp_c_w2, *tp_c_w5;
} PlenTd3CriticRows;
int plentd3_critic_rows(const PlenTd3CriticRows *args, void *stream);

/* td3.py:334-341 without the weight gradients, as plentd3_critic_rows: actor(s) -> a = max_a tanh(.) -> critic.Q1(s, a) -> the gradient of
 * -mean Q1 back to the actor's first hidden layer.  In: sa_pi [B][44] with the state columns filled (plentd3_critic_rows / plentd3_gather); its
 * action columns are written here.  Out for plentd3_wgrad: p1, p2 [B][256] (actor activations), dz [B][18], dp2, dp1 [B][256]; a_pi [B][18],
 * g1, dg2, dg1 [B][256] are scratch.  c_w1 / c_b1 = critic fc1 (the first 256 rows of W14), c_w2 / c_b2 = fc2, c_w3 = fc3.weight. */
typedef struct PlenTd3PolicyRows {
    const float *a_w1, *a_b1, *a_w2, *a_b2, *a_w3, *a_b3;
    const float *c_w1, *c_b1, *c_w2, *c_b2, *c_w3;
    float *sa_pi, *a_pi, *p1, *p2, *g1, *dg2, *dg1, *dz, *dp2, *dp1;
    float max_a;
    int B;
    /* plentd3_policy_team only (NULL otherwise): as PlenTd3CriticRows.adam_step for the actor's optimiser; done_count = device int, zero before the first call */
    float *adam_step; int *done_count;
    /* plentd3_policy_team only (NULL: row-major, parked through LDS): as PlenTd3CriticRows.tp_*; tp_c_w14 = the critic's stacked first layers (fc1 = its first 8 tiles) */
    const float *tp_a_w1, *tp_a_w2, *tp_a_w3, *tp_c_w14, *tp_c_w2;
} PlenTd3PolicyRows;
int plentd3_policy_rows(const PlenTd3PolicyRows *args, void *stream);

/* The same two passes for SMALL batches (the reference's recipe: batch 100, one update per env-step -- plen_td3.py:28, :119-120 -- where the latency
 * of one update is what counts): 4 batch rows per 512-thread workgroup (25 workgroups at batch 100) whose 8 waves split every layer's output columns (csrc/td3_team.hip).
 * Arguments, outputs and arithmetic as plentd3_critic_rows / plentd3_policy_rows (the critics' scalar heads sum in a different order), except that
 * loss[0] is STORED (per-workgroup partials summed in order by the last workgroup to finish), not added to: no zeroing, same bits every run. */
int plentd3_critic_team(const PlenTd3CriticRows *args, void *stream);
int plentd3_policy_team(const PlenTd3PolicyRows *args, void *stream);

/* Up to PLENTD3_WGRAD_JOBS weight gradients (each as plentd3_wgrad: dW[n][k] += sum_b dH[b][n] X[b][k], db[n] += sum_b dH[b][n] when db != NULL) over the
 * same B batch rows in one launch, the whole batch as a single reduction chunk (small batches).  tile0 is filled in by the call. */
#define PLENTD3_WGRAD_JOBS 6
typedef struct PlenTd3WgradJob {
    const float *dH; const float *X; float *dW; float *db; int ds, xs, dws, N, K, tile0;
    /* plentd3_wgrad_adam_group only (NULL / 0 otherwise): the team-order packed copies (PlenTd3PackJob.team = 1, pack_ns = ceil(K / 64)) of this job's weight matrix and
     * of its Polyak target -- every element the launch steps is also written into them, so the next pass finds them current without a packing launch */
    float *pack, *pack_t; int pack_ns;
} PlenTd3WgradJob;
typedef struct PlenTd3WgradGroup { PlenTd3WgradJob job[PLENTD3_WGRAD_JOBS]; int n_jobs, B; } PlenTd3WgradGroup;
int plentd3_wgrad_group(const PlenTd3WgradGroup *group, void *stream);

/* plentd3_wgrad_group with the optimiser step of plentd3_adam applied where each gradient element is produced (the batch is one reduction chunk, so
 * every element is owned by one workgroup): dW / db of the jobs must point INTO the flat gradient buffer g [n] (dws == K); the element's offset
 * there addresses p, m, v and target (NULL: no Polyak update).  The gradients themselves are not stored: g stays zero.  extra_off: offsets of
 * elements whose gradient is already in g (the critics' head biases): stepped and zeroed here.  step / done_count as plentd3_adam, unless step_advanced. */
#define PLENTD3_ADAM_EXTRAS 4
typedef struct PlenTd3AdamFused {
    float *p, *g, *m, *v, *step, *target; int *done_count;
    double lr, beta1, beta2; float eps, tau;
    int n, n_extra, extra_off[PLENTD3_ADAM_EXTRAS];
    int step_advanced;      /* 1: step[0] already holds this step's count (advanced by the pass kernel: adam_step); done_count is then unused */
} PlenTd3AdamFused;
int plentd3_wgrad_adam_group(const PlenTd3WgradGroup *group, const PlenTd3AdamFused *adam, void *stream);

/* The same two passes for LARGE batches (BASELINE.json configs[2]: batch 4096): 16 batch rows per workgroup, one workgroup per compute unit at batch 4096;
 * its waves (eight in the critic pass and the actor forward: two per SIMD inside 2 x 64 registers; four in the policy pass) split every layer's output features, the products are formed transposed (Y^T = W X^T) so that activations stay in LDS from
 * the gathered replay rows to the last gradient, and the weights are read PRE-PACKED in matrix-core operand order (csrc/td3_block.hip).
 * plentd3_pack writes that order: job j packs the N x K matrix M (element (i, k) at src[i rs + k cs]: rs / cs express W or W^T or a column block of it)
 * into dst, zero-padded to 16-row tiles and 16-k steps: dst float4 ((t KS + s) 64 + lane) = M[16 t + lane % 16][16 s + 4 (lane / 16) + (0..3)],
 * KS = ceil(K / 16); dst holds ceil(N / 16) KS 256 floats.  f4_0 is filled in by the call.  Run it after every optimiser / Polyak step that changes a
 * source matrix (3 us for all of a critic's and an actor's matrices). */
#define PLENTD3_PACK_JOBS 16
typedef struct PlenTd3PackJob {
    const float *src; float *dst; int rs, cs, N, K, f4_0;
    /* team = 1: the small-batch kernels' operand order (csrc/td3_team.hip: 32-column tiles, stages of 64 k, lane = 32 (k half) + column):
     *   dst float4 (((t NS + s) 8 + c) 64 + lane) = M[32 t + lane % 32][64 s + 32 (lane / 32) + 4 c + (0..3)],  NS = ceil(K / 64); dst holds ceil(N / 32) NS 2048 floats */
    int team;
} PlenTd3PackJob;
typedef struct PlenTd3PackGroup { PlenTd3PackJob job[PLENTD3_PACK_JOBS]; int n_jobs; } PlenTd3PackGroup;
int plentd3_pack(const PlenTd3PackGroup *group, void *stream);
/* rows: as plentd3_critic_team (idx / noise / adam_step honoured; loss[0] STORED; t0, t1, sa2 not written).  Packed operands: p_at_w1 (256 x 26), p_at_w2
 * (256 x 256), p_at_w3 (18 x 256) of the target actor; p_ct_w14 (512 x 44), p_ct_w2, p_ct_w5 of the target critic; p_c_w14, p_c_w2, p_c_w5 of the critic and
 * p_c_w2t, p_c_w5t = its second layers TRANSPOSED (element (i, k) = W[k][i]) for the input gradients.  partials: scratch float [4 ceil(B / 16)]. */
typedef struct PlenTd3CriticBlock {
    PlenTd3CriticRows rows;
    const float *p_at_w1, *p_at_w2, *p_at_w3, *p_ct_w14, *p_ct_w2, *p_ct_w5, *p_c_w14, *p_c_w2, *p_c_w5, *p_c_w2t, *p_c_w5t;
    float *partials;
} PlenTd3CriticBlock;
int plentd3_critic_block(const PlenTd3CriticBlock *args, void *stream);
/* rows: as plentd3_policy_team.  Packed operands: p_a_w1 (256 x 26), p_a_w2, p_a_w3 (18 x 256) of the actor; p_c_w14, p_c_w2 of the critic (Q1 = its first 16
 * tiles of W14 and fc2); transposed for the input gradients: p_c_w2t, p_c_w1ta = (i = action j, k = hidden) -> fc1.weight[k][26 + j] (18 x 256),
 * p_a_w3t = (i = hidden, k = action) -> fc3.weight[k][i] (256 x 18), p_a_w2t. */
typedef struct PlenTd3PolicyBlock {
    PlenTd3PolicyRows rows;
    const float *p_a_w1, *p_a_w2, *p_a_w3, *p_c_w14, *p_c_w2, *p_c_w2t, *p_c_w1ta, *p_a_w3t, *p_a_w2t;
} PlenTd3PolicyBlock;
int plentd3_policy_block(const PlenTd3PolicyBlock *args, void *stream);

/* Every weight gradient of a LARGE-batch pass in one launch, reduced deterministically in two stages (csrc/td3_block.hip: k_wgrad_big): the batch is cut into
 * `chunks` of rows_per_chunk rows; workgroup (job, 32 x 64 tile, chunk) STORES its partial dW[n][k] = sum_b dH[b][n] X[b][k] (and db[n] = sum_b dH[b][n] when boff >= 0)
 * at partial[chunk stride + goff + n K + k] (boff + n): goff / boff = the element offsets of dW / db in the network's flat gradient buffer, so partial[chunk] has
 * the bucket's layout (stride >= its length, a multiple of 4).  kind 1: a head row (N = 1, K <= 256, a multiple of 4; dH strided, X 16-byte aligned rows).
 * plentd3_adam_big then takes plentd3_adam's step on g[i] = bucket[i] + sum_chunks partial[chunk][i] (added in chunk order: same bits every run) and leaves the
 * bucket zero; reduce_only != 0: bucket[i] = that sum and no step (several ranks: all-reduce the bucket, then plentd3_adam).  partial must be zero wherever no
 * job writes (the head biases).  wg0 is filled in by the call. */
#define PLENTD3_WGRAD_BIG_JOBS 8
typedef struct PlenTd3WgradBigJob { const float *dH; const float *X; int ds, xs, N, K, goff, boff, kind, wg0; } PlenTd3WgradBigJob;
typedef struct PlenTd3WgradBig { PlenTd3WgradBigJob job[PLENTD3_WGRAD_BIG_JOBS]; int n_jobs, B, chunks, rows_per_chunk, stride; float *partial; } PlenTd3WgradBig;
int plentd3_wgrad_big(const PlenTd3WgradBig *group, void *stream);
int plentd3_adam_big(float *p, float *g, float *m, float *v, float *step, int *done_count, int n, double lr, double beta1, double beta2, float eps,
                     float *target, float tau, float *copy_out, const float *partial, int chunks, int stride, int reduce_only, void *stream);

/* plen_td3.py:101-104 for a whole vector step as one launch: action [B][18] = clamp(actor(state [B][26]) + N(0, sigma), +-max_a), the noise drawn as
 * plentd3_explore draws it (rng, bumped by the plentd3_store that follows); p1, p2 [B][256] are scratch. */
typedef struct PlenTd3ActorRows {
    const float *a_w1, *a_b1, *a_w2, *a_b2, *a_w3, *a_b3;
    const float *state; const uint64_t *rng;
    float *p1, *p2, *action;
    float sigma, max_a;
    int B;
} PlenTd3ActorRows;
int plentd3_actor_rows(const PlenTd3ActorRows *args, void *stream);

/* plentd3_actor_rows in the shape of the large-batch passes (16 envs per 256-thread workgroup, packed weights p_a_w1 (256 x 26), p_a_w2 (256 x 256), p_a_w3
 * (18 x 256) of the acting network, plentd3_pack): same draws, same arithmetic up to summation order; rows.p1 / rows.p2 are not written (may be NULL). */
typedef struct PlenTd3ActorBlock { PlenTd3ActorRows rows; const float *p_a_w1, *p_a_w2, *p_a_w3; } PlenTd3ActorBlock;
int plentd3_actor_block(const PlenTd3ActorBlock *args, void *stream);
/* development (scripts/gpu_clock_probe.py): `workgroups` single-wave workgroups that issue 32 iters matrix-core instructions each and nothing else;
 * sink: float [64 workgroups] or NULL */
int plentd3_dev_mfma_spin(int workgroups, int iters, float *sink, void *stream);
/* development: table[(*counter / div) % ring][idx] = the device's constant-rate clock (wall_clock64, 100 MHz) at this point of the stream (table is
 * [ring][nslots] uint64; counter NULL = row 0).  A graph node like the rest: successive replays fill successive rows, no profiler in the way */
int plentd3_stamp(uint64_t *table, const int64_t *counter, int64_t div, int ring, int nslots, int idx, void *stream);
const char *plentd3_version(void);

#ifdef __cplusplus
}
#endif
#endif
