/*
 * tetris_piclim.h -- C ABI of the MI355X-native batched Tetris-piclim environment.
 *
 * Drop-in boundary for the reference's hot path.  The reference has no FFI today: callers use the Python
 * object `Tetris` (game/tetris.py:140-214 of the upstream repo).  Each entry point below names the
 * reference interface it replaces; INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative tpl_status; tpl_last_error() gives the message of the
 *     calling thread's last failure.  No C++ exception crosses this boundary.
 *   - all data pointers are DEVICE pointers on the handle's GPU unless a parameter says "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only enqueue work: no
 *     host synchronisation happens inside tpl_step / tpl_move / tpl_reset / tpl_get_state / tpl_expand_obs,
 *     so they can be captured into a hipGraph.
 *   - one handle per GPU; a handle is not re-entrant.
 *
 * Interchange layout (what callers see; the resident HBM layout is private)
 *   board   uint16_t rows[20]  row 0 = top, bit x = column x        <- Tetris.board, 20x10 bool (:186)
 *   pieces  uint8_t  [M+1]     I0 L1 J2 T3 S4 Z5 O6, [0] falls first <- Tetris.pieces (:187), ids (:8-16)
 *   state   0 running / 1 won / 2 lost                              <- Tetris.state None/True/False (:151)
 *   action  rot*10 + loc, rot 0..3, loc 0..9                         <- move(rotations, location) (:354)
 */
#ifndef TETRIS_PICLIM_H
#define TETRIS_PICLIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPL_ROWS 20
#define TPL_COLS 10
#define TPL_OBS_DIM 217            /* Model(217, 14), model/train.py:26 */
#define TPL_NUM_ACTIONS 40

typedef struct tpl_env tpl_env;    /* opaque handle */

typedef enum {
    TPL_OK = 0,
    TPL_ERR_ARG = -1,              /* bad argument (null pointer, out-of-range L/M/n, ...) */
    TPL_ERR_HIP = -2,              /* a HIP runtime call failed; message carries hipGetErrorString */
    TPL_ERR_STATE = -3,            /* call order violated (e.g. reset before any configs were loaded) */
    TPL_ERR_NOMEM = -4
} tpl_status;

typedef enum { TPL_U8 = 0, TPL_I32 = 1, TPL_I64 = 2 } tpl_int_dtype;      /* dtype of action / rot / loc arrays */
typedef enum { TPL_F32 = 0, TPL_BF16 = 1 } tpl_obs_dtype;
typedef enum { TPL_ASSIGN_HASH = 0, TPL_ASSIGN_SEQUENTIAL = 1 } tpl_assign_mode;

/* Message of the last failure on this thread ("" if none). */
const char* tpl_last_error(void);

/* Library version string and the offload architecture it was built for ("gfx950"). */
const char* tpl_version(void);

/* Bytes of device memory a handle needs for `num_envs` boards with move limit M. */
size_t tpl_workspace_bytes(int64_t num_envs, int32_t M);
/* Bytes of device memory a pool of `n_cfg` prescribed configurations needs (a packed record of 64 / 128 / 256 bytes by M
 * and a 64-byte side record -- the same board unpacked, for the multi-step kernel -- per configuration). */
size_t tpl_pool_bytes(int64_t n_cfg, int32_t M);

/* Replaces Tetris.__init__(L, M, ...) (game/tetris.py:141-214) for `num_envs` boards on GPU `device_id`.
 * `global_offset` is the global index of this handle's board 0 (batch-index sharding over GPUs); `seed`
 * keys the configuration assignment.  `workspace` is caller-owned device memory of at least
 * tpl_workspace_bytes() (256-byte aligned), or NULL to let the library hipMalloc its own.
 * Limits: 1 <= L <= 250, 1 <= M <= 254.
 *
 * Which configuration an episode starts from: the pool entry is a function of (seed, global board index, the step
 * at which the episode begins) -- TPL_ASSIGN_HASH: a hash of the three; TPL_ASSIGN_SEQUENTIAL: (global index + step)
 * mod pool size, so boards that start together take adjacent entries.  Steps are counted on the device since the last
 * full reset (64 bits), so a board's sequence of configurations never repeats and does not depend on how the batch is
 * sharded over GPUs.  The whole resident state, step counters included, lives in the workspace. */
int tpl_create(tpl_env** out, int64_t num_envs, int32_t L, int32_t M, int32_t device_id,
               int64_t global_offset, uint64_t seed, void* workspace, size_t workspace_bytes);

/* Replaces Tetris.terminate() (game/tetris.py:451-470).  Frees only what the library allocated. */
int tpl_destroy(tpl_env* env);

/* auto_reset: a board that finishes during a step is re-initialised from the pool in that same step
 * (done is still reported 1 for that step).  Off: finished boards are frozen (reward 0, done 1).
 * reward = per_line * rows_cleared (+ win when the move wins) (+ lose when the move loses).
 * Changing assign_mode once a pool is loaded leaves the running boards without a way back to their piece lists: the
 * next call that advances boards fails with TPL_ERR_STATE until tpl_reset(env, NULL, stream) has run. */
int tpl_set_options(tpl_env* env, int32_t auto_reset, int32_t assign_mode,
                    float reward_per_line, float reward_win, float reward_lose);

/* Replaces the warm-reset supply (queue of (board, pieces) fed by two producers, game/tetris.py:195-211,445-449,
 * 473-488): uploads a pool of prescribed configurations that resets draw from.  rows: [n_cfg][20] uint16, pieces:
 * [n_cfg][M+1] uint8, both device pointers.  `pool_mem` is caller-owned device memory of tpl_pool_bytes() or NULL.
 *
 * The handle keeps TWO pool buffers so that the supply can be refreshed while boards run: the first call fills the
 * current buffer; every later call fills the other one (on `stream`, e.g. a side stream that a generator feeds) and
 * makes it current -- episodes that begin afterwards draw from it, boards that are mid-episode finish on the buffer
 * their configuration lives in (a board carries one bit for it).  A caller-owned `pool_mem` must therefore stay
 * alive until the call after next.  The buffer a call would overwrite must be out of use: the call fails with
 * TPL_ERR_STATE unless a full reset, or at least M + 1 steps enqueued through this API, have followed the previous
 * swap (tpl_pool_info tells).  The guard counts steps as they are ENQUEUED: when `stream` is not the stream the steps
 * run on, make it wait (an event) for the steps enqueued so far before this call, and make the stepping stream wait
 * for this call's work before the next step -- pool.py's PoolRefresher does exactly that.  Work that was captured
 * into a hipGraph keeps the pool pointers it was captured with: capture again after a swap. */
int tpl_load_configs(tpl_env* env, const uint16_t* rows, const uint8_t* pieces, int64_t n_cfg,
                     void* pool_mem, size_t pool_bytes, void* stream);

/* Which of the two pool buffers is current (0 / 1), the sizes of both, and how many more steps must be enqueued
 * before tpl_load_configs may overwrite the other one (0 = it may).  Any output may be NULL.  Host-side. */
int tpl_pool_info(tpl_env* env, int32_t* current_slot, int64_t* n_cfg_current, int64_t* n_cfg_other,
                  int64_t* steps_until_swap);
/* For a caller that puts a saved copy of the workspace back (the resident state lives in caller-owned memory): the
 * boards are then as old as they were when the copy was taken, so the swap guard goes back to the value
 * tpl_pool_info reported at that moment.  0 <= steps_until_swap <= M + 1. */
int tpl_pool_set_hold(tpl_env* env, int64_t steps_until_swap);
/* For a caller that REPLAYS a captured hipGraph of step launches: the guard above counts steps as they pass through
 * this API, which a replay does not; tell it how many steps the replay just enqueued.  Host-side. */
int tpl_note_steps(tpl_env* env, int64_t steps);

/* A stream for the producers of the supply to run on beside the stepping stream -- the reference runs its two
 * producers as separate PROCESSES that compete with the game for nothing but host cores (game/tetris.py:198-211); on the
 * GPU a generator kernel shares the compute units with the step kernel, whose whole grid is resident at once and slows
 * down when it is not.  cu_count > 0: work on this stream runs on `cu_count` compute units only (spread evenly over the
 * XCDs and shader engines), so the generator takes a bounded slice of the chip; cu_count == 0 and low_priority != 0: a
 * stream of the lowest priority; both 0: a plain non-blocking stream.  The stepping stream is not restricted. */
int tpl_stream_create(int32_t device_id, int32_t cu_count, int32_t low_priority, void** stream);
int tpl_stream_destroy(int32_t device_id, void* stream);

/* Replaces Tetris.reset() (game/tetris.py:438-443).  mask == NULL: the step counters and the statistics are
 * zeroed and every board starts an episode at step 0 (from the current pool buffer); otherwise boards with
 * mask[i] != 0 start an episode at the next step.  Unlike the reference, lines_cleared / moves_used / state ARE
 * zeroed. */
int tpl_reset(tpl_env* env, const uint8_t* mask, void* stream);

/* Replaces Tetris.move(rotations, location) (game/tetris.py:354-422), one move on every board.
 * rot/loc: arrays of `dtype`; rot is taken modulo the piece's rotation count, loc is right-clamped to
 * 10 - width (:364); values must be >= 0.  Outputs (each may be NULL): reward f32[n], done u8[n],
 * cleared u8[n] (rows cleared by this move). */
int tpl_move(tpl_env* env, const void* rot, const void* loc, int32_t dtype,
             float* reward, uint8_t* done, uint8_t* cleared, void* stream);

/* step(action): move(action / 10, action % 10).  The surface BASELINE.json's north_star names. */
int tpl_step(tpl_env* env, const void* action, int32_t dtype, float* reward, uint8_t* done, void* stream);

/* tpl_step and tpl_expand_obs in ONE launch -- north_star's step(action) -> (obs, reward, done) for a host-driven loop that
 * needs the observation at every step: the kernel that makes the move writes the [n][217] observation (obs_dtype TPL_F32 /
 * TPL_BF16; the boards as they stand after the step, a finished board's reset included) from the registers it holds,
 * instead of a second kernel reading the state back.  Same results as the two calls.  `obs` must be 16-byte aligned. */
int tpl_step_observe(tpl_env* env, const void* action, int32_t dtype, float* reward, uint8_t* done, void* obs,
                     int32_t obs_dtype, void* stream);

/* num_steps consecutive steps in one launch, equivalent to num_steps calls of tpl_step with
 * action = actions + k*action_stride (uint8, device).  The loop shape of game/performance_test.py:13-17
 * (move; reset when finished) with the board held in registers between moves.  Outputs, each optional:
 * reward_steps f32[num_steps][n], done_steps u8[num_steps][n], reward_sum f32[n] (sum over the steps in step
 * order), finished u32[n] (episodes the board finished). */
int tpl_rollout(tpl_env* env, const uint8_t* actions, int64_t action_stride, int32_t num_steps,
                float* reward_steps, uint8_t* done_steps, float* reward_sum, uint32_t* finished, void* stream);

/* tpl_rollout with the UNIFORM RANDOM POLICY drawn on the device -- the loop of game/performance_test.py:13-17, whose
 * moves are random.randint draws -- so that no actions need staging: step k plays, on every board, the action
 * tpl_explore_actions(epsilon = 1, seed, step0 + k) would put there (uniform in [0, 40), a hash of seed, global board
 * index and step).  actions_out u8[num_steps][n] (optional) records what was played; the other outputs as tpl_rollout. */
int tpl_rollout_random(tpl_env* env, uint64_t seed, uint32_t step0, int32_t num_steps, uint8_t* actions_out,
                       float* reward_steps, uint8_t* done_steps, float* reward_sum, uint32_t* finished, void* stream);

/* tpl_rollout / tpl_rollout_random with the COMPACT TRAJECTORY as the per-step output -- the loop of
 * game/performance_test.py:13-17 recorded for a learner at one byte per board-step instead of a float and a byte (which
 * cost the fused kernel a quarter of its rate).  trajectory u32[(num_steps + 3) / 4][n] (device): byte j of word w of
 * board i is step 4 w + j:  bits 0-2 rows cleared by the move (game/tetris.py:382-386), bits 3-4 how it ended (0 the game
 * goes on, 1 won :415-417, 2 lost at the move limit :391,419-421, 3 topped out :372-374), bit 5 the board was
 * re-initialised from the pool in this step (auto_reset), bit 6 the board was frozen (finished earlier, no auto_reset: no
 * move made).  Bytes past num_steps in the last word are zero.  Same moves, same statistics, same final state as
 * tpl_rollout.  tpl_decode_trajectory turns a trajectory into reward_steps f32[num_steps][n] / done_steps
 * u8[num_steps][n] (either may be NULL) with the handle's reward parameters -- bit for bit what tpl_rollout writes. */
int tpl_rollout_trajectory(tpl_env* env, const uint8_t* actions, int64_t action_stride, int32_t num_steps,
                           uint32_t* trajectory, uint32_t* finished, void* stream);
int tpl_rollout_random_trajectory(tpl_env* env, uint64_t seed, uint32_t step0, int32_t num_steps, uint8_t* actions_out,
                                  uint32_t* trajectory, uint32_t* finished, void* stream);
int tpl_decode_trajectory(tpl_env* env, const uint32_t* trajectory, int32_t num_steps, float* reward_steps,
                          uint8_t* done_steps, void* stream);

/* Replaces Tetris.get_state() (game/tetris.py:435-436) and the public attributes, batched and in the
 * interchange layout.  Any output may be NULL.  rows [n][20] u16; cur/nxt u8[n] (7 = no such piece);
 * lines/moves u8[n] (lines_cleared, moves_used -- L_rem = L - lines, M_rem = M - moves); state u8[n];
 * pieces_left u8[n] = len(Tetris.pieces). */
int tpl_get_state(tpl_env* env, uint16_t* rows, uint8_t* cur, uint8_t* nxt, uint8_t* lines,
                  uint8_t* moves, uint8_t* state, uint8_t* pieces_left, void* stream);

/* Replaces reading Tetris.board (game/tetris.py:186; the first element of get_state(), :435-436) for every board:
 * cells [n][20][10] uint8, 0 / 1 (the storage of a bool array), row 0 = top.  `cells` must be 16-byte aligned. */
int tpl_get_board(tpl_env* env, uint8_t* cells, void* stream);

/* Observation for Model(217, 14) (model/train.py:26): out [n][217] of `dtype`:
 * 200 cells row-major (y*10+x), one-hot current piece (7), one-hot next piece (7), L_rem, M_rem, terminal.
 * Any `out` is accepted; a 16-byte aligned one is written with 16-byte stores (the fast path). */
int tpl_expand_obs(tpl_env* env, void* out, int32_t dtype, void* stream);

/* Policy head -> action for Model(217, 14) (model/train.py:26; the reference never decodes its 14 outputs):
 * logits [n][14] of `dtype` (TPL_F32 / TPL_BF16); action[i] = argmax(logits[i][0:4]) * 10 + argmax(logits[i][4:14]),
 * lowest index on ties. */
int tpl_decode_actions(tpl_env* env, const void* logits, int32_t dtype, uint8_t* action, void* stream);

/* The same observation for `count` recorded 32-byte states (the states_a / states_b outputs of tpl_actor_rollout,
 * e.g. a minibatch gathered from a replay buffer) instead of the environment's resident boards.  L and M come
 * from `env`.  out: [count][217] of `dtype`. */
int tpl_expand_states(tpl_env* env, const void* states_a, const void* states_b, int64_t count, void* out,
                      int32_t dtype, void* stream);

/* ---- fused policy: observation -> Model(217, 14) -> action in one kernel (model/model.py:9-20, train.py:26) ----
 * tpl_policy_pack (host): the ten parameter arrays of the five Linear layers, float32, torch layout
 * (weight [out][in] row-major, bias [out]; sizes 128x217, 128x128 x3, 14x128) -> an image of
 * tpl_policy_image_bytes() bytes in HOST memory `image`; copy it to the device (16-byte aligned) once.
 * Weights are rounded to bf16 (nearest even), biases stay float32.  W1's columns are in the observation order of
 * tpl_expand_obs.
 * tpl_policy_act: for every board, logits = MLP(observation) with bf16 operands, float32 accumulation and bf16
 * activations between layers, on the matrix cores; action = argmax(logits[0:4])*10 + argmax(logits[4:14]) as
 * tpl_decode_actions.  `logits` ([n][14] float32) is optional.  The observation is never written to memory. */
size_t tpl_policy_image_bytes(void);
int tpl_policy_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                    const float* b3, const float* w4, const float* b4, const float* w5, const float* b5, void* image);
int tpl_policy_act(tpl_env* env, const void* image, uint8_t* action, float* logits, void* stream);
/* The same policy with FLOAT32 operands -- the arithmetic width of the reference's own nn.Linear stack
 * (model/model.py:9-20): weights, activations and accumulation in float32 on the matrix cores (each output is a
 * k-ordered chain of fused multiply-adds), so logits differ from a float32 torch module only by summation order.
 * tpl_policy_pack_f32 (host) takes the same ten arrays and fills an image of tpl_policy_image_bytes_f32() bytes;
 * tpl_policy_act_f32 is tpl_policy_act for such an image.  About 1/8 of the bf16 kernel's rate. */
size_t tpl_policy_image_bytes_f32(void);
int tpl_policy_pack_f32(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                        const float* b3, const float* w4, const float* b4, const float* w5, const float* b5, void* image);
int tpl_policy_act_f32(tpl_env* env, const void* image, uint8_t* action, float* logits, void* stream);
/* The same policy at FLOAT32 ACCURACY on the bf16 matrix pipe (csrc/policy_split.hip): every float32 weight and
 * activation is the sum of three bf16 numbers (8 + 8 + 8 significand bits), a product is six
 * v_mfma_f32_16x16x32_bf16 (three in layer 1, whose inputs are exact in bf16) accumulated in float32 -- 2.7 x less
 * matrix time than eight v_mfma_f32_16x16x4_f32, measured 2.1 x on the kernel.  Within the float32 kernel's tolerance of
 * a float64 evaluation of model/model.py:9-20 (2e-5 (1 + max|logit|)); not bit-identical to a float32 FMA chain.  The
 * image (tpl_policy_pack_split, same arguments as tpl_policy_pack) is tpl_policy_image_bytes_split() bytes. */
size_t tpl_policy_image_bytes_split(void);
int tpl_policy_pack_split(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                          const float* b3, const float* w4, const float* b4, const float* w5, const float* b5, void* image);
int tpl_policy_act_split(tpl_env* env, const void* image, uint8_t* action, float* logits, void* stream);
/* Epsilon-greedy exploration on an action array: with probability epsilon action[i] is replaced by a uniform
 * action in [0, 40), a function of (seed, global board index, step) alone: steps 2j and 2j + 1 of a board share one 32-bit
 * hash word of (seed, index, j) and take sixteen bits of it each, (bits * 40) >> 16 -- every action within 40 / 65536 of 1/40;
 * whether to replace is a hash of that word and the step's parity. */
int tpl_explore_actions(tpl_env* env, uint8_t* action, float epsilon, uint64_t seed, uint32_t step, void* stream);
/* num_steps iterations of (tpl_policy_act, tpl_explore_actions(step0 + t), tpl_step) in ONE launch: the weights
 * stay in LDS and the boards in registers; only the trajectory leaves the chip.  Outputs, each optional:
 * actions u8 / rewards f32 / dones u8 [num_steps][n], and states_a / states_b [num_steps][n] 16-byte words = the
 * resident state of every board BEFORE step t (a 32-byte observation for a replay buffer; layout in DESIGN.md). */
int tpl_actor_rollout(tpl_env* env, const void* image, int32_t num_steps, float epsilon, uint64_t seed, uint32_t step0,
                      uint8_t* actions, float* rewards, uint8_t* dones, void* states_a, void* states_b, void* stream);
/* tpl_actor_rollout for a float32 image (tpl_policy_pack_f32): num_steps iterations of (tpl_policy_act_f32,
 * tpl_explore_actions(step0 + t), tpl_step) in ONE launch -- the reference's nn.Linear arithmetic width
 * (model/model.py:9-20) as a multi-step loop: the boards stay in registers, the float32 weights (twice a CU's LDS) stream
 * through it once per step.  Same outputs, same results as the step-by-step calls. */
int tpl_actor_rollout_f32(tpl_env* env, const void* image, int32_t num_steps, float epsilon, uint64_t seed, uint32_t step0,
                          uint8_t* actions, float* rewards, uint8_t* dones, void* states_a, void* states_b, void* stream);
/* ... and for a split image (tpl_policy_pack_split): float32-grade decisions as a multi-step loop, at twice the float32
 * megakernel's rate.  num_steps iterations of (tpl_policy_act_split, tpl_explore_actions(step0 + t), tpl_step). */
int tpl_actor_rollout_split(tpl_env* env, const void* image, int32_t num_steps, float epsilon, uint64_t seed, uint32_t step0,
                            uint8_t* actions, float* rewards, uint8_t* dones, void* states_a, void* states_b, void* stream);

/* Statistics over episodes finished since the last full reset, reduced on the device into
 * out[4] (device pointer, uint64): {episodes, sum of lines_cleared at finish, wins, top-outs}. */
int tpl_get_stats(tpl_env* env, uint64_t* out, void* stream);

/* Host-side decode of the device shape table: get_tetromino(piece, rotations) (game/tetris.py:60-61).
 * masks[4]: row masks top->bottom (bit x = mask column x), revtopo[4]: reverse topography.  No GPU needed. */
int tpl_shape_info(int32_t piece, int32_t rotations, int32_t* h, int32_t* w, uint8_t* masks, uint8_t* revtopo);

/* Raw device pointers of the resident packed state (for zero-copy inspection; layout in DESIGN.md). */
int tpl_state_ptrs(tpl_env* env, void** plane_a, void** plane_b);
/* The step counters: `count` uint64 values on the device, one per group of 32 boards, each equal to the number of
 * steps since the last full reset. */
int tpl_clock_ptr(tpl_env* env, void** clock, int64_t* count);

/* Tuning knobs of the step kernel: boards handled per lane (1, 2 or 4; default 1 up to 2^19 boards, 2 beyond) and threads
 * per block (64, 128, 256 or 512; default 256).  Results do not depend on them. */
int tpl_set_tuning(tpl_env* env, int32_t boards_per_lane, int32_t block_threads);

/* Replaces the carving generator behind Tetris.reset() (game/tetris.py:226-352 with RandomPieceGenerator :64-108
 * and CheckpointManager :111-137; its worker process :473-479): produces `count` solvable prescribed
 * configurations on `threads` host threads (0 = all cores).  HOST pointers: rows [count][20] uint16,
 * pieces [count][M+1] uint8, and optionally the carved solution [count][M][2] (rotations, location) with
 * solution_len [count] (the reference's debug `solution`, :155-156).
 *
 * Two things here are this library's own definitions, not the reference's (csrc/tpl_device.h states both; the carving
 * logic itself is pinned by the reference's decision tapes): (1) the decision stream -- attempt a of configuration
 * first+i draws decision k as a 32-bit hash of key + k * stride, (key, stride) = the halves of rng(seed, 4, first+i, a),
 * top 24 bits reduced to [lo, hi] by a multiply; (2) the restart rule -- the reference's search loop has no bound and an
 * exponentially distributed length, so configuration first+i is the outcome of the FIRST attempt a = 0, 1, ... 23 that ends
 * within its iteration cut-off: `cutoff` for attempts 0-11, then doubling with every attempt (2 x for attempt 12, 4 x for 13,
 * ... 256 x for 19 and for 20-23; never above 2^28) -- with cutoff = 0, about twice the median search length at this L as
 * measured at M = 40 (an M close to the fewest pieces that can clear L rows searches far longer: the doubling is for that).
 * ABI NOTE (library version 0.2): this argument was `max_iters` through library 0.1 as shipped in round 3 -- ONE search per
 * configuration, 0 = unbounded.  It is now the restart rule's base cut-off, 0 = by L: a caller written against the old meaning
 * gets different configurations (the type is the same, nothing fails); tpl_version() tells the two apart.
 * The output depends on (L, M, seed, first+i, cutoff) only -- not on `threads`, and it is the same on the device.  Before a
 * batch goes out FOUR fixed pilot configurations are tried on the host (the verdict is kept per (L, M, cutoff)): TPL_ERR_ARG when M
 * is below the fewest pieces that can dig two columns of L cells (ceil(L / 2)), TPL_ERR_STATE when the 24 attempts of EVERY pilot run
 * into their cut-offs -- nothing is generated then (one pilot that caps refuses nothing: under a marginal cut-off the batch goes
 * ahead and reports its capped configurations one by one).  If all 24 attempts of a configuration of the batch run into their
 * cut-offs its outputs are zeroed and the call returns TPL_ERR_STATE after finishing the others ("did not finish within the
 * rule's bound", not "cannot be carved": a larger `cutoff` searches on).  1 <= L <= 16. */
int tpl_generate_configs(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count, int32_t threads,
                         int64_t cutoff, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                         int32_t* solution_len);

/* The same generator on the GPU: same decisions, same restart rule, same output as tpl_generate_configs, for
 * refreshing a device pool without the host (whose container may own only a few CPUs).  DEVICE pointers; `status`
 * [count] (optional) is 1 for a configuration whose 24 attempts all ran into their cut-off (outputs zeroed), else 0.
 * The same host-side pilot as tpl_generate_configs runs before the launch (an (L, M) that does not carve would otherwise be
 * a kernel that runs for minutes): TPL_ERR_ARG / TPL_ERR_STATE, nothing launched.
 * `work`: tpl_generate_configs_device_work_bytes(M, count) bytes, 8-byte aligned. */
size_t tpl_generate_configs_device_work_bytes(int32_t M, int64_t count);
/* The kernel is persistent: its lanes take configurations from a queue until none are left (a lane that held one
 * configuration for its whole life would idle while the slowest lane of its wave searches on), and once the queue is dry
 * the lanes without work run FURTHER attempts of the configurations still being searched -- the restart rule makes the
 * answer the lowest attempt that ends inside its cut-off, whichever lane ran it, so the end of a launch is bounded by
 * about two cut-offs instead of by the longest search of the batch.  `waves` of tpl_generate_configs_device_waves says
 * how many 64-lane waves share the queue (at most 4096): 0 = automatic (4096, or one lane per configuration if
 * that is fewer: the fastest for a generator that has the chip to itself); a small number -- 64 to 256 -- bounds the generator's footprint when it runs
 * BESIDE a stepping environment (every generator wave takes one of a SIMD's eight wave slots for milliseconds).  The
 * output does not depend on it.  count < 2^31. */
int tpl_generate_configs_device_waves(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count, int64_t cutoff,
                                      int32_t waves, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                                      int32_t* solution_len, int32_t* status, void* work, size_t work_bytes, void* stream);
int tpl_generate_configs_device(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count, int64_t cutoff,
                                uint16_t* rows, uint8_t* pieces, uint8_t* solution, int32_t* solution_len,
                                int32_t* status, void* work, size_t work_bytes, void* stream);

/* The same generator driven by CPython's `random` stream: configuration i is what the reference produces after
 * `random.seed(seeds[i]); Tetris(L, M, warm_reset=False)` (game/tetris.py:226-284 drawing through :85,93,250,253)
 * -- MT19937 seeded as random.seed(int) seeds it, randint/shuffle on _randbelow_with_getrandbits.  One search per
 * configuration, as in the reference (no restart rule); max_iters > 0 bounds it.  HOST pointers. */
int tpl_generate_configs_pyseed(int32_t L, int32_t M, const uint64_t* seeds, int64_t count, int32_t threads,
                                int64_t max_iters, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                                int32_t* solution_len);

/* Replaces the forward generator + solver behind Tetris.reset() (game/tetris_algo_main/: TetrisGameGenerator.py
 * :15-29,72-106, TetrisSolver.py:112-163, main.py generate_batch :29-74; its worker game/tetris.py:482-488):
 * for each seed, random-fill the board to `initial_height_max`, draw a 7-bag sequence of M pieces, and run the greedy
 * depth-first solver with `max_attempts`.  Game i equals the reference's TetrisGameGenerator(seed=seeds[i], goal=L,
 * tetrominoes=M, initial_height_max) and its TetrisSolver verdict.  HOST pointers; outputs for EVERY seed:
 * rows [count][20], sequence [count][M] (piece ids of Tetris.move, game/tetris.py:8-16), winnable [count],
 * failed_attempts [count] (optional), solution [count][M][2] as (rotations, location) of Tetris.move (optional),
 * solver_stack [count][M][3] as the solver's own (letter index in IJLOSTZ, rotation, column) (optional),
 * solution_len [count] (optional; 0 when not winnable).  The reference defaults are initial_height_max 4,
 * max_attempts 1000, seeds 0..99 (main.py:35-42). */
int tpl_forward_generate(int32_t L, int32_t M, int32_t initial_height_max, int32_t max_attempts, const uint64_t* seeds,
                         int64_t count, int32_t threads, uint16_t* rows, uint8_t* sequence, uint8_t* winnable,
                         int32_t* failed_attempts, uint8_t* solution, uint8_t* solver_stack, int32_t* solution_len);

/* The same generator + solver on the GPU, one game per lane (a lane carries its own CPython-compatible MT19937): the same
 * games, seed for seed, as tpl_forward_generate and as the reference.  DEVICE pointers throughout, `seeds` included; the
 * optional outputs may be null; rows of `solution` / `solver_stack` past solution_len are zero.  `work`:
 * tpl_forward_generate_device_work_bytes(M, count) bytes (the generator states and the solver's frames), 4-byte aligned.
 * count < 2^31.  Exists so that a pool blending both of the reference's producers (game/tetris.py:195-211, 482-488) can be
 * built without the host; not a fast kernel -- the reference runs this supplier over the same hundred seeds per batch
 * (tetris_algo_main/main.py:39-40). */
size_t tpl_forward_generate_device_work_bytes(int32_t M, int64_t count);
int tpl_forward_generate_device(int32_t L, int32_t M, int32_t initial_height_max, int32_t max_attempts, const uint64_t* seeds,
                                int64_t count, uint16_t* rows, uint8_t* sequence, uint8_t* winnable, int32_t* failed_attempts,
                                uint8_t* solution, uint8_t* solver_stack, int32_t* solution_len, void* work, size_t work_bytes,
                                void* stream);

/* Replaces Tetris.carve(piece, rotations, location, allow_partial) (game/tetris.py:286-352), the inverse of a move
 * and the carving generator's building block, on ONE board in HOST memory: rows[20] is modified in place when the
 * carve succeeds; *carved = 1 / 0.  (The reference does not clamp here: a location that puts the piece outside the
 * board is an argument error.) */
int tpl_carve(uint16_t* rows, int32_t piece, int32_t rotations, int32_t location, int32_t allow_partial,
              int32_t* carved);

/* Synthetic workload of SURVEY 8(d), generated on the device from a counter-based hash
 * keyed by (seed, stream, global board index, counter); DESIGN.md states the function. */
int tpl_synth_configs(tpl_env* env, uint64_t seed, int64_t first, int64_t count,
                      uint16_t* rows, uint8_t* pieces, void* stream);
int tpl_synth_actions(tpl_env* env, uint64_t seed, int64_t first, int64_t count, uint64_t step,
                      uint8_t* action, void* stream);

#ifdef __cplusplus
}
#endif
#endif
