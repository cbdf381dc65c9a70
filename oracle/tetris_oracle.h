/*
 * tetris_oracle.h -- CPU ORACLE for the Tetris-piclim board step.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar C restatement of the reference's hot path (game/tetris.py:23-61, 354-449 of the
 * upstream repo).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product (the HIP library behind include/tetris_piclim.h) never links, loads or calls it.
 *
 * Parity pin: tests/golden/ fixtures were produced by importing the reference's own Python module
 * (tests/golden/make_golden.py, run where /root/reference exists) and tests/test_oracle_golden.py checks
 * every function here against them.
 *
 * Interchange layout (shared with include/tetris_piclim.h):
 *   board   uint16_t rows[20]   row 0 = top, bit x = column x (bit 0 = leftmost column)
 *   pieces  uint8_t  [M+1]      ids I0 L1 J2 T3 S4 Z5 O6 (game/tetris.py:8-16); index 0 = first to fall
 *   state   0 running (None), 1 won (True), 2 lost (False)      (game/tetris.py:151,373,391,416,421)
 */
#ifndef TETRIS_ORACLE_H
#define TETRIS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TO_ROWS 20
#define TO_COLS 10
#define TO_FULL_ROW 0x3FFu

enum { TO_RUNNING = 0, TO_WON = 1, TO_LOST = 2 };

/* One game, mirroring the attributes of the reference's `Tetris` object (game/tetris.py:143-151,186-187). */
typedef struct {
    uint16_t rows[TO_ROWS];
    int32_t  lines_cleared;
    int32_t  moves_used;
    int32_t  state;      /* TO_RUNNING / TO_WON / TO_LOST */
    int32_t  cursor;     /* number of pieces popped so far == index of pieces[0] in the original list */
} to_game;

/* Shape table entry: get_tetromino(piece, rotations) (game/tetris.py:60-61). */
typedef struct {
    int32_t h, w;
    uint8_t mask[4];      /* row masks top->bottom, bit x = mask column x */
    uint8_t revtopo[4];   /* reverse_tetromino_topography */
} to_shape;

int  to_num_rotations(int piece);
void to_get_tetromino(int piece, int rotations, to_shape* out);

/* Tetris.move(rotations, location) (game/tetris.py:354-422) on one game.
 * `pieces` is the game's full piece list; the piece consumed is pieces[g->cursor].
 * Returns rows cleared by this move (0..4), or -1 when the move topped out (drop < 0).
 * Like the reference it does NOT look at g->state first (a finished game still mutates). */
int to_move(to_game* g, const uint8_t* pieces, int L, int M, int rotations, int location);

/* ---- build-defined batched environment semantics (same rules as the HIP library) -------------------
 * Finished boards are frozen (step is a no-op, reward 0, done 1) unless auto_reset is on, in which case
 * a board that finishes during a step is re-initialised from the config pool in that same step.       */
typedef struct {
    int64_t n;            /* boards */
    int32_t L, M;
    int64_t global_offset;/* global index of board 0 (multi-GPU sharding) */
    uint64_t seed;
    int32_t auto_reset;
    int32_t assign_mode;  /* 0 = hashed, 1 = sequential */
    float   reward_per_line, reward_win, reward_lose;
    /* two pool buffers: new episodes start from pool[cur_slot]; a pool set later goes into the other slot, which
     * becomes current.  (The oracle copies a configuration's piece list into the board at reset, so a board never
     * looks at its pool again -- which is what the device's slot bit and window refills must amount to.) */
    int64_t n_cfg[2];
    const uint16_t* pool_rows[2];    /* [n_cfg][20] */
    const uint8_t*  pool_pieces[2];  /* [n_cfg][M+1] */
    int32_t cur_slot;
    /* steps since the last full reset (the device keeps one such counter per 32 boards; they are all equal) */
    uint64_t clock;
    /* per-board state */
    to_game* games;       /* [n] */
    uint8_t* pieces;      /* [n][M+1]  current episode's list */
    uint64_t* birth;      /* [n] step at which the board's current episode began */
    /* statistics over finished episodes */
    uint64_t stat_episodes, stat_lines, stat_wins, stat_topouts;
} to_env;

to_env* to_env_create(int64_t n, int L, int M, int64_t global_offset, uint64_t seed);
void    to_env_destroy(to_env* e);
void    to_env_set_pool(to_env* e, const uint16_t* rows, const uint8_t* pieces, int64_t n_cfg); /* borrowed */
void    to_env_set_options(to_env* e, int auto_reset, int assign_mode, float per_line, float win, float lose);
/* pool entry (in the current slot) of the episode of `board` that begins at step `birth` */
int64_t to_env_assign(const to_env* e, int64_t board, uint64_t birth);
uint64_t to_env_clock(const to_env* e);
uint64_t to_env_birth(const to_env* e, int64_t board);
void    to_env_reset(to_env* e, const uint8_t* mask /* NULL = all */);
/* rot/loc are uint8 arrays of length n.  reward/done/cleared may be NULL. */
void    to_env_move(to_env* e, const uint8_t* rot, const uint8_t* loc, float* reward, uint8_t* done, uint8_t* cleared);
void    to_env_step(to_env* e, const uint8_t* action, float* reward, uint8_t* done);
void    to_env_get_state(const to_env* e, uint16_t* rows, uint8_t* cur, uint8_t* nxt,
                         uint8_t* lines, uint8_t* moves, uint8_t* state, uint8_t* pieces_left);
/* float observation [n][217]: 200 cells row-major, one-hot cur (7), one-hot next (7), L_rem, M_rem, terminal */
void    to_env_expand_obs(const to_env* e, float* out);
/* out = {episodes finished, sum of lines_cleared at finish, wins, top-outs} */
void    to_env_get_stats(const to_env* e, uint64_t out[4]);

/* Epsilon-greedy exploration (tpl_explore_actions; the uniform random policy of tpl_rollout_random at eps_q24 = 2^24):
 * action[b] is replaced by the draw of (seed, global_offset + b, step) with probability eps_q24 / 2^24. */
void to_explore_actions(uint8_t* action, int64_t n, int64_t global_offset, uint64_t seed, uint32_t step, uint32_t eps_q24);

/* ---- synthetic inputs: counter-based, identical on CPU and GPU ------------------------- */
uint64_t to_rng(uint64_t seed, uint64_t stream, uint64_t index, uint64_t counter);
void to_synth_boards(uint64_t seed, int64_t first, int64_t count, int L, uint16_t* rows /*[count][20]*/);
void to_synth_pieces(uint64_t seed, int64_t first, int64_t count, int M, uint8_t* pieces /*[count][M+1]*/);
void to_synth_actions(uint64_t seed, int64_t first, int64_t count, uint64_t step, uint8_t* action /*[count]*/);

/* ---- prescribed-configuration supply: the carving generator (game/tetris.py:64-137, 226-352) ---------------
 * Random decisions come through a callback so that the same code can be driven (a) by a tape of the decisions
 * the reference itself made (tests/golden/carving_*.npz -- pins the generator logic bit-exactly) or (b) by the
 * counter-based generator that the product uses.  randint(ctx, lo, hi) returns a value in [lo, hi]. */
typedef int32_t (*to_randint_fn)(void* ctx, int32_t lo, int32_t hi);
typedef void (*to_phase_fn)(void* ctx);

/* Tetris.carve(piece, rotations, location, allow_partial) (game/tetris.py:286-311) on rows[20]; returns 1/0. */
int to_carve(uint16_t* rows, int piece, int rotations, int location, int allow_partial);

/* Tetris._generate_initial_config (game/tetris.py:226-284) for one game.  rows[20], pieces[M+1],
 * solution[M][2] (rotations, location) and *sol_len are outputs.  Returns the number of loop iterations, or -1
 * if max_iters was hit (the reference has no bound).  `search_over` (may be NULL) is called once, when the loop of :234 has
 * ended and before the padding draws. */
int64_t to_generate_config(int L, int M, to_randint_fn randint, void* ctx, to_phase_fn search_over, int64_t max_iters,
                           uint16_t* rows, uint8_t* pieces, uint8_t* solution, int32_t* sol_len);

/* (a) tape-driven: tape = [n][3] int32 (lo, hi, value); returns iterations, -2 on a tape mismatch/underrun;
 * *consumed = decisions used. */
int64_t to_generate_config_tape(int L, int M, const int32_t* tape, int64_t n, int64_t* consumed,
                                uint16_t* rows, uint8_t* pieces, uint8_t* solution, int32_t* sol_len);
/* (b) counter-driven, with the build's restart rule (both stated in tetris_oracle.c above the function): the outcome of the
 * first of 24 attempts whose search ends within its iteration limit (`cutoff` > 0 overrides the per-L base limit).  Returns
 * the iterations spent (failed attempts included), -1 if every attempt ran into its limit (outputs all zero);
 * *attempt_out = the attempt that succeeded (24 when none did). */
int64_t to_carve_attempt_limit(int L, int64_t cutoff, int attempt);
int64_t to_generate_config_seeded(int L, int M, uint64_t seed, uint64_t index, int64_t cutoff,
                                  uint16_t* rows, uint8_t* pieces, uint8_t* solution, int32_t* sol_len, int32_t* attempt_out);

/* FNV-1a over the 20 rows (u16 units) -- per-step fingerprint used by the golden fixtures. */
uint64_t to_board_hash(const uint16_t* rows);

/* cpu_baseline helper: steps boards [first, first+count) of the synthetic workload for `steps` lockstep
 * steps with auto-reset from a synthetic pool of the same size; returns board-steps executed. */
int64_t to_bench_run(uint64_t seed, int64_t count, int L, int M, int64_t steps, int threads, double* seconds);

#ifdef __cplusplus
}
#endif
#endif
