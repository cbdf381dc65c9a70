// tetris_piclim.hip -- kernels and C ABI (include/tetris_piclim.h) of the batched Tetris-piclim environment.
// gfx950 only.  The per-lane move lives in tpl_device.h; this file holds the kernels around it, the handle
// and the extern "C" entry points.  Citations "(:NNN)" are lines of the reference's game/tetris.py.
#include "tpl_internal.h"
#include "tpl_observe.h"
#include "tpl_step.h"

#include <hip/hip_bf16.h>

#include <type_traits>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

namespace tpl {

// ---------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

int fail_msg(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// A state plane holds whole blocks of the widest step-kernel geometry: boards past num_envs are padding that the
// step kernel may read and write back but never advances (they are created finished).
constexpr size_t kPlanePad = 512 * 4;
static inline size_t plane_bytes(int64_t n) { return align_up((size_t)n, kPlanePad) * sizeof(uint4); }
static inline size_t clock_bytes(int64_t n) { return align_up((size_t)n, kPlanePad) / kClockGroup * sizeof(unsigned long long); }
// piece words of a configuration: word w = entries [10w, 10w+12) is loaded when the cursor reaches 10w <= M, and
// entries up to M+1 (pieces[1] after the last move) lie inside word M/10
static inline int piece_words(int M) { return M / kWindowStride + 1; }
// record stride: a power of two (64, 128 or 256 bytes), so that a record's address is one shift-and-add
static inline uint32_t record_stride_shift(int M) {
    const size_t need = 32 + 8 * (size_t)(piece_words(M) - 1);
    uint32_t shift = 6;
    while (((size_t)1 << shift) < need) ++shift;
    return shift;
}
static inline size_t record_stride(int M) { return (size_t)1 << record_stride_shift(M); }

// ---------------------------------------------------------------------------------------------- kernels
// One Tetris.move per board (:354-422).  ACTION form: act0 = rot*10+loc; MOVE form: act0 = rot, act1 = loc.
// Each lane owns kBpl boards (block-strided, so every load is still 1 KiB per wave).  The kernel is written in
// phases so that a wave makes as few dependent trips to memory as the rules allow -- the whole grid is resident at
// once, so the time of a launch is the length of one wave's chain, not a sum over rounds of waves:
//   0  every board's state words and actions are requested
//   1  the piece-word gathers of the boards whose window runs out with this move are requested
//   2  the moves are computed (registers and LDS only)
//   3  reward / done / cleared are written
//   4  the pool records of the boards that finished are requested (auto-reset)
//   5  the state words are written
#ifdef TPL_DIAG_CLOCK
// diagnostic build only (tools/step_timeline.py): every wave stamps the 100 MHz real-time counter
#define TPL_STAMP(k) stamp[k] = __builtin_amdgcn_s_memrealtime()   /* wave-uniform: stays in scalar registers */
#else
#define TPL_STAMP(k) do { } while (0)
#endif

//
// Obs = float / __hip_bfloat16 makes it step-AND-observe (tpl_step_observe): after the state has been written back the
// wave expands its 64 boards -- as they stand after the move and the reset -- into their [64][217] span of p.obs, with
// the two stages of observe.hip (tpl_observe.h), from the registers it already holds.  A host-driven loop that needs the
// observation every step (north_star's step(action) -> obs, reward, done) then makes ONE launch per iteration and reads
// the 32-byte state once.  One board per lane in this form (a wave's boards must be 64 neighbours); the column words of
// the moving board sit at the head of the wave's observation bytes, which are written only after the move is over.
template <bool kActionForm, bool kAutoReset, int kBpl, int kThreads, typename Obs = void>
__global__ __launch_bounds__(kThreads) void step_kernel(const StepArgs p) {
    constexpr bool kObserve = !std::is_void<Obs>::value;
    static_assert(!kObserve || kBpl == 1, "step-and-observe: one board per lane");
    static_assert((int)sizeof(uint32_t) * kLdsCols * kLdsStride <= obs::kWaveLds, "the column words fit the wave's observation bytes");
    __shared__ ShapeWord s_shape[32];
    __shared__ uint32_t s_stat[4];
    // a board's column words while it moves (tpl_device.h); with kObserve they live inside s_rows
    __shared__ uint32_t s_cols[kObserve ? 1 : kThreads / 64][kObserve ? 1 : kLdsCols][kObserve ? 1 : kLdsStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[kObserve ? kThreads / 64 : 1][kObserve ? obs::kWaveLds : 16];
#ifdef TPL_DIAG_CLOCK
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
#endif
    TPL_STAMP(0);

    // phase 0.  The shape table's trip to LDS rides in the shadow of the board loads instead of ahead of them.
    const int64_t base = (int64_t)blockIdx.x * (kThreads * kBpl) + threadIdx.x;
    uint4 A[kBpl], B[kBpl];
    uint32_t a0[kBpl], a1[kBpl];
    unsigned long long clock[kBpl];
    bool valid[kBpl];
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        // No branch around the loads, so all of a lane's requests leave before the first wait.  The planes are
        // padded to whole blocks with boards that are finished for good (tpl_create), so a lane past the end
        // reads one of those; its action comes from the last real board and is never used.
        const int64_t i = base + (int64_t)k * kThreads;
        valid[k] = i < p.n;
        const int64_t j = valid[k] ? i : p.n - 1;
        A[k] = p.plane_a[i];
        B[k] = p.plane_b[i];
        // the index of this step: the clock of the wave's first group, read through the scalar cache (the wave owns both
        // of its groups' clocks and writes both back; all clocks are equal by construction)
        const int64_t wave_first = (int64_t)blockIdx.x * (kThreads * kBpl) + (int64_t)k * kThreads +
                                   (int64_t)(__builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63);
        clock[k] = p.clock[wave_first >> kClockShift];
        a0[k] = load_int(p.act0, p.int_shift, j);
        a1[k] = kActionForm ? 0u : load_int(p.act1, p.int_shift, j);
    }
    if (threadIdx.x < 32) s_shape[threadIdx.x] = kShapeTable[threadIdx.x];
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    __syncthreads();

    // phase 1.  pieces.pop(0) (:356) moves the cursor to moves_used + 1 whatever the move does.  When that is a
    // multiple of ten the window is down to its last two entries and piece word cursor/10 replaces it.
    uint64_t word[kBpl];
    uint32_t moves[kBpl], refill_word[kBpl];
    bool live[kBpl], refill[kBpl];
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        // an empty statement that needs every phase-0 result: it pins the one wait for phase 0 here, ahead of all
        // the gathers (left alone, the compiler waits for board k only after the gather of board k-1 has left,
        // which then has to be waited for as well)
        asm volatile("" ::"v"(A[k].x), "v"(B[k].x), "v"(a0[k]), "v"(a1[k]), "s"(clock[k]));
    }
    TPL_STAMP(1);
    // From here on the wave issues ahead of any wave of another kernel on its SIMD.  A supply generator running beside
    // the environment (PoolRefresher: long-lived, vector-ALU-bound waves) otherwise wins the issue arbitration as the
    // OLDEST wave of every SIMD it sits on, and a launch whose whole grid is resident at once lasts as long as its
    // slowest SIMD: 19.1 us per step instead of 15.4 with 1024 generator waves, 18.0 even with 32; 16.3 / 15.9 with this
    // line (profiles/r03_live_supply).  Raised only now, after the state words have arrived: at kernel entry it delays the
    // loads of the waves that start later (+0.35 us on the launch alone); here it costs nothing alone.
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        moves[k] = packed_moves(A[k]);
        const uint32_t tenth = tenths(moves[k] + 1u);
        refill_word[k] = window_word(tenth);
        live[k] = packed_state(B[k]) == ST_RUNNING;
        refill[k] = live[k] && window_runs_out(tenth) && (p.n_cfg[0] | p.n_cfg[1]) != 0u;
    }
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        const int64_t i = base + (int64_t)k * kThreads;
        word[k] = 0;
        if (refill[k]) {
            // the board's episode began at step clock - moves_used, in the pool buffer the board carries
            const uint32_t slot = packed_slot(B[k]);
            const uint32_t cfg = config_of(p, (uint32_t)i, clock[k] - moves[k], slot);
            word[k] = piece_word_at(pool_record(p, slot, cfg), refill_word[k]);
        }
    }

    // phase 2
    float reward[kBpl];
    uint32_t n_clear[kBpl];
    bool done[kBpl], reload[kBpl];
    bool finished = false;
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        reward[k] = 0.0f;
        n_clear[k] = 0;
        done[k] = true;
        reload[k] = false;
        if (!live[k]) continue;                               // frozen: reward 0, done, state untouched
        uint32_t rot, loc;
        if (kActionForm) {
            split_action(a0[k], rot, loc);
        } else {
            rot = a0[k];
            loc = a1[k];
        }
        Board s;
        bool topout;
        // the move indexes the board's columns by the piece's position: through LDS, where a lane can (tpl_device.h;
        // 0.1 us on the launch against the all-registers form, profiles/r02_step/ab_lds_move.log)
        unpack_board<true>(A[k], B[k], s);
        uint32_t* cols = kObserve ? (uint32_t*)s_rows[threadIdx.x >> 6] + (threadIdx.x & 63)
                                  : &s_cols[threadIdx.x >> 6][0][threadIdx.x & 63];
        lds_store_cols(cols, s.c);
        if (k == 0) {
#pragma unroll
            for (int c = kCols; c < kLdsCols; ++c) cols[c * kLdsStride] = kSentinelBit;
        }
        n_clear[k] = move_board_lds(s, cols, s_shape, rot, loc, p.L, p.M, topout);
        lds_load_cols(cols, s.c);
        next_window(s, refill[k], word[k]);                    // the falling piece is consumed even on a top-out

        reward[k] = step_reward(p, n_clear[k], s.state);
        done[k] = s.state != ST_RUNNING;
        if (done[k]) {
            // adds of a constant collapse to one popcount per wave; the line sum is rarely non-zero
            finished = true;
            atomicAdd(&s_stat[0], 1u);
            if (s.lines) atomicAdd(&s_stat[1], s.lines);
            if (s.state == ST_WON) atomicAdd(&s_stat[2], 1u);
            if (s.state == ST_LOST_TOPOUT) atomicAdd(&s_stat[3], 1u);
        }
        reload[k] = kAutoReset && done[k];
        pack_board<true>(s, A[k], B[k]);
    }

    TPL_STAMP(2);
    // phase 3
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        const int64_t i = base + (int64_t)k * kThreads;
        if (!valid[k]) continue;
        if (p.reward) p.reward[i] = reward[k];
        if (p.done) p.done[i] = done[k] ? 1 : 0;
        if (p.cleared) p.cleared[i] = (uint8_t)n_clear[k];
    }

    // phase 4: reset()/load_warm_reset() (:438-449) of the boards that finished.  One divergent region for all of
    // a lane's boards, with the records landing in fresh registers: every request leaves before the first wait.
    // (A lane in the region for one board only reads record 0 for its others and drops it.)
    if (kAutoReset) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < kBpl; ++k) any = any || reload[k];
        if (any) {
            uint4 RA[kBpl], RB[kBpl];
#pragma unroll
            for (int k = 0; k < kBpl; ++k) {
                const int64_t i = base + (int64_t)k * kThreads;
                // the new episode's first move is the next step: birth = clock + 1, from the current pool buffer
                const uint32_t cfg = config_of(p, (uint32_t)i, clock[k] + 1u, p.cur_slot);
                const uint4* rec = (const uint4*)pool_record(p, p.cur_slot, reload[k] ? cfg : 0u);
                RA[k] = rec[0];
                RB[k] = rec[1];
            }
#pragma unroll
            for (int k = 0; k < kBpl; ++k) {
                if (!reload[k]) continue;
                A[k] = RA[k];
                // the record carries slot 0: stamp the current one (as load_config does)
                B[k] = make_uint4(RB[k].x, RB[k].y | (p.cur_slot << 30), RB[k].z, RB[k].w);
            }
        }
    }

#ifdef TPL_DIAG_CLOCK
#pragma unroll
    for (int k = 0; k < kBpl; ++k) asm volatile("" ::"v"(A[k].x), "v"(B[k].x));
#endif
    TPL_STAMP(3);
    // phase 5.  With auto-reset every lane stores (a frozen or padding board is written back as it was read): no
    // branch between the stores, so none of them waits for an earlier one to be acknowledged.
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        const int64_t i = base + (int64_t)k * kThreads;
        if (!kAutoReset && !live[k]) continue;
        p.plane_a[i] = A[k];
        p.plane_b[i] = B[k];
    }
    // the groups' next step.  Every lane stores (the 32 lanes of a group write the same value to the same word: one
    // 16-byte write per wave): a branch on the lane index here would put a divergent region, and with it the
    // compiler's full memory wait, between the state stores.
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        const int64_t i = base + (int64_t)k * kThreads;
        p.clock[i >> kClockShift] = clock[k] + 1u;
    }

    TPL_STAMP(4);
    // the observation of the boards as they now stand (get_state() :435-436 as a vector; Model(217, 14), model/train.py:26)
    if constexpr (kObserve) {
        const int lane = threadIdx.x & 63;
        const int64_t wave_base = (int64_t)blockIdx.x * kThreads + (threadIdx.x & ~63);
        const int64_t left = p.n - wave_base;
        const int count = left >= 64 ? 64 : left > 0 ? (int)left : 0;          // wave-uniform
        uint8_t* const rows = s_rows[threadIdx.x >> 6];
        int lines_left = 0;
        if (lane < count) {
            Board s;
            unpack_board(A[0], B[0], s);
            lines_left = obs::board_to_bytes(s, p.L, p.M, rows, lane);
        }
        if (count > 0) obs::store_span<Obs>(rows, lane, count, wave_base, lines_left, (Obs*)p.obs);
    }
    // per-block statistics of the episodes that finished in this step -> one sharded 64-bit atomic per counter
    if (__syncthreads_or(finished ? 1 : 0)) {
        if (threadIdx.x < 4) {
            const uint32_t v = s_stat[threadIdx.x];
            if (v) atomicAdd(&p.stats[(size_t)(blockIdx.x % kStatShards) * kStatStride + threadIdx.x],
                             (unsigned long long)v);
        }
    }
#ifdef TPL_DIAG_CLOCK
    __builtin_amdgcn_s_waitcnt(0);           // the stores have been acknowledged
    TPL_STAMP(5);
    if (p.diag && (threadIdx.x & 63) == 0) {
        unsigned long long* d = p.diag + 6 * ((size_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6));
        for (int k = 0; k < 6; ++k) d[k] = stamp[k];
    }
#endif
}

// K consecutive moves per board in ONE launch: the board stays unpacked in registers between
// moves, so the 64 B/step state round trip of step_kernel is paid once per K steps.  Actions are pre-staged
// as actions[k][n] (uint8, stride `action_stride` between steps).  Semantics are exactly K calls of tpl_step:
// same freeze / auto-reset rules, same statistics, and -- when asked for -- the same per-step reward/done.
struct RolloutArgs {
    StepArgs s;
    const uint8_t* actions;     // [K][stride]; null in the random form
    uint64_t random_seed;       // random form: action of step k = explore(seed, global board index, step0 + k), uniform
    uint32_t step0;
    uint8_t* actions_out;       // random form: [K][n] or null, the actions that were played
    int64_t action_stride;
    uint32_t K;
    float* reward_steps;        // [K][n] or null
    uint8_t* done_steps;        // [K][n] or null
    float* reward_sum;          // [n] or null: sum over the K steps, accumulated in step order
    uint32_t* finished;         // [n] or null: episodes this board finished during the K steps
    uint32_t* trajectory;       // compact form only: [ceil(K / 4)][n], byte j of word w = the step 4 w + j (tpl_step.h)
};

// kCompact: the per-step outputs are ONE byte per board-step (what happened, not what it was worth: trajectory_code in
// tpl_step.h), gathered in a register and written as one dword per lane every fourth step -- against a float and a byte
// per step, which cost the kernel a quarter of its rate (two stores, the reward's arithmetic and a running sum in every
// wave-step of a loop that is bound by instruction issue).  No reward is computed in this form at all.
template <bool kAutoReset, bool kRandom, bool kCompact>
__global__ __launch_bounds__(kBlock) void rollout_kernel(const RolloutArgs q) {
    const StepArgs& p = q.s;
    __shared__ ShapeWord s_shape[32];
    __shared__ uint32_t s_stat[4];
    __shared__ uint32_t s_cols[kBlock / 64][kLdsCols][kLdsStride];      // the boards' column words between moves (tpl_device.h)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < p.n;
    uint4 A = make_uint4(0, 0, 0, 0), B = A;
    uint32_t act = 0;
    unsigned long long clock = 0;
    if (valid) {
        A = p.plane_a[i];
        B = p.plane_b[i];
        if (!kRandom) act = q.actions[i];
    }
    // the index of the first step: the clock of the wave's first group, through the scalar cache (as in step_kernel)
    const int64_t wave_first = (int64_t)blockIdx.x * kBlock + (int64_t)(__builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63);
    if (wave_first < p.n) clock = p.clock[wave_first >> kClockShift];
    if (threadIdx.x < 32) s_shape[threadIdx.x] = kShapeTable[threadIdx.x];
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    __syncthreads();

    RareTally tally;
    if (valid) {
        Board s;
        unpack_board<true>(A, B, s);
        uint32_t* cols = &s_cols[threadIdx.x >> 6][0][threadIdx.x & 63];
        lds_store_cols(cols, s.c);
#pragma unroll
        for (int k = kCols; k < kLdsCols; ++k) cols[k * kLdsStride] = kSentinelBit;
        const uint8_t* next_word = next_piece_word(current_record(s, p, (uint32_t)i, clock), s.moves);
        uint32_t until_refill = moves_until_refill(s.moves);
        // random form: the part of the exploration draw that does not depend on the step, and the pair word in use
        const uint32_t draw_base = kRandom ? explore_base(q.random_seed, (uint64_t)(p.global_offset + i)) : 0u;
        uint32_t draw_pair = 0;
        float rsum = 0.0f;
        // The per-step streams are addressed as a wave-uniform row pointer (the block's first board in step k's row,
        // moved on by the scalar unit) plus the lane's constant offset in the block: no vector instruction per step
        // and stream (a running per-lane pointer costs a 64-bit add each).
        const int64_t block_first = (int64_t)blockIdx.x * kBlock;
        const uint32_t in_block = threadIdx.x;
        const uint8_t* act_row = q.actions + block_first;
        uint8_t* act_out_row = q.actions_out + block_first;
        float* reward_row = q.reward_steps + block_first;
        uint8_t* done_row = q.done_steps + block_first;
        const bool want_reward = !kCompact && q.reward_steps != nullptr, want_done = !kCompact && q.done_steps != nullptr;   // wave-uniform
        uint32_t* traj_row = q.trajectory + block_first;
        uint32_t traj_word = 0;
        // The first action must have ARRIVED before the loop is entered.  Otherwise its register is "possibly still
        // being loaded" at the loop header on one of the two ways in, and the compiler puts a full memory wait at
        // the top of every iteration -- right behind the requests (next action, window word) that iteration has
        // just issued to ride under the move.
        asm volatile("" ::"v"(act));
        // A step's reward / done are STORED AT THE TOP OF THE NEXT ITERATION.  The compiler cannot count what is in flight
        // across the divergent regions of a move, so wherever a load is waited for it waits for everything -- stores
        // included.  Issued at the end of a step they would be waited for (their acknowledgement comes from memory) a few
        // instructions later, at the loop's bottom; issued here they have a whole move to complete under.
        float reward_prev = 0.0f;
        bool done_prev = false;
        for (uint32_t k = 0; k < q.K; ++k) {
            if (!kCompact && k > 0) {
                // (non-temporal: a trajectory is written once and consumed later, by someone else)
                if (want_reward) { __builtin_nontemporal_store(reward_prev, reward_row + in_block); reward_row += p.n; }
                if (want_done) { __builtin_nontemporal_store((uint8_t)(done_prev ? 1 : 0), done_row + in_block); done_row += p.n; }
            }
            if (kCompact && k > 0 && (k & 3u) == 0u) {                      // steps k-4 .. k-1 (a wave-uniform test)
                __builtin_nontemporal_store(traj_word, traj_row + in_block);
                traj_row += p.n;
                traj_word = 0;
            }
            // next step's action is independent of the board: fetch it under this step's move
            uint32_t act_next = 0;
            if (kRandom) {
                // the uniform random policy, drawn on the device (what tpl_explore_actions gives at epsilon = 1): one hash
                // serves two steps, so it is taken at even steps and at the first step of the launch (a wave-uniform test)
                const uint32_t step = q.step0 + k;
                if (k == 0 || (step & 1u) == 0u) draw_pair = explore_pair(draw_base, q.random_seed, step);
                act = explore_pick(draw_pair, step);
                if (q.actions_out) { act_out_row[in_block] = (uint8_t)act; act_out_row += p.n; }
            } else {
                act_row += q.action_stride;
                if (k + 1 < q.K) act_next = act_row[in_block];
            }
            uint32_t rot, loc;
            split_small_action(act, rot, loc);
            float reward;
            uint32_t code;
            const bool done = advance_board_lds<kAutoReset>(s, cols, next_word, until_refill, rot, loc, p, (uint32_t)i, clock + k,
                                                            s_shape, reward, code, tally);
            if (kCompact) traj_word |= code << (8u * (k & 3u));              // the shift is wave-uniform
            else {
                rsum = rsum + reward;
                reward_prev = reward;
                done_prev = done;
            }
            act = act_next;
        }
        if (want_reward) __builtin_nontemporal_store(reward_prev, reward_row + in_block);
        if (want_done) __builtin_nontemporal_store((uint8_t)(done_prev ? 1 : 0), done_row + in_block);
        if (kCompact) __builtin_nontemporal_store(traj_word, traj_row + in_block);      // the last word, whole or not
        lds_load_cols(cols, s.c);
        pack_board<true>(s, A, B);
        p.plane_a[i] = A;
        p.plane_b[i] = B;
        if ((threadIdx.x & (kClockGroup - 1)) == 0) p.clock[i >> kClockShift] = clock + q.K;
        if (!kCompact && q.reward_sum) q.reward_sum[i] = rsum;
        if (q.finished) q.finished[i] = tally.episodes;
    }
    flush_tally(tally, s_stat, p.stats);
}

// Tetris.reset() (:438-443) for every board (mask == null: the host has zeroed the step clocks, every episode begins
// at step 0) or the masked ones (their next episode begins at the group's next step).
__global__ __launch_bounds__(kBlock) void reset_kernel(const StepArgs p, const uint8_t* mask) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n) return;
    unsigned long long birth = 0;
    if (mask) {
        if (!mask[i]) return;
        birth = p.clock[i >> kClockShift];
    }
    const uint32_t cfg = config_of(p, (uint32_t)i, birth, p.cur_slot);
    uint4 A, B;
    load_config(p, cfg, A, B);
    p.plane_a[i] = A;
    p.plane_b[i] = B;
}

// interchange (rows u16[20], pieces u8[M+1]) -> pool records in the resident layout
__device__ __forceinline__ uint64_t piece_word(const uint8_t* pc, uint32_t M, uint32_t w) {
    uint64_t word = 0;
    for (uint32_t j = 0; j < (uint32_t)kWindowEntries; ++j) {
        const uint32_t idx = w * kWindowStride + j;
        word |= (uint64_t)(idx <= M ? (uint32_t)(pc[idx] & 7u) : 7u) << (3u * j);
    }
    return word;
}

__global__ __launch_bounds__(kBlock) void pack_configs_kernel(const uint16_t* rows, const uint8_t* pieces, int64_t n_cfg,
                                                             uint32_t M, uint32_t words, uint8_t* pool, uint32_t stride) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_cfg) return;
    uint16_t r[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) r[k] = rows[i * kRows + k];
    Board s;
    rows_to_cols(r, s.c);
    const uint8_t* pc = pieces + (size_t)i * (M + 1);
    set_window(s, piece_word(pc, M, 0));
    s.state = ST_RUNNING; s.lines = 0; s.moves = 0; s.slot = 0;
    uint4 A, B;
    pack_board(s, A, B);
    uint8_t* rec = pool + (size_t)i * stride;
    ((uint4*)rec)[0] = A;
    ((uint4*)rec)[1] = B;
    for (uint32_t w = 1; w < words; ++w) ((uint64_t*)(rec + 32))[w - 1] = piece_word(pc, M, w);
}

// packed records -> side records (the board unpacked, column words with the sentinel bit: tpl_device.h)
__global__ __launch_bounds__(kBlock) void build_side_kernel(const uint8_t* pool, uint32_t stride_shift, int64_t n_cfg, uint8_t* side) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_cfg) return;
    const uint4* rec = (const uint4*)(pool + ((size_t)i << stride_shift));
    Board s;
    unpack_board<true>(rec[0], rec[1], s);
    uint4* sr = (uint4*)(side + ((size_t)i << kSideShift));
    sr[0] = make_uint4(s.c[0], s.c[1], s.c[2], s.c[3]);
    sr[1] = make_uint4(s.c[4], s.c[5], s.c[6], s.c[7]);
    sr[2] = make_uint4(s.c[8], s.c[9], s.window, s.window_hi);
    sr[3] = make_uint4(0, 0, 0, 0);
}

// get_state (:435-436) + public attributes, resident layout -> interchange layout
__global__ __launch_bounds__(kBlock) void export_kernel(const uint4* plane_a, const uint4* plane_b, int64_t n, uint32_t M,
                                                       uint16_t* rows, uint8_t* cur, uint8_t* nxt, uint8_t* lines,
                                                       uint8_t* moves, uint8_t* state, uint8_t* pieces_left) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Board s;
    unpack_board(plane_a[i], plane_b[i], s);
    if (rows) {
#pragma unroll
        for (int r = 0; r < kRows; ++r) rows[i * kRows + r] = (uint16_t)row_of_cols(s.c, r);
    }
    if (cur) cur[i] = (uint8_t)(s.window & 7u);
    if (nxt) nxt[i] = (uint8_t)((s.window >> 3) & 7u);
    if (lines) lines[i] = (uint8_t)s.lines;
    if (moves) moves[i] = (uint8_t)s.moves;
    if (state) state[i] = (uint8_t)(s.state == ST_LOST_TOPOUT ? ST_LOST_LIMIT : s.state);
    if (pieces_left) pieces_left[i] = (uint8_t)(M + 1u - s.moves - (s.state == ST_LOST_TOPOUT ? 1u : 0u));
}

// Observation [n][217], element-wise form (the fast form is observe.hip; this one serves outputs that are not
// 16-byte aligned): a wave expands its 64 boards cooperatively so that every store instruction writes 256
// contiguous bytes.  Each lane first turns its own board into a row-major 200-bit cell vector
// (7 words) + one feature word in LDS; then for each board all 64 lanes emit its 217 values.
template <typename T>
__device__ __forceinline__ T obs_cast(float v);
template <> __device__ __forceinline__ float obs_cast<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 obs_cast<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename T>
__global__ __launch_bounds__(kBlock) void expand_obs_kernel(const uint4* plane_a, const uint4* plane_b, int64_t n,
                                                           uint32_t L, uint32_t M, T* out) {
    __shared__ uint32_t s_bits[kBlock / 64][64][9];   // 7 cell words + features, padded to 9 (odd stride)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t base = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * 64;
    const int64_t i = base + lane;
    if (i < n) {
        Board s;
        unpack_board(plane_a[i], plane_b[i], s);
        uint32_t words[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const uint32_t row = row_of_cols(s.c, r);
            const int bit = r * 10;
            words[bit >> 5] |= row << (bit & 31);
            if ((bit & 31) > 22) words[(bit >> 5) + 1] |= row >> (32 - (bit & 31));
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) s_bits[wave][lane][k] = words[k];
        const uint32_t terminal = s.state != ST_RUNNING ? 1u : 0u;
        s_bits[wave][lane][7] = (s.window & 63u) | (terminal << 6) | (s.lines << 8) | (s.moves << 16);
    }
    __syncthreads();
    const int64_t count = (n - base) < 64 ? (n - base) : 64;
    for (int64_t j = 0; j < count; ++j) {
        const uint32_t* bits = s_bits[wave][j];
        T* o = out + (base + j) * TPL_OBS_DIM;
        const uint32_t f = bits[7];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = lane + 64 * t;
            if (e < 200) {
                o[e] = obs_cast<T>((float)((bits[e >> 5] >> (e & 31)) & 1u));
            } else if (e < TPL_OBS_DIM) {
                float v;
                if (e < 207) v = ((f & 7u) == (uint32_t)(e - 200)) ? 1.0f : 0.0f;
                else if (e < 214) v = (((f >> 3) & 7u) == (uint32_t)(e - 207)) ? 1.0f : 0.0f;
                else if (e == 214) v = (float)((int)L - (int)((f >> 8) & 0xFFu));
                else if (e == 215) v = (float)((int)M - (int)((f >> 16) & 0xFFu));
                else v = (float)((f >> 6) & 1u);
                o[e] = obs_cast<T>(v);
            }
        }
    }
}

// 14 policy outputs -> one action: argmax over the 4 rotation logits, argmax over the 10 location logits
// (lowest index wins ties; NaN never wins), action = rot*10 + loc.  Model(217, 14): model/train.py:26.
template <typename T>
__device__ __forceinline__ float logit_f32(T v);
template <> __device__ __forceinline__ float logit_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float logit_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

template <typename T>
__global__ __launch_bounds__(kBlock) void decode_actions_kernel(const T* logits, int64_t n, uint8_t* action) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const T* row = logits + i * 14;
    uint32_t rot = 0, loc = 0;
    float best = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float v = logit_f32<T>(row[k]);
        if (v > best) { best = v; rot = (uint32_t)k; }
    }
    best = -INFINITY;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float v = logit_f32<T>(row[4 + k]);
        if (v > best) { best = v; loc = (uint32_t)k; }
    }
    action[i] = (uint8_t)(rot * 10u + loc);
}

__global__ void reduce_stats_kernel(const unsigned long long* shards, unsigned long long* out) {
    const int k = threadIdx.x;   // 4 threads
    unsigned long long s = 0;
    for (int i = 0; i < kStatShards; ++i) s += shards[(size_t)i * kStatStride + k];
    out[k] = s;
}

// synthetic workload of SURVEY 8(d): boards, 7-bag piece lists, uniform actions from the counter-based generator
__global__ __launch_bounds__(kBlock) void synth_configs_kernel(uint64_t seed, int64_t first, int64_t count, uint32_t L,
                                                              uint32_t M, uint16_t* rows, uint8_t* pieces) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= count) return;
    const uint64_t g = (uint64_t)(first + b);
    if (rows) {
        const uint32_t filled = L < (uint32_t)kRows ? L : (uint32_t)kRows;
        for (uint32_t r = 0; r < (uint32_t)kRows; ++r) {
            uint32_t v = 0;
            if (r >= kRows - filled) {
                const uint64_t h = rng(seed, 0, g, r);
                v = (uint32_t)(h & 0x3FFu);
                if (v == 0x3FFu) v &= ~(1u << (uint32_t)((h >> 10) % 10u));
            }
            rows[b * kRows + r] = (uint16_t)v;
        }
    }
    if (pieces) {
        const uint32_t len = M + 1;
        uint8_t* out = pieces + (size_t)b * len;
        uint32_t produced = 0;
        for (uint32_t bag = 0; produced < len; ++bag) {
            const uint64_t h = rng(seed, 1, g, bag);
            // the 7-bag as seven 3-bit fields of one register; Fisher-Yates by field swaps
            uint32_t a = 0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15 | 6u << 18;
            for (int j = 6; j >= 1; --j) {
                const uint32_t k = (uint32_t)((h >> (8 * (6 - j))) & 0xFFu) % (uint32_t)(j + 1);
                const uint32_t vj = (a >> (3 * j)) & 7u, vk = (a >> (3 * k)) & 7u;
                a = (a & ~(7u << (3 * j))) | (vk << (3 * j));
                a = (a & ~(7u << (3 * k))) | (vj << (3 * k));
            }
            for (int j = 0; j < 7 && produced < len; ++j) out[produced++] = (uint8_t)((a >> (3 * j)) & 7u);
        }
    }
}

__global__ __launch_bounds__(kBlock) void synth_actions_kernel(uint64_t seed, int64_t first, int64_t count, uint64_t step,
                                                              uint8_t* action) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= count) return;
    const uint64_t h = rng(seed, 2, (uint64_t)(first + b), step);
    const uint32_t rot = (uint32_t)(h & 3u);
    const uint32_t loc = (uint32_t)((h >> 8) & 0xFFFFu) % 10u;
    action[b] = (uint8_t)(rot * 10u + loc);
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

template <typename Obs>
static void launch_step_observe(bool auto_reset, const StepArgs& a, hipStream_t stream) {
    const dim3 grid((unsigned)((a.n + kBlock - 1) / kBlock)), block(kBlock);
    if (auto_reset) hipLaunchKernelGGL((step_kernel<true, true, 1, kBlock, Obs>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((step_kernel<true, false, 1, kBlock, Obs>), grid, block, 0, stream, a);
}

template <int kBpl, int kThreads>
static void launch_step_cfg(bool action_form, bool auto_reset, const StepArgs& a, hipStream_t stream) {
    const int64_t per_block = (int64_t)kThreads * kBpl;
    const dim3 grid((unsigned)((a.n + per_block - 1) / per_block)), block(kThreads);
    if (action_form) {
        if (auto_reset) hipLaunchKernelGGL((step_kernel<true, true, kBpl, kThreads>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((step_kernel<true, false, kBpl, kThreads>), grid, block, 0, stream, a);
    } else {
        if (auto_reset) hipLaunchKernelGGL((step_kernel<false, true, kBpl, kThreads>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((step_kernel<false, false, kBpl, kThreads>), grid, block, 0, stream, a);
    }
}

template <int kBpl>
static void launch_step_bpl(int threads, bool action_form, bool auto_reset, const StepArgs& a, hipStream_t stream) {
    switch (threads) {
        case 64: launch_step_cfg<kBpl, 64>(action_form, auto_reset, a, stream); break;
        case 128: launch_step_cfg<kBpl, 128>(action_form, auto_reset, a, stream); break;
        case 512: launch_step_cfg<kBpl, 512>(action_form, auto_reset, a, stream); break;
        default: launch_step_cfg<kBpl, 256>(action_form, auto_reset, a, stream); break;
    }
}

// what every board-advancing entry point checks before it launches, and the bookkeeping of the pool swap rule
int check_can_advance(tpl_env* e) {
    if (e->auto_reset && e->pool[e->cur_slot].n_cfg == 0) return fail_msg(TPL_ERR_STATE, "auto_reset needs tpl_load_configs first");
    if (e->assign_dirty)
        return fail_msg(TPL_ERR_STATE, "the assignment mode changed while boards were running: call tpl_reset(env, NULL, stream) first");
    return TPL_OK;
}

void count_steps(tpl_env* e, int64_t steps) {
    e->steps_since_swap += steps;
    // every episode that began before the swap has ended (or frozen) after M + 1 steps
    if (e->other_slot_live && e->steps_since_swap > (int64_t)e->M) e->other_slot_live = false;
}

static int launch_step(tpl_env* e, const void* act0, const void* act1, int32_t dtype, float* reward, uint8_t* done,
                       uint8_t* cleared, hipStream_t stream) {
    if (dtype != TPL_U8 && dtype != TPL_I32 && dtype != TPL_I64) return fail_msg(TPL_ERR_ARG, "unknown integer dtype %d", dtype);
    if (int rc = check_can_advance(e)) return rc;
    StepArgs a = make_args(e);
    a.act0 = act0; a.act1 = act1; a.int_shift = dtype == TPL_U8 ? 0u : dtype == TPL_I32 ? 2u : 3u; a.reward = reward; a.done = done; a.cleared = cleared;
    const bool action_form = act1 == nullptr;
    switch (e->boards_per_lane) {
        case 1: launch_step_bpl<1>(e->block_threads, action_form, e->auto_reset != 0, a, stream); break;
        case 2: launch_step_bpl<2>(e->block_threads, action_form, e->auto_reset != 0, a, stream); break;
        default: launch_step_bpl<4>(e->block_threads, action_form, e->auto_reset != 0, a, stream); break;
    }
    TPL_HIP(hipGetLastError());
    count_steps(e, 1);
    return TPL_OK;
}

template <bool kRandom, bool kCompact = false>
static void launch_rollout(const tpl_env* e, const RolloutArgs& q, hipStream_t stream) {
    const dim3 grid(blocks_for(e->n)), block(kBlock);
    if (e->auto_reset) hipLaunchKernelGGL((rollout_kernel<true, kRandom, kCompact>), grid, block, 0, stream, q);
    else hipLaunchKernelGGL((rollout_kernel<false, kRandom, kCompact>), grid, block, 0, stream, q);
}

// the learner's side of the compact trajectory: byte -> (reward, done), the arithmetic of step_reward (one rounded multiply,
// at most one rounded add)
__global__ __launch_bounds__(kBlock) void decode_trajectory_kernel(const uint32_t* trajectory, int64_t n, uint32_t K, float r_line,
                                                                  float r_win, float r_lose, float* reward_steps,
                                                                  uint8_t* done_steps) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t words = (K + 3u) / 4u;
    for (uint32_t w = 0; w < words; ++w) {
        const uint32_t word = trajectory[(int64_t)w * n + i];
        for (uint32_t j = 0; j < 4u && 4u * w + j < K; ++j) {
            const uint32_t code = (word >> (8u * j)) & 0xFFu, end = (code >> 3) & 3u;
            const bool frozen = (code & kTrajFrozen) != 0u;
            float reward = r_line * (float)(code & 7u);
            if (end == 1u) reward = reward + r_win;
            if (end >= 2u) reward = reward + r_lose;
            const int64_t at = (int64_t)(4u * w + j) * n + i;
            if (reward_steps) reward_steps[at] = frozen ? 0.0f : reward;
            if (done_steps) done_steps[at] = (uint8_t)((frozen || end != 0u) ? 1 : 0);
        }
    }
}

// The resets of the multi-step kernel read the current pool's side records: written once per pool, on first use (a
// caller that only steps never pays for them).  Under stream capture the build is only RECORDED into the graph -- it
// has not run when the call returns and runs again with every replay -- so the pool is not marked: the next eager
// rollout, or a rollout captured into another graph, builds them itself.
static int ensure_side_records(tpl_env* e, hipStream_t stream) {
    Pool& cur = e->pool[e->cur_slot];
    if (cur.n_cfg == 0 || cur.side_ready) return TPL_OK;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    TPL_HIP(hipStreamIsCapturing(stream, &capturing));
    hipLaunchKernelGGL(build_side_kernel, dim3(blocks_for(cur.n_cfg)), dim3(kBlock), 0, stream, cur.rec, e->stride_shift,
                       cur.n_cfg, cur.side);
    TPL_HIP(hipGetLastError());
    cur.side_ready = capturing == hipStreamCaptureStatusNone;
    return TPL_OK;
}

}  // namespace tpl

using namespace tpl;

// ============================================================================================== C ABI
extern "C" {

const char* tpl_last_error(void) { return g_err; }

const char* tpl_version(void) { return "tetris_piclim 0.2.0 (gfx950)"; }

size_t tpl_workspace_bytes(int64_t num_envs, int32_t M) {
    if (num_envs <= 0 || M < 1) return 0;
    return plane_bytes(num_envs) * 2 + align_up((size_t)kStatShards * kStatStride * sizeof(unsigned long long), 256) +
           align_up(clock_bytes(num_envs), 256);
}

size_t tpl_pool_bytes(int64_t n_cfg, int32_t M) {
    if (n_cfg <= 0 || M < 1) return 0;
    return (size_t)n_cfg * (record_stride(M) + ((size_t)1 << kSideShift));   // records, then the side records
}

int tpl_create(tpl_env** out, int64_t num_envs, int32_t L, int32_t M, int32_t device_id, int64_t global_offset,
               uint64_t seed, void* workspace, size_t workspace_bytes) {
    if (!out) return fail_msg(TPL_ERR_ARG, "out is null");
    *out = nullptr;
    if (num_envs <= 0 || num_envs > ((int64_t)1 << 31)) return fail_msg(TPL_ERR_ARG, "num_envs %lld out of range", (long long)num_envs);
    if (L < 1 || L > 250) return fail_msg(TPL_ERR_ARG, "L=%d out of range [1, 250]", L);
    if (M < 1 || M > 254) return fail_msg(TPL_ERR_ARG, "M=%d out of range [1, 254]", M);
    if (global_offset < 0) return fail_msg(TPL_ERR_ARG, "global_offset is negative");
    int ndev = 0;
    TPL_HIP(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail_msg(TPL_ERR_ARG, "device %d not in [0, %d)", device_id, ndev);
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail_msg(TPL_ERR_HIP, "hipSetDevice(%d) failed", device_id);

    const size_t need = tpl_workspace_bytes(num_envs, M);
    tpl_env* e = new (std::nothrow) tpl_env();
    if (!e) return fail_msg(TPL_ERR_NOMEM, "host allocation failed");
    e->n = num_envs; e->L = L; e->M = M; e->device = device_id;
    e->global_offset = global_offset; e->seed = seed;
    // step-kernel geometry: one board per lane while that still puts the whole grid on the chip at once (2^19 boards = 8 waves
    // on each of the 1024 SIMDs), two beyond: 6.3 against 6.7 us per step at 262,144 boards, 8.4 against 8.8 at 524,288,
    // 12.8 against 11.9 at 786,432 (round-3 tuning runs, profiles/NOTES.md).  tpl_set_tuning overrides.
    e->boards_per_lane = num_envs <= ((int64_t)1 << 19) ? 1 : 2;
    char* base = (char*)workspace;
    if (base) {
        if (workspace_bytes < need) { delete e; return fail_msg(TPL_ERR_ARG, "workspace has %zu bytes, need %zu", workspace_bytes, need); }
        if (((uintptr_t)base & 255u) != 0) { delete e; return fail_msg(TPL_ERR_ARG, "workspace must be 256-byte aligned"); }
    } else {
        hipError_t err = hipMalloc((void**)&base, need);
        if (err != hipSuccess) { delete e; return fail_msg(TPL_ERR_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(err)); }
        e->owned = base;
    }
    const size_t n = (size_t)num_envs, padded = plane_bytes(num_envs) / sizeof(uint4);
    e->plane_a = (uint4*)base; base += plane_bytes(num_envs);
    e->plane_b = (uint4*)base; base += plane_bytes(num_envs);
    e->stats = (unsigned long long*)base; base += align_up((size_t)kStatShards * kStatStride * sizeof(unsigned long long), 256);
    e->clock = (unsigned long long*)base;
    e->stride_shift = record_stride_shift(M);
    // every board: empty, running, no pieces; every padding board: lost (the state field sits in the second word of
    // a plane-B entry; the fill puts the same pattern in all four, which a padding board never looks at).  Step clocks
    // zero.  Synchronised, because the caller's later work may run on a stream that does not order itself against the
    // null stream.
    hipError_t err = hipMemset(e->plane_a, 0, need);
    if (err == hipSuccess && padded > n)
        err = hipMemsetD32((hipDeviceptr_t)(e->plane_b + n), (int)((uint32_t)ST_LOST_LIMIT << 28), (padded - n) * 4);
    if (err == hipSuccess) err = hipStreamSynchronize(nullptr);
    if (err != hipSuccess) {
        if (e->owned) (void)hipFree(e->owned);
        delete e;
        return fail_msg(TPL_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(err));
    }
    *out = e;
    return TPL_OK;
}

int tpl_destroy(tpl_env* e) {
    if (!e) return TPL_OK;
    DeviceGuard guard(e->device);
    if (e->owned) (void)hipFree(e->owned);
    for (int k = 0; k < 2; ++k)
        if (e->pool[k].owned) (void)hipFree(e->pool[k].owned);
    delete e;
    return TPL_OK;
}

int tpl_set_options(tpl_env* e, int32_t auto_reset, int32_t assign_mode, float per_line, float win, float lose) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (assign_mode != TPL_ASSIGN_HASH && assign_mode != TPL_ASSIGN_SEQUENTIAL) return fail_msg(TPL_ERR_ARG, "unknown assign_mode %d", assign_mode);
    // a running board finds its pool entry again through the assignment function (tpl_device.h): changing the
    // function under running boards would hand them the piece lists of other configurations
    if (assign_mode != e->assign_mode && e->pool[e->cur_slot].n_cfg != 0) e->assign_dirty = true;
    e->auto_reset = auto_reset ? 1 : 0; e->assign_mode = assign_mode;
    e->r_line = per_line; e->r_win = win; e->r_lose = lose;
    return TPL_OK;
}

int tpl_load_configs(tpl_env* e, const uint16_t* rows, const uint8_t* pieces, int64_t n_cfg, void* pool_mem,
                     size_t pool_bytes, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!rows || !pieces) return fail_msg(TPL_ERR_ARG, "rows/pieces is null");
    if (n_cfg <= 0 || n_cfg >= ((int64_t)1 << 32)) return fail_msg(TPL_ERR_ARG, "n_cfg %lld out of range", (long long)n_cfg);
    DeviceGuard guard(e->device);
    // the first pool goes into the current slot; every later one into the OTHER slot, which then becomes current:
    // boards that are mid-episode keep refilling their piece windows from the buffer their configuration lives in
    const bool first = e->pool[e->cur_slot].n_cfg == 0;
    const int target = first ? e->cur_slot : e->cur_slot ^ 1;
    if (!first && e->other_slot_live)
        return fail_msg(TPL_ERR_STATE,
                        "boards that began before the previous tpl_load_configs may still be running on the buffer this call would "
                        "overwrite: step %lld more time(s) (M + 1 after a swap) or call tpl_reset(env, NULL, stream) first",
                        (long long)((int64_t)e->M + 1 - e->steps_since_swap));
    const size_t need = tpl_pool_bytes(n_cfg, e->M);
    char* base = (char*)pool_mem;
    void* newly_owned = nullptr;
    if (base) {
        if (pool_bytes < need) return fail_msg(TPL_ERR_ARG, "pool_mem has %zu bytes, need %zu", pool_bytes, need);
        if (((uintptr_t)base & 255u) != 0) return fail_msg(TPL_ERR_ARG, "pool_mem must be 256-byte aligned");
    } else {
        hipError_t err = hipMalloc((void**)&base, need);
        if (err != hipSuccess) return fail_msg(TPL_ERR_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(err));
        newly_owned = base;
    }
    Pool& slot = e->pool[target];
    if (slot.owned) {
        // no board refers to it any more, but launches that read it may still be enqueued
        TPL_HIP(hipStreamSynchronize((hipStream_t)stream));
        (void)hipFree(slot.owned);
    }
    slot.owned = newly_owned;
    slot.rec = (uint8_t*)base;
    slot.side = (uint8_t*)base + (size_t)n_cfg * record_stride(e->M);
    slot.side_ready = false;
    slot.n_cfg = n_cfg;
    TPL_HIP(hipMemsetAsync(base, 0, (size_t)n_cfg * record_stride(e->M), (hipStream_t)stream));     // record padding reads as zero
    hipLaunchKernelGGL(pack_configs_kernel, dim3(blocks_for(n_cfg)), dim3(kBlock), 0, (hipStream_t)stream, rows, pieces,
                       n_cfg, (uint32_t)e->M, (uint32_t)piece_words(e->M), slot.rec, (uint32_t)record_stride(e->M));
    TPL_HIP(hipGetLastError());
    if (!first) {
        e->cur_slot = target;
        e->steps_since_swap = 0;
        e->other_slot_live = true;
    }
    return TPL_OK;
}

int tpl_pool_info(tpl_env* e, int32_t* current_slot, int64_t* n_cfg_current, int64_t* n_cfg_other, int64_t* steps_until_swap) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (current_slot) *current_slot = e->cur_slot;
    if (n_cfg_current) *n_cfg_current = e->pool[e->cur_slot].n_cfg;
    if (n_cfg_other) *n_cfg_other = e->pool[e->cur_slot ^ 1].n_cfg;
    if (steps_until_swap) *steps_until_swap = e->other_slot_live ? (int64_t)e->M + 1 - e->steps_since_swap : 0;
    return TPL_OK;
}

int tpl_pool_set_hold(tpl_env* e, int64_t steps_until_swap) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (steps_until_swap < 0 || steps_until_swap > (int64_t)e->M + 1) return fail_msg(TPL_ERR_ARG, "steps_until_swap out of [0, M + 1]");
    e->other_slot_live = steps_until_swap > 0;
    e->steps_since_swap = (int64_t)e->M + 1 - steps_until_swap;
    return TPL_OK;
}

int tpl_stream_create(int32_t device_id, int32_t cu_count, int32_t low_priority, void** stream) {
    if (!stream) return fail_msg(TPL_ERR_ARG, "stream is null");
    *stream = nullptr;
    int ndev = 0;
    TPL_HIP(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail_msg(TPL_ERR_ARG, "device %d not in [0, %d)", device_id, ndev);
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail_msg(TPL_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    hipDeviceProp_t prop;
    TPL_HIP(hipGetDeviceProperties(&prop, device_id));
    const int cus = prop.multiProcessorCount;
    if (cu_count < 0 || cu_count > cus) return fail_msg(TPL_ERR_ARG, "cu_count %d not in [0, %d]", cu_count, cus);
    hipStream_t s = nullptr;
    if (cu_count > 0) {
        // The driver deals the mask's bits out round-robin: bit i -> XCD i mod 8, then shader engine, then CU -- so the
        // LOW cu_count bits are an even spread over the chip's XCDs and shader engines (32 = one CU in each of the four
        // shader engines of each of the eight XCDs).
        uint32_t mask[32] = {0};
        const uint32_t words = (uint32_t)(cus + 31) / 32u;
        for (int i = 0; i < cu_count; ++i) mask[i >> 5] |= 1u << (i & 31);
        TPL_HIP(hipExtStreamCreateWithCUMask(&s, words, mask));
    } else if (low_priority) {
        int least = 0, greatest = 0;
        TPL_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        TPL_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least));
    } else {
        TPL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    *stream = s;
    return TPL_OK;
}

int tpl_stream_destroy(int32_t device_id, void* stream) {
    if (!stream) return TPL_OK;
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail_msg(TPL_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    TPL_HIP(hipStreamSynchronize((hipStream_t)stream));
    TPL_HIP(hipStreamDestroy((hipStream_t)stream));
    return TPL_OK;
}

int tpl_note_steps(tpl_env* e, int64_t steps) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (steps < 0) return fail_msg(TPL_ERR_ARG, "steps is negative");
    count_steps(e, steps);
    return TPL_OK;
}

int tpl_reset(tpl_env* e, const uint8_t* mask, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (e->pool[e->cur_slot].n_cfg == 0) return fail_msg(TPL_ERR_STATE, "tpl_reset needs tpl_load_configs first");
    if (mask && e->assign_dirty)
        return fail_msg(TPL_ERR_STATE, "the assignment mode changed while boards were running: a FULL reset (mask NULL) is due");
    DeviceGuard guard(e->device);
    if (!mask) {
        TPL_HIP(hipMemsetAsync(e->stats, 0, (size_t)kStatShards * kStatStride * sizeof(unsigned long long), (hipStream_t)stream));
        TPL_HIP(hipMemsetAsync(e->clock, 0, clock_bytes(e->n), (hipStream_t)stream));
    }
    hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(e->n)), dim3(kBlock), 0, (hipStream_t)stream, make_args(e), mask);
    TPL_HIP(hipGetLastError());
    if (!mask) {
        // every board now lives in the current slot and starts from step 0 under the current assignment mode
        e->other_slot_live = false;
        e->assign_dirty = false;
    }
    return TPL_OK;
}

int tpl_move(tpl_env* e, const void* rot, const void* loc, int32_t dtype, float* reward, uint8_t* done,
             uint8_t* cleared, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!rot || !loc) return fail_msg(TPL_ERR_ARG, "rot/loc is null");
    DeviceGuard guard(e->device);
    return launch_step(e, rot, loc, dtype, reward, done, cleared, (hipStream_t)stream);
}

int tpl_step(tpl_env* e, const void* action, int32_t dtype, float* reward, uint8_t* done, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!action) return fail_msg(TPL_ERR_ARG, "action is null");
    DeviceGuard guard(e->device);
    return launch_step(e, action, nullptr, dtype, reward, done, nullptr, (hipStream_t)stream);
}

int tpl_step_observe(tpl_env* e, const void* action, int32_t dtype, float* reward, uint8_t* done, void* obs, int32_t obs_dtype,
                     void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!action || !obs) return fail_msg(TPL_ERR_ARG, "action/obs is null");
    if (dtype != TPL_U8 && dtype != TPL_I32 && dtype != TPL_I64) return fail_msg(TPL_ERR_ARG, "unknown integer dtype %d", dtype);
    if (obs_dtype != TPL_F32 && obs_dtype != TPL_BF16) return fail_msg(TPL_ERR_ARG, "unknown observation dtype %d", obs_dtype);
    if (((uintptr_t)obs & 15u) != 0)
        return fail_msg(TPL_ERR_ARG, "obs must be 16-byte aligned (tpl_step followed by tpl_expand_obs takes any address)");
    if (int rc = check_can_advance(e)) return rc;
    DeviceGuard guard(e->device);
    StepArgs a = make_args(e);
    a.act0 = action; a.act1 = nullptr; a.int_shift = dtype == TPL_U8 ? 0u : dtype == TPL_I32 ? 2u : 3u;
    a.reward = reward; a.done = done; a.cleared = nullptr; a.obs = obs;
    if (obs_dtype == TPL_F32) launch_step_observe<float>(e->auto_reset != 0, a, (hipStream_t)stream);
    else launch_step_observe<__hip_bfloat16>(e->auto_reset != 0, a, (hipStream_t)stream);
    TPL_HIP(hipGetLastError());
    count_steps(e, 1);
    return TPL_OK;
}

int tpl_rollout(tpl_env* e, const uint8_t* actions, int64_t action_stride, int32_t num_steps, float* reward_steps,
                uint8_t* done_steps, float* reward_sum, uint32_t* finished, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!actions) return fail_msg(TPL_ERR_ARG, "actions is null");
    if (num_steps < 1) return fail_msg(TPL_ERR_ARG, "num_steps must be >= 1");
    if (action_stride < e->n) return fail_msg(TPL_ERR_ARG, "action_stride %lld is smaller than num_envs", (long long)action_stride);
    if (int rc = check_can_advance(e)) return rc;
    DeviceGuard guard(e->device);
    if (int rc = ensure_side_records(e, (hipStream_t)stream)) return rc;
    RolloutArgs q{};
    q.s = make_args(e);
    q.actions = actions; q.action_stride = action_stride; q.K = (uint32_t)num_steps;
    q.reward_steps = reward_steps; q.done_steps = done_steps; q.reward_sum = reward_sum; q.finished = finished;
    launch_rollout<false>(e, q, (hipStream_t)stream);
    TPL_HIP(hipGetLastError());
    count_steps(e, num_steps);
    return TPL_OK;
}

int tpl_rollout_random(tpl_env* e, uint64_t seed, uint32_t step0, int32_t num_steps, uint8_t* actions_out, float* reward_steps,
                       uint8_t* done_steps, float* reward_sum, uint32_t* finished, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (num_steps < 1) return fail_msg(TPL_ERR_ARG, "num_steps must be >= 1");
    if (int rc = check_can_advance(e)) return rc;
    DeviceGuard guard(e->device);
    if (int rc = ensure_side_records(e, (hipStream_t)stream)) return rc;
    RolloutArgs q{};
    q.s = make_args(e);
    q.actions = nullptr; q.action_stride = 0; q.K = (uint32_t)num_steps;
    q.random_seed = seed; q.step0 = step0; q.actions_out = actions_out;
    q.reward_steps = reward_steps; q.done_steps = done_steps; q.reward_sum = reward_sum; q.finished = finished;
    launch_rollout<true>(e, q, (hipStream_t)stream);
    TPL_HIP(hipGetLastError());
    count_steps(e, num_steps);
    return TPL_OK;
}

int tpl_rollout_trajectory(tpl_env* e, const uint8_t* actions, int64_t action_stride, int32_t num_steps, uint32_t* trajectory,
                           uint32_t* finished, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!actions || !trajectory) return fail_msg(TPL_ERR_ARG, "actions / trajectory is null");
    if (num_steps < 1) return fail_msg(TPL_ERR_ARG, "num_steps must be >= 1");
    if (action_stride < e->n) return fail_msg(TPL_ERR_ARG, "action_stride %lld is smaller than num_envs", (long long)action_stride);
    if (int rc = check_can_advance(e)) return rc;
    DeviceGuard guard(e->device);
    if (int rc = ensure_side_records(e, (hipStream_t)stream)) return rc;
    RolloutArgs q{};
    q.s = make_args(e);
    q.actions = actions; q.action_stride = action_stride; q.K = (uint32_t)num_steps;
    q.trajectory = trajectory; q.finished = finished;
    launch_rollout<false, true>(e, q, (hipStream_t)stream);
    TPL_HIP(hipGetLastError());
    count_steps(e, num_steps);
    return TPL_OK;
}

int tpl_rollout_random_trajectory(tpl_env* e, uint64_t seed, uint32_t step0, int32_t num_steps, uint8_t* actions_out,
                                  uint32_t* trajectory, uint32_t* finished, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!trajectory) return fail_msg(TPL_ERR_ARG, "trajectory is null");
    if (num_steps < 1) return fail_msg(TPL_ERR_ARG, "num_steps must be >= 1");
    if (int rc = check_can_advance(e)) return rc;
    DeviceGuard guard(e->device);
    if (int rc = ensure_side_records(e, (hipStream_t)stream)) return rc;
    RolloutArgs q{};
    q.s = make_args(e);
    q.K = (uint32_t)num_steps;
    q.random_seed = seed; q.step0 = step0; q.actions_out = actions_out;
    q.trajectory = trajectory; q.finished = finished;
    launch_rollout<true, true>(e, q, (hipStream_t)stream);
    TPL_HIP(hipGetLastError());
    count_steps(e, num_steps);
    return TPL_OK;
}

int tpl_decode_trajectory(tpl_env* e, const uint32_t* trajectory, int32_t num_steps, float* reward_steps, uint8_t* done_steps,
                          void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!trajectory) return fail_msg(TPL_ERR_ARG, "trajectory is null");
    if (num_steps < 1) return fail_msg(TPL_ERR_ARG, "num_steps must be >= 1");
    DeviceGuard guard(e->device);
    hipLaunchKernelGGL(decode_trajectory_kernel, dim3(blocks_for(e->n)), dim3(kBlock), 0, (hipStream_t)stream, trajectory, e->n,
                       (uint32_t)num_steps, e->r_line, e->r_win, e->r_lose, reward_steps, done_steps);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

int tpl_get_state(tpl_env* e, uint16_t* rows, uint8_t* cur, uint8_t* nxt, uint8_t* lines, uint8_t* moves,
                  uint8_t* state, uint8_t* pieces_left, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    DeviceGuard guard(e->device);
    hipLaunchKernelGGL(export_kernel, dim3(blocks_for(e->n)), dim3(kBlock), 0, (hipStream_t)stream, e->plane_a, e->plane_b,
                       e->n, (uint32_t)e->M, rows, cur, nxt, lines, moves, state, pieces_left);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

int tpl_get_board(tpl_env* e, uint8_t* cells, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!cells) return fail_msg(TPL_ERR_ARG, "cells is null");
    if (((uintptr_t)cells & 15u) != 0) return fail_msg(TPL_ERR_ARG, "cells must be 16-byte aligned");
    DeviceGuard guard(e->device);
    return launch_cells(e->plane_a, e->plane_b, e->n, cells, (hipStream_t)stream);
}

static int launch_expand(tpl_env* e, const uint4* plane_a, const uint4* plane_b, int64_t n, void* out, int32_t dtype,
                         hipStream_t stream) {
    if (dtype != TPL_F32 && dtype != TPL_BF16) return fail_msg(TPL_ERR_ARG, "unknown observation dtype %d", dtype);
    if (observe_fast_path(out)) return launch_observe(plane_a, plane_b, n, (uint32_t)e->L, (uint32_t)e->M, out, dtype, stream);
    // an output that is not 16-byte aligned (a slice of a larger tensor): element-wise stores
    const dim3 grid(blocks_for(n)), block(kBlock);
    if (dtype == TPL_F32)
        hipLaunchKernelGGL(expand_obs_kernel<float>, grid, block, 0, stream, plane_a, plane_b, n, (uint32_t)e->L,
                           (uint32_t)e->M, (float*)out);
    else if (dtype == TPL_BF16)
        hipLaunchKernelGGL(expand_obs_kernel<__hip_bfloat16>, grid, block, 0, stream, plane_a, plane_b, n, (uint32_t)e->L,
                           (uint32_t)e->M, (__hip_bfloat16*)out);
    else
        return fail_msg(TPL_ERR_ARG, "unknown observation dtype %d", dtype);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

int tpl_expand_obs(tpl_env* e, void* out, int32_t dtype, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!out) return fail_msg(TPL_ERR_ARG, "out is null");
    DeviceGuard guard(e->device);
    return launch_expand(e, e->plane_a, e->plane_b, e->n, out, dtype, (hipStream_t)stream);
}

int tpl_expand_states(tpl_env* e, const void* states_a, const void* states_b, int64_t count, void* out, int32_t dtype,
                      void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!states_a || !states_b || !out) return fail_msg(TPL_ERR_ARG, "states/out is null");
    if (count < 1) return fail_msg(TPL_ERR_ARG, "count must be positive");
    DeviceGuard guard(e->device);
    return launch_expand(e, (const uint4*)states_a, (const uint4*)states_b, count, out, dtype, (hipStream_t)stream);
}

int tpl_decode_actions(tpl_env* e, const void* logits, int32_t dtype, uint8_t* action, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!logits || !action) return fail_msg(TPL_ERR_ARG, "logits/action is null");
    DeviceGuard guard(e->device);
    const dim3 grid(blocks_for(e->n)), block(kBlock);
    if (dtype == TPL_F32)
        hipLaunchKernelGGL(decode_actions_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)logits, e->n, action);
    else if (dtype == TPL_BF16)
        hipLaunchKernelGGL(decode_actions_kernel<__hip_bfloat16>, grid, block, 0, (hipStream_t)stream,
                           (const __hip_bfloat16*)logits, e->n, action);
    else
        return fail_msg(TPL_ERR_ARG, "unknown logits dtype %d", dtype);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

int tpl_get_stats(tpl_env* e, uint64_t* out, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!out) return fail_msg(TPL_ERR_ARG, "out is null");
    DeviceGuard guard(e->device);
    hipLaunchKernelGGL(reduce_stats_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, e->stats, (unsigned long long*)out);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

int tpl_shape_info(int32_t piece, int32_t rotations, int32_t* h, int32_t* w, uint8_t* masks, uint8_t* revtopo) {
    if (piece < 0 || piece > 6 || rotations < 0) return fail_msg(TPL_ERR_ARG, "piece %d / rotations %d out of range", piece, rotations);
    const ShapeWord sh = kShapeTableHost[piece * 4 + (rotations & 3)];
    const int ww = (int)((sh.x >> 16) & 7u), hh = (int)((sh.x >> 19) & 7u);
    if (h) *h = hh;
    if (w) *w = ww;
    for (int k = 0; k < 4; ++k) {
        if (masks) {
            uint32_t m = 0;
            for (int c = 0; c < 4; ++c) m |= ((sh.x >> (4 * c + k)) & 1u) << c;
            masks[k] = (uint8_t)m;
        }
        if (revtopo) revtopo[k] = k < ww ? (uint8_t)(3u - ((sh.y >> (8 * k)) & 0xFFu)) : 0;
    }
    return TPL_OK;
}

int tpl_state_ptrs(tpl_env* e, void** plane_a, void** plane_b) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (plane_a) *plane_a = e->plane_a;
    if (plane_b) *plane_b = e->plane_b;
    return TPL_OK;
}

int tpl_clock_ptr(tpl_env* e, void** clock, int64_t* count) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (clock) *clock = e->clock;
    if (count) *count = (e->n + kClockGroup - 1) / kClockGroup;
    return TPL_OK;
}

int tpl_set_tuning(tpl_env* e, int32_t boards_per_lane, int32_t block_threads) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (boards_per_lane != 1 && boards_per_lane != 2 && boards_per_lane != 4) return fail_msg(TPL_ERR_ARG, "boards_per_lane must be 1, 2 or 4");
    if (block_threads != 64 && block_threads != 128 && block_threads != 256 && block_threads != 512)
        return fail_msg(TPL_ERR_ARG, "block_threads must be 64, 128, 256 or 512");
    e->boards_per_lane = boards_per_lane;
    e->block_threads = block_threads;
    return TPL_OK;
}


int tpl_synth_configs(tpl_env* e, uint64_t seed, int64_t first, int64_t count, uint16_t* rows, uint8_t* pieces,
                      void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (count <= 0) return fail_msg(TPL_ERR_ARG, "count must be positive");
    DeviceGuard guard(e->device);
    hipLaunchKernelGGL(synth_configs_kernel, dim3(blocks_for(count)), dim3(kBlock), 0, (hipStream_t)stream, seed, first,
                       count, (uint32_t)e->L, (uint32_t)e->M, rows, pieces);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

int tpl_synth_actions(tpl_env* e, uint64_t seed, int64_t first, int64_t count, uint64_t step, uint8_t* action,
                      void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (count <= 0 || !action) return fail_msg(TPL_ERR_ARG, "bad count/action");
    DeviceGuard guard(e->device);
    hipLaunchKernelGGL(synth_actions_kernel, dim3(blocks_for(count)), dim3(kBlock), 0, (hipStream_t)stream, seed, first,
                       count, step, action);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

#ifdef TPL_DIAG_CLOCK
// diagnostic build only: where the step kernel writes its per-wave stamps (6 x uint64 per wave), or null
int tpl_dev_set_step_diag(tpl_env* e, void* diag) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    e->step_diag = (unsigned long long*)diag;
    return TPL_OK;
}
#endif

}  // extern "C"
