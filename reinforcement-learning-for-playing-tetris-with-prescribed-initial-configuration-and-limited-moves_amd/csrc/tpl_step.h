// tpl_step.h -- device-side pieces shared by the kernels that advance boards (step, rollout, actor rollout).
#pragma once

#include "tpl_internal.h"

namespace tpl {

struct StepArgs {
    uint4* plane_a;
    uint4* plane_b;
    unsigned long long* clock; // step clocks, one per kClockGroup boards (next to the planes: the kernels ask for it first)
    int64_t n;
    uint32_t L, M;
    const void* act0;          // action, or rot
    const void* act1;          // loc (move form) or null (action form)
    uint32_t int_shift;        // 0, 2 or 3: log2 of the width of the little-endian integers act0/act1 point to
    float* reward;
    uint8_t* done;
    uint8_t* cleared;
    void* obs;                 // step-and-observe form: the [n][217] observation (float or bf16 by the kernel's type), else null
    float r_line, r_win, r_lose;
    // configuration pools (auto-reset, window refills): new episodes start from pool[cur_slot], a running board
    // refills its window from the slot it carries
    const uint8_t* pool[2];
    const uint8_t* side[2];    // the pools' side records (unpacked boards, for the multi-step kernel's resets)
    uint32_t n_cfg[2];
    uint32_t offset_mod[2];    // global_offset mod n_cfg
    uint32_t stride_shift;     // records are 1 << stride_shift bytes apart
    uint32_t cur_slot;
    int32_t assign_mode;
    uint32_t seed_mix;         // assign_seed(seed)
    int64_t global_offset;
    unsigned long long* stats;
#ifdef TPL_DIAG_CLOCK
    unsigned long long* diag;  // diagnostic build: per-wave clock stamps of the step kernel
#endif
};

// host: the arguments every board-advancing kernel takes from the handle
inline StepArgs make_args(const tpl_env* e) {
    StepArgs a{};
    a.plane_a = e->plane_a; a.plane_b = e->plane_b;
    a.n = e->n; a.L = (uint32_t)e->L; a.M = (uint32_t)e->M;
    a.r_line = e->r_line; a.r_win = e->r_win; a.r_lose = e->r_lose;
    for (int k = 0; k < 2; ++k) {
        a.pool[k] = e->pool[k].rec;
        a.side[k] = e->pool[k].side;
        a.n_cfg[k] = (uint32_t)e->pool[k].n_cfg;
        a.offset_mod[k] = e->pool[k].n_cfg ? (uint32_t)((uint64_t)e->global_offset % (uint64_t)e->pool[k].n_cfg) : 0u;
    }
    a.stride_shift = e->stride_shift; a.cur_slot = (uint32_t)e->cur_slot;
    a.assign_mode = e->assign_mode; a.seed_mix = assign_seed(e->seed);
    a.global_offset = e->global_offset; a.stats = e->stats; a.clock = e->clock;
#ifdef TPL_DIAG_CLOCK
    a.diag = e->step_diag;
#endif
    return a;
}

// Element i of an array of 1-, 4- or 8-byte little-endian integers (1 << shift bytes each), as its low 32 bits,
// without a branch on the width: the aligned word that holds the element's first byte is read and shifted.  (For
// one-byte elements that word may reach up to three bytes past the end of the array -- never past the 4-byte unit
// the last element is in.)
__device__ __forceinline__ uint32_t load_int(const void* p, uint32_t shift, int64_t i) {
    const char* q = (const char*)p + (i << shift);
    const uint32_t skew = (uint32_t)((uintptr_t)q & 3u);
    const uint32_t w = *(const uint32_t*)(q - skew);
    const uint32_t v = w >> (8u * skew);
    return shift == 0u ? (v & 0xFFu) : v;
}

// action = rot * 10 + loc, exact for any 32-bit action, without an integer multiply beyond the one multiply-high
__device__ __forceinline__ void split_action(uint32_t action, uint32_t& rot, uint32_t& loc) {
    rot = __umulhi(action, 0xCCCCCCCDu) >> 3;
    loc = action - ((rot << 3) + (rot << 1));
}

// A pool record's address.  `slot` is a per-lane value (0 / 1): the two bases are selected, not indexed.
__device__ __forceinline__ const uint8_t* pool_record(const StepArgs& p, uint32_t slot, uint32_t cfg) {
    const uint8_t* base = slot ? p.pool[1] : p.pool[0];
    return base + ((size_t)cfg << p.stride_shift);
}

// pool entry of the episode of board i that begins at step `birth`, in pool buffer `slot`
__device__ __forceinline__ uint32_t config_of(const StepArgs& p, uint32_t i, uint64_t birth, uint32_t slot) {
    const uint32_t n_cfg = slot ? p.n_cfg[1] : p.n_cfg[0];
    const uint32_t offset_mod = slot ? p.offset_mod[1] : p.offset_mod[0];
    return n_cfg ? assign_config(p.global_offset, offset_mod, i, birth, p.seed_mix, n_cfg, p.assign_mode) : 0u;
}

// piece word `w` >= 1 of a record (the window refill)
__device__ __forceinline__ uint64_t piece_word_at(const uint8_t* rec, uint32_t w) {
    return *(const uint64_t*)(rec + 32u + 8u * (w - 1u));
}

// the same for an action below 1029 (a uint8 action, a policy's action): one 24-bit multiply and a shift instead of a
// multiply-high by a 32-bit reciprocal and its correction
__device__ __forceinline__ void split_small_action(uint32_t action, uint32_t& rot, uint32_t& loc) {
    rot = __umul24(action, 205u) >> 11;
    loc = action - __umul24(rot, 10u);
}

// Epsilon-greedy exploration (tpl_explore_actions): with probability eps_q24 / 2^24 the action is replaced by a uniform
// draw from [0, 40).  With eps_q24 = 2^24 it is the uniform random policy on the device (tpl_rollout_random).
//
// The draws come in PAIRS: steps 2j and 2j + 1 of a board share ONE 32-bit hash of (seed, global board index, j) and take
// sixteen bits of it each, reduced to [0, 40) by a 24-bit multiply: (half * 40) >> 16, every action within 40 / 65536 of
// 1/40.  (Round 2 spent a hash and a multiply-high on every step -- five multiplies and the shifts and xors between them, a fifth of
// the instructions of a wave-step of the multi-step kernel; this is one hash every other step -- and still a function of the step index
// alone, so any step can be drawn without the ones before it.)  The decision at epsilon < 1 is a hash of its own, of
// the pair word and the step's parity: it costs nothing when every action is replaced.
__device__ __forceinline__ uint32_t explore_base(uint64_t seed, uint64_t gidx) {
    return fmix32((uint32_t)gidx ^ ((uint32_t)(gidx >> 32) * 0x9E3779B9u) ^ (uint32_t)seed ^ 0x51ED270Bu);
}
__device__ __forceinline__ uint32_t explore_pair(uint32_t base, uint64_t seed, uint32_t step) {
    return fmix32(base + (step >> 1) * 0x9E3779B1u + (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t explore_pick(uint32_t pair, uint32_t step) {
    return __umul24((pair >> (16u * (step & 1u))) & 0xFFFFu, 40u) >> 16;
}
__device__ __forceinline__ uint32_t explore(uint32_t action, uint64_t seed, uint64_t gidx, uint32_t step, uint32_t eps_q24) {
    const uint32_t pair = explore_pair(explore_base(seed, gidx), seed, step);
    const uint32_t replacement = explore_pick(pair, step);
    if (eps_q24 >= (1u << 24)) return replacement;
    const uint32_t decision = fmix32(pair ^ (0x2545F491u + (step & 1u))) >> 8;
    return decision < eps_q24 ? replacement : action;
}

// (re)initialise a board from pool entry `cfg` of the current slot.  reset()/load_warm_reset() (:438-449), with the
// counters zeroed; the record's two state words are one 32-B read.
__device__ __forceinline__ void load_config(const StepArgs& p, uint32_t cfg, uint4& A, uint4& B) {
    const uint4* rec = (const uint4*)pool_record(p, p.cur_slot, cfg);
    A = rec[0];
    const uint4 pb = rec[1];
    B = make_uint4(pb.x, pb.y | (p.cur_slot << 30), pb.z, pb.w);     // the record carries slot 0: stamp the current one
}

// reward = per_line * rows_cleared (+ win when the move wins) (+ lose when the move loses): one rounded multiply,
// then at most one rounded add.  Contraction is switched off for this function: left alone the compiler fuses the
// pair into an FMA, whose single rounding differs from the CPU's two when per_line * 3 is not exact (0.1f * 3 - 0.3f
// is 0 in two roundings and -7.45e-9 fused).  (`__fmul_rn` / `__fadd_rn` do NOT prevent it: in this ROCm's headers they
// are a plain `*` and `+`; round 2 relied on them and shipped the FMA -- tests/test_gpu_parity.py now has the case.)
__device__ __forceinline__ float step_reward(const StepArgs& p, uint32_t n_clear, uint32_t state) {
#pragma clang fp contract(off)
    float reward = p.r_line * (float)n_clear;
    if (state == ST_WON) reward = reward + p.r_win;
    if (state >= ST_LOST_LIMIT) reward = reward + p.r_lose;
    return reward;
}

// episodes a lane finished, accumulated in registers across the steps of one launch
struct Tally { uint32_t episodes = 0, lines = 0, wins = 0, topouts = 0; };
// The multi-step kernel's form: top-outs are what is left of the episodes once the wins and the losses at the move limit
// are taken off -- those two are the rare ways to finish, and each is counted under a branch that the whole wave skips
// when no lane of it finished that way.
struct RareTally { uint32_t episodes = 0, lines = 0, wins = 0, limits = 0; };

// The pool record of the board's current episode at step `clock` (kept in registers by the multi-step kernels: the
// window refill needs it every tenth move, and hashing the entry again each time costs more than the refill itself).
__device__ __forceinline__ const uint8_t* current_record(const Board& s, const StepArgs& p, uint32_t i, uint64_t clock) {
    return pool_record(p, s.slot, config_of(p, i, clock - s.moves, s.slot));
}

// pool entry of the board's current episode at step `clock` (the actor kernel keeps this index, one register, where the
// K-step kernel keeps the record's address: its registers are the scarcer)
__device__ __forceinline__ uint32_t current_config(const Board& s, const StepArgs& p, uint32_t i, uint64_t clock) {
    return config_of(p, i, clock - s.moves, s.slot);
}

// One step of one unpacked board held in registers: Tetris.move (:354-422) + the window pop/refill + the
// build's freeze / auto-reset rules + reward.  `cfg` = current_config() of the board, updated on a reset; `clock` = the
// index of this step (the group's step clock on entry + the steps already done in this launch).  Returns done (state !=
// running after the move, before a reset).
template <bool kAutoReset>
__device__ __forceinline__ bool advance_board(Board& s, uint32_t& cfg, uint32_t rot, uint32_t loc, const StepArgs& p,
                                              uint32_t i, uint64_t clock, const ShapeWord* shape, float& reward, Tally& tally) {
    reward = 0.0f;
    if (s.state != ST_RUNNING) return true;      // frozen
    // pieces.pop(0) (:356) moves the cursor to moves_used + 1 whatever the move does; at a multiple of ten the
    // window is down to its last two entries and piece word cursor/10 replaces it (gather issued before the move)
    const uint32_t tenth = tenths(s.moves + 1u);
    const bool refill = window_runs_out(tenth) && (p.n_cfg[0] | p.n_cfg[1]) != 0u;
    uint64_t word = 0;
    if (refill) word = piece_word_at(pool_record(p, s.slot, cfg), window_word(tenth));
    bool topout;
    const uint32_t n_clear = move_board(s, shape, rot, loc, p.L, p.M, topout);
    next_window(s, refill, word);
    reward = step_reward(p, n_clear, s.state);
    const bool done = s.state != ST_RUNNING;
    if (done) {
        tally.episodes += 1u;
        tally.lines += s.lines;
        tally.wins += s.state == ST_WON ? 1u : 0u;
        tally.topouts += s.state == ST_LOST_TOPOUT ? 1u : 0u;
        if (kAutoReset) {
            // the new episode's first move is the next step; it starts from the current pool buffer
            cfg = config_of(p, i, clock + 1u, p.cur_slot);
            uint4 A2, B2;
            load_config(p, cfg, A2, B2);
            unpack_board(A2, B2, s);
        }
    }
    return done;
}

// advance_board with the board's column words in LDS (move_board_lds), written for the multi-step kernel's loop:
//   * how the move ended stays in flags (lane masks), not in a state word;
//   * `until_refill` counts the moves left before the piece window runs out and `next_word` points at the piece word
//     that will replace it: the test on the way in is one compare, the refill one load and one add;
//   * the reward's conditional add happens where only the finished boards are active: one select between the two
//     constants and one add (a finished board has won or lost, never both: the same single rounded add as step_reward);
//   * `code` is the step's trajectory byte (trajectory_code below); a caller that uses only one of `reward` / `code` does not
//     pay for the other (the function is inlined, the unused one's arithmetic is dropped).
// The compact trajectory: what happened to a board in one step, in one byte -- rows cleared (bits 0-2), how the move ended
// (bits 3-4: 0 the game goes on, 1 won, 2 lost at the move limit, 3 topped out), whether the board was re-initialised from
// the pool in this step (bit 5), whether it was frozen, i.e. had finished earlier and is not auto-reset (bit 6: no move was
// made, reward 0, done).  reward and done follow from it and the handle's reward parameters (decode_trajectory_kernel).
constexpr uint32_t kTrajWon = 1u << 3, kTrajLimit = 2u << 3, kTrajTopout = 3u << 3, kTrajReset = 1u << 5, kTrajFrozen = 1u << 6;

template <bool kAutoReset>
__device__ __forceinline__ bool advance_board_lds(Board& s, uint32_t* cols, const uint8_t*& next_word, uint32_t& until_refill,
                                                  uint32_t rot, uint32_t loc, const StepArgs& p, uint32_t i, uint64_t clock,
                                                  const ShapeWord* shape, float& reward, uint32_t& code, RareTally& tally) {
    reward = 0.0f;
    code = kTrajFrozen;
    if (s.state != ST_RUNNING) return true;      // frozen
    // pieces.pop(0) (:356) moves the cursor on by one whatever the move does; when it reaches a multiple of ten the
    // window is down to its last two entries and the next piece word replaces it
    until_refill -= 1u;
    const bool refill = until_refill == 0u && (p.n_cfg[0] | p.n_cfg[1]) != 0u;
    // (read only where `refill` is set: any value will do elsewhere -- initialising it costs two instructions a step.  "Any
    // value" is said to the compiler in so many words: an unspecified but valid value, not an uninitialised read)
    uint64_t word = __builtin_nondeterministic_value(word);
    if (refill) {
        word = *(const uint64_t*)next_word;
        next_word += 8;
        until_refill = (uint32_t)kWindowStride;
    }
    // A board that tops out (nineteen finishes in twenty under random play) is known to have finished as soon as its drop
    // is: its next configuration's side record is sent for THEN, and travels under the rest of the move.
    MoveEnd end;
    // (each of these is written on the one path that reads it: left to any value, not zeroed six times a step)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 s0, s1, s2;
    uint32_t cfg;
    const auto send_for_next = [&]() {
        // the new episode's first move is the next step; the board comes from the side record, already unpacked
        cfg = config_of(p, i, clock + 1u, p.cur_slot);
        const u32x4* side = (const u32x4*)((p.cur_slot ? p.side[1] : p.side[0]) + ((size_t)cfg << kSideShift));
        s0 = side[0]; s1 = side[1]; s2 = side[2];
    };
    const uint32_t n_clear = move_board_lds(s, cols, shape, rot, loc, p.L, p.M, end, [&](bool topout) {
        if (kAutoReset && topout) send_for_next();
    });
    next_window(s, refill, word);
    const bool done = end.topout || end.won || end.limit;
    if (!kAutoReset) s.state = end.topout ? ST_LOST_TOPOUT : end.won ? ST_WON : end.limit ? ST_LOST_LIMIT : ST_RUNNING;
    const float base_reward = p.r_line * (float)n_clear;
    reward = base_reward;
    code = n_clear;
    if (done) {
#pragma clang fp contract(off)
        tally.episodes += 1u;
        tally.lines += s.lines;
        // A finished board has lost unless it has won: the common add is the loss's, and the two rare ways to finish sit
        // under branches that a wave skips (on its lane mask) when none of its boards finished that way.  The empty asm
        // statements keep them branches: turned into selects they would cost every wave-step two instructions each.
        reward = base_reward + p.r_lose;
        code = n_clear + (kTrajTopout | (kAutoReset ? kTrajReset : 0u));
        if (end.won) {
            asm volatile("");
            reward = base_reward + p.r_win;
            code = n_clear + (kTrajWon | (kAutoReset ? kTrajReset : 0u));
            tally.wins += 1u;
        }
        if (end.limit) {
            asm volatile("");
            code = n_clear + (kTrajLimit | (kAutoReset ? kTrajReset : 0u));
            tally.limits += 1u;
        }
        if (kAutoReset) {
            if (!end.topout) {                    // won, or lost at the move limit: known only now
                asm volatile("");
                send_for_next();
            }
            next_word = ((p.cur_slot ? p.pool[1] : p.pool[0]) + 32) + ((size_t)cfg << p.stride_shift);   // piece word 1 of the record
            const uint32_t c[kCols] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y};
            lds_store_cols(cols, c);
            s.window = s2.z; s.window_hi = s2.w;
            s.moves = 0; s.lines = 0; s.slot = p.cur_slot;
            until_refill = (uint32_t)kWindowStride;
        }
    }
    return done;
}

// For a board at `moves` moves (cursor = moves; a board that topped out is finished and never asks): the moves left before
// its window runs out, 10 - cursor % 10, and the piece word that will then replace it, word cursor / 10 + 1 of its record
// (words 1.. sit behind the record's 32 state bytes).
__device__ __forceinline__ uint32_t moves_until_refill(uint32_t moves) {
    return (uint32_t)kWindowStride - (moves - window_word(tenths(moves)) * (uint32_t)kWindowStride);
}
__device__ __forceinline__ const uint8_t* next_piece_word(const uint8_t* rec, uint32_t moves) {
    return rec + 32u + 8u * window_word(tenths(moves));
}

// block-level flush of the lanes' tallies: LDS atomics, then one sharded 64-bit global atomic per counter.
// Every thread of the block must call it (it contains a barrier); s_stat[4] must have been zeroed before.
__device__ __forceinline__ void flush_tally(const Tally& t, uint32_t* s_stat, unsigned long long* stats) {
    if (t.episodes) {
        atomicAdd(&s_stat[0], t.episodes);
        if (t.lines) atomicAdd(&s_stat[1], t.lines);
        if (t.wins) atomicAdd(&s_stat[2], t.wins);
        if (t.topouts) atomicAdd(&s_stat[3], t.topouts);
    }
    if (__syncthreads_or(t.episodes ? 1 : 0)) {
        if (threadIdx.x < 4) {
            const uint32_t v = s_stat[threadIdx.x];
            if (v) atomicAdd(&stats[(size_t)(blockIdx.x % kStatShards) * kStatStride + threadIdx.x], (unsigned long long)v);
        }
    }
}

// the multi-step kernel's tally: top-outs = episodes - wins - losses at the move limit
__device__ __forceinline__ void flush_tally(const RareTally& r, uint32_t* s_stat, unsigned long long* stats) {
    Tally t;
    t.episodes = r.episodes; t.lines = r.lines; t.wins = r.wins; t.topouts = r.episodes - r.wins - r.limits;
    flush_tally(t, s_stat, stats);
}

}  // namespace tpl
