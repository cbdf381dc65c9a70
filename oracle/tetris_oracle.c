/*
 * tetris_oracle.c -- CPU ORACLE for the Tetris-piclim board step.  TEST INFRASTRUCTURE ONLY
 * (see tetris_oracle.h for who may use it).  Every function cites the reference lines it restates;
 * citations are relative to the upstream repo root (game/tetris.py unless another file is named).
 *
 * Pinned by the tests/golden/ fixtures, which were generated from the imported reference.
 */
#include "tetris_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * Shape table: `tetrominos` (game/tetris.py:23-57).  Each entry is (mask rows top->bottom,
 * reverse_topography).  Row masks use bit x = mask column x, so the reference's
 * ((False, False, True), (True, True, True)) for L rot 0 is {0b100, 0b111} = {4, 7}.
 * Piece ids follow piece_translations (game/tetris.py:8-16): I0 L1 J2 T3 S4 Z5 O6.
 * ---------------------------------------------------------------------------------------------- */
static const int N_ROT[7] = {2, 4, 4, 4, 2, 2, 1};

static const to_shape SHAPES[7][4] = {
    /* I  :24-27 */ {{1, 4, {15, 0, 0, 0}, {0, 0, 0, 0}}, {4, 1, {1, 1, 1, 1}, {3, 0, 0, 0}}, {0}, {0}},
    /* L  :28-33 */ {{2, 3, {4, 7, 0, 0}, {1, 1, 1, 0}}, {3, 2, {3, 2, 2, 0}, {0, 2, 0, 0}},
                     {2, 3, {7, 1, 0, 0}, {1, 0, 0, 0}}, {3, 2, {1, 1, 3, 0}, {2, 2, 0, 0}}},
    /* J  :34-39 */ {{2, 3, {1, 7, 0, 0}, {1, 1, 1, 0}}, {3, 2, {2, 2, 3, 0}, {2, 2, 0, 0}},
                     {2, 3, {7, 4, 0, 0}, {0, 0, 1, 0}}, {3, 2, {3, 1, 1, 0}, {2, 0, 0, 0}}},
    /* T  :40-45 */ {{2, 3, {2, 7, 0, 0}, {1, 1, 1, 0}}, {3, 2, {2, 3, 2, 0}, {1, 2, 0, 0}},
                     {2, 3, {7, 2, 0, 0}, {0, 1, 0, 0}}, {3, 2, {1, 3, 1, 0}, {2, 1, 0, 0}}},
    /* S  :46-49 */ {{2, 3, {6, 3, 0, 0}, {1, 1, 0, 0}}, {3, 2, {1, 3, 2, 0}, {1, 2, 0, 0}}, {0}, {0}},
    /* Z  :50-53 */ {{2, 3, {3, 6, 0, 0}, {0, 1, 1, 0}}, {3, 2, {2, 3, 1, 0}, {2, 1, 0, 0}}, {0}, {0}},
    /* O  :54-56 */ {{2, 2, {3, 3, 0, 0}, {1, 1, 0, 0}}, {0}, {0}, {0}},
};

int to_num_rotations(int piece) { return N_ROT[piece]; }

/* get_tetromino (game/tetris.py:60-61): tetrominos[piece][rotations % len(tetrominos[piece])] */
void to_get_tetromino(int piece, int rotations, to_shape* out) {
    *out = SHAPES[piece][rotations % N_ROT[piece]];
}

/* ------------------------------------------------------------------------------------------------
 * Tetris.move (game/tetris.py:354-422) with calculate_drop_deltas (:427-433) and calculate_drop
 * (:424-425).  Same operation order as the reference.
 * ---------------------------------------------------------------------------------------------- */
int to_move(to_game* g, const uint8_t* pieces, int L, int M, int rotations, int location) {
    /* :356  piece = self.pieces.pop(0) -- consumed before anything can fail */
    int piece = pieces[g->cursor];
    g->cursor += 1;

    /* :359-360 */
    to_shape s;
    to_get_tetromino(piece, rotations, &s);

    /* :363-364  right clamp only */
    if (location > TO_COLS - s.w) location = TO_COLS - s.w;

    /* :427-433  board topography under the piece's columns, minus the piece's reverse topography */
    int min_delta = 1 << 20;
    for (int c = 0; c < s.w; ++c) {
        int top = TO_ROWS; /* :431  20 when the column is empty */
        for (int r = 0; r < TO_ROWS; ++r) {
            if ((g->rows[r] >> (location + c)) & 1u) { top = r; break; }
        }
        int delta = top - (int)s.revtopo[c];
        if (delta < min_delta) min_delta = delta;
    }
    /* :424-425 */
    int drop = min_delta - 1;

    /* :372-374  top-out: state False, nothing else changes (moves_used is NOT incremented) */
    if (drop < 0) {
        g->state = TO_LOST;
        return -1;
    }

    /* :377-379  lock */
    for (int i = 0; i < s.h; ++i) g->rows[drop + i] |= (uint16_t)((uint16_t)s.mask[i] << location);
    g->moves_used += 1;

    /* :382-386  only the piece's rows are tested */
    int full[4] = {0, 0, 0, 0};
    int rows_cleared = 0;
    for (int i = 0; i < s.h; ++i) {
        full[i] = (g->rows[drop + i] == TO_FULL_ROW);
        rows_cleared += full[i];
    }

    /* :389-394 */
    if (rows_cleared == 0) {
        if (g->moves_used >= M) g->state = TO_LOST;
        return 0;
    }

    /* :397-407  keep the rows that are not cleared, in order, under `rows_cleared` empty rows */
    uint16_t kept[TO_ROWS];
    int nk = 0;
    for (int r = 0; r < TO_ROWS; ++r) {
        int cleared = (r >= drop && r < drop + s.h && full[r - drop]);
        if (!cleared) kept[nk++] = g->rows[r];
    }
    for (int r = 0; r < rows_cleared; ++r) g->rows[r] = 0;
    for (int r = 0; r < nk; ++r) g->rows[rows_cleared + r] = kept[r];

    /* :409 */
    g->lines_cleared += rows_cleared;

    /* :415-417  win is tested before the move limit */
    if (g->lines_cleared >= L) {
        g->state = TO_WON;
        return rows_cleared;
    }
    /* :420-422 */
    if (g->moves_used >= M) g->state = TO_LOST;
    return rows_cleared;
}

/* ------------------------------------------------------------------------------------------------
 * Counter-based generator for the synthetic workload.  splitmix64 finaliser chained
 * over (seed, stream, index, counter).  The HIP library implements the same function.
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t sm64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

uint64_t to_rng(uint64_t seed, uint64_t stream, uint64_t index, uint64_t counter) {
    uint64_t h = sm64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
    h = sm64(h ^ index);
    return sm64(h ^ counter);
}

enum { STREAM_BOARD = 0, STREAM_PIECES = 1, STREAM_ACTION = 2, STREAM_ASSIGN = 3 };

/* rows 20-L..19: ten Bernoulli(1/2) cells; a row that comes out full has one hashed cell cleared */
void to_synth_boards(uint64_t seed, int64_t first, int64_t count, int L, uint16_t* rows) {
    int filled = L < TO_ROWS ? L : TO_ROWS;
    for (int64_t b = 0; b < count; ++b) {
        uint16_t* out = rows + b * TO_ROWS;
        for (int r = 0; r < TO_ROWS; ++r) {
            uint16_t v = 0;
            if (r >= TO_ROWS - filled) {
                uint64_t h = to_rng(seed, STREAM_BOARD, (uint64_t)(first + b), (uint64_t)r);
                v = (uint16_t)(h & TO_FULL_ROW);
                if (v == TO_FULL_ROW) v &= (uint16_t)~(1u << ((h >> 10) % 10u));
            }
            out[r] = v;
        }
    }
}

/* ceil((M+1)/7) Fisher-Yates-shuffled 7-bags truncated to M+1 -- the statistical form of
 * RandomPieceGenerator.get_random_sequence (game/tetris.py:91-102). */
void to_synth_pieces(uint64_t seed, int64_t first, int64_t count, int M, uint8_t* pieces) {
    int len = M + 1;
    for (int64_t b = 0; b < count; ++b) {
        uint8_t* out = pieces + b * len;
        int produced = 0;
        for (int bag = 0; produced < len; ++bag) {
            uint64_t h = to_rng(seed, STREAM_PIECES, (uint64_t)(first + b), (uint64_t)bag);
            uint8_t a[7] = {0, 1, 2, 3, 4, 5, 6};
            for (int j = 6; j >= 1; --j) {
                unsigned k = (unsigned)((h >> (8 * (6 - j))) & 0xFFu) % (unsigned)(j + 1);
                uint8_t t = a[j]; a[j] = a[k]; a[k] = t;
            }
            for (int j = 0; j < 7 && produced < len; ++j) out[produced++] = a[j];
        }
    }
}

/* uniform rot 0..3, loc 0..9, keyed by (board, step); action = rot*10 + loc */
void to_synth_actions(uint64_t seed, int64_t first, int64_t count, uint64_t step, uint8_t* action) {
    for (int64_t b = 0; b < count; ++b) {
        uint64_t h = to_rng(seed, STREAM_ACTION, (uint64_t)(first + b), step);
        unsigned rot = (unsigned)(h & 3u);
        unsigned loc = (unsigned)((h >> 8) & 0xFFFFu) % 10u;
        action[b] = (uint8_t)(rot * 10u + loc);
    }
}

uint64_t to_board_hash(const uint16_t* rows) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int r = 0; r < TO_ROWS; ++r) h = (h ^ (uint64_t)rows[r]) * 0x100000001b3ULL;
    return h;
}

/* ------------------------------------------------------------------------------------------------
 * Batched environment (build-defined rules; the HIP library follows the same ones).
 * ---------------------------------------------------------------------------------------------- */
to_env* to_env_create(int64_t n, int L, int M, int64_t global_offset, uint64_t seed) {
    to_env* e = (to_env*)calloc(1, sizeof(to_env));
    e->n = n; e->L = L; e->M = M; e->global_offset = global_offset; e->seed = seed;
    e->auto_reset = 0; e->assign_mode = 0;
    e->reward_per_line = 1.0f; e->reward_win = 0.0f; e->reward_lose = 0.0f;
    e->games = (to_game*)calloc((size_t)n, sizeof(to_game));
    e->pieces = (uint8_t*)calloc((size_t)n * (size_t)(M + 1), 1);
    e->birth = (uint64_t*)calloc((size_t)n, sizeof(uint64_t));
    return e;
}

void to_env_destroy(to_env* e) {
    if (!e) return;
    free(e->games); free(e->pieces); free(e->birth); free(e);
}

/* the first pool goes into the current slot; a later one into the other slot, which becomes current (running boards
 * keep the piece lists they copied at reset) */
void to_env_set_pool(to_env* e, const uint16_t* rows, const uint8_t* pieces, int64_t n_cfg) {
    if (e->n_cfg[e->cur_slot] != 0) e->cur_slot ^= 1;
    e->pool_rows[e->cur_slot] = rows; e->pool_pieces[e->cur_slot] = pieces; e->n_cfg[e->cur_slot] = n_cfg;
}

void to_env_set_options(to_env* e, int auto_reset, int assign_mode, float per_line, float win, float lose) {
    e->auto_reset = auto_reset; e->assign_mode = assign_mode;
    e->reward_per_line = per_line; e->reward_win = win; e->reward_lose = lose;
}

/* Which pool entry a board's episode starts from.  32-bit mixing (murmur3 finaliser) so the device
 * pays a handful of integer ops per reset; the range reduction is a multiply-high, not a modulo. */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

/* hash mode: (g, birth, seed) folded into 32 bits -- for one board the word is birth * odd + const, so distinct births
 * give distinct words and the sequence of entries has no period -- then one finaliser round and a multiply-high.
 * sequential mode: (g + birth) mod n_cfg on the low 32 bits of birth. */
int64_t to_env_assign(const to_env* e, int64_t board, uint64_t birth) {
    uint64_t g = (uint64_t)(e->global_offset + board);
    uint64_t n_cfg = (uint64_t)e->n_cfg[e->cur_slot];
    if (e->assign_mode == 1) return (int64_t)((g % n_cfg + (uint64_t)((uint32_t)birth % (uint32_t)n_cfg)) % n_cfg);
    /* 24-bit multiplies take the low 24 bits of each factor and keep the low 32 bits of the product */
    uint32_t x = ((uint32_t)g + (uint32_t)birth * 0x9E3779B1u) ^ (((uint32_t)(g >> 32) & 0xFFFFFFu) * 0x85EBCBu)
                 ^ (((uint32_t)(birth >> 32) & 0xFFFFFFu) * 0xC2B2AFu) ^ (uint32_t)(sm64(e->seed) >> 32);
    uint32_t h = fmix32(x);
    return (int64_t)(((uint64_t)h * (uint64_t)(uint32_t)n_cfg) >> 32);
}

/* Epsilon-greedy exploration -- the build's own definition, restated here so that the device's
 * draws can be checked against something that does not share its code.  Steps 2j and 2j + 1 of a board share one
 * 32-bit word w = fmix32(base + j * 0x9E3779B1 + (seed >> 32)), base = fmix32(low(g) ^ high(g) * 0x9E3779B9 ^ low(seed)
 * ^ 0x51ED270B); the even step takes its low sixteen bits, the odd step its high sixteen, and the replacement action is
 * (those sixteen bits * 40) >> 16.  With epsilon = eps_q24 / 2^24 < 1 the action is replaced iff the top 24 bits of
 * fmix32(w ^ (0x2545F491 + step parity)) are below eps_q24. */
void to_explore_actions(uint8_t* action, int64_t n, int64_t global_offset, uint64_t seed, uint32_t step, uint32_t eps_q24) {
    for (int64_t b = 0; b < n; ++b) {
        const uint64_t g = (uint64_t)(global_offset + b);
        const uint32_t base = fmix32((uint32_t)g ^ ((uint32_t)(g >> 32) * 0x9E3779B9u) ^ (uint32_t)seed ^ 0x51ED270Bu);
        const uint32_t w = fmix32(base + (step / 2u) * 0x9E3779B1u + (uint32_t)(seed >> 32));
        const uint32_t sixteen = (step % 2u) ? (w >> 16) : (w & 0xFFFFu);
        const uint32_t replacement = (sixteen * 40u) >> 16;
        if (eps_q24 < (1u << 24)) {
            const uint32_t decision = fmix32(w ^ (0x2545F491u + (step % 2u))) >> 8;
            if (decision >= eps_q24) continue;
        }
        action[b] = (uint8_t)replacement;
    }
}

uint64_t to_env_clock(const to_env* e) { return e->clock; }
uint64_t to_env_birth(const to_env* e, int64_t board) { return e->birth[board]; }

/* reset()/load_warm_reset() (game/tetris.py:438-449): clear, then take the next (board, pieces) -- the pool entry that
 * belongs to an episode of this board beginning at step `birth`.
 * Unlike the reference (which leaves lines_cleared/moves_used/state untouched -- SURVEY 3.3) the
 * counters and the state are zeroed. */
static void load_config(to_env* e, int64_t b, uint64_t birth) {
    int64_t cfg = to_env_assign(e, b, birth);
    to_game* g = &e->games[b];
    e->birth[b] = birth;
    memcpy(g->rows, e->pool_rows[e->cur_slot] + cfg * TO_ROWS, sizeof(g->rows));
    memcpy(e->pieces + b * (e->M + 1), e->pool_pieces[e->cur_slot] + cfg * (e->M + 1), (size_t)(e->M + 1));
    g->lines_cleared = 0; g->moves_used = 0; g->state = TO_RUNNING; g->cursor = 0;
}

void to_env_reset(to_env* e, const uint8_t* mask) {
    /* a full reset() starts the step count over; a masked reset starts the masked boards' next episode at the next step */
    if (!mask) e->clock = 0;
    for (int64_t b = 0; b < e->n; ++b) {
        if (mask && !mask[b]) continue;
        load_config(e, b, e->clock);
    }
    if (!mask) { e->stat_episodes = e->stat_lines = e->stat_wins = e->stat_topouts = 0; }
}

static void env_move_one(to_env* e, int64_t b, int rot, int loc, float* reward, uint8_t* done, uint8_t* cleared) {
    to_game* g = &e->games[b];
    float r = 0.0f; int n = 0;
    if (g->state == TO_RUNNING) {
        int res = to_move(g, e->pieces + b * (e->M + 1), e->L, e->M, rot, loc);
        n = res < 0 ? 0 : res;
        r = e->reward_per_line * (float)n;
        if (g->state == TO_WON) r = r + e->reward_win;
        if (g->state == TO_LOST) r = r + e->reward_lose;
        if (g->state != TO_RUNNING) {
            e->stat_episodes += 1;
            e->stat_lines += (uint64_t)g->lines_cleared;
            e->stat_wins += (g->state == TO_WON);
            e->stat_topouts += (res < 0);
            if (done) done[b] = 1;
            if (e->auto_reset) load_config(e, b, e->clock + 1);   /* its first move is the next step */
        } else if (done) {
            done[b] = 0;
        }
    } else if (done) {
        done[b] = 1;   /* frozen */
    }
    if (reward) reward[b] = r;
    if (cleared) cleared[b] = (uint8_t)n;
}

void to_env_move(to_env* e, const uint8_t* rot, const uint8_t* loc, float* reward, uint8_t* done, uint8_t* cleared) {
    for (int64_t b = 0; b < e->n; ++b) env_move_one(e, b, rot[b], loc[b], reward, done, cleared);
    e->clock += 1;
}

/* action = rot*10 + loc (4 rotation x 10 location choices) */
void to_env_step(to_env* e, const uint8_t* action, float* reward, uint8_t* done) {
    for (int64_t b = 0; b < e->n; ++b) env_move_one(e, b, action[b] / 10, action[b] % 10, reward, done, NULL);
    e->clock += 1;
}

/* get_state (game/tetris.py:435-436), batched; a missing piece (list exhausted) reads as 7 */
void to_env_get_state(const to_env* e, uint16_t* rows, uint8_t* cur, uint8_t* nxt,
                      uint8_t* lines, uint8_t* moves, uint8_t* state, uint8_t* pieces_left) {
    int len = e->M + 1;
    for (int64_t b = 0; b < e->n; ++b) {
        const to_game* g = &e->games[b];
        const uint8_t* p = e->pieces + b * len;
        if (rows) memcpy(rows + b * TO_ROWS, g->rows, sizeof(g->rows));
        if (cur) cur[b] = g->cursor < len ? p[g->cursor] : 7;
        if (nxt) nxt[b] = g->cursor + 1 < len ? p[g->cursor + 1] : 7;
        if (lines) lines[b] = (uint8_t)g->lines_cleared;
        if (moves) moves[b] = (uint8_t)g->moves_used;
        if (state) state[b] = (uint8_t)g->state;
        if (pieces_left) pieces_left[b] = (uint8_t)(len - g->cursor);
    }
}

void to_env_expand_obs(const to_env* e, float* out) {
    int len = e->M + 1;
    for (int64_t b = 0; b < e->n; ++b) {
        const to_game* g = &e->games[b];
        const uint8_t* p = e->pieces + b * len;
        float* o = out + b * 217;
        for (int y = 0; y < TO_ROWS; ++y)
            for (int x = 0; x < TO_COLS; ++x) o[y * 10 + x] = (float)((g->rows[y] >> x) & 1u);
        int cur = g->cursor < len ? p[g->cursor] : 7;
        int nxt = g->cursor + 1 < len ? p[g->cursor + 1] : 7;
        for (int k = 0; k < 7; ++k) { o[200 + k] = (cur == k) ? 1.0f : 0.0f; o[207 + k] = (nxt == k) ? 1.0f : 0.0f; }
        o[214] = (float)(e->L - g->lines_cleared);
        o[215] = (float)(e->M - g->moves_used);
        o[216] = g->state != TO_RUNNING ? 1.0f : 0.0f;
    }
}

void to_env_get_stats(const to_env* e, uint64_t out[4]) {
    out[0] = e->stat_episodes; out[1] = e->stat_lines; out[2] = e->stat_wins; out[3] = e->stat_topouts;
}

/* ------------------------------------------------------------------------------------------------
 * Carving generator (game/tetris.py:64-137, 226-352), restated on uint16 rows.
 * ---------------------------------------------------------------------------------------------- */

/* calculate_drop_deltas (:427-433): deltas[c] = top of board column location+c (20 if empty) - revtopo[c] */
static void drop_deltas(const uint16_t* rows, int location, const to_shape* s, int* deltas) {
    for (int c = 0; c < s->w; ++c) {
        int top = TO_ROWS;
        for (int r = 0; r < TO_ROWS; ++r)
            if ((rows[r] >> (location + c)) & 1u) { top = r; break; }
        deltas[c] = top - (int)s->revtopo[c];
    }
}

static int min_delta(const int* d, int w, int* argmin) {
    int best = d[0], at = 0;
    for (int c = 1; c < w; ++c)
        if (d[c] < best) { best = d[c]; at = c; }   /* np.argmin: first index of the minimum */
    if (argmin) *argmin = at;
    return best;
}

/* calculate_carve (:313-352) */
static int calculate_carve(uint16_t* rows, int drop, int location, const to_shape* s, int allow_partial) {
    /* :317-318 */
    if (drop + s->h > TO_ROWS) return 0;
    if (drop < 0) return 0;   /* the reference would wrap around with a negative slice start; not reachable for L <= 16 */
    /* :321-329  without partial carving every piece cell must be filled on the board */
    if (!allow_partial) {
        for (int i = 0; i < s->h; ++i) {
            uint16_t m = (uint16_t)((uint16_t)s->mask[i] << location);
            if ((rows[drop + i] & m) != m) return 0;
        }
    }
    /* :332-337  save the slice, carve */
    uint16_t saved[4];
    for (int i = 0; i < s->h; ++i) {
        saved[i] = rows[drop + i];
        rows[drop + i] &= (uint16_t)~((uint16_t)s->mask[i] << location);
    }
    /* :341-349  the piece dropped onto the carved board must land exactly where it was carved */
    int d[4];
    drop_deltas(rows, location, s, d);
    int new_drop = min_delta(d, s->w, NULL) - 1;
    if (new_drop != drop) {
        for (int i = 0; i < s->h; ++i) rows[drop + i] = saved[i];
        return 0;
    }
    return 1;
}

/* carve (:286-311) */
int to_carve(uint16_t* rows, int piece, int rotations, int location, int allow_partial) {
    to_shape s;
    to_get_tetromino(piece, rotations, &s);
    int d[4], at;
    drop_deltas(rows, location, &s, d);
    int drop = min_delta(d, s.w, &at) - 1;           /* :293-295 */
    int push = (int)s.revtopo[at] + 1;               /* :298 */
    drop += push;                                    /* :301 */
    int increments = allow_partial ? s.h : 1;        /* :304 */
    for (int k = 0; k < increments; ++k) {           /* :305-308 */
        if (calculate_carve(rows, drop, location, &s, allow_partial)) return 1;
        drop -= 1;
    }
    return 0;                                        /* :311 */
}

/* random.shuffle as CPython implements it, on top of randint: for i = n-1 .. 1: j = randint(0, i); swap */
static void shuffle(uint8_t* a, int n, to_randint_fn randint, void* ctx) {
    for (int i = n - 1; i >= 1; --i) {
        int j = randint(ctx, 0, i);
        uint8_t t = a[i]; a[i] = a[j]; a[j] = t;
    }
}

typedef struct {
    uint16_t rows[TO_ROWS];
    uint8_t pieces[256];
    uint8_t sol[256][2];
    int n_pieces;
} carve_checkpoint;

int64_t to_generate_config(int L, int M, to_randint_fn randint, void* ctx, to_phase_fn search_over, int64_t max_iters,
                           uint16_t* rows, uint8_t* pieces_out, uint8_t* solution, int32_t* sol_len) {
    /* game state: board, pieces (front = first to fall), solution (parallel to the carved prefix of pieces) */
    uint8_t pieces[256];
    uint8_t sol[256][2];
    int n_pieces = 0;
    /* RandomPieceGenerator (:64-108) */
    uint8_t bag[7];
    int n_bag = 0;
    /* CheckpointManager (:111-137) */
    carve_checkpoint* cps = NULL;
    int n_cps = 0, cap_cps = 0;
    int attempts = 0, checkpoint_uses = 0;
    const int max_attempts = 40, max_checkpoint_uses = 10;

    /* :228 */
    for (int r = 0; r < TO_ROWS; ++r) rows[r] = (r >= TO_ROWS - L) ? TO_FULL_ROW : 0;

    int64_t iters = 0;
    /* :234  until at least two cells of the bottom row are carved */
    while (__builtin_popcount(rows[TO_ROWS - 1]) > 8) {
        if (max_iters > 0 && iters >= max_iters) { free(cps); return -1; }
        ++iters;
        /* :236  get_random_piece through the _regenerate wrapper (:71-86) */
        int regenerated = 0;
        if (n_bag == 0) {
            for (int k = 0; k < 7; ++k) bag[k] = (uint8_t)k;
            n_bag = 7;
            regenerated = 1;
        }
        int idx = randint(ctx, 0, n_bag - 1);
        int piece = bag[idx];
        /* :239-247  a fresh bag marks a checkpoint */
        if (regenerated) {
            if (n_cps == cap_cps) {
                cap_cps = cap_cps ? 2 * cap_cps : 8;
                cps = (carve_checkpoint*)realloc(cps, (size_t)cap_cps * sizeof(carve_checkpoint));
            }
            carve_checkpoint* c = &cps[n_cps++];
            memcpy(c->rows, rows, sizeof(c->rows));
            memcpy(c->pieces, pieces, (size_t)n_pieces);
            memcpy(c->sol, sol, (size_t)n_pieces * 2);
            c->n_pieces = n_pieces;
        }
        /* :250-253 */
        int rotations = randint(ctx, 0, 3);
        to_shape s;
        to_get_tetromino(piece, rotations, &s);
        int location = randint(ctx, 0, TO_COLS - s.w);
        /* :257-262 */
        if (n_pieces < M && to_carve(rows, piece, rotations, location, n_pieces == 0)) {
            memmove(pieces + 1, pieces, (size_t)n_pieces);
            memmove(sol + 1, sol, (size_t)n_pieces * 2);
            pieces[0] = (uint8_t)piece;
            sol[0][0] = (uint8_t)rotations; sol[0][1] = (uint8_t)location;
            ++n_pieces;
            memmove(bag + idx, bag + idx + 1, (size_t)(n_bag - idx - 1));   /* delete_index (:88-89) */
            --n_bag;
        } else {
            /* :268  add_attempt (:121-123) is only evaluated when the move limit has not been reached */
            int reload = (n_pieces >= M);
            if (!reload) { attempts += 1; reload = attempts > max_attempts; }
            if (reload) {
                /* load_checkpoint (:128-137) */
                attempts = 0;
                if (n_cps > 1 && checkpoint_uses > max_checkpoint_uses) { --n_cps; checkpoint_uses = 0; }
                else checkpoint_uses += 1;
                const carve_checkpoint* c = &cps[n_cps - 1];
                memcpy(rows, c->rows, sizeof(c->rows));            /* :275-276 */
                memcpy(pieces, c->pieces, (size_t)c->n_pieces);
                memcpy(sol, c->sol, (size_t)c->n_pieces * 2);
                n_pieces = c->n_pieces;
                for (int k = 0; k < 7; ++k) bag[k] = (uint8_t)k;   /* :278 generate_pieces */
                n_bag = 7;
            }
        }
    }
    free(cps);
    if (search_over) search_over(ctx);               /* the source of decisions is told that the loop of :234 has ended */
    if (sol_len) *sol_len = n_pieces;
    if (solution) memcpy(solution, sol, (size_t)n_pieces * 2);
    /* :281-284  pad with get_random_sequence(M - len + 1) (:95-102): the leftover bag is shuffled first */
    if (n_pieces <= M) {
        int need = M - n_pieces + 1;
        while (need > 0) {
            if (n_bag == 0) { for (int k = 0; k < 7; ++k) bag[k] = (uint8_t)k; n_bag = 7; }   /* _regenerate wrapper */
            shuffle(bag, n_bag, randint, ctx);                                             /* :93 */
            int take = need < 7 ? need : 7;                                                /* [:min(length - len, 7)] */
            if (take > n_bag) take = n_bag;
            memcpy(pieces + n_pieces, bag, (size_t)take);
            n_pieces += take; need -= take;
            n_bag = 0;                                                                     /* :100 */
        }
    }
    memcpy(pieces_out, pieces, (size_t)(M + 1));
    return iters;
}

typedef struct { const int32_t* tape; int64_t n, pos; int bad; } tape_ctx;

static int32_t tape_randint(void* vctx, int32_t lo, int32_t hi) {
    tape_ctx* t = (tape_ctx*)vctx;
    if (t->pos >= t->n) { t->bad = 1; return lo; }
    const int32_t* e = t->tape + 3 * t->pos++;
    if (e[0] != lo || e[1] != hi) t->bad = 1;      /* the reference asked a different question here */
    return e[2];
}

int64_t to_generate_config_tape(int L, int M, const int32_t* tape, int64_t n, int64_t* consumed,
                                uint16_t* rows, uint8_t* pieces, uint8_t* solution, int32_t* sol_len) {
    tape_ctx t = {tape, n, 0, 0};
    int64_t it = to_generate_config(L, M, tape_randint, &t, NULL, 0, rows, pieces, solution, sol_len);
    if (consumed) *consumed = t.pos;
    return t.bad ? -2 : it;
}

/* The counter-driven decision stream and the restart rule (the build's own definitions; the carving
 * logic above is the reference's).
 *
 * Stream: attempt `a` of configuration `index` owns the word  w = to_rng(seed, 4, index, a);  its k-th stream word is the
 * 32-bit murmur3 finaliser of  low32(w) + k * (high32(w) | 1).  While the search loop (:234) runs, ONE stream word serves the
 * three decisions of a trip, asked for in the reference's order: bag index (:85) = (bits 31-20 * n) >> 12, rotations (:250) =
 * bits 19-18, location (:253) = (bits 17-0 * n) >> 18, n = hi - lo + 1.  After the loop (the shuffles of :93) every decision
 * takes a stream word of its own, its top 24 bits reduced to [lo, hi] as lo + ((top24 * (hi - lo + 1)) >> 24).
 *
 * Restart rule: the configuration is what the FIRST attempt a = 0, 1, ... 23 builds whose search loop (:234) ends within
 * limit(a) trips -- base for attempts 0-11, then doubling with every attempt (2 base for 12, 4 base for 13, ... 256 base for
 * 19 and for 20-23; never above 2^28) -- base = `cutoff` if positive, else the table
 * below (about twice the median search length at that L).  Every attempt starts from the full stack with its own stream.  If all 24 attempts run into their limit the
 * configuration is capped: all-zero rows and pieces, no solution, return value -1. */
typedef struct { uint32_t key, stride, counter, word; int asked, searching; } seeded_ctx;

static int32_t seeded_randint(void* vctx, int32_t lo, int32_t hi) {
    seeded_ctx* c = (seeded_ctx*)vctx;
    uint32_t n = (uint32_t)(hi - lo + 1);
    if (!c->searching) {
        uint32_t top24 = fmix32(c->key + c->counter * c->stride) >> 8;
        c->counter += 1;
        return lo + (int32_t)(((uint64_t)top24 * n) >> 24);
    }
    int which = c->asked++ % 3;                      /* the loop asks three times a trip, always in this order */
    if (which == 0) {
        c->word = fmix32(c->key + c->counter * c->stride);
        c->counter += 1;
        return lo + (int32_t)(((c->word >> 20) * n) >> 12);
    }
    if (which == 1) return lo + (int32_t)((c->word >> 18) & 3u);          /* asked as randint(0, 3) */
    return lo + (int32_t)(((c->word & 0x3FFFFu) * n) >> 18);
}
static void seeded_search_over(void* vctx) { ((seeded_ctx*)vctx)->searching = 0; }

#define TO_CARVE_ATTEMPTS 24
static const int64_t carve_base_limit[17] = {0, 64, 64, 64, 128, 256, 384, 512, 768, 1792, 3328, 5376, 9216, 17408, 36000,
                                             64000, 132000};

int64_t to_carve_attempt_limit(int L, int64_t cutoff, int attempt) {
    int64_t base = cutoff > 0 ? cutoff : carve_base_limit[L < 1 ? 1 : L > 16 ? 16 : L];
    int64_t grown = base << (attempt < 12 ? 0 : attempt - 11 < 8 ? attempt - 11 : 8);
    return grown < ((int64_t)1 << 28) ? grown : ((int64_t)1 << 28);
}

int64_t to_generate_config_seeded(int L, int M, uint64_t seed, uint64_t index, int64_t cutoff,
                                  uint16_t* rows, uint8_t* pieces, uint8_t* solution, int32_t* sol_len, int32_t* attempt_out) {
    int64_t total = 0;
    for (int a = 0; a < TO_CARVE_ATTEMPTS; ++a) {
        uint64_t w = to_rng(seed, 4, index, (uint64_t)a);
        seeded_ctx c = {(uint32_t)w, (uint32_t)(w >> 32) | 1u, 0, 0, 0, 1};
        int64_t limit = to_carve_attempt_limit(L, cutoff, a);
        int64_t it = to_generate_config(L, M, seeded_randint, &c, seeded_search_over, limit, rows, pieces, solution, sol_len);
        if (it >= 0) {
            if (attempt_out) *attempt_out = a;
            return total + it;                       /* iterations spent on this configuration, failed attempts included */
        }
        total += limit;
    }
    memset(rows, 0, TO_ROWS * sizeof(uint16_t));
    memset(pieces, 0, (size_t)M + 1);
    if (sol_len) *sol_len = 0;
    if (attempt_out) *attempt_out = TO_CARVE_ATTEMPTS;
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * cpu_baseline: the loop shape of game/performance_test.py:13-17 (move; reset when finished) over
 * `count` synthetic boards in lockstep, one contiguous slice per thread.
 * ---------------------------------------------------------------------------------------------- */
static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int64_t to_bench_run(uint64_t seed, int64_t count, int L, int M, int64_t steps, int threads, double* seconds) {
    uint16_t* rows = (uint16_t*)malloc((size_t)count * TO_ROWS * sizeof(uint16_t));
    uint8_t* pcs = (uint8_t*)malloc((size_t)count * (size_t)(M + 1));
    uint8_t* act = (uint8_t*)malloc((size_t)count * (size_t)steps);
    if (threads < 1) threads = 1;
    int64_t per = (count + threads - 1) / threads;
    /* inputs are generated before the clock starts, one slice per thread */
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static, 1)
#endif
    for (int t = 0; t < threads; ++t) {
        int64_t lo = t * per, hi = lo + per > count ? count : lo + per;
        if (hi <= lo) continue;
        to_synth_boards(seed, lo, hi - lo, L, rows + lo * TO_ROWS);
        to_synth_pieces(seed, lo, hi - lo, M, pcs + lo * (M + 1));
        for (int64_t s = 0; s < steps; ++s) to_synth_actions(seed, lo, hi - lo, (uint64_t)s, act + s * count + lo);
    }
    to_env** envs = (to_env**)calloc((size_t)threads, sizeof(to_env*));
    for (int t = 0; t < threads; ++t) {
        int64_t lo = t * per, hi = lo + per > count ? count : lo + per;
        if (hi <= lo) continue;
        envs[t] = to_env_create(hi - lo, L, M, lo, seed);
        to_env_set_pool(envs[t], rows, pcs, count);
        to_env_set_options(envs[t], 1, 1, 1.0f, 0.0f, 0.0f);
        to_env_reset(envs[t], NULL);
    }
    float* reward = (float*)malloc((size_t)count * sizeof(float));
    uint8_t* done = (uint8_t*)malloc((size_t)count);
    double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static, 1)
#endif
    for (int t = 0; t < threads; ++t) {
        if (!envs[t]) continue;
        int64_t lo = t * per;
        for (int64_t s = 0; s < steps; ++s) to_env_step(envs[t], act + s * count + lo, reward + lo, done + lo);
    }
    double t1 = now_s();
    if (seconds) *seconds = t1 - t0;
    for (int t = 0; t < threads; ++t) to_env_destroy(envs[t]);
    free(envs); free(rows); free(pcs); free(act); free(reward); free(done);
    return count * steps;
}
