// carve_device.hip -- the carving generator (game/tetris.py:226-352) on the GPU: each lane builds one configuration at a time.
//
// Same algorithm, same counter-based decision stream and therefore the same output, bit for bit, as the host
// generator in carve_generator.hip (`tpl_generate_configs`); it exists because the container of a GPU box may use
// only a handful of host CPUs, while the pool of a million-board environment wants refreshing on the device.
//
// Per lane: the board as ten column words in LDS, lane-major (column k of lane l at word 64 k + l: whatever column a lane
// asks for, its bank is its lane number), because LDS is memory a lane can INDEX: a carve reads the four columns under
// the piece at a per-lane address, works on that local copy -- everything a carve looks at lies in those four columns --
// and writes them back when it stands (through round 2 the columns sat in registers: a 23-blend network to pick the four,
// a 64-bit shift and ten extracts to put the piece back).  The number of filled cells in the bottom row, which ends the
// search at eight (:234), is kept as a count.  Every column word carries a sentinel bit at row 20 (an empty column's top
// reads 20 with no preparation), and the four columns under a piece are ranked by ONE minimum of per-column keys (CarveShape
// below): the loop is bound by the number of vector instructions a trip takes (profiles/NOTES.md, round 4, steps 7-13), and
// these two took a fifth of them.  The 7-bag is seven 3-bit fields of one register, and a slice of `work`
// memory holds the piece list, the solution and the checkpoints.  The lists are kept in REVERSE order: the reference
// prepends each carved piece (`insert(0, ...)`, :258-260), so the list at any checkpoint is a suffix of every later list;
// appending to the reversed arrays and truncating on a reload gives the same lists without ever copying them, and a
// checkpoint is just the ten columns, a length and the bottom-row count.
//
// The search loop is data dependent per lane -- a configuration takes 2,400 iterations on average at L = 10, close to
// exponentially distributed -- so a wave that held 64 configurations for their whole life would run at the pace of its
// slowest lane with the others idle.  The kernel is therefore PERSISTENT with a work queue: a launch has fewer lanes than
// configurations, and a lane that finishes one takes the index of the next from a counter in memory (one atomic add) until
// none are left.  Which lane builds a configuration does not matter -- configuration k is a function of (seed, first + k).
//
// That leaves the END of a launch: when the queue runs dry every lane is in the middle of a search whose remaining length is
// again exponential, and a wave lasts until the slowest of its 64 is through -- 28,000 iterations against 7,400 to drain the
// queue at 2^20 configurations, a third of the lanes active on average (profiles/r03_carve: 23.8 of 64; the schedule replayed
// from the measured histogram gives 0.35).  The restart rule (tpl_device.h) is what shortens that: a configuration is DEFINED
// as the outcome of the first attempt a = 0, 1, ... that ends within its iteration cut-off, every attempt a pure function of
// (seed, index, a).  While the queue has work, the lane that took a configuration runs its attempts one after the other, as
// the host generator does.  Once the queue is dry, the lanes of a wave that have nothing left run FURTHER attempts of the
// configurations their own wave is still searching -- a wave costs the same per trip whether one of its lanes works or all
// do, so these attempts are free, and because helping stays inside the wave, everything it needs sits in LDS (a state word
// and a ticket counter per configuration, updated with LDS atomics); no wave ever waits for another.  An attempt that ends
// reports into the configuration's state word (a mask of failed attempts, a mask of finished ones); the finished attempt with
// the lowest number wins as soon as every lower one has failed -- until then its lane holds the result (columns in LDS, lists
// in its slice) and polls; an attempt above a finished one is dropped.  The answer is therefore the sequential definition's,
// whatever ran where, and a wave outlives the queue by about two cut-offs instead of by its longest search.
// (A first version let any lane of the launch help any configuration through a table in global memory: correct, and twice
// as SLOW as no helping at all -- with every lane of the chip kept busy on speculative attempts the few that mattered ran at
// a quarter of the speed, and thousands of waves polling one counter did the rest: profiles/r04_carve/NOTES.)
//
// The checkpoint stack lives in the lane's slice of global memory, but its TOP entry is mirrored in LDS: a reload (every 41
// failed carves, so in every trip of a 64-lane wave) is then LDS traffic, and the loop is free of global loads -- it was the
// wait for those eleven words that four waves per SIMD were needed to hide.
#include "tpl_internal.h"

namespace tpl {
namespace {

constexpr uint32_t kAttemptMask = (1u << kCarveAttempts) - 1u;
constexpr int kHelpersPerConfiguration = 12;     // attempts of one configuration in flight at a time, at most: those at the base cut-off

struct CarveArgs {
    int32_t L, M;
    uint64_t seed;
    int64_t first, count, cutoff;
    uint16_t* rows;        // [count][20]
    uint8_t* pieces;       // [count][M+1]
    uint8_t* solution;     // [count][M][2] or null
    int32_t* solution_len; // [count] or null
    int32_t* status;       // [count] or null: 0 finished, 1 capped (every attempt ran into its cut-off)
    uint8_t* work;         // [lanes][work_stride]: one slice per LANE of the launch
    int64_t work_stride;
    unsigned long long* next;   // the queue: index of the next configuration nobody has taken yet (zeroed before the launch)
};

// A shape as this kernel reads it (32 bytes of the wave's LDS table, made from kShapeTable at the kernel's start):
//   x    = column nibbles (16 bits) | places the piece can stand, 10 - w + 1, << 16 | 20 - h << 24 (a byte each)
//   cols = the column nibbles again, a BYTE each: a byte of a register is an operand (SDWA), a nibble takes an extract
//   B[k] = per column k the constant 128 bias_k + 32 k,  bias_k = 3 - reverse topography, or 64 past the width (a word each:
//          an operand as it is read).
// With c_k = the top of column k (v_ffbl: the columns carry a sentinel bit at row 20, so an empty one reads 20), the KEY
// 129 c_k + B_k = 128 (c_k + bias_k) + 32 k + c_k orders the columns by c_k + bias_k, equal sums by k (c_k <= 20 < 32), and
// carries c_k in its low five bits: the minimum of four keys is np.argmin's column (:298), its c_k and its sum in one go --
// three instructions a column where compare / minimum / select on two values were six.
struct alignas(16) CarveShape { uint32_t x, cols, spare0, spare1, B[4]; };
struct DShape { uint32_t pat16, places, room, cols, B[4]; };              // room = 20 - h: the deepest drop that stays inside

__device__ __forceinline__ CarveShape carve_shape(const ShapeWord sw) {
    const uint32_t w = (sw.x >> 16) & 7u, h = (sw.x >> 19) & 7u;
    uint32_t B[4];
    for (int k = 0; k < 4; ++k) B[k] = ((sw.y >> (8 * k)) & 0xFFu) * 128u + 32u * (uint32_t)k;
    uint32_t cols = 0;
    for (int k = 0; k < 4; ++k) cols |= ((sw.x >> (4 * k)) & 0xFu) << (8 * k);
    return CarveShape{(sw.x & 0xFFFFu) | ((uint32_t)kCols - w + 1u) << 16 | ((uint32_t)kRows - h) << 24, cols, 0u, 0u,
                      {B[0], B[1], B[2], B[3]}};
}

__device__ __forceinline__ DShape shape_of(const CarveShape* table, uint32_t piece, uint32_t rotations) {
    const CarveShape e = table[piece * 4u + (rotations & 3u)];             // get_tetromino (:60-61); `table` = the wave's LDS copy
    return DShape{e.x & 0xFFFFu, (e.x >> 16) & 0xFFu, e.x >> 24, e.cols, {e.B[0], e.B[1], e.B[2], e.B[3]}};
}

constexpr int kPadCols = 3;                // columns behind column 9 (empty: the sentinel alone), for pieces narrower than four at the right edge
constexpr int kColStride = 64;             // words between consecutive columns of a lane
constexpr uint32_t kFloor = 1u << kRows;   // the sentinel every column word carries in LDS

__device__ __forceinline__ uint32_t column_key(uint32_t column, const DShape& s, int k) {
    return __umul24((uint32_t)__builtin_ctz(column), 129u) + s.B[k];
}
__device__ __forceinline__ uint32_t least_key(const uint32_t* d, const DShape& s) {
    uint32_t best = column_key(d[0], s, 0);
#pragma unroll
    for (int k = 1; k < 4; ++k) { const uint32_t v = column_key(d[k], s, k); best = v < best ? v : best; }
    return best;
}
// calculate_drop_deltas + calculate_drop (:424-433) on the four columns under the piece (d = c[loc .. loc+3]), then :298-301:
// drop + reverse_topography of the first column that attains the minimum (np.argmin) + 1 = (c + bias - 4) + (3 - bias) + 1 = c
__device__ __forceinline__ int carve_depth(const uint32_t* d, const DShape& s) { return (int)(least_key(d, s) & 31u); }
// where the piece comes to rest on the columns d
__device__ __forceinline__ int rest_of(const uint32_t* d, const DShape& s) { return (int)(least_key(d, s) >> 7) - 4; }

// calculate_carve (:313-352) on the local copy d of the piece's four columns; `after` = the columns with the piece taken out.
// Straight-line: its three tests (inside the board :317-318, every cell of the piece filled :321-329, the piece comes to rest
// where it was carved :341-349) are all computed and combined.  Early exits here were nested divergent regions that a wave
// of 64 searches entered on every trip anyway -- some lane passes each test -- at a dozen scalar instructions and two branches
// a test.  (A drop outside the board -- an empty column under the piece reads 20 -- would shift the piece onto the sentinel:
// the sentinel is put back with the same instruction that takes the piece out, so that no column is ever zero when its top is
// asked for, and the carve is refused by the first test whatever the others say.)
__device__ __forceinline__ bool try_carve(const uint32_t* d, int drop, const DShape& s, bool allow_partial, uint32_t* after) {
    const bool inside = (uint32_t)drop <= s.room;                           // :317-318: 0 <= drop and drop + h <= 20, as one comparison
    const uint32_t shift = (uint32_t)drop & 31u;
    uint32_t missing = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = ((s.cols >> (8 * k)) & 0xFFu) << shift;
        missing |= m & ~d[k];
        after[k] = (d[k] & ~m) | kFloor;                                    // :332-337 (one three-input bit operation; see below)
    }
    const bool rests = rest_of(after, s) == drop;                           // :341-349: the piece must come to rest there
    return inside & (allow_partial | (missing == 0u)) & rests;              // :321-329
}

// carve (:286-311) on the lane's columns in LDS, as a TEST: `after` = the four columns under the piece with the piece taken
// out, `drop` = where; the caller writes them back if the carve stands (take_out).  Every lane of a wave runs this whether its
// list has room for another piece or not (the caller refuses the result if it has none): the columns are there to be read,
// and a region under `n < M` -- true for all but a lane or two -- only cost its entry, exit and the copies around it.
__device__ __forceinline__ bool carve(const uint32_t* col, const DShape& s, uint32_t loc, bool allow_partial, uint32_t* after, int& drop) {
    const uint32_t* under = col + loc * kColStride;
    uint32_t d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = under[k * kColStride];
    drop = carve_depth(d, s);                                               // :298-301
    // :304 -- the first piece of a configuration may stick out of the stack and is tried at h depths; every other carve has
    // ONE try, and that is the path every trip takes: it is written without the loop (whose bookkeeping -- a dozen scalar
    // instructions and two branches -- would be paid by every carve of every trip)
    bool ok = try_carve(d, drop, s, allow_partial, after);
    if (!ok && allow_partial) {
        const int h = (int)kRows - (int)s.room;
        for (int t = 1; t < h && !ok; ++t) {
            --drop;
            ok = try_carve(d, drop, s, true, after);
        }
    }
    return ok;
}

// the carve stands: the columns go back, and `bottom` (filled cells of the bottom row) follows
__device__ __forceinline__ void take_out(uint32_t* col, uint32_t& bottom, const DShape& s, uint32_t loc, const uint32_t* after, int drop) {
    uint32_t* under = col + loc * kColStride;
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) under[k * kColStride] = after[k];
        // Cells taken out of the bottom row = cells of the piece that lie in it: bit 19 - drop of each column nibble (every cell
        // of the piece was filled, or -- a configuration's first piece -- the board is the full stack, whose bottom row is full).
        // Shifted left by drop - 16 (mod 32) that bit sits at the top of its nibble for drop 16..19 and past bit 15 for any other.
        bottom -= (uint32_t)__builtin_popcount((s.pat16 << (((uint32_t)drop - 16u) & 31u)) & 0x8888u);
    }
}

constexpr uint32_t kFullBag = 0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15 | 6u << 18;

// one configuration under construction: the board's columns in LDS, the rest in the registers of its lane
struct Search {
    uint32_t* col;              // this lane's column 0 in LDS; column k at col[64 k]
    uint32_t* top;              // this lane's copy of the TOP checkpoint in LDS: ten columns, then length | bottom << 16
    const CarveShape* shapes;   // the shape table in LDS (a per-lane indexed read of constant memory is a vector memory load)
    uint32_t bottom;            // filled cells of the bottom row (the search ends at eight, :234)
    uint32_t bag;               // the 7-bag as 3-bit fields
    int n_bag, n, n_cp, attempts, uses;
    uint32_t iters;             // trips of this attempt (a cut-off is at most 2^28 << 2)
    DecisionStream rnd;         // this attempt's decision stream (tpl_device.h)
    __device__ __forceinline__ int randint(int lo, int hi) { return decision(rnd, lo, hi); }
};

// this lane's slice of the work memory: reversed piece list, reversed solution, checkpoints (entries of 11 words: ten
// columns, then list length | bottom-row count << 16).  Held as 32-bit byte offsets from CarveArgs::work (the work memory of
// a full launch is 0.3-0.7 GB): an access is then the uniform base plus a 32-bit lane offset, with no 64-bit lane arithmetic
struct Slice { uint32_t pieces_rev, sol_rev, cps; };
__device__ __forceinline__ uint8_t* bytes_at(const CarveArgs& p, uint32_t offset) { return p.work + (size_t)offset; }
__device__ __forceinline__ uint32_t* words_at(const CarveArgs& p, uint32_t offset) { return (uint32_t*)(p.work + (size_t)offset); }

__device__ __forceinline__ void begin_search(Search& g, const CarveArgs& p, int64_t k, int attempt) {
    const uint32_t filled = (p.L >= kRows ? kColMask : (((1u << p.L) - 1u) << (kRows - p.L))) | kFloor;
    g.rnd = decision_stream(p.seed, (uint64_t)(p.first + k), (uint32_t)attempt);
#pragma unroll
    for (int x = 0; x < kCols; ++x) g.col[x * kColStride] = filled;         // :228
    g.bottom = (uint32_t)kCols;                                             // L >= 1: the bottom row is full
    g.bag = 0; g.n_bag = 0; g.n = 0; g.n_cp = 0; g.attempts = 0; g.uses = 0; g.iters = 0;
}

__device__ __forceinline__ bool solved(const Search& g) { return g.bottom <= 8u; }     // :234

// one trip of the reference's while loop (:234-279)
__device__ __forceinline__ void search_iteration(Search& g, const Slice& w, const CarveArgs& p) {
    const int max_cps = p.M / 7 + 3;
    bool fresh = false;                                                     // _regenerate (:71-81)
    if (g.n_bag == 0) { g.bag = kFullBag; g.n_bag = 7; fresh = true; }
    const uint32_t word = decision_word(g.rnd);                             // this trip's three decisions (tpl_device.h)
    const int idx = word_bag_index(word, g.n_bag);                          // :85
    const uint32_t piece = (g.bag >> (3 * idx)) & 7u;
    if (fresh && g.n_cp < max_cps) {                                        // :239-247
        uint32_t* e = words_at(p, w.cps + __umul24((uint32_t)g.n_cp, 44u));
        const uint32_t tail = (uint32_t)g.n | (g.bottom << 16);
#pragma unroll
        for (int x = 0; x < kCols; ++x) {
            const uint32_t c = g.col[x * kColStride];
            e[x] = c;                                                       // the stack (global memory: written, rarely read)
            g.top[x * kColStride] = c;                                      // its top entry again, in LDS
        }
        e[10] = tail;
        g.top[kCols * kColStride] = tail;
        ++g.n_cp;
    }
    const int rotations = word_rotations(word);                             // :250
    const DShape s = shape_of(g.shapes, piece, (uint32_t)rotations);
    const int loc = word_location(word, (int)s.places);                     // :253
    uint32_t after[4];
    int drop;
    const bool stands = carve(g.col, s, (uint32_t)loc, g.n == 0, after, drop) & (g.n < p.M);       // :257
    if (stands) {
        take_out(g.col, g.bottom, s, (uint32_t)loc, after, drop);
        *bytes_at(p, w.pieces_rev + (uint32_t)g.n) = (uint8_t)piece;        // insert(0, ...) (:258-260), reversed
        uint8_t* sol = bytes_at(p, w.sol_rev + 2u * (uint32_t)g.n);
        sol[0] = (uint8_t)rotations;
        sol[1] = (uint8_t)loc;
        ++g.n;
        const uint32_t low = g.bag & ((1u << (3 * idx)) - 1u);              // delete_index (:262)
        g.bag = low | ((g.bag >> (3 * (idx + 1))) << (3 * idx));
        --g.n_bag;
    } else if (g.n >= p.M || ++g.attempts > 40) {                           // :268, add_attempt (:121-123)
        g.attempts = 0;                                                     // load_checkpoint (:128-137)
        if (g.n_cp > 1 && g.uses > 10) {                                    // drop the top entry: the one below becomes the top
            --g.n_cp; g.uses = 0;
            const uint32_t* e = words_at(p, w.cps + __umul24((uint32_t)(g.n_cp - 1), 44u));
#pragma unroll
            for (int x = 0; x <= kCols; ++x) g.top[x * kColStride] = e[x];
        } else ++g.uses;
#pragma unroll
        for (int x = 0; x < kCols; ++x) g.col[x * kColStride] = g.top[x * kColStride];     // :275-276
        const uint32_t tail = g.top[kCols * kColStride];
        g.n = (int)(tail & 0xFFFFu);
        g.bottom = tail >> 16;
        g.bag = kFullBag; g.n_bag = 7;                                      // :278
    }
}

// Configuration k is finished.  Its outputs are written by the WHOLE wave for one winning lane at a time: a lane on its own
// would un-reverse its lists one dependent global load after the other (some thirty round trips of a microsecond with the
// other 63 lanes waiting: at L = 5, where a wave finishes a configuration every third trip, that was nine tenths of the
// launch), the wave reads them in one go.  `src` = the winning lane, `cfg` / `n` its configuration and list length
// (wave-uniform).  Lists un-reversed (the reference prepends, :258-260); the board in the interchange layout.
__device__ __forceinline__ void write_lists_and_board(const CarveArgs& p, int lane, int src, int64_t cfg, int n,
                                                      const uint8_t* slice, const uint32_t* col_of_src) {
    const uint8_t* pieces_rev = slice;
    const uint8_t* sol_rev = slice + 256;
    for (int i = lane; i < n; i += 64) {
        p.pieces[cfg * (p.M + 1) + i] = pieces_rev[n - 1 - i];
        if (p.solution) {
            p.solution[(cfg * p.M + i) * 2 + 0] = sol_rev[2 * (n - 1 - i)];
            p.solution[(cfg * p.M + i) * 2 + 1] = sol_rev[2 * (n - 1 - i) + 1];
        }
    }
    if (lane < kRows) {                                                     // lane r builds row r
        uint32_t v = 0;
#pragma unroll
        for (int x = 0; x < kCols; ++x) v |= ((col_of_src[x * kColStride] >> lane) & 1u) << x;
        p.rows[cfg * kRows + lane] = (uint16_t)v;
    }
}

// the winning lane alone: status, solution length, and the piece list filled up to M + 1 (:281-284) -- stores only
__device__ __forceinline__ void write_padding(Search& g, const CarveArgs& p, int64_t k) {
    if (p.status) p.status[k] = 0;
    if (p.solution_len) p.solution_len[k] = g.n;
    uint8_t* out = p.pieces + k * (p.M + 1);
    int n = g.n;
    int need = p.M - n + 1;                                                 // get_random_sequence (:95-102)
    while (need > 0) {
        if (g.n_bag == 0) { g.bag = kFullBag; g.n_bag = 7; }
        for (int i = g.n_bag - 1; i >= 1; --i) {                            // random.shuffle (:93)
            const int j = g.randint(0, i);
            const uint32_t vi = (g.bag >> (3 * i)) & 7u, vj = (g.bag >> (3 * j)) & 7u;
            g.bag = (g.bag & ~(7u << (3 * i))) | (vj << (3 * i));
            g.bag = (g.bag & ~(7u << (3 * j))) | (vi << (3 * j));
        }
        const int take = need < g.n_bag ? need : g.n_bag;
        for (int i = 0; i < take; ++i) out[n + i] = (uint8_t)((g.bag >> (3 * i)) & 7u);
        n += take; need -= take;
        g.n_bag = 0;                                                        // :100
    }
}

// every attempt of configuration k ran into its cut-off: all-zero outputs, status 1
__device__ __forceinline__ void write_capped(const CarveArgs& p, int64_t k) {
    if (p.status) p.status[k] = 1;
    if (p.solution_len) p.solution_len[k] = 0;
    for (int i = 0; i <= p.M; ++i) p.pieces[k * (p.M + 1) + i] = 0;
    for (int r = 0; r < kRows; ++r) p.rows[k * kRows + r] = 0;
}

// a configuration's state word (LDS): attempts that failed (bits 0..23), attempts that finished (24..47), answered (63)
constexpr unsigned long long kAnswered = 1ULL << 63;
__device__ __forceinline__ uint32_t failed_of(unsigned long long st) { return (uint32_t)st & kAttemptMask; }
__device__ __forceinline__ uint32_t finished_of(unsigned long long st) { return (uint32_t)(st >> kCarveAttempts) & kAttemptMask; }
template <typename T> __device__ __forceinline__ T lds_load(const T* p) {             // never kept in a register across trips
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename T> __device__ __forceinline__ void lds_store(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Blocks of kWavesPerBlock waves that share nothing but the block: a CU spreads the waves of one block over its four SIMDs,
// while 64-thread blocks land wherever a slot is free -- with 4096 of them some SIMDs ran six waves and some two, and a
// persistent wave keeps its place for the whole launch (trips of 1.2 to 2.7 us side by side, profiles/r04_carve/NOTES).
constexpr int kWavesPerBlock = 4;
// what one wave keeps in LDS
struct WaveLds {
    uint32_t col[kCols + kPadCols][kColStride];     // the boards' columns, lane-major
    uint32_t top[kCols + 1][kColStride];            // the top checkpoint of every lane: ten columns, then length | bottom << 16
    // per configuration taken from the queue by lane i of this wave (indexed by i, its "home"): state word, tickets handed out
    unsigned long long state[64];
    uint32_t ticket[64];
    CarveShape shape[32];                           // the shape table
};
constexpr int kQueueChunk = 16;            // configurations a wave takes from the queue per atomic
// trips of the search between two looks at the queue, the helpers and the state words: 32 where a search takes hundreds of
// trips or more (L >= 8: median 400 and up), 8 below (measured at L = 10, 2^20 configurations: 4 -> 50 M/s, 8 -> 55, 16 -> 56,
// 32 -> 57; at L = 5 a search is 190 trips and a lane whose attempt ends sits out the rest of its burst)
__device__ __forceinline__ int burst_shift(int L) { return L >= 8 ? 5 : 3; }

// waves_per_eu(4, 4): the register count is reported high enough that a SIMD holds no more than four of these waves --
// with the LDS padding at the launch (four blocks to a CU) the only placement left is four waves on every SIMD
#ifndef TPL_CARVE_WAVES_PER_SIMD
#define TPL_CARVE_WAVES_PER_SIMD 4
#endif
constexpr int kWavesPerSimd = TPL_CARVE_WAVES_PER_SIMD;
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(kWavesPerSimd, kWavesPerSimd))) void carve_kernel(const CarveArgs p) {
    const int lane = (int)threadIdx.x & 63, wave_in_block = (int)threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block;
    const int64_t slot = wave * 64 + lane;
    Slice w;
    w.pieces_rev = (uint32_t)(slot * p.work_stride);
    w.sol_rev = w.pieces_rev + 256u;
    w.cps = w.sol_rev + 512u;

    // everything in LDS is per WAVE (no barrier anywhere: a wave runs in lockstep with itself)
    __shared__ WaveLds s_lds[kWavesPerBlock];
    WaveLds& mine = s_lds[wave_in_block];
    uint32_t (*s_col)[kColStride] = mine.col;
    uint32_t (*s_top)[kColStride] = mine.top;
    unsigned long long* s_state = mine.state;
    uint32_t* s_ticket = mine.ticket;
    CarveShape* s_shape = mine.shape;
    if (lane < 32) s_shape[lane] = carve_shape(kShapeTable[lane]);
    Search g;
    g.col = &s_col[0][lane];
    g.top = &s_top[0][lane];
    g.shapes = s_shape;
#pragma unroll
    for (int x = kCols; x < kCols + kPadCols; ++x) g.col[x * kColStride] = kFloor;
    begin_search(g, p, 0, 0);

    enum : int { kIdle = 0, kRun = 1, kHold = 2 };
    int mode = kIdle;
    bool dry = false;                      // this lane found the queue empty
    bool owns = false;                     // this lane took a configuration from the queue that has no answer yet ...
    int32_t own_k = 0;                     // ... this one (its state word and ticket counter are s_state[lane], s_ticket[lane])
    int32_t k = 0;                         // what this lane is running: configuration,
    int attempt = 0, home = lane;          // attempt, and the lane that took the configuration from the queue
    uint32_t limit = 0;                    // iterations this attempt may use
    uint32_t trip = 0;                     // bursts so far
    const int shift = burst_shift(p.L);    // log2 of a burst's trips
    // the wave's share of the queue: configurations [res_lo, res_hi) are its own to hand to its lanes (wave-uniform values)
    int64_t res_lo = 0, res_hi = 0;
    bool exhausted = false;                // the queue had nothing left when this wave last asked
#ifdef TPL_CARVE_DIAG
    // diagnostic build (tools/carve_diag.py): clocks and counters into the spare words of the queue counter's line
    unsigned long long* diag = p.next + 2;
    unsigned long long d_iters = 0, d_helped = 0, d_dropped = 0;
    atomicMax(&diag[5], ~wall_clock64());                                    // earliest start, as a maximum of the complement
    unsigned long long w_dry = 0, w_tail_trips = 0;                          // per wave: when its first lane found the queue dry
#endif

    auto begin_attempt = [&](int32_t cfg, int a, int at) {
        k = cfg; attempt = a; home = at; limit = (uint32_t)carve_cutoff(p.L, p.cutoff, a);
        begin_search(g, p, cfg, a);
        mode = kRun;
    };

    for (;;) {
        // A SIMD issues for its OLDEST wave first: of the four waves a SIMD holds, the one of the first quarter of the grid ran a
        // trip in 1.2 us and the one of the last quarter in 2.5 (profiles/r04_carve/NOTES), which the queue evens out while it
        // has work and nothing does afterwards -- the launch then waits for the youngest waves.  So the issue priority goes
        // round: every 256 trips a wave moves on to the next of four phases (levels 0, 1, 2, 2), a quarter of the grid in each.
        if ((trip & ((256u >> shift) - 1u)) == 0u) {
            switch (((uint32_t)blockIdx.x / (gridDim.x / 4u + 1u) + (trip >> (8 - shift))) & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(2); break;      // never 3: that level is the step kernel's (tetris_piclim.hip),
                                                                   // which must win the issue against a generator beside it
            }
        }
        ++trip;
        // an answered configuration no longer keeps its home lane (or the wave) in the loop
        if (owns && (lds_load(&s_state[lane]) & kAnswered)) owns = false;
        // (1) the queue: one atomic per wave per kQueueChunk configurations (one per lane per configuration put a million
        // atomics on one address: at L = 5, where a search is short, they WERE the launch)
        const unsigned long long need = __ballot(mode == kIdle && !dry && !owns);
        if (need != 0ULL) {
            if (res_lo >= res_hi && !exhausted) {
                const int leader = __builtin_ctzll(need);
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(p.next, (unsigned long long)kQueueChunk);
                const int64_t got = (int64_t)(((unsigned long long)(uint32_t)__shfl((int)(base >> 32), leader) << 32) |
                                              (uint32_t)__shfl((int)base, leader));
                if (got >= p.count) exhausted = true;
                else { res_lo = got; res_hi = got + kQueueChunk < p.count ? got + kQueueChunk : p.count; }
            }
            if ((need >> lane) & 1ULL) {
                const int64_t q = res_lo + __popcll(need & ((1ULL << lane) - 1ULL));
                if (q < res_hi) {
                    owns = true; own_k = (int32_t)q;
                    lds_store(&s_state[lane], 0ULL);
                    lds_store(&s_ticket[lane], 0u);
                    begin_attempt((int32_t)q, 0, lane);
                } else if (exhausted) {                                      // else: the next trip refills the wave's share
                    dry = true;
#ifdef TPL_CARVE_DIAG
                    atomicMax(&diag[0], ~wall_clock64());                     // the first lane to find the queue empty
#endif
                }
            }
            res_lo += __popcll(need);
            if (res_lo > res_hi) res_lo = res_hi;
        }
        const unsigned long long open = __ballot(owns);
#ifdef TPL_CARVE_DIAG
        if (w_dry == 0 && __ballot(dry) != 0ULL) w_dry = wall_clock64();
        if (w_dry != 0) w_tail_trips += 1u << shift;
#endif
        if (open == 0ULL && __ballot(!dry) == 0ULL) break;                   // wave-uniform: nothing left here, nothing to take
        // (2) lanes with nothing to do (the queue is dry) run a further attempt of a configuration of this wave that has no
        // finished attempt yet and fewer than kHelpersPerConfiguration in flight
        const unsigned long long idle = __ballot(mode == kIdle && dry);
        if (idle != 0ULL && open != 0ULL) {
            const unsigned long long st = lds_load(&s_state[lane]);
            const int handed = (int)lds_load(&s_ticket[lane]);               // attempts 0 .. handed are out
            const int flying = handed + 1 - __popc(failed_of(st) | finished_of(st));       // its attempts still running
            const bool wants = owns && finished_of(st) == 0u && handed + 1 < kCarveAttempts && flying < kHelpersPerConfiguration;
            // A further attempt is needed only if every one already running fails, so it is worth most where the fewest are
            // running: this trip serves the configurations at the lowest such number (the next trip the next).  Replaying
            // the measured search lengths, that ends a wave 17 % earlier than serving them in lane order.
            unsigned long long cand = 0ULL;
            for (int level = 0; level < kHelpersPerConfiguration && cand == 0ULL; ++level) cand = __ballot(wants && flying <= level);
            if (cand != 0ULL) {
                // the r-th idle lane takes the r-th candidate (one new attempt per configuration per trip).  The shuffle is
                // executed by the whole wave: a lane that is switched off hands nothing to ds_bpermute
                const int rank = __popcll(idle & ((1ULL << lane) - 1ULL));
                int skip = rank < __popcll(cand) ? rank : 0;
                unsigned long long m = cand;
                while (skip-- > 0) m &= m - 1ULL;
                const int src = __builtin_ctzll(m);
                const int32_t cfg = __shfl(own_k, src);
                if (mode == kIdle && dry && rank < __popcll(cand)) {
                    const int a = (int)atomicAdd(&s_ticket[src], 1u) + 1;
                    // a ticket below kCarveAttempts is RUN (or dropped only because a lower attempt has finished): a ticket
                    // nobody ran would stand between a finished attempt and its win for ever
                    if (a < kCarveAttempts) {
                        begin_attempt(cfg, a, src);
#ifdef TPL_CARVE_DIAG
                        ++d_helped;
#endif
                    }
                }
            }
        }
        bool won = false;                  // this lane's attempt is the answer: it has finished and every lower one has failed
        // (3) has a lower attempt of my configuration finished (mine cannot be the answer then), or -- holding a finished
        // attempt -- have all lower ones failed?
        if (mode != kIdle) {
            const unsigned long long st = lds_load(&s_state[home]);
            const uint32_t below = (1u << attempt) - 1u;
            if (finished_of(st) & below) {
                mode = kIdle;
#ifdef TPL_CARVE_DIAG
                ++d_dropped;
#endif
            } else if (mode == kHold && (failed_of(st) & below) == below) won = true;
        }
        // (4) a burst of the search -- the bookkeeping above is some forty instructions, a trip of the search two hundred: it is
        // paid once per burst (a lane whose attempt ends inside a burst sits out the rest of it: sixteen trips in 2,500) --
        // then, if the attempt is over, its end
        if (mode == kRun) {
            // attempts begin between bursts, so within one every running lane is at trip `iters + r` of its attempt: the cut-off
            // is a comparison of the burst's own (scalar) counter with what the lane has left, and `iters` moves once per burst
            const uint32_t left = limit - g.iters;
            const uint32_t stop = left < (1u << shift) ? left : 1u << shift;  // the burst's end and the cut-off as ONE per-lane bound
#pragma unroll 1
            for (uint32_t r = 0;; ++r) {
                if (solved(g) || r >= stop) break;
#ifdef TPL_CARVE_DIAG
                ++d_iters;
#endif
                search_iteration(g, w, p);
            }
            const bool done = solved(g), out = !done && left <= (1u << shift);
            g.iters += 1u << shift;                                          // (only read again if the attempt goes on)
            if (!done && !out) {
                // the attempt goes on
            } else if (done) {
                const unsigned long long bit = 1ULL << (kCarveAttempts + attempt);
                const unsigned long long st = atomicOr(&s_state[home], bit) | bit;
                const uint32_t below = (1u << attempt) - 1u;
                if (finished_of(st) & below) mode = kIdle;
                else if ((failed_of(st) & below) == below) won = true;
                else mode = kHold;
            } else {
                const unsigned long long bit = 1ULL << attempt;
                const unsigned long long st = atomicOr(&s_state[home], bit) | bit;
                mode = kIdle;
                if (failed_of(st) == kAttemptMask) {                        // the last of the attempts to fail
                    write_capped(p, k);
                    atomicOr(&s_state[home], kAnswered);
                } else if (home == lane && finished_of(st) == 0u) {        // the next attempt of the configuration I took
                    const int a = (int)atomicAdd(&s_ticket[lane], 1u) + 1;
                    if (a < kCarveAttempts) begin_attempt(k, a, lane);
                }
            }
        }
        unsigned long long winners = __ballot(won);
        if (winners != 0ULL) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");           // the winners' list stores, before other lanes read them
            do {
                const int src = __builtin_ctzll(winners);
                winners &= winners - 1ULL;
                write_lists_and_board(p, lane, src, (int64_t)__shfl(k, src), __shfl(g.n, src),
                                      p.work + (wave * 64 + src) * p.work_stride, &s_col[0][src]);
            } while (winners != 0ULL);
            if (won) {
                write_padding(g, p, k);
                atomicOr(&s_state[home], kAnswered);
                mode = kIdle;
            }
        }
    }
#ifdef TPL_CARVE_DIAG
    atomicMax(&diag[1], wall_clock64());
    atomicAdd(&diag[2], d_iters);
    atomicAdd(&diag[3], d_helped);
    atomicAdd(&diag[4], d_dropped);
    if (lane == 0) {                                                         // per wave, behind the queue's line
        unsigned long long* wd = p.next + 8 + (size_t)wave * 4;
        wd[0] = w_dry; wd[1] = wall_clock64(); wd[2] = trip; wd[3] = w_tail_trips;
    }
#endif
}

}  // namespace
}  // namespace tpl

using namespace tpl;

// work memory: one slice per lane of the launch (never more lanes than configurations, rounded up to whole waves, and never
// more than kMaxWaves waves: four to a SIMD of the chip), then the queue's counter on a line of its own
constexpr int64_t kMaxWaves = 1024 * kWavesPerSimd;
static size_t work_stride_bytes(int32_t M) { return (256 + 512 + (size_t)(M / 7 + 3) * 44 + 63) / 64 * 64; }
static size_t work_slices(int64_t count) {                        // whole blocks of four waves
    int64_t waves = ((count + 63) / 64 + 3) / 4 * 4;
    return (size_t)(waves < kMaxWaves ? waves : kMaxWaves) * 64;
}
#ifdef TPL_CARVE_DIAG
static size_t control_bytes(int64_t) { return 64 + kMaxWaves * 32; }     // + four words per wave (tools/carve_diag.py)
#else
static size_t control_bytes(int64_t) { return 64; }
#endif

extern "C" size_t tpl_generate_configs_device_work_bytes(int32_t M, int64_t count) {
    if (M < 1 || count < 1) return 0;
    return work_stride_bytes(M) * work_slices(count) + control_bytes(count);
}

extern "C" int tpl_generate_configs_device_waves(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count,
                                                 int64_t cutoff, int32_t waves, uint16_t* rows, uint8_t* pieces,
                                                 uint8_t* solution, int32_t* solution_len, int32_t* status, void* work,
                                                 size_t work_bytes, void* stream) {
    if (L < 1 || L > 16) return fail_msg(TPL_ERR_ARG, "carving needs 1 <= L <= 16 (got %d)", L);
    if (M < 1 || M > 254) return fail_msg(TPL_ERR_ARG, "M=%d out of range [1, 254]", M);
    if (count < 1 || count > 0x7FFFFFFF || first < 0 || !rows || !pieces)
        return fail_msg(TPL_ERR_ARG, "bad count / first / output pointers");
    if (waves < 0 || cutoff < 0 || cutoff > ((int64_t)1 << 28)) return fail_msg(TPL_ERR_ARG, "waves is negative / cutoff outside [0, 2^28]");
    const size_t need = tpl_generate_configs_device_work_bytes(M, count);
    if (!work || work_bytes < need) return fail_msg(TPL_ERR_ARG, "work has %zu bytes, need %zu", work_bytes, need);
    if (((uintptr_t)work & 7u) != 0) return fail_msg(TPL_ERR_ARG, "work must be 8-byte aligned");
    { const int rc = carve_pilot(L, M, cutoff); if (rc != TPL_OK) return rc; }     // the pilot configurations on the host first
    if (work_stride_bytes(M) * work_slices(count) >= ((size_t)1 << 32))             // the kernel's slice offsets are 32-bit
        return fail_msg(TPL_ERR_STATE, "work slices of %zu bytes outgrew 32-bit offsets", work_stride_bytes(M));
    // how many waves share the queue.  Automatic: as many as the chip runs at full rate (four per SIMD), or one lane per
    // configuration if that is fewer -- lanes without a configuration of their own run further attempts of their wave's.
    const int64_t most = (count + 63) / 64;
    int64_t launch = waves > 0 ? waves : most;
    if (launch > most) launch = most;
    if (launch > kMaxWaves) launch = kMaxWaves;
    const int64_t blocks = launch < 1 ? 1 : (launch + 3) / 4;       // blocks of four waves (lanes beyond `count` only ever help)
    CarveArgs p{};
    p.L = L; p.M = M; p.seed = seed; p.first = first; p.count = count; p.cutoff = cutoff;
    p.rows = rows; p.pieces = pieces; p.solution = solution; p.solution_len = solution_len; p.status = status;
    p.work = (uint8_t*)work; p.work_stride = (int64_t)work_stride_bytes(M);
    p.next = (unsigned long long*)((uint8_t*)work + work_stride_bytes(M) * work_slices(count));
    TPL_HIP(hipMemsetAsync(p.next, 0, 64, (hipStream_t)stream));
    // LDS per block padded to a quarter of a CU's 160 KB: no CU takes more than four blocks (one wave of each per SIMD), so a
    // full launch of 1024 blocks sits four to every CU instead of three here and five there
    constexpr size_t kLdsPerBlock = 160 * 1024 / kWavesPerSimd, kLdsStatic = sizeof(WaveLds) * kWavesPerBlock;
    static_assert(kLdsStatic <= kLdsPerBlock, "the kernel's LDS arrays outgrew the padding");
    hipLaunchKernelGGL(carve_kernel, dim3((unsigned)blocks), dim3(256), kLdsPerBlock - kLdsStatic, (hipStream_t)stream, p);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

extern "C" int tpl_generate_configs_device(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count,
                                           int64_t cutoff, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                                           int32_t* solution_len, int32_t* status, void* work, size_t work_bytes,
                                           void* stream) {
    return tpl_generate_configs_device_waves(L, M, seed, first, count, cutoff, 0, rows, pieces, solution, solution_len,
                                             status, work, work_bytes, stream);
}
