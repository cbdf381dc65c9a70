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
// search at eight (:234), is kept as a count.  The 7-bag is seven 3-bit fields of one register, and a slice of `work`
// memory holds the piece list, the solution and the checkpoints.  The lists are kept in REVERSE order: the reference
// prepends each carved piece (`insert(0, ...)`, :258-260), so the list at any checkpoint is a suffix of every later list;
// appending to the reversed arrays and truncating on a reload gives the same lists without ever copying them, and a
// checkpoint is just the ten columns, a length and the bottom-row count.
//
// The search loop is data dependent per lane -- a configuration takes 2,400 iterations on average at L = 10 and ten times
// that now and then -- so a wave that held 64 configurations for their whole life would run at the pace of its slowest
// lane with the others idle (measured: 12 of 64 lanes active per vector instruction, profiles/r03_carve).  The kernel is
// therefore PERSISTENT with a work queue: a launch has fewer lanes than configurations, and a lane that finishes one
// takes the index of the next from a counter in memory (one atomic add) until none are left.  Which lane builds a
// configuration does not matter -- configuration k is a function of (seed, first + k) alone.  The wave leaves when all
// its lanes have found the queue empty, which every lane does after finitely many iterations: the loop of one
// configuration has no natural bound (the reference's has none either), so every configuration stops after
// `max_iters` iterations (a hard cap applies when the caller passes 0) and reports it in `status`.
#include "tpl_internal.h"

namespace tpl {
namespace {

constexpr int64_t kHardIterationCap = 1 << 22;

struct CarveArgs {
    int32_t L, M;
    uint64_t seed;
    int64_t first, count, max_iters;
    uint16_t* rows;        // [count][20]
    uint8_t* pieces;       // [count][M+1]
    uint8_t* solution;     // [count][M][2] or null
    int32_t* solution_len; // [count] or null
    int32_t* status;       // [count] or null: 0 finished, 1 stopped at the iteration cap
    uint8_t* work;         // [lanes][work_stride]: one slice per LANE of the launch
    int64_t work_stride;
    unsigned long long* next;   // the queue: index of the next configuration nobody has taken yet (zeroed before the launch)
};

struct DShape { uint32_t pat16, w, h; uint32_t bias; };   // column nibbles, width, height, per-column 3 - revtopo bytes

__device__ __forceinline__ DShape shape_of(uint32_t piece, uint32_t rotations) {
    const ShapeWord sw = kShapeTable[piece * 4u + (rotations & 3u)];       // get_tetromino (:60-61)
    return DShape{sw.x & 0xFFFFu, (sw.x >> 16) & 7u, (sw.x >> 19) & 7u, sw.y};
}

constexpr int kPadCols = 3;                // empty columns behind column 9, for pieces narrower than four at the right edge
constexpr int kColStride = 64;             // words between consecutive columns of a lane

// calculate_drop_deltas + calculate_drop (:424-433) on the four columns under the piece (d = c[loc .. loc+3]): drop, and
// reverse_topography of the first column that attains the minimum (np.argmin, :298)
__device__ __forceinline__ int drop_of(const uint32_t* d, const DShape& s, uint32_t& revtopo_at_min) {
    uint32_t best = 0xFFu, at_bias = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t bias = (s.bias >> (8 * k)) & 0xFFu;                 // 3 - revtopo, or 64 past the width
        const uint32_t v = (uint32_t)__builtin_ctz(d[k] | (1u << kRows)) + bias;
        if (v < best) { best = v; at_bias = bias; }
    }
    revtopo_at_min = 3u - at_bias;
    return (int)best - 4;
}

// calculate_carve (:313-352) on the local copy d of the piece's four columns; `after` = the columns with the piece taken out
__device__ __forceinline__ bool try_carve(const uint32_t* d, int drop, const DShape& s, bool allow_partial, uint32_t* after) {
    if (drop + (int)s.h > kRows || drop < 0) return false;                  // :317-318
    uint32_t missing = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = ((s.pat16 >> (4 * k)) & 0xFu) << (uint32_t)drop;
        missing |= m & ~d[k];
        after[k] = d[k] & ~m;                                               // :332-337
    }
    if (!allow_partial && missing) return false;                            // :321-329
    uint32_t unused;
    return drop_of(after, s, unused) == drop;                               // :341-349: the piece must come to rest there
}

// carve (:286-311) on the lane's columns in LDS; `bottom` = filled cells of the bottom row, kept up to date
__device__ __forceinline__ bool carve(uint32_t* col, uint32_t& bottom, const DShape& s, uint32_t loc, bool allow_partial) {
    uint32_t* under = col + loc * kColStride;
    uint32_t d[4], after[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = under[k * kColStride];
    uint32_t revtopo;
    int drop = drop_of(d, s, revtopo);
    drop += (int)revtopo + 1;                                               // :298-301
    const int tries = allow_partial ? (int)s.h : 1;                         // :304
    for (int t = 0; t < tries; ++t, --drop) {
        if (try_carve(d, drop, s, allow_partial, after)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                under[k * kColStride] = after[k];
                bottom -= ((d[k] ^ after[k]) >> (kRows - 1)) & 1u;
            }
            return true;
        }
    }
    return false;
}

constexpr uint32_t kFullBag = 0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15 | 6u << 18;

// one configuration under construction: the board's columns in LDS, the rest in the registers of its lane
struct Search {
    uint32_t* col;              // this lane's column 0 in LDS; column k at col[64 k]
    uint32_t bottom;            // filled cells of the bottom row (the search ends at eight, :234)
    uint32_t bag;               // the 7-bag as 3-bit fields
    int n_bag, n, n_cp, attempts, uses;
    int64_t iters;
    uint32_t key;               // the configuration's decision stream (decision(), tpl_device.h)
    __device__ __forceinline__ int randint(int lo, int hi) { return decision(key, lo, hi); }
};

// this lane's slice of the work memory: reversed piece list, reversed solution, checkpoints (entries of 11 words: ten
// columns, then list length | bottom-row count << 16)
struct Slice { uint8_t* pieces_rev; uint8_t* sol_rev; uint32_t* cps; };

__device__ __forceinline__ void begin_search(Search& g, const CarveArgs& p, int64_t k) {
    const uint32_t filled = p.L >= kRows ? kColMask : (((1u << p.L) - 1u) << (kRows - p.L));
    g.key = decision_key(p.seed, (uint64_t)(p.first + k));
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
    const int idx = g.randint(0, g.n_bag - 1);                              // :85
    const uint32_t piece = (g.bag >> (3 * idx)) & 7u;
    if (fresh && g.n_cp < max_cps) {                                        // :239-247
        uint32_t* e = w.cps + g.n_cp * 11;
#pragma unroll
        for (int x = 0; x < kCols; ++x) e[x] = g.col[x * kColStride];
        e[10] = (uint32_t)g.n | (g.bottom << 16);
        ++g.n_cp;
    }
    const int rotations = g.randint(0, 3);                                  // :250
    const DShape s = shape_of(piece, (uint32_t)rotations);
    const int loc = g.randint(0, kCols - (int)s.w);                         // :253
    if (g.n < p.M && carve(g.col, g.bottom, s, (uint32_t)loc, g.n == 0)) {  // :257
        w.pieces_rev[g.n] = (uint8_t)piece;                                 // insert(0, ...) (:258-260), reversed
        w.sol_rev[2 * g.n] = (uint8_t)rotations;
        w.sol_rev[2 * g.n + 1] = (uint8_t)loc;
        ++g.n;
        const uint32_t low = g.bag & ((1u << (3 * idx)) - 1u);              // delete_index (:262)
        g.bag = low | ((g.bag >> (3 * (idx + 1))) << (3 * idx));
        --g.n_bag;
    } else if (g.n >= p.M || ++g.attempts > 40) {                           // :268, add_attempt (:121-123)
        g.attempts = 0;                                                     // load_checkpoint (:128-137)
        if (g.n_cp > 1 && g.uses > 10) { --g.n_cp; g.uses = 0; }
        else ++g.uses;
        const uint32_t* e = w.cps + (g.n_cp - 1) * 11;
#pragma unroll
        for (int x = 0; x < kCols; ++x) g.col[x * kColStride] = e[x];       // :275-276
        g.n = (int)(e[10] & 0xFFFFu);
        g.bottom = e[10] >> 16;
        g.bag = kFullBag; g.n_bag = 7;                                      // :278
    }
}

// configuration k is finished (or gave up at the cap): lists un-reversed, the piece list filled up to M + 1 (:281-284),
// the board in the interchange layout
__device__ __forceinline__ void write_configuration(Search& g, const Slice& w, const CarveArgs& p, int64_t k, bool capped) {
    if (p.status) p.status[k] = capped ? 1 : 0;
    if (p.solution_len) p.solution_len[k] = g.n;
    uint8_t* out = p.pieces + k * (p.M + 1);
    int n = g.n;
    for (int i = 0; i < n; ++i) {
        out[i] = w.pieces_rev[n - 1 - i];
        if (p.solution) {
            p.solution[(k * p.M + i) * 2 + 0] = w.sol_rev[2 * (n - 1 - i)];
            p.solution[(k * p.M + i) * 2 + 1] = w.sol_rev[2 * (n - 1 - i) + 1];
        }
    }
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
    uint32_t c[kCols];
#pragma unroll
    for (int x = 0; x < kCols; ++x) c[x] = g.col[x * kColStride];
    uint16_t* rows = p.rows + k * kRows;
#pragma unroll
    for (int r = 0; r < kRows; ++r) rows[r] = (uint16_t)row_of_cols(c, r);
}

__global__ __launch_bounds__(64) void carve_kernel(const CarveArgs p) {
    const int64_t slot = (int64_t)blockIdx.x * 64 + threadIdx.x;
    Slice w;
    w.pieces_rev = p.work + slot * p.work_stride;
    w.sol_rev = w.pieces_rev + 256;
    w.cps = (uint32_t*)(w.sol_rev + 512);
    const int64_t cap = p.max_iters > 0 ? p.max_iters : kHardIterationCap;

    __shared__ uint32_t s_col[kCols + kPadCols][kColStride];
    Search g;
    g.col = &s_col[0][threadIdx.x];
#pragma unroll
    for (int x = kCols; x < kCols + kPadCols; ++x) g.col[x * kColStride] = 0u;
    begin_search(g, p, 0);
    int64_t k = 0;                                                          // the configuration this lane is building
    bool busy = false, dry = false;                                         // dry: this lane found the queue empty
    for (;;) {
        if (!busy && !dry) {
            k = (int64_t)atomicAdd(p.next, 1ULL);
            dry = k >= p.count;
            busy = !dry;
            if (busy) begin_search(g, p, k);
        }
        // every lane of the wave stays in the loop until all of them are dry: the exit is wave-uniform
        if (__ballot(busy) == 0ULL) break;
        if (busy) {
            const bool done = solved(g), capped = !done && g.iters >= cap;
            if (!done && !capped) {
                ++g.iters;
                search_iteration(g, w, p);
            } else {
                write_configuration(g, w, p, k, capped);
                busy = false;
            }
        }
    }
}

}  // namespace
}  // namespace tpl

using namespace tpl;

// work memory: one slice per lane of the launch (never more lanes than configurations, rounded up to whole waves, and never
// more than kMaxWaves waves: four to a SIMD of the chip), then the queue's counter on a line of its own
constexpr int64_t kMaxWaves = 4096;
static size_t work_stride_bytes(int32_t M) { return (256 + 512 + (size_t)(M / 7 + 3) * 44 + 63) / 64 * 64; }
static size_t work_slices(int64_t count) {
    const int64_t waves = (count + 63) / 64;
    return (size_t)(waves < kMaxWaves ? waves : kMaxWaves) * 64;
}

extern "C" size_t tpl_generate_configs_device_work_bytes(int32_t M, int64_t count) {
    if (M < 1 || count < 1) return 0;
    return work_stride_bytes(M) * work_slices(count) + 64;
}

extern "C" int tpl_generate_configs_device_waves(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count,
                                                 int64_t max_iters, int32_t waves, uint16_t* rows, uint8_t* pieces,
                                                 uint8_t* solution, int32_t* solution_len, int32_t* status, void* work,
                                                 size_t work_bytes, void* stream) {
    if (L < 1 || L > 16) return fail_msg(TPL_ERR_ARG, "carving needs 1 <= L <= 16 (got %d)", L);
    if (M < 1 || M > 254) return fail_msg(TPL_ERR_ARG, "M=%d out of range [1, 254]", M);
    if (count < 1 || first < 0 || !rows || !pieces) return fail_msg(TPL_ERR_ARG, "bad count / first / output pointers");
    if (waves < 0) return fail_msg(TPL_ERR_ARG, "waves is negative");
    const size_t need = tpl_generate_configs_device_work_bytes(M, count);
    if (!work || work_bytes < need) return fail_msg(TPL_ERR_ARG, "work has %zu bytes, need %zu", work_bytes, need);
    if (((uintptr_t)work & 7u) != 0) return fail_msg(TPL_ERR_ARG, "work must be 8-byte aligned");
    // how many waves share the queue.  Automatic: four configurations per lane on average (the wave's tail is then one
    // configuration out of four or more); never more than four waves per SIMD of the chip (2^20 configurations: 14.8 M/s
    // on 1024 waves, 16.7 on 2048, 18.6 on 4096).
    const int64_t most = (count + 63) / 64;
    int64_t launch = waves > 0 ? waves : (count + 255) / 256;
    if (launch > most) launch = most;
    if (launch > kMaxWaves) launch = kMaxWaves;
    if (launch < 1) launch = 1;
    CarveArgs p{};
    p.L = L; p.M = M; p.seed = seed; p.first = first; p.count = count; p.max_iters = max_iters;
    p.rows = rows; p.pieces = pieces; p.solution = solution; p.solution_len = solution_len; p.status = status;
    p.work = (uint8_t*)work; p.work_stride = (int64_t)work_stride_bytes(M);
    p.next = (unsigned long long*)((uint8_t*)work + work_stride_bytes(M) * work_slices(count));
    TPL_HIP(hipMemsetAsync(p.next, 0, sizeof(unsigned long long), (hipStream_t)stream));
    hipLaunchKernelGGL(carve_kernel, dim3((unsigned)launch), dim3(64), 0, (hipStream_t)stream, p);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

extern "C" int tpl_generate_configs_device(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count,
                                           int64_t max_iters, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                                           int32_t* solution_len, int32_t* status, void* work, size_t work_bytes,
                                           void* stream) {
    return tpl_generate_configs_device_waves(L, M, seed, first, count, max_iters, 0, rows, pieces, solution, solution_len,
                                             status, work, work_bytes, stream);
}
