// carve_device.hip -- the carving generator (game/tetris.py:226-352) on the GPU: one configuration per lane.
//
// Same algorithm, same counter-based decision stream and therefore the same output, bit for bit, as the host
// generator in carve_generator.hip (`tpl_generate_configs`); it exists because the container of a GPU box may use
// only a handful of host CPUs, while the pool of a million-board environment wants refreshing on the device.
//
// Per lane: the board as ten column words in registers (the four columns under the piece come out of the same
// blend network as in the step kernel, carving is an AND with the piece's column patterns moved into place by one
// 64-bit shift), the 7-bag as seven 3-bit fields of one register, and a slice of `work` memory holding the piece
// list, the solution and the checkpoints.  The lists are kept in REVERSE order: the reference prepends each carved
// piece (`insert(0, ...)`, :258-260), so the list at any checkpoint is a suffix of every later list; appending to
// the reversed arrays and truncating on a reload gives the same lists without ever copying them, and a checkpoint
// is just the ten columns and a length.
//
// The search loop is data dependent per lane and has no natural bound (the reference's has none either): every lane
// stops after `max_iters` iterations (a hard cap applies when the caller passes 0) and reports it in `status`.
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
    uint8_t* work;         // [count][work_stride]
    int64_t work_stride;
};

struct DShape { uint32_t pat16, w, h; uint32_t bias; };   // column nibbles, width, height, per-column 3 - revtopo bytes

__device__ __forceinline__ DShape shape_of(uint32_t piece, uint32_t rotations) {
    const ShapeWord sw = kShapeTable[piece * 4u + (rotations & 3u)];       // get_tetromino (:60-61)
    return DShape{sw.x & 0xFFFFu, (sw.x >> 16) & 7u, (sw.x >> 19) & 7u, sw.y};
}

// c[loc .. loc+3] (see move_board in tpl_device.h)
__device__ __forceinline__ void select4(const uint32_t* c, uint32_t loc, uint32_t* d) {
    const uint32_t m0 = 0u - (loc & 1u), m1 = 0u - ((loc >> 1) & 1u), m2 = 0u - ((loc >> 2) & 1u), m3 = 0u - ((loc >> 3) & 1u);
    uint32_t a[10], b[10];
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = blend(m0, c[k + 1], c[k]);
    a[9] = c[9];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = blend(m1, a[k + 2], a[k]);
    b[8] = a[8]; b[9] = a[9];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = blend(m2, b[k + 4], b[k]);
    d[0] = blend(m3, b[8], d[0]);
    d[1] = blend(m3, b[9], d[1]);
}

// calculate_drop_deltas + calculate_drop (:424-433): drop, and reverse_topography of the first column that attains
// the minimum (np.argmin, :298)
__device__ __forceinline__ int drop_of(const uint32_t* c, uint32_t loc, const DShape& s, uint32_t& revtopo_at_min) {
    uint32_t d[4];
    select4(c, loc, d);
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

// the piece's column patterns, shifted down by `drop`, spread over the ten columns
__device__ __forceinline__ void piece_columns(const DShape& s, uint32_t loc, uint32_t drop, uint32_t* m) {
    const uint64_t placed = (uint64_t)s.pat16 << (4u * loc);
    const uint32_t lo = (uint32_t)placed, hi = (uint32_t)(placed >> 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = ((lo >> (4 * k)) & 0xFu) << drop;
    m[8] = (hi & 0xFu) << drop;
    m[9] = ((hi >> 4) & 0xFu) << drop;
}

// calculate_carve (:313-352)
__device__ __forceinline__ bool try_carve(uint32_t* c, int drop, uint32_t loc, const DShape& s, bool allow_partial) {
    if (drop + (int)s.h > kRows || drop < 0) return false;                  // :317-318
    uint32_t m[10];
    piece_columns(s, loc, (uint32_t)drop, m);
    if (!allow_partial) {                                                   // :321-329
        uint32_t missing = 0;
#pragma unroll
        for (int k = 0; k < kCols; ++k) missing |= m[k] & ~c[k];
        if (missing) return false;
    }
    uint32_t saved[10];
#pragma unroll
    for (int k = 0; k < kCols; ++k) { saved[k] = c[k]; c[k] &= ~m[k]; }     // :332-337
    uint32_t unused;
    if (drop_of(c, loc, s, unused) != drop) {                               // :341-349
#pragma unroll
        for (int k = 0; k < kCols; ++k) c[k] = saved[k];
        return false;
    }
    return true;
}

// carve (:286-311)
__device__ __forceinline__ bool carve(uint32_t* c, const DShape& s, uint32_t loc, bool allow_partial) {
    uint32_t revtopo;
    int drop = drop_of(c, loc, s, revtopo);
    drop += (int)revtopo + 1;                                               // :298-301
    const int tries = allow_partial ? (int)s.h : 1;                         // :304
    for (int k = 0; k < tries; ++k, --drop)
        if (try_carve(c, drop, loc, s, allow_partial)) return true;
    return false;
}

constexpr uint32_t kFullBag = 0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15 | 6u << 18;

__global__ __launch_bounds__(64) void carve_kernel(const CarveArgs p) {
    const int64_t k = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (k >= p.count) return;
    const uint64_t base = rng_base(p.seed, 4, (uint64_t)(p.first + k));
    uint64_t counter = 0;
    auto randint = [&](int lo, int hi) { return rng_range(rng_at(base, counter++), lo, hi); };

    // this lane's slice of the work memory: reversed piece list, reversed solution, checkpoints
    uint8_t* pieces_rev = p.work + k * p.work_stride;
    uint8_t* sol_rev = pieces_rev + 256;
    uint32_t* cps = (uint32_t*)(sol_rev + 512);                             // entries of 11 words: ten columns, length
    const int max_cps = p.M / 7 + 3;

    uint32_t c[kCols];
    const uint32_t filled = p.L >= kRows ? kColMask : (((1u << p.L) - 1u) << (kRows - p.L));
#pragma unroll
    for (int x = 0; x < kCols; ++x) c[x] = filled;                          // :228
    uint32_t bag = 0;
    int n_bag = 0, n = 0, n_cp = 0, attempts = 0, uses = 0;
    int64_t iters = 0;
    const int64_t cap = p.max_iters > 0 ? p.max_iters : kHardIterationCap;
    bool capped = false;

    for (;;) {
        int bottom = 0;
#pragma unroll
        for (int x = 0; x < kCols; ++x) bottom += (c[x] >> (kRows - 1)) & 1u;
        if (bottom <= 8) break;                                             // :234
        if (iters++ >= cap) { capped = true; break; }
        bool fresh = false;                                                 // _regenerate (:71-81)
        if (n_bag == 0) { bag = kFullBag; n_bag = 7; fresh = true; }
        const int idx = randint(0, n_bag - 1);                              // :85
        const uint32_t piece = (bag >> (3 * idx)) & 7u;
        if (fresh && n_cp < max_cps) {                                      // :239-247
            uint32_t* e = cps + n_cp * 11;
#pragma unroll
            for (int x = 0; x < kCols; ++x) e[x] = c[x];
            e[10] = (uint32_t)n;
            ++n_cp;
        }
        const int rotations = randint(0, 3);                                // :250
        const DShape s = shape_of(piece, (uint32_t)rotations);
        const int loc = randint(0, kCols - (int)s.w);                       // :253
        if (n < p.M && carve(c, s, (uint32_t)loc, n == 0)) {                // :257
            pieces_rev[n] = (uint8_t)piece;                                 // insert(0, ...) (:258-260), reversed
            sol_rev[2 * n] = (uint8_t)rotations;
            sol_rev[2 * n + 1] = (uint8_t)loc;
            ++n;
            const uint32_t low = bag & ((1u << (3 * idx)) - 1u);            // delete_index (:262)
            bag = low | ((bag >> (3 * (idx + 1))) << (3 * idx));
            --n_bag;
        } else if (n >= p.M || ++attempts > 40) {                           // :268, add_attempt (:121-123)
            attempts = 0;                                                   // load_checkpoint (:128-137)
            if (n_cp > 1 && uses > 10) { --n_cp; uses = 0; }
            else ++uses;
            const uint32_t* e = cps + (n_cp - 1) * 11;
#pragma unroll
            for (int x = 0; x < kCols; ++x) c[x] = e[x];                    // :275-276
            n = (int)e[10];
            bag = kFullBag; n_bag = 7;                                      // :278
        }
    }

    if (p.status) p.status[k] = capped ? 1 : 0;
    if (p.solution_len) p.solution_len[k] = n;
    uint8_t* out = p.pieces + k * (p.M + 1);
    for (int i = 0; i < n; ++i) {                                           // un-reverse
        out[i] = pieces_rev[n - 1 - i];
        if (p.solution) {
            p.solution[(k * p.M + i) * 2 + 0] = sol_rev[2 * (n - 1 - i)];
            p.solution[(k * p.M + i) * 2 + 1] = sol_rev[2 * (n - 1 - i) + 1];
        }
    }
    int need = p.M - n + 1;                                                 // :281-284, get_random_sequence (:95-102)
    while (need > 0) {
        if (n_bag == 0) { bag = kFullBag; n_bag = 7; }
        for (int i = n_bag - 1; i >= 1; --i) {                              // random.shuffle (:93)
            const int j = randint(0, i);
            const uint32_t vi = (bag >> (3 * i)) & 7u, vj = (bag >> (3 * j)) & 7u;
            bag = (bag & ~(7u << (3 * i))) | (vj << (3 * i));
            bag = (bag & ~(7u << (3 * j))) | (vi << (3 * j));
        }
        const int take = need < n_bag ? need : n_bag;
        for (int i = 0; i < take; ++i) out[n + i] = (uint8_t)((bag >> (3 * i)) & 7u);
        n += take; need -= take;
        n_bag = 0;                                                          // :100
    }
    uint16_t* rows = p.rows + k * kRows;
#pragma unroll
    for (int r = 0; r < kRows; ++r) rows[r] = (uint16_t)row_of_cols(c, r);
}

}  // namespace
}  // namespace tpl

using namespace tpl;

extern "C" size_t tpl_generate_configs_device_work_bytes(int32_t M, int64_t count) {
    if (M < 1 || count < 1) return 0;
    const size_t stride = (256 + 512 + (size_t)(M / 7 + 3) * 44 + 63) / 64 * 64;
    return stride * (size_t)count;
}

extern "C" int tpl_generate_configs_device(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count,
                                           int64_t max_iters, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                                           int32_t* solution_len, int32_t* status, void* work, size_t work_bytes,
                                           void* stream) {
    if (L < 1 || L > 16) return fail_msg(TPL_ERR_ARG, "carving needs 1 <= L <= 16 (got %d)", L);
    if (M < 1 || M > 254) return fail_msg(TPL_ERR_ARG, "M=%d out of range [1, 254]", M);
    if (count < 1 || first < 0 || !rows || !pieces) return fail_msg(TPL_ERR_ARG, "bad count / first / output pointers");
    const size_t need = tpl_generate_configs_device_work_bytes(M, count);
    if (!work || work_bytes < need) return fail_msg(TPL_ERR_ARG, "work has %zu bytes, need %zu", work_bytes, need);
    if (((uintptr_t)work & 3u) != 0) return fail_msg(TPL_ERR_ARG, "work must be 4-byte aligned");
    CarveArgs p{};
    p.L = L; p.M = M; p.seed = seed; p.first = first; p.count = count; p.max_iters = max_iters;
    p.rows = rows; p.pieces = pieces; p.solution = solution; p.solution_len = solution_len; p.status = status;
    p.work = (uint8_t*)work; p.work_stride = (int64_t)(need / (size_t)count);
    hipLaunchKernelGGL(carve_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}
