// forward_device.hip -- the reference's second configuration supplier (game/tetris_algo_main/) on the GPU, one game per lane:
// random fill of the board to a height limit (TetrisGameGenerator.py:72-86), a 7-bag piece sequence (:91-106) and the greedy
// depth-first solver that keeps only winnable games (TetrisSolver.py:112-163).  Same games, seed for seed, as the host form
// in forward_generator.hip and as the reference: every random decision is drawn from CPython's `random` stream
// (random.seed(seed); choice / randint / shuffle), so a lane carries its own MT19937.
//
// What a lane keeps, and where:
//   * the MT19937 state (624 words) and the solver's frames in a slice of `work` memory, LANE-MAJOR within the wave's region
//     (word j of lane l at region[64 j + l]): seeding and the twist walk the state in the same order on every lane, so those
//     2,500 read-modify-writes per lane are coalesced 256-byte accesses;
//   * the board as twenty row words in LDS, lane-major (row y of lane l at word 64 y + l: whatever row a lane indexes, its bank
//     is its lane number) -- the generator and the solver place pieces by ROW masks and clear EVERY full row of the board
//     (TetrisSolver.py:62-76), unlike Tetris.move, so the column form of the step kernels does not serve here;
//   * the solver's recursion (TetrisSolver.py:112-163 calls itself once per piece of the sequence) as an explicit stack: frame
//     d = the board and line count BEFORE piece d was placed (ten words of two rows each) and (lines, letter, rotation, column)
//     in an eleventh.  Returning from a failed child is "pop, restore, count the failure the reference counts there".
//
// This is not a fast kernel and does not need to be: the reference runs this supplier over the SAME hundred seeds 0..99 for
// every batch (tetris_algo_main/main.py:39-40,63), and its yield is 22 % at L = 5 and nothing at L = 10.  It exists so that
// the whole supply side of a batched environment -- carved and forward-generated games blended into one pool, as the
// reference's two producers feed one queue (game/tetris.py:195-211, 482-488) -- can be produced without the host, and so
// that SURVEY 8(f-4) has a HIP form whose output is pinned by the reference's own games (tests/golden/forward_L*.npz).
#include "tpl_internal.h"

namespace tpl {
namespace {

constexpr int kH = 20, kW = 10;
constexpr int kMtWords = 624;
constexpr int kFrameWords = 11;
constexpr uint32_t kFullRow = 0x3FFu;

// tetromino_shapes (TetrisGameGenerator.py:6-13 = TetrisSolver.py:5-13), letters in the order of tetrominoes_names (:23)
// I J L O S T Z; one word per (letter, rotation): row i (top -> bottom, bit x = column x) in nibble i, height << 16, width << 20
constexpr uint32_t fs(int h, int w, int r0, int r1 = 0, int r2 = 0, int r3 = 0) {
    return (uint32_t)r0 | (uint32_t)r1 << 4 | (uint32_t)r2 << 8 | (uint32_t)r3 << 12 | (uint32_t)h << 16 | (uint32_t)w << 20;
}
__device__ __constant__ const uint32_t kForwardShape[7][4] = {
    /* I */ {fs(1, 4, 15), fs(4, 1, 1, 1, 1, 1), 0, 0},
    /* J */ {fs(2, 3, 1, 7), fs(3, 2, 3, 1, 1), fs(2, 3, 7, 4), fs(3, 2, 2, 2, 3)},
    /* L */ {fs(2, 3, 4, 7), fs(3, 2, 1, 1, 3), fs(2, 3, 7, 1), fs(3, 2, 3, 2, 2)},
    /* O */ {fs(2, 2, 3, 3), 0, 0, 0},
    /* S */ {fs(2, 3, 6, 3), fs(3, 2, 1, 3, 2), 0, 0},
    /* T */ {fs(2, 3, 2, 7), fs(3, 2, 1, 3, 1), fs(2, 3, 7, 2), fs(3, 2, 2, 3, 2)},
    /* Z */ {fs(2, 3, 3, 6), fs(3, 2, 2, 3, 1), 0, 0},
};
__device__ __constant__ const uint8_t kForwardRot[8] = {2, 4, 4, 1, 2, 4, 2, 0};
// piece_translations (game/tetris.py:8-16): letter -> id used by Tetris.move
__device__ __constant__ const uint8_t kLetterToId[8] = {/*I*/ 0, /*J*/ 2, /*L*/ 1, /*O*/ 6, /*S*/ 4, /*T*/ 3, /*Z*/ 5, 0};

struct ForwardArgs {
    int32_t L, M, height_max, max_attempts;
    int64_t count;
    const uint64_t* seeds;
    uint16_t* rows;          // [count][20]
    uint8_t* sequence;       // [count][M] ids of Tetris.move
    uint8_t* winnable;       // [count]
    int32_t* failed;         // [count] or null
    uint8_t* solution;       // [count][M][2] or null
    uint8_t* stack;          // [count][M][3] or null
    int32_t* sol_len;        // [count] or null
    uint32_t* work;          // [waves][words_per_lane][64]
    int64_t words_per_lane;
    uint8_t move_rot[28];    // (letter, rotation) -> the rotation count of Tetris.move that shows the same shape
};

// ---- CPython's random stream (py_random.h, on a lane-major state in global memory) ----------------------------------
struct Mt {
    uint32_t* s;             // word i of this lane's state at s[64 i]
    int idx;
};
__device__ __forceinline__ uint32_t& mtw(const Mt& m, int i) { return m.s[(size_t)i * 64]; }

__device__ void mt_seed(Mt& m, uint64_t seed) {                   // random.seed(int): init_by_array over the 32-bit digits
    uint32_t prev = 19650218u;
    mtw(m, 0) = prev;
    for (int i = 1; i < kMtWords; ++i) { prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i; mtw(m, i) = prev; }
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int len = key[1] ? 2 : 1;
    int i = 1, j = 0;
    prev = mtw(m, 0);
    for (int k = kMtWords; k; --k) {
        prev = (mtw(m, i) ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        mtw(m, i) = prev;
        if (++i >= kMtWords) { mtw(m, 0) = prev; i = 1; }
        if (++j >= len) j = 0;
    }
    for (int k = kMtWords - 1; k; --k) {
        prev = (mtw(m, i) ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        mtw(m, i) = prev;
        if (++i >= kMtWords) { mtw(m, 0) = prev; i = 1; }
    }
    mtw(m, 0) = 0x80000000u;
    m.idx = kMtWords;
}
__device__ uint32_t mt_next(Mt& m) {
    if (m.idx >= kMtWords) {                                      // the twist, in place and in the generator's own order
        for (int k = 0; k < kMtWords; ++k) {
            const uint32_t y = (mtw(m, k) & 0x80000000u) | (mtw(m, k + 1 < kMtWords ? k + 1 : 0) & 0x7FFFFFFFu);
            mtw(m, k) = mtw(m, k + 397 < kMtWords ? k + 397 : k + 397 - kMtWords) ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
        }
        m.idx = 0;
    }
    uint32_t y = mtw(m, m.idx++);
    y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
    return y;
}
__device__ uint32_t mt_randbelow(Mt& m, uint32_t n) {             // _randbelow_with_getrandbits: k = n.bit_length()
    const int k = 32 - __builtin_clz(n);
    uint32_t r;
    do r = mt_next(m) >> (32 - k); while (r >= n);
    return r;
}
__device__ __forceinline__ int mt_randint(Mt& m, int lo, int hi) { return lo + (int)mt_randbelow(m, (uint32_t)(hi - lo + 1)); }

// ---- the board: twenty row words of this lane in LDS, at b[64 y] -----------------------------------------------------
struct Shape { uint32_t rows; int h, w; };
__device__ __forceinline__ Shape shape_of(int letter, int rotation) {
    const uint32_t e = kForwardShape[letter][rotation];
    return Shape{e & 0xFFFFu, (int)((e >> 16) & 7u), (int)((e >> 20) & 7u)};
}
__device__ __forceinline__ uint32_t shape_row(const Shape& s, int i, int col) { return ((s.rows >> (4 * i)) & 0xFu) << col; }

__device__ __forceinline__ bool overlaps(const uint32_t* b, const Shape& s, int r, int col) {
    bool hit = false;
    for (int i = 0; i < s.h; ++i) hit |= (b[64 * (r + i)] & shape_row(s, i, col)) != 0u;
    return hit;
}
// calculate_placement_height (TetrisGameGenerator.py:60-69, TetrisSolver.py:100-109): rows descended from the top until the
// shape would overlap or leave the board
__device__ int placement_height(const uint32_t* b, const Shape& s, int col) {
    int height = 0;
    while (height + s.h <= kH && !overlaps(b, s, height, col)) ++height;
    return height;
}
// place_tetromino + clear_lines (TetrisGameGenerator.py:43-57, TetrisSolver.py:62-76): drop from row 0, lock, clear EVERY
// full row of the board; returns rows cleared
__device__ int place(uint32_t* b, const Shape& s, int col) {
    const int r = placement_height(b, s, col) - 1;
    for (int i = 0; i < s.h; ++i) b[64 * (r + i)] |= shape_row(s, i, col);
    int w = kH - 1;
    for (int y = kH - 1; y >= 0; --y) {
        const uint32_t v = b[64 * y];
        if (v != kFullRow) { b[64 * w] = v; --w; }
    }
    const int cleared = w + 1;
    for (; w >= 0; --w) b[64 * w] = 0u;
    return cleared;
}

__global__ __launch_bounds__(64) void forward_kernel(const ForwardArgs p) {
    __shared__ uint32_t s_rows[kH][64];
    const int lane = (int)threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    if (k >= p.count) return;                                     // no barrier anywhere below: a wave runs in lockstep with itself
    uint32_t* b = &s_rows[0][lane];
    uint32_t* slice = p.work + (size_t)blockIdx.x * (size_t)p.words_per_lane * 64 + lane;
    Mt rnd{slice, kMtWords};
    uint32_t* frames = slice + (size_t)kMtWords * 64;                       // frame d, word j at frames[64 (11 d + j)]
    uint32_t* letters = frames + (size_t)kFrameWords * p.M * 64;            // four letters a word
    const int M = p.M;

    // ---- TetrisGameGenerator.__init__ (:15-29): seed, fill_grid, generate_tetromino_sequence
    mt_seed(rnd, p.seeds[k]);
    for (int y = 0; y < kH; ++y) b[64 * y] = 0u;
    for (;;) {                                                              // fill_grid (:72-86)
        const int t = (int)mt_randbelow(rnd, 7u);                           // random.choice(self.tetrominoes_names)
        const int rot = mt_randint(rnd, 0, (int)kForwardRot[t] - 1);
        const Shape s = shape_of(t, rot);
        const int col = mt_randint(rnd, 0, kW - s.w);
        if (!overlaps(b, s, 0, col)) {                                      // is_valid_move(shape, 0, col) (:31-41)
            const int height = placement_height(b, s, col);
            if (kH + 1 - height <= p.height_max) place(b, s, col);
            else break;
        }
    }
    for (int y = 0; y < kH; ++y) p.rows[k * kH + y] = (uint16_t)b[64 * y];
    // generate_tetromino_sequence (:91-106): shuffled 7-bags (the S/Z re-shuffle condition of :100 compares two DIFFERENT
    // entries of a bag for equality and so never fires)
    for (int produced = 0; produced < M;) {
        uint32_t bag = 0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15 | 6u << 18;     // seven 3-bit entries
        for (int i = 6; i >= 1; --i) {                                      // random.shuffle
            const int j = (int)mt_randbelow(rnd, (uint32_t)(i + 1));
            const uint32_t a = (bag >> (3 * i)) & 7u, c = (bag >> (3 * j)) & 7u;
            bag = (bag & ~(7u << (3 * i)) & ~(7u << (3 * j))) | c << (3 * i) | a << (3 * j);
        }
        for (int q = 0; q < 7 && produced < M; ++q, ++produced) {
            const uint32_t letter = (bag >> (3 * q)) & 7u;
            uint32_t& word = letters[64 * (produced >> 2)];
            word = (produced & 3) == 0 ? letter : word | letter << (8 * (produced & 3));
            p.sequence[k * M + produced] = kLetterToId[letter];
        }
    }
    auto letter_at = [&](int i) { return (int)((letters[64 * (i >> 2)] >> (8 * (i & 3))) & 0xFFu); };

    // ---- TetrisSolver.solve (:112-163), the recursion as a stack of frames.  `rotation` = the rotation the frame at `depth`
    // tries next; a frame whose rotations are used up (or that finds failed >= max_attempts, :119-121) returns False to its
    // caller, which restores the board it saved (:141-150) and counts what the reference counts there (:157-160).
    int depth = 0, next = 1, lines = 0, failed = 0, rotation = 0, current = letter_at(0);
    bool won = false;
    for (;;) {
        bool returns = rotation >= (int)kForwardRot[current];
        Shape s{};
        int col = 0;
        if (!returns) {
            s = shape_of(current, rotation);
            // evaluate_columns(...)[:1] (:90-98): the column with the greatest placement height, leftmost on ties
            int best = -1;
            for (int c = 0; c <= kW - s.w; ++c) {
                const int ph = placement_height(b, s, c);
                if (ph > best) { best = ph; col = c; }
            }
            returns = failed >= p.max_attempts;                             // :119-121
        }
        if (returns) {
            if (depth == 0) break;                                          // the outermost call returns False
            --depth;                                                        // back in the caller, behind `if self.solve(...)`
            --next;
            uint32_t* f = frames + (size_t)64 * kFrameWords * depth;
            for (int j = 0; j < 10; ++j) {                                  // self.board = saved (:148-150)
                const uint32_t two = f[64 * j];
                b[64 * (2 * j)] = two & 0xFFFFu;
                b[64 * (2 * j + 1)] = two >> 16;
            }
            const uint32_t tag = f[64 * 10];
            lines = (int)(tag & 0xFFu);
            current = (int)((tag >> 8) & 0xFFu);
            rotation = (int)((tag >> 16) & 0xFFu);
            const int at = (int)(tag >> 24);
            // :157-160  `rotation == len(current) - 1`: current is a one-letter string, so this is rotation == 0
            if (rotation == 0 && at == kW - shape_of(current, 0).w) ++failed;
            ++rotation;
            continue;
        }
        uint32_t* f = frames + (size_t)64 * kFrameWords * depth;
        for (int j = 0; j < 10; ++j) f[64 * j] = b[64 * (2 * j)] | b[64 * (2 * j + 1)] << 16;     // saved = board (:123-124)
        f[64 * 10] = (uint32_t)lines | (uint32_t)current << 8 | (uint32_t)rotation << 16 | (uint32_t)col << 24;
        const int saved_lines = lines;
        auto restore = [&] {
            for (int j = 0; j < 10; ++j) {
                const uint32_t two = f[64 * j];
                b[64 * (2 * j)] = two & 0xFFFFu;
                b[64 * (2 * j + 1)] = two >> 16;
            }
            lines = saved_lines;
        };
        if (overlaps(b, s, 0, col)) { ++failed; ++rotation; continue; }     // :127-129
        lines += place(b, s, col);                                          // :125-126
        if (b[0] != 0u) { restore(); ++failed; ++rotation; continue; }      // is_game_over (:87-88), :131-135
        if (lines >= p.L) { won = true; break; }                            // :137-139 (the frame at `depth` is the last move)
        if (next < M) {                                                     // :141-150: the next piece of the sequence
            current = letter_at(next++);
            ++depth;
            rotation = 0;
            continue;
        }
        restore();                                                          // :152-155
        ++failed;
        if (rotation == 0 && col == kW - s.w) ++failed;                     // :157-160
        ++rotation;
    }
    p.winnable[k] = won ? 1 : 0;
    if (p.failed) p.failed[k] = failed;
    const int len = won ? depth + 1 : 0;
    if (p.sol_len) p.sol_len[k] = len;
    for (int i = 0; i < M; ++i) {
        uint32_t letter = 0, rot = 0, at = 0, mrot = 0;
        if (i < len) {
            const uint32_t tag = frames[(size_t)64 * (kFrameWords * i + 10)];
            letter = (tag >> 8) & 0xFFu; rot = (tag >> 16) & 0xFFu; at = tag >> 24;
            mrot = p.move_rot[letter * 4 + rot];
        }
        if (p.stack) { p.stack[(k * M + i) * 3 + 0] = (uint8_t)letter; p.stack[(k * M + i) * 3 + 1] = (uint8_t)rot; p.stack[(k * M + i) * 3 + 2] = (uint8_t)at; }
        if (p.solution) { p.solution[(k * M + i) * 2 + 0] = (uint8_t)mrot; p.solution[(k * M + i) * 2 + 1] = (uint8_t)at; }
    }
}

size_t words_per_lane(int32_t M) { return (size_t)kMtWords + (size_t)kFrameWords * M + ((size_t)M + 3) / 4; }

}  // namespace
}  // namespace tpl

using namespace tpl;

extern "C" size_t tpl_forward_generate_device_work_bytes(int32_t M, int64_t count) {
    if (M < 1 || count < 1) return 0;
    return words_per_lane(M) * 64 * sizeof(uint32_t) * (size_t)((count + 63) / 64);
}

extern "C" int tpl_forward_generate_device(int32_t L, int32_t M, int32_t initial_height_max, int32_t max_attempts,
                                           const uint64_t* seeds, int64_t count, uint16_t* rows, uint8_t* sequence,
                                           uint8_t* winnable, int32_t* failed_attempts, uint8_t* solution, uint8_t* solver_stack,
                                           int32_t* solution_len, void* work, size_t work_bytes, void* stream) {
    if (L < 1 || L > 250) return fail_msg(TPL_ERR_ARG, "L=%d out of range [1, 250]", L);
    if (M < 1 || M > 254) return fail_msg(TPL_ERR_ARG, "M=%d out of range [1, 254]", M);
    if (initial_height_max < 1 || initial_height_max > 16) return fail_msg(TPL_ERR_ARG, "initial_height_max must be in [1, 16]");
    if (max_attempts < 1) return fail_msg(TPL_ERR_ARG, "max_attempts must be positive");
    if (!seeds || count < 1 || count > 0x7FFFFFFF || !rows || !sequence || !winnable)
        return fail_msg(TPL_ERR_ARG, "bad seeds / count / output pointers");
    const size_t need = tpl_forward_generate_device_work_bytes(M, count);
    if (!work || work_bytes < need) return fail_msg(TPL_ERR_ARG, "work has %zu bytes, need %zu", work_bytes, need);
    if (((uintptr_t)work & 3u) != 0) return fail_msg(TPL_ERR_ARG, "work must be 4-byte aligned");
    ForwardArgs p{};
    p.L = L; p.M = M; p.height_max = initial_height_max; p.max_attempts = max_attempts; p.count = count;
    p.seeds = seeds; p.rows = rows; p.sequence = sequence; p.winnable = winnable; p.failed = failed_attempts;
    p.solution = solution; p.stack = solver_stack; p.sol_len = solution_len;
    p.work = (uint32_t*)work; p.words_per_lane = (int64_t)words_per_lane(M);
    for (int letter = 0; letter < 7; ++letter)
        for (int r = 0; r < 4; ++r) p.move_rot[letter * 4 + r] = (uint8_t)forward_move_rotations(letter, r);
    hipLaunchKernelGGL(forward_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}
