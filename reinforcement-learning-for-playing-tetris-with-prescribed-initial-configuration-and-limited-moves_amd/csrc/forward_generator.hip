// forward_generator.hip -- the reference's second configuration supplier (game/tetris_algo_main/): random fill
// of the board to a height limit (TetrisGameGenerator.py:72-86), a 7-bag piece sequence (:91-106) and the
// greedy depth-first solver that keeps only winnable games (TetrisSolver.py:112-163; batch driver main.py:7-31,
// 62-74).  Host code, one game per thread task, like carve_generator.hip.
//
// Every random decision goes through the same CPython-compatible stream the reference uses
// (random.seed(seed); choice / randint / shuffle), so game `seed` here IS game `seed` there: same board, same
// sequence, same verdict, same failed-attempt count, same solution.
//
// This sub-package of the reference has its own shape tables, with a rotation order different from game/tetris.py
// (SURVEY section 2 row 8); solutions are reported both in the solver's own (letter index, rotation, column) and
// translated to move()'s (rotations, location) through piece_translations (game/tetris.py:8-16).
#include "tpl_internal.h"
#include "py_random.h"

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

namespace tpl {
namespace {

constexpr int kH = 20, kW = 10;

struct FShape { int h, w; uint16_t row[4]; };

// tetromino_shapes (TetrisGameGenerator.py:6-13, identical in TetrisSolver.py:5-13); letters in the order of
// tetrominoes_names (:23) I J L O S T Z; row masks top->bottom, bit x = column x
const FShape kForward[7][4] = {
    /* I */ {{1, 4, {15}}, {4, 1, {1, 1, 1, 1}}},
    /* J */ {{2, 3, {1, 7}}, {3, 2, {3, 1, 1}}, {2, 3, {7, 4}}, {3, 2, {2, 2, 3}}},
    /* L */ {{2, 3, {4, 7}}, {3, 2, {1, 1, 3}}, {2, 3, {7, 1}}, {3, 2, {3, 2, 2}}},
    /* O */ {{2, 2, {3, 3}}},
    /* S */ {{2, 3, {6, 3}}, {3, 2, {1, 3, 2}}},
    /* T */ {{2, 3, {2, 7}}, {3, 2, {1, 3, 1}}, {2, 3, {7, 2}}, {3, 2, {2, 3, 2}}},
    /* Z */ {{2, 3, {3, 6}}, {3, 2, {2, 3, 1}}},
};
const int kForwardRot[7] = {2, 4, 4, 1, 2, 4, 2};
// piece_translations (game/tetris.py:8-16): letter -> id used by Tetris.move
const int kLetterToId[7] = {/*I*/ 0, /*J*/ 2, /*L*/ 1, /*O*/ 6, /*S*/ 4, /*T*/ 3, /*Z*/ 5};

struct Grid {
    uint16_t row[kH];
    bool overlaps(const FShape& s, int r, int col) const {
        for (int i = 0; i < s.h; ++i)
            if (row[r + i] & (uint16_t)(s.row[i] << col)) return true;
        return false;
    }
    // calculate_placement_height (TetrisGameGenerator.py:60-69, TetrisSolver.py:100-109): rows descended from the
    // top until the shape would overlap or leave the board
    int placement_height(const FShape& s, int col) const {
        int height = 0;
        while (height + s.h <= kH && !overlaps(s, height, col)) ++height;
        return height;
    }
    // place_tetromino + clear_lines (TetrisGameGenerator.py:43-57, TetrisSolver.py:62-76): drop from row 0, lock,
    // clear EVERY full row of the board; returns rows cleared
    int place(const FShape& s, int col) {
        const int r = placement_height(s, col) - 1;
        for (int i = 0; i < s.h; ++i) row[r + i] |= (uint16_t)(s.row[i] << col);
        uint16_t kept[kH];
        int nk = 0;
        for (int y = 0; y < kH; ++y)
            if (row[y] != 0x3FFu) kept[nk++] = row[y];
        const int cleared = kH - nk;
        for (int y = 0; y < cleared; ++y) row[y] = 0;
        for (int y = 0; y < nk; ++y) row[cleared + y] = kept[y];
        return cleared;
    }
};

// TetrisGameGenerator.__init__ (:15-29): seed, fill_grid, generate_tetromino_sequence
void generate_game(PyRandom& rnd, int M, int initial_height_max, Grid& g, uint8_t* letters) {
    std::memset(g.row, 0, sizeof(g.row));
    for (;;) {                                                       // fill_grid (:72-86)
        const int t = (int)rnd.randbelow(7);                         // random.choice(self.tetrominoes_names)
        const int rot = rnd.randint(0, kForwardRot[t] - 1);
        const FShape& s = kForward[t][rot];
        const int col = rnd.randint(0, kW - s.w);
        if (!g.overlaps(s, 0, col)) {                                // is_valid_move(shape, 0, col) (:31-41)
            const int height = g.placement_height(s, col);
            if (kH + 1 - height <= initial_height_max) g.place(s, col);
            else break;
        }
    }
    // generate_tetromino_sequence (:91-106): shuffled 7-bags; the S/Z re-shuffle condition (:100) compares two
    // DIFFERENT entries of a bag for equality and so never fires
    int produced = 0;
    while (produced < M) {
        uint8_t bag[7] = {0, 1, 2, 3, 4, 5, 6};
        for (int i = 6; i >= 1; --i) {                               // random.shuffle
            const int j = (int)rnd.randbelow((uint32_t)(i + 1));
            const uint8_t tmp = bag[i]; bag[i] = bag[j]; bag[j] = tmp;
        }
        for (int k = 0; k < 7 && produced < M; ++k) letters[produced++] = bag[k];
    }
}

// TetrisSolver.solve (:112-163), iteration for iteration
struct Solver {
    Grid board;
    const uint8_t* seq;
    int n_seq, next = 0;           // the deque: seq[next..)
    int lines = 0, failed = 0, goal, max_attempts;
    uint8_t stack[256][3];
    int depth = 0;

    bool solve(int current) {
        const int n_rot = kForwardRot[current];
        for (int rotation = 0; rotation < n_rot; ++rotation) {
            const FShape& s = kForward[current][rotation];
            // evaluate_columns(...)[:1] (:90-98): the column with the greatest placement height, leftmost on ties
            int col = 0, best = -1;
            for (int c = 0; c <= kW - s.w; ++c) {
                const int ph = board.placement_height(s, c);
                if (ph > best) { best = ph; col = c; }
            }
            if (failed >= max_attempts) return false;                                   // :119-121
            const Grid saved = board;
            const int saved_lines = lines;
            if (!board.overlaps(s, 0, col)) lines += board.place(s, col);                // :125-126
            else { ++failed; continue; }                                                // :127-129
            if (board.row[0] != 0) {                                                    // is_game_over (:87-88)
                board = saved; lines = saved_lines; ++failed; continue;                 // :131-135
            } else if (lines >= goal) {                                                 // :137-139
                push(current, rotation, col);
                return true;
            } else if (next < n_seq) {                                                  // :141-150
                push(current, rotation, col);
                const int nxt = seq[next++];
                if (solve(nxt)) return true;
                --next;
                --depth;
                lines = saved_lines;
                board = saved;
            } else {                                                                    // :152-155
                board = saved; lines = saved_lines; ++failed;
            }
            // :157-160  `rotation == len(current) - 1`: current is a one-letter string, so this is rotation == 0
            if (rotation == 0 && col == kW - s.w) { ++failed; board = saved; lines = saved_lines; }
        }
        return false;
    }
    void push(int letter, int rotation, int col) {
        stack[depth][0] = (uint8_t)letter; stack[depth][1] = (uint8_t)rotation; stack[depth][2] = (uint8_t)col;
        ++depth;
    }
};

// the rotation count of Tetris.move that shows the same shape as the solver's (letter, rotation)
int to_move_rotations(int letter, int rotation) {
    const FShape& s = kForward[letter][rotation];
    const int id = kLetterToId[letter];
    for (int r = 0; r < 4; ++r) {
        const ShapeWord sw = kShapeTableHost[id * 4 + r];
        if ((int)((sw.x >> 16) & 7u) != s.w || (int)((sw.x >> 19) & 7u) != s.h) continue;
        bool same = true;
        for (int i = 0; i < s.h && same; ++i) {
            uint32_t m = 0;
            for (int c = 0; c < 4; ++c) m |= ((sw.x >> (4 * c + i)) & 1u) << c;
            same = m == s.row[i];
        }
        if (same) return r;
    }
    return 0;
}

}  // namespace

// for forward_device.hip: the same translation, as a table it hands to its kernel
int forward_move_rotations(int letter, int rotation) {
    if (letter < 0 || letter > 6 || rotation < 0 || rotation >= kForwardRot[letter]) return 0;
    return to_move_rotations(letter, rotation);
}

}  // namespace tpl

extern "C" int tpl_forward_generate(int32_t L, int32_t M, int32_t initial_height_max, int32_t max_attempts,
                                    const uint64_t* seeds, int64_t count, int32_t threads, uint16_t* rows,
                                    uint8_t* sequence, uint8_t* winnable, int32_t* failed_attempts, uint8_t* solution,
                                    uint8_t* solver_stack, int32_t* solution_len) {
    using namespace tpl;
    if (L < 1 || L > 250) return fail_msg(TPL_ERR_ARG, "L=%d out of range [1, 250]", L);
    if (M < 1 || M > 254) return fail_msg(TPL_ERR_ARG, "M=%d out of range [1, 254]", M);
    if (initial_height_max < 1 || initial_height_max > 16) return fail_msg(TPL_ERR_ARG, "initial_height_max must be in [1, 16]");
    if (max_attempts < 1) return fail_msg(TPL_ERR_ARG, "max_attempts must be positive");
    if (!seeds || count < 1 || !rows || !sequence || !winnable) return fail_msg(TPL_ERR_ARG, "bad seeds / count / output pointers");
    if (threads < 1) threads = (int32_t)host_cpu_budget();
    if ((int64_t)threads > count) threads = (int32_t)count;
    std::atomic<int64_t> next{0};
    auto work = [&] {
        std::vector<uint8_t> letters((size_t)M);
        for (;;) {
            const int64_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= count) return;
            PyRandom rnd(seeds[k]);
            Solver sv;
            generate_game(rnd, M, initial_height_max, sv.board, letters.data());
            std::memcpy(rows + k * kH, sv.board.row, sizeof(sv.board.row));
            for (int i = 0; i < M; ++i) sequence[k * M + i] = (uint8_t)kLetterToId[letters[i]];
            // solve_game (main.py:7-19): TetrisSolver(board, sequence, goal, max_attempts).solve()
            sv.seq = letters.data(); sv.n_seq = M; sv.goal = L; sv.max_attempts = max_attempts;
            const int first = sv.seq[sv.next++];
            const bool ok = sv.solve(first);
            winnable[k] = ok ? 1 : 0;
            if (failed_attempts) failed_attempts[k] = sv.failed;
            const int len = ok ? sv.depth : 0;
            if (solution_len) solution_len[k] = len;
            for (int i = 0; i < len; ++i) {
                if (solver_stack) std::memcpy(solver_stack + (k * M + i) * 3, sv.stack[i], 3);
                if (solution) {
                    solution[(k * M + i) * 2 + 0] = (uint8_t)to_move_rotations(sv.stack[i][0], sv.stack[i][1]);
                    solution[(k * M + i) * 2 + 1] = sv.stack[i][2];
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return TPL_OK;
}
