// carve_generator.hip -- supply of prescribed initial configurations: the reference's carving generator
// (game/tetris.py:64-137 RandomPieceGenerator/CheckpointManager, :226-284 _generate_initial_config, :286-352
// carve/calculate_carve) as a multi-threaded native producer on the host cores.  Host code only: the search is
// serial with backtracking per configuration, and configurations are independent, so one configuration per
// thread task fills a pool of a million entries in about a second on the GPU box's cores.
//
// The board is kept in the same column form as on the device (ten 20-bit words, bit r = row r), with the same
// encoded shape table (tpl_device.h), so a column's top is one count-trailing-zeros and a carve is one AND per
// piece column.  Random decisions are counter-based (decision_stream() / decision(), tpl_device.h) and a configuration is
// the outcome of the first attempt that ends within the cut-off (the restart rule, tpl_device.h) -- independent of the
// thread count, and the same configuration the device generator builds.
#include "tpl_internal.h"
#include "py_random.h"

#include <sched.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

namespace tpl {

namespace {

// The two sources of decisions.  A trip of the search asks for a piece and its rotations, then -- the piece's width known --
// for a location; the padding asks for single numbers.
struct Decisions {                                       // the build's counter-based stream: one word per trip (tpl_device.h)
    DecisionStream s;
    uint32_t word = 0;
    Decisions(uint64_t seed, uint64_t index, uint32_t attempt) : s(decision_stream(seed, index, attempt)) {}
    void piece_and_rotations(int n_bag, int& idx, int& rotations) {
        word = decision_word(s);
        idx = word_bag_index(word, n_bag);
        rotations = word_rotations(word);
    }
    int location(int places) { return word_location(word, places); }
    int randint(int lo, int hi) { return decision(s, lo, hi); }
};
struct PyDecisions {                                     // CPython's stream, drawn from in the reference's order (:85, :250, :253)
    PyRandom r;
    explicit PyDecisions(uint64_t seed) : r(seed) {}
    void piece_and_rotations(int n_bag, int& idx, int& rotations) {
        idx = r.randint(0, n_bag - 1);
        rotations = r.randint(0, 3);
    }
    int location(int places) { return r.randint(0, places - 1); }
    int randint(int lo, int hi) { return r.randint(lo, hi); }
};

struct Shape {
    int h, w;
    uint32_t col[4];   // column c of the piece as a bit-per-row pattern
    int revtopo[4];
};

inline Shape shape_of(int piece, int rotations) {
    const ShapeWord sw = kShapeTableHost[piece * 4 + (rotations & 3)];   // get_tetromino (:60-61)
    Shape s;
    s.w = (int)((sw.x >> 16) & 7u);
    s.h = (int)((sw.x >> 19) & 7u);
    for (int c = 0; c < 4; ++c) {
        s.col[c] = (sw.x >> (4 * c)) & 0xFu;
        s.revtopo[c] = c < s.w ? 3 - (int)((sw.y >> (8 * c)) & 0xFFu) : 0;
    }
    return s;
}

struct Game {
    uint32_t col[kCols];
    uint8_t pieces[256];
    uint8_t sol[256][2];
    int n = 0;
};

// calculate_drop_deltas + calculate_drop (:424-433): returns drop, and the first column that attains the minimum
inline int drop_of(const uint32_t* col, int loc, const Shape& s, int* argmin) {
    int best = 1 << 20, at = 0;
    for (int c = 0; c < s.w; ++c) {
        const int top = __builtin_ctz(col[loc + c] | (1u << kRows));
        const int d = top - s.revtopo[c];
        if (d < best) { best = d; at = c; }
    }
    if (argmin) *argmin = at;
    return best - 1;
}

// calculate_carve (:313-352)
inline bool try_carve(uint32_t* col, int drop, int loc, const Shape& s, bool allow_partial) {
    if (drop + s.h > kRows || drop < 0) return false;                       // :317-318
    if (!allow_partial)                                                     // :321-329 every piece cell is filled
        for (int c = 0; c < s.w; ++c)
            if ((col[loc + c] & (s.col[c] << drop)) != (s.col[c] << drop)) return false;
    uint32_t saved[4];
    for (int c = 0; c < s.w; ++c) {                                         // :332-337
        saved[c] = col[loc + c];
        col[loc + c] &= ~(s.col[c] << drop);
    }
    if (drop_of(col, loc, s, nullptr) != drop) {                            // :341-349 must land where it was carved
        for (int c = 0; c < s.w; ++c) col[loc + c] = saved[c];
        return false;
    }
    return true;
}

// carve (:286-311)
inline bool carve(uint32_t* col, int piece, int rotations, int loc, bool allow_partial) {
    const Shape s = shape_of(piece, rotations);
    int at;
    int drop = drop_of(col, loc, s, &at);
    drop += s.revtopo[at] + 1;                                              // :298-301 push the piece into the stack
    const int tries = allow_partial ? s.h : 1;                              // :304
    for (int k = 0; k < tries; ++k, --drop)
        if (try_carve(col, drop, loc, s, allow_partial)) return true;
    return false;
}

// _generate_initial_config (:226-284) for one configuration.  Returns false if max_iters (> 0) was reached.
template <typename Random>
bool generate_one(int L, int M, Random& rnd, int64_t max_iters, uint16_t* rows_out, uint8_t* pieces_out,
                  uint8_t* sol_out, int32_t* sol_len, const std::atomic<bool>* stop = nullptr) {
    Game g;
    const uint32_t filled = L >= kRows ? kColMask : (((1u << L) - 1u) << (kRows - L));
    for (int c = 0; c < kCols; ++c) g.col[c] = filled;                      // :228 L full rows
    uint8_t bag[7];
    int n_bag = 0;
    // CheckpointManager (:111-137).  A checkpoint is added only when a fresh bag is opened, i.e. after at least
    // seven successful carves on top of the previous one (or a reload), so M/7 + 2 entries always suffice; kept on
    // the stack so that hundreds of generator threads do not meet in the allocator.
    constexpr int kMaxCheckpoints = 254 / 7 + 3;
    Game checkpoints[kMaxCheckpoints];
    int n_cp = 0;
    int attempts = 0, uses = 0;
    int64_t iters = 0;

    auto bottom_cells = [&] { int k = 0; for (int c = 0; c < kCols; ++c) k += (g.col[c] >> (kRows - 1)) & 1u; return k; };
    while (bottom_cells() > 8) {                                            // :234
        if (max_iters > 0 && iters++ >= max_iters) return false;
        if (stop && (iters & 4095) == 0 && stop->load(std::memory_order_relaxed)) return false;   // the pilot's other attempts
        bool fresh_bag = false;                                             // _regenerate (:71-81)
        if (n_bag == 0) { for (int k = 0; k < 7; ++k) bag[k] = (uint8_t)k; n_bag = 7; fresh_bag = true; }
        int idx, rotations;
        rnd.piece_and_rotations(n_bag, idx, rotations);                     // :85, :250
        const int piece = bag[idx];
        if (fresh_bag && n_cp < kMaxCheckpoints) checkpoints[n_cp++] = g;   // :239-247
        const int width = shape_of(piece, rotations).w;
        const int loc = rnd.location(kCols - width + 1);                    // :253
        if (g.n < M && carve(g.col, piece, rotations, loc, g.n == 0)) {     // :257
            std::memmove(g.pieces + 1, g.pieces, (size_t)g.n);              // insert(0, ...) (:258-260)
            std::memmove(g.sol + 1, g.sol, (size_t)g.n * 2);
            g.pieces[0] = (uint8_t)piece;
            g.sol[0][0] = (uint8_t)rotations; g.sol[0][1] = (uint8_t)loc;
            ++g.n;
            std::memmove(bag + idx, bag + idx + 1, (size_t)(n_bag - idx - 1));   // delete_index (:262)
            --n_bag;
        } else if (g.n >= M || ++attempts > 40) {                           // :268, add_attempt (:121-123)
            attempts = 0;                                                   // load_checkpoint (:128-137)
            if (n_cp > 1 && uses > 10) { --n_cp; uses = 0; }
            else ++uses;
            g = checkpoints[n_cp - 1];                                      // :275-276
            for (int k = 0; k < 7; ++k) bag[k] = (uint8_t)k;                // :278
            n_bag = 7;
        }
    }
    if (sol_len) *sol_len = g.n;
    if (sol_out) std::memcpy(sol_out, g.sol, (size_t)g.n * 2);
    int need = M - g.n + 1;                                                 // :281-284 pad to M+1 pieces
    while (need > 0) {                                                      // get_random_sequence (:95-102)
        if (n_bag == 0) { for (int k = 0; k < 7; ++k) bag[k] = (uint8_t)k; n_bag = 7; }
        for (int i = n_bag - 1; i >= 1; --i) {                              // random.shuffle (:93)
            const int j = rnd.randint(0, i);
            const uint8_t t = bag[i]; bag[i] = bag[j]; bag[j] = t;
        }
        const int take = need < n_bag ? need : n_bag;
        std::memcpy(g.pieces + g.n, bag, (size_t)take);
        g.n += take; need -= take;
        n_bag = 0;                                                          // :100
    }
    std::memcpy(pieces_out, g.pieces, (size_t)(M + 1));
    for (int r = 0; r < kRows; ++r) {                                       // columns -> interchange rows
        uint32_t v = 0;
        for (int c = 0; c < kCols; ++c) v |= ((g.col[c] >> r) & 1u) << c;
        rows_out[r] = (uint16_t)v;
    }
    return true;
}

}  // namespace
}  // namespace tpl


// Host threads worth starting when the caller names none: the affinity mask, capped by the cgroup CPU quota when there is one
// (the rule of _lib.cpu_budget() on the Python side; hardware_concurrency() reports every CPU of the machine, a container's
// share may be a sixteenth of that).
int tpl::host_cpu_budget() {
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = (int)std::thread::hardware_concurrency();
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
            const long long q = (std::atoll(quota) + period / 2) / period;
            if (q >= 1 && q < n) n = (int)q;
        }
        std::fclose(f);
    }
    return n < 1 ? 1 : n;
}

// A few configurations tried on the host before a batch goes out (tpl_internal.h).  The restart rule bounds every
// configuration at some 1,500 base cut-offs; a batch of an (L, M) that cannot be carved at all would spend that on EVERY
// configuration -- minutes of host threads, or a kernel that runs for minutes -- so the generators first run the rule on
// fixed pilot configurations (seed 0x7E7215, indices 0..kPilots-1; the verdict is kept per (L, M, cut-off)).  The batch is
// refused only when EVERY pilot runs into every cut-off: under a marginal cut-off, where a configuration in a hundred caps
// (the callers' status[] reports those one by one), a single unlucky pilot must not refuse the ninety-nine.  Per pilot:
// attempts 0-11 one after the other (a normal (L, M) is through with the first attempt of the first pilot: a third of a
// millisecond at L = 10), the doubled ones side by side on the host's CPU budget.
int tpl::carve_pilot(int32_t L, int32_t M, int64_t cutoff) {
    if (M < carve_fewest_pieces(L))
        return fail_msg(TPL_ERR_ARG, "L=%d cannot be carved with M=%d pieces: two columns of %d cells need at least %d", L, M, L,
                        carve_fewest_pieces(L));
    static std::mutex mu;
    static std::map<std::tuple<int32_t, int32_t, int64_t>, bool> verdicts;
    const auto key = std::make_tuple(L, M, cutoff);
    bool known = false, ok = false;
    {
        std::lock_guard<std::mutex> hold(mu);
        const auto it = verdicts.find(key);
        if (it != verdicts.end()) { known = true; ok = it->second; }
    }
    constexpr int kPilots = 4;
    if (!known) {
        constexpr uint64_t kPilotSeed = 0x7E7215ULL;
        uint16_t rows[kRows];
        uint8_t pieces[256];
        for (uint64_t pilot = 0; pilot < (uint64_t)kPilots && !ok; ++pilot) {
            for (int a = 0; a < 12 && !ok; ++a) {
                Decisions rnd(kPilotSeed, pilot, (uint32_t)a);
                ok = generate_one(L, M, rnd, carve_cutoff(L, cutoff, a), rows, pieces, nullptr, nullptr);
            }
            if (ok) break;
            std::atomic<bool> found{false};
            std::atomic<int> next{12};
            auto work = [&] {
                uint16_t r[kRows];
                uint8_t p[256];
                for (;;) {
                    const int a = next.fetch_add(1);
                    if (a >= kCarveAttempts || found.load()) return;
                    Decisions rnd(kPilotSeed, pilot, (uint32_t)a);
                    if (generate_one(L, M, rnd, carve_cutoff(L, cutoff, a), r, p, nullptr, nullptr, &found)) found.store(true);
                }
            };
            int threads = host_cpu_budget();
            threads = threads > kCarveAttempts - 12 ? kCarveAttempts - 12 : threads;
            std::vector<std::thread> pool;
            for (int t = 1; t < threads; ++t) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
            ok = found.load();
        }
        std::lock_guard<std::mutex> hold(mu);
        verdicts[key] = ok;
    }
    if (!ok)
        return fail_msg(TPL_ERR_STATE, "L=%d M=%d: none of the %d pilot configurations finished within %d attempts (base cut-off %lld "
                        "trips, the last ones at %lld): not finished within the restart rule's bound -- a larger `cutoff` searches on",
                        L, M, kPilots, kCarveAttempts, (long long)carve_cutoff(L, cutoff, 0),
                        (long long)carve_cutoff(L, cutoff, kCarveAttempts - 1));
    return TPL_OK;
}

// `build(k)` fills configuration k's outputs and says whether it finished
template <typename BuildOne>
static int run_generator(int32_t L, int32_t M, int64_t count, int32_t threads, uint16_t* rows, uint8_t* pieces, BuildOne build) {
    using namespace tpl;
    if (L < 1 || L > 16) return fail_msg(TPL_ERR_ARG, "carving needs 1 <= L <= 16 (got %d)", L);
    if (M < 1 || M > 254) return fail_msg(TPL_ERR_ARG, "M=%d out of range [1, 254]", M);
    if (count < 1 || !rows || !pieces) return fail_msg(TPL_ERR_ARG, "bad count / output pointers");
    if (threads < 1) threads = (int32_t)host_cpu_budget();
    if ((int64_t)threads > count) threads = (int32_t)count;
    std::atomic<int64_t> next{0};
    std::atomic<int64_t> failed{-1};
    auto work = [&] {
        for (;;) {
            const int64_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= count) return;
            if (!build(k)) failed.store(k, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (failed.load() >= 0)
        return fail_msg(TPL_ERR_STATE, "configuration %lld did not finish: every attempt of the restart rule ran into its cut-off "
                        "(L=%d M=%d; a larger `cutoff` searches on)", (long long)failed.load(), L, M);
    return TPL_OK;
}

extern "C" int tpl_generate_configs(int32_t L, int32_t M, uint64_t seed, int64_t first, int64_t count, int32_t threads,
                                    int64_t cutoff, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                                    int32_t* solution_len) {
    using namespace tpl;
    if (first < 0) return fail_msg(TPL_ERR_ARG, "first is negative");
    if (cutoff < 0 || cutoff > ((int64_t)1 << 28)) return fail_msg(TPL_ERR_ARG, "cutoff outside [0, 2^28]");
    if (L >= 1 && L <= 16 && M >= 1 && M <= 254) { const int rc = carve_pilot(L, M, cutoff); if (rc != TPL_OK) return rc; }
    return run_generator(L, M, count, threads, rows, pieces, [=](int64_t k) {
        uint8_t* sol = solution ? solution + k * (int64_t)M * 2 : nullptr;
        int32_t* len = solution_len ? solution_len + k : nullptr;
        for (int a = 0; a < kCarveAttempts; ++a) {          // the restart rule (tpl_device.h): first attempt inside its cut-off
            Decisions rnd(seed, (uint64_t)(first + k), (uint32_t)a);
            if (generate_one(L, M, rnd, carve_cutoff(L, cutoff, a), rows + k * kRows, pieces + k * (M + 1), sol, len)) return true;
        }
        std::memset(rows + k * kRows, 0, kRows * sizeof(uint16_t));         // capped: all-zero outputs
        std::memset(pieces + k * (M + 1), 0, (size_t)M + 1);
        if (len) *len = 0;
        return false;
    });
}

extern "C" int tpl_generate_configs_pyseed(int32_t L, int32_t M, const uint64_t* seeds, int64_t count, int32_t threads,
                                           int64_t max_iters, uint16_t* rows, uint8_t* pieces, uint8_t* solution,
                                           int32_t* solution_len) {
    using namespace tpl;
    if (!seeds) return fail_msg(TPL_ERR_ARG, "seeds is null");
    // CPython's stream is the reference's own: one search, no restarts (max_iters > 0 only bounds it)
    return run_generator(L, M, count, threads, rows, pieces, [=](int64_t k) {
        PyDecisions rnd(seeds[k]);
        return generate_one(L, M, rnd, max_iters, rows + k * kRows, pieces + k * (M + 1),
                            solution ? solution + k * (int64_t)M * 2 : nullptr, solution_len ? solution_len + k : nullptr);
    });
}

// Tetris.carve(piece, rotations, location, allow_partial) (:286-311) on one board in the interchange layout (HOST
// memory, modified in place when the carve succeeds): the inverse of a move, the generator's building block.
extern "C" int tpl_carve(uint16_t* rows, int32_t piece, int32_t rotations, int32_t location, int32_t allow_partial,
                         int32_t* carved) {
    using namespace tpl;
    if (!rows || !carved) return fail_msg(TPL_ERR_ARG, "rows/carved is null");
    if (piece < 0 || piece > 6 || rotations < 0) return fail_msg(TPL_ERR_ARG, "piece %d / rotations %d out of range", piece, rotations);
    const int w = (int)((kShapeTableHost[piece * 4 + (rotations & 3)].x >> 16) & 7u);
    if (location < 0 || location + w > kCols) return fail_msg(TPL_ERR_ARG, "location %d puts the piece outside the board", location);
    uint32_t col[kCols] = {0};
    for (int r = 0; r < kRows; ++r)
        for (int x = 0; x < kCols; ++x) col[x] |= (uint32_t)((rows[r] >> x) & 1u) << r;
    const bool ok = carve(col, piece, rotations, location, allow_partial != 0);
    if (ok)
        for (int r = 0; r < kRows; ++r) {
            uint32_t v = 0;
            for (int x = 0; x < kCols; ++x) v |= ((col[x] >> r) & 1u) << x;
            rows[r] = (uint16_t)v;
        }
    *carved = ok ? 1 : 0;
    return TPL_OK;
}
