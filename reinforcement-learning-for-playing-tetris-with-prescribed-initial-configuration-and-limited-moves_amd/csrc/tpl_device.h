// tpl_device.h -- resident board layout and the per-lane move for gfx950 (CDNA4, wave64).
//
// One board per lane.  The board is held TRANSPOSED: ten 20-bit column words (bit r = row r, row 0 = top),
// because the reference's drop rule is column-top based (game/tetris.py:427-433): a column's top is one
// v_ffbl on its word.  Citations "(:NNN)" are lines of the reference's game/tetris.py.
//
// Resident state, 32 B per board in two uint4 planes (SoA, plane[i] is board i -> 1 KiB per wave-load).
// 200 board bits + 18 counter bits + a 36-bit window of the piece list + 1 pool-slot bit (+ 1 spare) = 256:
//   A.x = col0 | col1<<20     A.y = col1>>12 | col2<<8 | moves[3:0]<<28     (three 20-bit columns per 64 bits)
//   A.z = col3 | col4<<20     A.w = col4>>12 | col5<<8 | moves[7:4]<<28
//   B.x = col6 | col7<<20     B.y = col7>>12 | col8<<8 | state<<28 | slot<<30
//         (state: 0 run, 1 won, 2 lost@limit, 3 lost@top-out; slot: which of the two pool buffers the board's
//          configuration lives in)
//   B.z = col9 | lines_cleared<<20 | window[35:32]<<28
//   B.w = window[31:0]        piece window: twelve 3-bit ids, entry 0 = pieces[0], entry 1 = pieces[1]
// Piece list of a configuration, in the pool: 64-bit words of twelve 3-bit ids with a stride of TEN entries,
// word w = entries [10w, 10w+12), ids past the end of the list read 7.  A board carries one such word as its
// window: every move shifts it down by one entry, and when the cursor reaches a multiple of ten the two
// entries left in the window are exactly the first two of word cursor/10, which is then loaded whole.  So
// pieces[0] and pieces[1] are always in the state itself and the common step does one round trip to HBM.
// Pool record (AoS, `stride` bytes, 64-B aligned): plane-A word, plane-B word (window = word 0), 64-bit words 1..
//
// Side record (64 B per configuration, behind the records): the ten column words as the multi-step kernel keeps them
// (bit 20 set), then the piece window (low word, high 4 bits), zero padding.  A reset in that kernel is three 16-byte
// loads and five LDS writes -- no unpacking (the packed record costs it sixteen instructions per reset, and in a wave of
// 64 some lane resets at almost every step).  The step kernel never touches it.
//
// A board does not store which pool entry it was started from, nor an episode number.  Every group of 32 boards
// has a STEP CLOCK in memory (uint64, advanced by the wave that owns the group: +1 per step launch, +K per
// K-step launch, zeroed by a full reset -- so every clock equals the number of steps since the last full reset).
// A running board's episode began at step  birth = clock - moves_used  (a running board makes exactly one move
// per step), and its pool entry is assign_config(global board index, birth, seed) in the slot it carries -- a
// function of the 64-bit birth step, so a board's sequence of configurations never repeats.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tpl {

constexpr int kRows = 20;
constexpr int kCols = 10;
constexpr uint32_t kColMask = 0xFFFFFu;
constexpr int kWindowStride = 10;    // entries between the starts of consecutive piece words
constexpr int kWindowEntries = 12;
constexpr int kSideShift = 6;        // side records are 64 bytes
constexpr int kClockGroup = 32;      // boards per step clock (log2 = kClockShift)
constexpr int kClockShift = 5;

enum : uint32_t { ST_RUNNING = 0, ST_WON = 1, ST_LOST_LIMIT = 2, ST_LOST_TOPOUT = 3 };

struct Board {
    uint32_t c[kCols];   // column words
    uint32_t window;     // piece window, entries 0..9 (+ 2 bits of entry 10): entry 0 = the piece that falls next
    uint32_t window_hi;  //   bits 32..35 of the window (rest of entry 10, entry 11)
    uint32_t state, lines, moves, slot;
};

__device__ __forceinline__ void set_window(Board& s, uint64_t word) {
    s.window = (uint32_t)word;
    s.window_hi = (uint32_t)(word >> 32) & 0xFu;
}
// the window after pieces.pop(0): one entry down (a 36-bit shift), or -- when it has run out -- the piece word that was
// fetched for it.  Selects, not a branch: a divergent region here costs every wave two exec-mask branches and the
// compiler's full memory wait.
__device__ __forceinline__ void next_window(Board& s, bool refill, uint64_t word) {
    const uint32_t lo = __builtin_amdgcn_alignbit(s.window_hi, s.window, 3);
    const uint32_t hi = s.window_hi >> 3;
    s.window = refill ? (uint32_t)word : lo;
    s.window_hi = refill ? ((uint32_t)(word >> 32) & 0xFu) : hi;
}
// cursor / 10 and cursor % 10 == 0 for cursor < 256 from one 24-bit multiply (full rate): cursor * 205 = 2048 * (cursor / 10)
// + a remainder that is below 205 exactly when cursor is a multiple of ten (checked exhaustively in tests/test_boundary.py)
__device__ __forceinline__ uint32_t tenths(uint32_t cursor) { return __umul24(cursor, 205u); }
__device__ __forceinline__ bool window_runs_out(uint32_t t) { return (t & 2047u) < 205u; }
__device__ __forceinline__ uint32_t window_word(uint32_t t) { return t >> 11; }

// ---- shape table --------------------------------------------------------------------------------------
// `tetrominos` (:23-57) re-encoded per (piece, rotations & 3) for the column layout.  Two words per entry:
//   .x  bits 0-15: four nibbles, nibble c = the piece's column c as a bit-per-row pattern (bit i = mask row i)
//       bits 16-18: width, bits 19-21: height, bits 22-25: 10 - width (the right clamp of :364 as one v_min),
//       bits 26-29: (1 << height) - 1 (the piece's rows as a mask, for the full-row test of :382-386)
//   .y  byte c = 3 - reverse_topography[c] for a covered column, 64 for columns past the piece's width
// so that  drop = min_c(top[c] + byte_c) - 4  ==  min_c(top[c] - revtopo[c]) - 1   (:424-425, :433).
// get_tetromino's `rotations % len` (:60-61) is folded in: len is 1, 2 or 4, so (rotations & 3) % len
// indexes the same entry as rotations % len for any non-negative rotations.
struct ShapeWord { uint32_t x, y; };

__host__ __device__ constexpr ShapeWord make_shape(int h, int w, int m0, int m1, int m2, int m3,
                                                   int t0, int t1, int t2, int t3) {
    const int m[4] = {m0, m1, m2, m3};
    const int t[4] = {t0, t1, t2, t3};
    uint32_t x = 0, y = 0;
    for (int c = 0; c < 4; ++c) {
        uint32_t colbits = 0;
        for (int i = 0; i < 4; ++i) colbits |= (uint32_t)((m[i] >> c) & 1) << i;
        x |= colbits << (4 * c);
        y |= (uint32_t)(c < w ? 3 - t[c] : 64) << (8 * c);
    }
    x |= (uint32_t)w << 16 | (uint32_t)h << 19 | (uint32_t)(kCols - w) << 22 | ((1u << h) - 1u) << 26;
    return ShapeWord{x, y};
}

// rows: (h, w, row masks top->bottom with bit x = mask column x, reverse topography)
#define TPL_I0 make_shape(1, 4, 15, 0, 0, 0, 0, 0, 0, 0)   /* :25 */
#define TPL_I1 make_shape(4, 1, 1, 1, 1, 1, 3, 0, 0, 0)    /* :26 */
#define TPL_L0 make_shape(2, 3, 4, 7, 0, 0, 1, 1, 1, 0)    /* :29 */
#define TPL_L1 make_shape(3, 2, 3, 2, 2, 0, 0, 2, 0, 0)    /* :30 */
#define TPL_L2 make_shape(2, 3, 7, 1, 0, 0, 1, 0, 0, 0)    /* :31 */
#define TPL_L3 make_shape(3, 2, 1, 1, 3, 0, 2, 2, 0, 0)    /* :32 */
#define TPL_J0 make_shape(2, 3, 1, 7, 0, 0, 1, 1, 1, 0)    /* :35 */
#define TPL_J1 make_shape(3, 2, 2, 2, 3, 0, 2, 2, 0, 0)    /* :36 */
#define TPL_J2 make_shape(2, 3, 7, 4, 0, 0, 0, 0, 1, 0)    /* :37 */
#define TPL_J3 make_shape(3, 2, 3, 1, 1, 0, 2, 0, 0, 0)    /* :38 */
#define TPL_T0 make_shape(2, 3, 2, 7, 0, 0, 1, 1, 1, 0)    /* :41 */
#define TPL_T1 make_shape(3, 2, 2, 3, 2, 0, 1, 2, 0, 0)    /* :42 */
#define TPL_T2 make_shape(2, 3, 7, 2, 0, 0, 0, 1, 0, 0)    /* :43 */
#define TPL_T3 make_shape(3, 2, 1, 3, 1, 0, 2, 1, 0, 0)    /* :44 */
#define TPL_S0 make_shape(2, 3, 6, 3, 0, 0, 1, 1, 0, 0)    /* :47 */
#define TPL_S1 make_shape(3, 2, 1, 3, 2, 0, 1, 2, 0, 0)    /* :48 */
#define TPL_Z0 make_shape(2, 3, 3, 6, 0, 0, 0, 1, 1, 0)    /* :51 */
#define TPL_Z1 make_shape(3, 2, 2, 3, 1, 0, 2, 1, 0, 0)    /* :52 */
#define TPL_O0 make_shape(2, 2, 3, 3, 0, 0, 1, 1, 0, 0)    /* :55 */

// [piece][rotations & 3]; piece ids I0 L1 J2 T3 S4 Z5 O6 (:8-16); slot 7 is the "no piece" id and is never used
// by a running board.
#define TPL_SHAPE_LIST                                                                                    \
    TPL_I0, TPL_I1, TPL_I0, TPL_I1, TPL_L0, TPL_L1, TPL_L2, TPL_L3, TPL_J0, TPL_J1, TPL_J2, TPL_J3, TPL_T0,  \
    TPL_T1, TPL_T2, TPL_T3, TPL_S0, TPL_S1, TPL_S0, TPL_S1, TPL_Z0, TPL_Z1, TPL_Z0, TPL_Z1, TPL_O0, TPL_O0,  \
    TPL_O0, TPL_O0, TPL_O0, TPL_O0, TPL_O0, TPL_O0
__device__ __constant__ const ShapeWord kShapeTable[32] = {TPL_SHAPE_LIST};
constexpr ShapeWord kShapeTableHost[32] = {TPL_SHAPE_LIST};   // same entries, for tpl_shape_info on the host

// ---- pack / unpack ------------------------------------------------------------------------------------
__device__ __forceinline__ void pack3(uint32_t a, uint32_t b, uint32_t c, uint32_t& lo, uint32_t& hi) {
    lo = a | (b << 20);
    hi = (b >> 12) | (c << 8);
}

// kSentinel: the multi-step kernel keeps every column word with bit 20 set (the top of a column, 20 when it is empty, is
// then one v_ffbl with no preparation); unpack_board<true> produces that form at no extra cost -- a bit-field insert
// where the plain form has an `and` -- move_board_lds expects it and pack_board<true> strips it.
constexpr uint32_t kSentinelBit = 1u << kRows;

// (mask & x) | (~mask & y) as ONE v_bfi_b32.  With both constants written as literals the compiler emits an `and` and an
// `or` (a VOP3 instruction takes one literal); handing them over in scalar registers it cannot see through gives the
// bit-field insert.
__device__ __forceinline__ uint32_t opaque_sgpr(uint32_t v) {
    uint32_t r;
    asm("s_mov_b32 %0, %1" : "=s"(r) : "i"(v));
    return r;
}

template <bool kSentinel>
__device__ __forceinline__ void unpack3(uint32_t lo, uint32_t hi, uint32_t& a, uint32_t& b, uint32_t& c) {
    if (kSentinel) {
        const uint32_t m = opaque_sgpr(kColMask), top = opaque_sgpr(kSentinelBit);
        a = (m & lo) | (~m & top);
        b = (m & __builtin_amdgcn_alignbit(hi, lo, 20)) | (~m & top);
        c = (m & (hi >> 8)) | (~m & top);
    } else {
        a = lo & kColMask;
        b = __builtin_amdgcn_alignbit(hi, lo, 20) & kColMask;   // ({hi,lo} >> 20)
        c = (hi >> 8) & kColMask;
    }
}

template <bool kSentinel = false>
__device__ __forceinline__ void unpack_board(const uint4& A, const uint4& B, Board& s) {
    unpack3<kSentinel>(A.x, A.y, s.c[0], s.c[1], s.c[2]);
    unpack3<kSentinel>(A.z, A.w, s.c[3], s.c[4], s.c[5]);
    unpack3<kSentinel>(B.x, B.y, s.c[6], s.c[7], s.c[8]);
    if (kSentinel) {
        const uint32_t m = opaque_sgpr(kColMask), top = opaque_sgpr(kSentinelBit);
        s.c[9] = (m & B.z) | (~m & top);
    } else {
        s.c[9] = B.z & kColMask;
    }
    s.lines = (B.z >> 20) & 0xFFu;
    s.moves = (A.y >> 28) | ((A.w >> 28) << 4);
    s.window = B.w;
    s.window_hi = B.z >> 28;
    s.state = (B.y >> 28) & 3u;
    s.slot = (B.y >> 30) & 1u;
}

// the fields a step needs before it unpacks the board
__device__ __forceinline__ uint32_t packed_state(const uint4& B) { return (B.y >> 28) & 3u; }
__device__ __forceinline__ uint32_t packed_slot(const uint4& B) { return (B.y >> 30) & 1u; }
__device__ __forceinline__ uint32_t packed_moves(const uint4& A) { return (A.y >> 28) | ((A.w >> 28) << 4); }

template <bool kSentinel = false>
__device__ __forceinline__ void pack_board(const Board& s, uint4& A, uint4& B) {
    uint32_t c[kCols];
#pragma unroll
    for (int k = 0; k < kCols; ++k) c[k] = kSentinel ? (s.c[k] & kColMask) : s.c[k];
    pack3(c[0], c[1], c[2], A.x, A.y);
    pack3(c[3], c[4], c[5], A.z, A.w);
    pack3(c[6], c[7], c[8], B.x, B.y);
    A.y |= s.moves << 28;              // low nibble (the high one shifts out)
    A.w |= (s.moves >> 4) << 28;
    B.y |= (s.state << 28) | (s.slot << 30);
    B.z = c[9] | (s.lines << 20) | (s.window_hi << 28);
    B.w = s.window;
}

// rows (interchange, u16[20], bit x = column x) <-> columns
__device__ __forceinline__ void rows_to_cols(const uint16_t* rows, uint32_t* c) {
#pragma unroll
    for (int x = 0; x < kCols; ++x) c[x] = 0;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        uint32_t v = rows[r];
#pragma unroll
        for (int x = 0; x < kCols; ++x) c[x] |= ((v >> x) & 1u) << r;
    }
}

__device__ __forceinline__ uint32_t row_of_cols(const uint32_t* c, int r) {
    uint32_t v = 0;
#pragma unroll
    for (int x = 0; x < kCols; ++x) v |= ((c[x] >> r) & 1u) << x;
    return v;
}

// (m & x) | (~m & y) with m all-ones or zero: v_bfi_b32
__device__ __forceinline__ uint32_t blend(uint32_t m, uint32_t x, uint32_t y) { return (m & x) | (~m & y); }

// ---- the move ------------------------------------------------------------------------------------------
// Tetris.move(rotations, location) (:354-422) on the lane's board; the piece is entry 0 of the window, which the
// caller pops (:356).  `shape` is the LDS-resident table.  Returns rows cleared (0..4); sets
// `topout` when drop < 0 (:372-374), in which case the board and moves_used are left unchanged.
__device__ __forceinline__ uint32_t move_board(Board& s, const ShapeWord* shape, uint32_t rot, uint32_t loc,
                                               uint32_t L, uint32_t M, bool& topout) {
    // get_tetromino (:60-61, :359-360)
    const ShapeWord sh = shape[(s.window & 7u) * 4u + (rot & 3u)];
    const uint32_t w = (sh.x >> 16) & 7u;
    const uint32_t h = (sh.x >> 19) & 7u;

    // right clamp only (:363-364)
    loc = min(loc, (uint32_t)kCols - w);

    // calculate_drop_deltas (:427-433).  The top of EVERY column (20 when empty, via a sentinel bit) is one
    // v_ffbl each; the ten tops are packed a byte apiece into an 80-bit string, of which the piece needs the four
    // bytes starting at byte `loc` -- a word select on loc>>2 and one v_alignbyte.  (Selecting the four column
    // words themselves and then taking their tops costs a 23-blend network; and a `c[loc + k]` form is turned into
    // a runtime-indexed array by the compiler and lands in scratch.)  Bytes past column 9 read 0 and belong to
    // piece columns past its width, whose table bias of 64 keeps them out of the minimum.
    uint32_t t[kCols];
#pragma unroll
    for (int k = 0; k < kCols; ++k) t[k] = (uint32_t)__builtin_ctz(s.c[k] | kSentinelBit);
    uint32_t p0 = t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
    uint32_t p1 = t[4] | (t[5] << 8) | (t[6] << 16) | (t[7] << 24);
    uint32_t p2 = t[8] | (t[9] << 8);
    // keep the three words straight-line: left alone, the compiler sinks the tops of columns 0-3 into a divergent
    // branch on loc < 4 (saving five instructions for some lanes, paying two exec-mask branches in every wave)
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));
    const uint32_t word = loc >> 2;                               // loc <= 9 after the clamp
    const uint32_t lo = word == 0u ? p0 : word == 1u ? p1 : p2;
    const uint32_t hi = word == 0u ? p1 : word == 1u ? p2 : 0u;
    const uint32_t tops = __builtin_amdgcn_alignbyte(hi, lo, loc & 3u);
    // top - reverse_topography per piece column, as four byte sums (no carry: a top is at most 20, a bias at most 64)
    const uint32_t sums = tops + sh.y;
    const uint32_t best = min(min(sums & 0xFFu, (sums >> 8) & 0xFFu), min((sums >> 16) & 0xFFu, sums >> 24));
    // calculate_drop (:424-425): min(deltas) - 1, with the table's +3 bias removed
    const int drop = (int)best - 4;

    // top-out (:372-374): state False; board and moves_used untouched
    topout = drop < 0;
    const uint32_t dshift = topout ? 0u : (uint32_t)drop;

    // lock (:377-378): OR the piece's column patterns, shifted down by `drop`, into columns loc..loc+w-1.
    // The four nibbles are moved to their columns with one 64-bit shift, then each column takes its nibble.
    uint64_t placed = (uint64_t)(sh.x & 0xFFFFu) << (4u * loc);
    if (topout) placed = 0;
    const uint32_t plo = (uint32_t)placed, phi = (uint32_t)(placed >> 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) s.c[k] |= ((plo >> (4 * k)) & 0xFu) << dshift;
    s.c[8] |= (phi & 0xFu) << dshift;
    s.c[9] |= ((phi >> 4) & 0xFu) << dshift;
    s.moves += topout ? 0u : 1u;                                                  // (:379)

    // full rows among the piece's rows only (:382-386)
    uint32_t full = s.c[0];
#pragma unroll
    for (int k = 1; k < kCols; ++k) full &= s.c[k];
    uint32_t clear = topout ? 0u : (full & (((1u << h) - 1u) << dshift));
    const uint32_t n = (uint32_t)__builtin_popcount(clear);

    // compaction (:397-407): drop each cleared row, rows above it move down one, an empty row enters at the top
    while (clear) {
        const uint32_t r = (uint32_t)__builtin_ctz(clear);
        clear &= clear - 1u;
        const uint32_t above = (1u << r) - 1u;          // rows 0..r-1
        const uint32_t keep = ~((above << 1) | 1u);     // rows r+1..
#pragma unroll
        for (int k = 0; k < kCols; ++k) s.c[k] = (s.c[k] & keep) | ((s.c[k] & above) << 1);
    }
    s.lines += n;                                                                 // (:409)

    // terminal tests: no-clear exit (:389-394), win before move limit (:415-422)
    const bool won = !topout && n != 0u && s.lines >= L;
    const bool limit = !topout && !won && s.moves >= M;
    s.state = topout ? ST_LOST_TOPOUT : won ? ST_WON : limit ? ST_LOST_LIMIT : ST_RUNNING;
    return n;
}

// ---- the move with the board's columns in LDS (the K-steps-per-launch kernel) ----------------------------------
// move_board pays for the fact that a lane cannot index its registers: the tops of all ten columns are taken and packed
// so that the piece's four can be picked out, and the piece is ORed into all ten columns through ten nibble extracts.
// LDS is memory a lane CAN index.  The multi-step kernel keeps the column words in LDS between moves, lane-major
// (column k of lane l at word k * 64 + l: whatever column a lane asks for, its bank is its lane number -- no
// conflicts): a move reads the four columns under the piece, takes four tops, writes the four columns back with the
// piece in, then reads all ten for the full-row test.  The words carry the sentinel bit; three pad columns (holding
// only the sentinel) sit behind column 9 for pieces narrower than four at the right edge.
constexpr int kLdsCols = kCols + 3;
constexpr int kLdsStride = 64;            // words between consecutive columns of a lane

__device__ __forceinline__ void lds_store_cols(uint32_t* cols, const uint32_t (&c)[kCols]) {
#pragma unroll
    for (int k = 0; k < kCols; ++k) cols[k * kLdsStride] = c[k];
}
__device__ __forceinline__ void lds_load_cols(const uint32_t* cols, uint32_t (&c)[kCols]) {
#pragma unroll
    for (int k = 0; k < kCols; ++k) c[k] = cols[k * kLdsStride];
}

// Tetris.move (:354-422) as move_board, on the column words at `cols` (this lane's column 0).  `s.c` is not used, and
// `s.state` is not written: how the move ended comes back as three flags (at most one set), which the multi-step
// kernel keeps as lane masks -- its tallies, its reward and its reset test are then scalar-unit work.
struct MoveEnd { bool topout, won, limit; };

// `after_drop(topout)` runs as soon as the drop is known -- before the lock and the full-row test: the multi-step kernel
// uses it to send for a topped-out board's next configuration a lock, five LDS reads and a full-row test earlier.
template <typename AfterDrop>
__device__ __forceinline__ uint32_t move_board_lds(Board& s, uint32_t* cols, const ShapeWord* shape, uint32_t rot, uint32_t loc,
                                                   uint32_t L, uint32_t M, MoveEnd& end, AfterDrop&& after_drop) {
    const ShapeWord sh = shape[(s.window & 7u) * 4u + (rot & 3u)];                 // get_tetromino (:60-61, :359-360)
    loc = min(loc, (sh.x >> 22) & 15u);                                            // right clamp only (:363-364)

    // calculate_drop_deltas (:427-433) on the piece's own columns
    uint32_t* under = cols + loc * kLdsStride;
    uint32_t c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = under[j * kLdsStride];
    uint32_t best = 0xFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) best = min(best, (uint32_t)__builtin_ctz(c[j]) + ((sh.y >> (8 * j)) & 0xFFu));
    const int drop = (int)best - 4;                                                // calculate_drop (:424-425)
    const bool topout = drop < 0;                                                  // (:372-374)
    const uint32_t dshift = topout ? 0u : (uint32_t)drop;
    after_drop(topout);

    // lock (:377-378): the four columns go back with the piece's column patterns ORed in (nothing on a top-out)
    const uint32_t pattern = topout ? 0u : sh.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) under[j * kLdsStride] = c[j] | (((pattern >> (4 * j)) & 0xFu) << dshift);
    s.moves += topout ? 0u : 1u;                                                   // (:379)

    // full rows among the piece's rows only (:382-386)
    uint32_t b[kCols];
    lds_load_cols(cols, b);
    uint32_t full = b[0];
#pragma unroll
    for (int k = 1; k < kCols; ++k) full &= b[k];
    uint32_t clear = topout ? 0u : (full & (((sh.x >> 26) & 15u) << dshift));
    const uint32_t n = (uint32_t)__builtin_popcount(clear);

    // compaction (:397-407), on the registers, written back when it ran
    if (clear) {
        do {
            const uint32_t r = (uint32_t)__builtin_ctz(clear);
            clear &= clear - 1u;
            const uint32_t above = (1u << r) - 1u;          // rows 0..r-1
            const uint32_t keep = ~((above << 1) | 1u);     // rows r+1.. (and the sentinel)
#pragma unroll
            for (int k = 0; k < kCols; ++k) b[k] = (b[k] & keep) | ((b[k] & above) << 1);
        } while (clear);
        lds_store_cols(cols, b);
    }
    s.lines += n;                                                                  // (:409)

    // terminal tests: no-clear exit (:389-394), win before move limit (:415-422)
    end.topout = topout;
    end.won = !topout && n != 0u && s.lines >= L;
    end.limit = !topout && !end.won && s.moves >= M;
    return n;
}

__device__ __forceinline__ uint32_t move_board_lds(Board& s, uint32_t* cols, const ShapeWord* shape, uint32_t rot, uint32_t loc,
                                                   uint32_t L, uint32_t M, MoveEnd& end) {
    return move_board_lds(s, cols, shape, rot, loc, L, M, end, [](bool) {});
}

// the same with the state written, for the kernels that make one move per launch
__device__ __forceinline__ uint32_t move_board_lds(Board& s, uint32_t* cols, const ShapeWord* shape, uint32_t rot, uint32_t loc,
                                                   uint32_t L, uint32_t M, bool& topout) {
    MoveEnd end;
    const uint32_t n = move_board_lds(s, cols, shape, rot, loc, L, M, end);
    topout = end.topout;
    s.state = end.topout ? ST_LOST_TOPOUT : end.won ? ST_WON : end.limit ? ST_LOST_LIMIT : ST_RUNNING;
    return n;
}

// ---- counter-based generator (the synthetic workload of SURVEY 8d; DESIGN.md states the function) --------
__host__ __device__ __forceinline__ uint64_t sm64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// rng(seed, stream, index, counter) in two stages: the part that is fixed per (seed, stream, index) and the draw
__host__ __device__ __forceinline__ uint64_t rng_base(uint64_t seed, uint64_t stream, uint64_t index) {
    return sm64(sm64(seed ^ (stream * 0xD1B54A32D192ED03ULL)) ^ index);
}
__host__ __device__ __forceinline__ uint64_t rng_at(uint64_t base, uint64_t counter) { return sm64(base ^ counter); }
__host__ __device__ __forceinline__ uint64_t rng(uint64_t seed, uint64_t stream, uint64_t index, uint64_t counter) {
    return rng_at(rng_base(seed, stream, index), counter);
}
// a draw reduced to [lo, hi] by multiply-high of its upper half (no division on the device)
__host__ __device__ __forceinline__ int rng_range(uint64_t draw, int lo, int hi) {
    return lo + (int)(((draw >> 32) * (uint64_t)(hi - lo + 1)) >> 32);
}

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// The decision stream of the carving generators (host and device).  A configuration is built by ATTEMPTS (the restart
// rule below); attempt `a` of configuration `index` owns the 64-bit word  w = rng(seed, 4, index, a)  and its decision k is
// fmix32(key_k),  key_k = low32(w) + k * stride  (a running add),  stride = high32(w) | 1  -- both halves of the word are
// used, so two configurations (or two attempts) share a stream only if their 63 bits agree; a stream's period is 2^32
// words.  A word serves one trip of the search loop (decision_word below) or one other decision, reduced to [lo, hi] as
// lo + (((d >> 8) * (hi - lo + 1)) >> 24)  -- a 24-bit multiply; the ranges drawn from have at most ten values (a bias below
// 10 / 2^24), and at most 255 fit.
// (Through round 2 every decision was a 64-bit splitmix round and a 64-bit multiply-high: some seven 32-bit multiplies and
// twice as many shifts, xors and carries each, three decisions per iteration of the search; in round 3 the stride was one constant for every stream,
// so all streams walked one 2^32-cycle from different offsets.)
struct DecisionStream { uint32_t key, stride; };
__host__ __device__ __forceinline__ DecisionStream decision_stream(uint64_t seed, uint64_t index, uint32_t attempt) {
    const uint64_t w = rng(seed, 4, index, attempt);
    return DecisionStream{(uint32_t)w, (uint32_t)(w >> 32) | 1u};
}
// One trip of the search loop (:234-279) makes three decisions -- which piece of the bag (:85), how many rotations (:250), where
// (:253) -- and takes ONE word of the stream for them: bits 31-20 -> the bag index, (field * n_bag) >> 12; bits 19-18 -> the
// rotations; bits 17-0 -> the location, (field * places) >> 18 (biases below 7 / 4096 and 10 / 2^18).  Three hashes a trip
// were a fifth of a trip's instructions on the device.  Every other decision (the shuffles
// of the padding, :93) takes a word of its own through decision().
__host__ __device__ __forceinline__ uint32_t decision_word(DecisionStream& s) {
    const uint32_t w = fmix32(s.key);
    s.key += s.stride;
    return w;
}
__host__ __device__ __forceinline__ int word_bag_index(uint32_t w, int n_bag) {
#ifdef __HIP_DEVICE_COMPILE__
    return (int)(__umul24(w >> 20, (uint32_t)n_bag) >> 12);        // twelve bits by three: the product fits the 24-bit multiply
#else
    return (int)(((w >> 20) * (uint32_t)n_bag) >> 12);
#endif
}
__host__ __device__ __forceinline__ int word_rotations(uint32_t w) { return (int)((w >> 18) & 3u); }
__host__ __device__ __forceinline__ int word_location(uint32_t w, int places) { return (int)(((w & 0x3FFFFu) * (uint32_t)places) >> 18); }

__host__ __device__ __forceinline__ int decision(DecisionStream& s, int lo, int hi) {
    const uint32_t d = fmix32(s.key) >> 8;
    s.key += s.stride;
#ifdef __HIP_DEVICE_COMPILE__
    return lo + (int)(__umul24(d, (uint32_t)(hi - lo + 1)) >> 24);
#else
    return lo + (int)((d * (uint32_t)(hi - lo + 1)) >> 24);
#endif
}

// The restart rule of the carving generators (build-defined, like the decision stream; the carving logic itself is the
// reference's, pinned by its decision tapes).  The reference's search (game/tetris.py:234-279) is a Las-Vegas loop with no
// bound; its length is close to exponentially distributed (mean 1.45 x median, p99 6.5 x median at L = 10:
// profiles/r04_carve/iteration_histogram.json), so a batch of 2^20 configurations waits for one that takes 14 means.
// Configuration `index` is therefore DEFINED as the outcome of the first attempt a = 0, 1, ... (each with its own decision
// stream, each starting from the full stack) that ends within  carve_cutoff(L, cutoff, a)  trips of the while loop.  A
// cut-off of about twice the median search costs 1-3 % more iterations in total (a memoryless search loses only its
// warm-up when restarted) and bounds every attempt -- which is what lets the device generator run the attempts of one
// straggling configuration on many lanes at once and still return exactly this configuration (carve_device.hip).
// The table's cut-offs were measured at M = 40.  Where M is close to the fewest pieces that can dig two columns down to the
// bottom row (L = 15 with M = 16: a search of a million trips against the table's 64,000) almost every attempt at the base
// cut-off fails, so from the thirteenth attempt on the cut-off DOUBLES with every attempt until it is 256 times the base
// (attempts 19-23; never above 2^28 trips): the trips lost to failed attempts stay within a small multiple of the one that
// succeeds, and a search may be 256 times as long as the table expects (2^24 trips at L = 15) before anything is given up.
// Only when all kCarveAttempts attempts have run into their cut-offs is the configuration reported as capped (all-zero
// outputs): "this (L, M) did not finish within 256 x the base cut-off" -- not "cannot be carved": a caller who wants to
// search on passes a larger `cutoff` (up to 2^28).  The bound is there because an unbounded search on the DEVICE is a kernel
// that may never end; the generators also try four pilot configurations on the host before a batch goes out (carve_pilot).
constexpr int kCarveAttempts = 24;
constexpr int kCarveDoublings = 8;
constexpr int64_t kCarveCutoffMax = (int64_t)1 << 28;
// iterations allowed to attempt `attempt`: base = `cutoff` if the caller gave one, else about twice the measured median search
// length at this L (M = 40)
__host__ __device__ __forceinline__ int64_t carve_cutoff(int L, int64_t cutoff, int attempt) {
    int64_t c = cutoff;
    if (c <= 0) {                                          // a switch, not a table load: no constant memory on the device side
        switch (L) {
            case 1: case 2: case 3: c = 64; break;
            case 4: c = 128; break;     case 5: c = 256; break;     case 6: c = 384; break;     case 7: c = 512; break;
            case 8: c = 768; break;     case 9: c = 1792; break;    case 10: c = 3328; break;   case 11: c = 5376; break;
            case 12: c = 9216; break;   case 13: c = 17408; break;  case 14: c = 36000; break;  case 15: c = 64000; break;
            default: c = 132000; break;
        }
    }
    // twelve attempts at the base cut-off, then 2, 4, 8, ... 256 times it (attempts 12 ... 19), 256 times for the last four.  (Doubling from the seventh attempt on, as this
    // first was, made the unluckiest configuration of EVERY large batch -- one in 4,000 fails six times -- wait for an attempt
    // of 2 c trips: the slowest wave of a launch ran 10,000 trips past the queue's end where the median ran 5,100.  Twelve
    // failures at a cut-off of twice the median happen to one configuration in 10^7.)
    const int64_t grown = c << (attempt < 12 ? 0 : attempt - 11 < kCarveDoublings ? attempt - 11 : kCarveDoublings);
    return grown < kCarveCutoffMax ? grown : kCarveCutoffMax;
}
// fewest pieces with which the search loop (:234) can end at all: it ends when two cells of the bottom row are gone, a cell can
// only be carved once its column is empty above it (the piece must come to rest where it is taken out, :341-349), and a piece
// takes four cells: 2 L cells, ceil(L / 2) pieces.  A necessary condition only -- the generators refuse what fails it and
// find out about the rest by the restart rule.
__host__ __device__ __forceinline__ int carve_fewest_pieces(int L) { return (L + 1) / 2; }

// which pool entry the episode of global board `g` = global_offset + i that begins at step `birth` starts from.
// hash mode: (g, birth, seed) folded into 32 bits, one round of a 32-bit finaliser (a bijection), range-reduced with
// a multiply-high.  For a fixed board the folded word is birth * odd + const, so distinct births (mod 2^32) give
// distinct words: a board's configuration sequence has no period.  `seed_mix` = assign_seed(seed), computed once on the host.
// sequential mode: (g + birth) mod n_cfg on the low 32 bits of birth, done with 32-bit remainders only (offset_mod =
// global_offset mod n_cfg comes from the host) because a 64-bit remainder is a long software routine on the GPU;
// boards that start together take adjacent entries.
__host__ __device__ __forceinline__ uint32_t assign_seed(uint64_t seed) { return (uint32_t)(sm64(seed) >> 32); }

__device__ __forceinline__ uint32_t assign_config(int64_t global_offset, uint32_t offset_mod, uint32_t i, uint64_t birth,
                                                  uint32_t seed_mix, uint32_t n_cfg, int mode) {
    if (mode == 1) {
        uint64_t t = (uint64_t)offset_mod + (uint64_t)(i % n_cfg);
        if (t >= n_cfg) t -= n_cfg;
        t += (uint32_t)birth % n_cfg;
        if (t >= n_cfg) t -= n_cfg;
        return (uint32_t)t;
    }
    const uint64_t g = (uint64_t)global_offset + i;
    const uint32_t x = ((uint32_t)g + (uint32_t)birth * 0x9E3779B1u) ^ __umul24((uint32_t)(g >> 32), 0x85EBCBu) ^
                       __umul24((uint32_t)(birth >> 32), 0xC2B2AFu) ^ seed_mix;
    return __umulhi(fmix32(x), n_cfg);
}

}  // namespace tpl
