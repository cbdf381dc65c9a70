// tpl_observe.h -- the two stages of the [N,217] observation (observe.hip states the layout and why), as device
// functions shared by observe_kernel and by the step kernel's step-and-observe form (tetris_piclim.hip).
#pragma once

#include "tpl_device.h"
#include "../../include/tetris_piclim.h"

#include <hip/hip_bf16.h>

namespace tpl {
namespace obs {

constexpr int kObs = TPL_OBS_DIM;          // 217
constexpr int kPitch = 224;                // LDS row pitch
constexpr int kWaveLds = 64 * kPitch;      // 14,336 B per wave
constexpr int kObsWaves = 4;               // waves per block
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// byte `slice` (0, 1, 2) of four column words, packed into one register (column a in byte 0 ... column d in byte 3)
template <int kSlice>
__device__ __forceinline__ uint32_t pack_slice(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    // v_perm_b32 selector: 0-3 = bytes of the second operand, 4-7 = bytes of the first
    constexpr uint32_t pair = 0x0C0C0000u | ((4u + kSlice) << 8) | (uint32_t)kSlice;     // {0, 0, hi.slice, lo.slice}
    const uint32_t ab = perm(b, a, pair), cd = perm(d, c, pair);
    return perm(cd, ab, 0x05040100u);
}

template <typename T>
struct Chunk;                               // elements per 16-byte chunk and the conversion of a chunk's bytes

template <>
struct Chunk<float> {
    static constexpr int kElems = 4;
    using Bytes = uint32_t;
    static __device__ __forceinline__ uint4 convert(uint32_t b) {
        return make_uint4(__float_as_uint((float)(b & 0xFFu)), __float_as_uint((float)((b >> 8) & 0xFFu)),
                          __float_as_uint((float)((b >> 16) & 0xFFu)), __float_as_uint((float)(b >> 24)));
    }
    static __device__ __forceinline__ float from_int(int v) { return (float)v; }
};

template <>
struct Chunk<__hip_bfloat16> {
    static constexpr int kElems = 8;
    using Bytes = uint2;
    static __device__ __forceinline__ uint32_t pair(uint32_t lo_byte, uint32_t hi_byte) {
        // an integer below 256 has at most 8 significant bits: its bf16 is exactly the upper half of its f32
        return perm(__float_as_uint((float)hi_byte), __float_as_uint((float)lo_byte), 0x07060302u);
    }
    static __device__ __forceinline__ uint4 convert(uint2 b) {
        return make_uint4(pair(b.x & 0xFFu, (b.x >> 8) & 0xFFu), pair((b.x >> 16) & 0xFFu, b.x >> 24),
                          pair(b.y & 0xFFu, (b.y >> 8) & 0xFFu), pair((b.y >> 16) & 0xFFu, b.y >> 24));
    }
    static __device__ __forceinline__ __hip_bfloat16 from_int(int v) { return __float2bfloat16((float)v); }
};

template <typename Bytes>
__device__ __forceinline__ Bytes lds_bytes(const uint8_t* p) {        // unaligned LDS read of 4 or 8 bytes
    Bytes v;
    __builtin_memcpy(&v, p, sizeof(Bytes));
    return v;
}

// stage A: the board of lane `lane` -> 217 bytes at rows[lane * 224] (+ the tail of the previous lane's row).  Returns
// lines left, which a byte cannot carry when it is negative (a frozen board whose last clear overshot L).
__device__ __forceinline__ int board_to_bytes(const Board& s, uint32_t L, uint32_t M, uint8_t* rows, int lane) {
    uint32_t d[kCols];                                        // columns moved down one row
#pragma unroll
    for (int x = 0; x < kCols; ++x) d[x] = s.c[x] >> 1;
    uint32_t* const row32 = (uint32_t*)(rows + lane * kPitch);
    uint32_t first[2] = {0, 0};                               // cells 0..7, for the previous row's tail
    // Rows come in pairs (20 bytes = 5 dwords): [r: x0-3] [r: x4-7] [r: x8,9 | r+1: x0,1] [r+1: x2-5] [r+1: x6-9].
    // With slice = rows 8*slice .. 8*slice+7 of four columns in one register, the dword of row pair (r, r+1),
    // r = 8*slice + u, is (group >> u) & 0x01010101 -- the last three groups are built from the columns moved
    // down one row, so the same u serves row r+1.
#define TPL_OBS_SLICE(SL, PAIRS)                                                                                  \
    {                                                                                                             \
        const uint32_t g0 = pack_slice<SL>(s.c[0], s.c[1], s.c[2], s.c[3]);                                       \
        const uint32_t g1 = pack_slice<SL>(s.c[4], s.c[5], s.c[6], s.c[7]);                                       \
        const uint32_t g2 = pack_slice<SL>(s.c[8], s.c[9], d[0], d[1]);                                           \
        const uint32_t g3 = pack_slice<SL>(d[2], d[3], d[4], d[5]);                                               \
        const uint32_t g4 = pack_slice<SL>(d[6], d[7], d[8], d[9]);                                               \
        _Pragma("unroll") for (int k = 0; k < PAIRS; ++k) {                                                       \
            const int u = 2 * k, q = 5 * (4 * SL + k);                                                            \
            row32[q + 0] = (g0 >> u) & 0x01010101u;                                                               \
            row32[q + 1] = (g1 >> u) & 0x01010101u;                                                               \
            row32[q + 2] = (g2 >> u) & 0x01010101u;                                                               \
            row32[q + 3] = (g3 >> u) & 0x01010101u;                                                               \
            row32[q + 4] = (g4 >> u) & 0x01010101u;                                                               \
        }                                                                                                         \
        if (SL == 0) { first[0] = g0 & 0x01010101u; first[1] = g1 & 0x01010101u; }                                \
    }
    TPL_OBS_SLICE(0, 4)          // rows 0-7
    TPL_OBS_SLICE(1, 4)          // rows 8-15
    TPL_OBS_SLICE(2, 2)          // rows 16-19
#undef TPL_OBS_SLICE
    // features 200..215 as one 16-byte string: one-hot current piece (7 bytes), one-hot next piece (7 bytes),
    // lines left, moves left.  Piece id 7 ("none") has no byte in either.
    const uint32_t cur = s.window & 7u, nxt = (s.window >> 3) & 7u;
    const uint64_t c1 = ((uint64_t)1 << (8u * cur)) & 0x00FFFFFFFFFFFFFFull;
    const uint64_t n1 = ((uint64_t)1 << (8u * nxt)) & 0x00FFFFFFFFFFFFFFull;
    const int lines_left = (int)L - (int)s.lines;             // negative after a clear that overshoots L
    const uint32_t moves_left = M - s.moves;
    const uint32_t t0 = (uint32_t)c1;
    const uint32_t t1 = (uint32_t)(c1 >> 32) | ((uint32_t)n1 << 24);
    const uint32_t t2 = (uint32_t)(n1 >> 8);
    const uint32_t t3 = (uint32_t)(n1 >> 40) | ((uint32_t)(lines_left < 0 ? 0 : lines_left) << 16) | (moves_left << 24);
    row32[50] = t0; row32[51] = t1; row32[52] = t2; row32[53] = t3;
    rows[lane * kPitch + 216] = s.state != ST_RUNNING ? 1 : 0;
    if (lane > 0) {                                           // my first seven cells close the previous row
        uint8_t* tail = rows + (lane - 1) * kPitch + kObs;    // offsets 217 (1 byte), 218 (2), 220 (4)
        tail[0] = (uint8_t)first[0];
        *(uint16_t*)(tail + 1) = (uint16_t)(first[0] >> 8);
        *(uint32_t*)(tail + 3) = __builtin_amdgcn_alignbyte(first[1], first[0], 3);
    }
    return lines_left;
}

// stage B: the wave's span of the output (the rows of boards base .. base + count - 1), 16 bytes per lane per store,
// from the bytes stage A left at `rows`.  Called by the whole wave; lanes >= count pass lines_left = 0.
template <typename T>
__device__ __forceinline__ void store_span(const uint8_t* rows, int lane, int count, int64_t base, int lines_left, T* out) {
    // one wave wrote, the same wave reads: LDS operations of a wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int kE = Chunk<T>::kElems;
    const int total = count * kObs;                               // elements in the span
    const int chunks = total / kE;
    uint4* const span = (uint4*)(out + base * kObs);
    // (row, offset) of my chunk's first element, carried along instead of divided out: a step of 64 chunks is
    // 64*kE elements = kAdvRows whole rows + kAdvOff
    constexpr int kAdvRows = 64 * kE / kObs, kAdvOff = 64 * kE - kAdvRows * kObs;
    int row = 0, off = lane * kE;
    while (off >= kObs) { off -= kObs; ++row; }
    for (int c = lane; c < chunks; c += 64) {
        const typename Chunk<T>::Bytes b = lds_bytes<typename Chunk<T>::Bytes>(rows + row * kPitch + off);
        // non-temporal: the observation is written once and read by someone else, much later; keeping it out of the
        // caches' way is worth 20 % on the bf16 stream (455 MB at 2^20 boards: 100 -> 80 us)
        const uint4 v = Chunk<T>::convert(b);
        u32x4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, (u32x4*)(span + c));
        row += kAdvRows; off += kAdvOff;
        if (off >= kObs) { off -= kObs; ++row; }
    }
    // a partial wave's span need not end on a chunk
    const int e = chunks * kE + lane;
    if (lane < kE && e < total) {
        const int r = e / kObs, f = e - r * kObs;
        out[base * kObs + e] = Chunk<T>::from_int((int)rows[r * kPitch + f]);
    }
    // the one value a byte cannot carry: lines left below zero (a frozen board whose last clear overshot L)
    if (__any(lines_left < 0)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // (compiler ordering only)
        __builtin_amdgcn_s_waitcnt(0);                            // the span's stores are done: this one lands last
        if (lines_left < 0) out[(base + lane) * kObs + 214] = Chunk<T>::from_int(lines_left);
    }
}

}  // namespace obs
}  // namespace tpl
