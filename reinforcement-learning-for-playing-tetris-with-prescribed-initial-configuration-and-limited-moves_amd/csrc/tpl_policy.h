// tpl_policy.h -- device pieces shared by the three policy kernels (policy_mlp.hip: bf16 operands; policy_f32.hip: f32
// operands; policy_split.hip: three bf16 pieces per number): the board's features as bits, the cross-lane fetch of the partner
// board, the argmax of the 14 outputs.
// Geometry of a wave: the float32 and split kernels run 32 boards = two N tiles of 16 -- lane l = (c = l & 15, g = l >> 4)
// owns board (t = g >> 1, c), lanes g and g ^ 1 carry identical copies of it (both_features below fetches the partner tile's
// board from lane l ^ 32).  The bf16 kernel runs 64 boards = four N tiles, one board per lane (t = g; its own column_features /
// own_action are in policy_mlp.hip).  pick_action serves both: all four lanes of a column return the action of the tile asked for.
#pragma once

#include "tpl_internal.h"

#include <cstring>
#include "tpl_step.h"

namespace tpl {
namespace p16 {

// arguments of the actor kernels (policy -> epsilon-greedy -> step, T times in one launch), bf16 and float32 alike
struct ActorArgs {
    StepArgs s;
    const uint4* image;
    uint32_t T, step0, eps_q24;
    uint64_t explore_seed;
    uint8_t* actions;
    float* rewards;
    uint8_t* dones;
    uint4* states_a;
    uint4* states_b;
};


typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kHidden = 128, kObs = 217, kOut = 14;

// internal layer-1 feature -> tpl_expand_obs index, -1 for a pad.  Internal order: k = 20x + y for the cell in row y,
// column x -- how the state stores the board -- then the 17 extras at 200..216.
static inline int std_feature(int k) {
    if (k < 200) return (k % 20) * 10 + (k / 20);
    return k < kObs ? k : -1;
}

// host side of the packers: float -> bf16, round to nearest even
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// k of element j of lane group g in k-step s.  Hidden layers: dictated by the accumulator hand-off.
static inline int frag_k(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }
// Layer 1 is free to choose, because its B fragments are made from bits: the two elements of register i come from
// bits 4g + i and 16 + 4g + i of feature word s, so that one shift and one mask turn the word into the register.
static inline int frag_k1(int s, int g, int j) { return 32 * s + 4 * g + (j >> 1) + 16 * (j & 1); }

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// a board -> its cell bit vector in the internal order (bit 20x + y; extras as bits 200..213 and 216) and the two
// numeric features (L_rem, M_rem) as a packed bf16 pair; f[7] = that pair
__device__ __forceinline__ void board_features(const Board& s, int L, int M, uint32_t (&f)[8]) {
    f[0] = s.c[0] | (s.c[1] << 20);
    f[1] = (s.c[1] >> 12) | (s.c[2] << 8) | (s.c[3] << 28);
    f[2] = (s.c[3] >> 4) | (s.c[4] << 16);
    f[3] = (s.c[4] >> 16) | (s.c[5] << 4) | (s.c[6] << 24);
    f[4] = (s.c[6] >> 8) | (s.c[7] << 12);
    f[5] = s.c[8] | (s.c[9] << 20);
    const uint32_t cur = s.window & 7u, nxt = (s.window >> 3) & 7u;
    f[6] = (s.c[9] >> 12) | ((1u << (8 + cur)) & 0x7F00u) | ((1u << (15 + nxt)) & 0x3F8000u) |
           (s.state != ST_RUNNING ? 1u << 24 : 0u);
    f[7] = pack_bf16((float)(L - (int)s.lines), (float)(M - (int)s.moves));   // features 214, 215
}

// argmax of outputs 0..3 (rotation, on the g = 0 lane) and of outputs 4..13 (location, spread over g = 1, 2, 3),
// lowest index on ties, NaN never wins.  All four lanes of a board return its action.
__device__ __forceinline__ uint32_t pick_action(const f32x4& c, int g, int lane) {
    float rv = -INFINITY; int ri = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c[k] > rv) { rv = c[k]; ri = k; }
    float lv = -INFINITY; int li = 99;
    const int first = 4 * g - 4;                         // location index of c[0] on this lane (g >= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (g >= 1 && first + k < 10 && c[k] > lv) { lv = c[k]; li = first + k; }
#pragma unroll
    for (int step = 16; step <= 32; step <<= 1) {
        const float ov = __shfl_xor(lv, step);
        const int oi = __shfl_xor(li, step);
        if (ov > lv || (ov == lv && oi < li)) { lv = ov; li = oi; }
    }
    if (li == 99) li = 0;
    const int rot = __shfl(ri, lane & 15);               // the g = 0 lane of this column
    return (uint32_t)(rot * 10 + li);
}

// features of both boards of column c: the lane's own board (t = g >> 1) and the one held by lane ^ 32
__device__ __forceinline__ void both_features(const uint32_t (&own)[8], int g, uint32_t (&fb)[2][8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t other = __shfl_xor(own[k], 32);
        fb[0][k] = (g >> 1) == 0 ? own[k] : other;
        fb[1][k] = (g >> 1) == 1 ? own[k] : other;
    }
}

}  // namespace p16
}  // namespace tpl
