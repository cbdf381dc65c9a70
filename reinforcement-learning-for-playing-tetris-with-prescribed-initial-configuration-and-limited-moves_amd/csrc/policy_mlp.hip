// policy_mlp.hip -- obs -> MLP -> action in ONE kernel (BASELINE configs[4]).
//
// Policy: Model(217, 14) of the reference (model/model.py:9-20, model/train.py:26): Linear 217-128, 128-128,
// 128-128, 128-128, 128-14 with ReLU between them.  This is the only GEMM-shaped work on the path, so it runs
// on the matrix cores: v_mfma_f32_32x32x16_bf16, bf16 operands, f32 accumulation, activations rounded to bf16
// between layers (what a bf16 torch module does).
//
// Everything is computed TRANSPOSED: boards run along the MFMA's N dimension (the lane), features along M/K.
//   H_{l+1}^T [out x boards] = W_{l+1} [out x in] . H_l^T [in x boards]
// With that orientation the 32x32 accumulator tile of one layer IS the B operand of the next (its column is on
// the lane, its rows are the next product's k): registers 8s..8s+7 of a tile, converted pairwise to bf16, are the
// fragment of k-step s -- no LDS round trip, no lane movement.  The k order inside such a fragment is permuted
// (element j of lane half h is row 16s + 8(j>>2) + 4h + (j&3)); the weights are pre-packed on the host in the
// same permuted order, lane-major, so an A fragment is one conflict-free ds_read_b128.
//
// All weights (158 KB of bf16 + 2 KB of f32 biases) stay resident in the CU's 160 KB LDS; a 256-thread workgroup
// (one wave per SIMD) loads them once and then loops over board tiles.  The observation is never materialised:
// each lane turns its board's 32-B state into the layer-1 B fragments directly (cells are 0/1, so a 4-bit
// nibble becomes two packed bf16 registers with two multiplies).
//
// Internal feature order of layer 1 (the packer permutes W1's columns, so callers keep the standard order of
// tpl_expand_obs): k = 20*x + y for the cell in row y, column x (that is how the state stores the board), then
// the 17 extras in their standard positions 200..216, then 7 zero pads.
#include "tpl_internal.h"
#include "tpl_step.h"

#include <cstring>
#include <vector>

namespace tpl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kHidden = 128;
constexpr int kObs = 217;
constexpr int kOut = 14;
constexpr int kKs1 = 14;                         // k-steps of layer 1: 224 = 217 padded to a multiple of 16
constexpr int kKsH = 8;                          // k-steps of a hidden layer: 128 / 16
constexpr int kMt = 4;                           // 32-row output tiles of a 128-wide layer

// byte offsets inside the packed image
constexpr int kOffW1 = 0;
constexpr int kOffW2 = kOffW1 + kMt * kKs1 * 1024;             // 57344
constexpr int kOffW3 = kOffW2 + kMt * kKsH * 1024;
constexpr int kOffW4 = kOffW3 + kMt * kKsH * 1024;
constexpr int kOffW5 = kOffW4 + kMt * kKsH * 1024;             // 155648
constexpr int kOffB = kOffW5 + kKsH * 512;                     // 159744: biases f32: 4 x 128, then 16
constexpr int kImageBytes = kOffB + (4 * kHidden + 16) * 4;    // 161856 <= 163840
static_assert(kImageBytes <= 160 * 1024, "policy image must fit the CU's LDS");
static_assert(kImageBytes % 16 == 0, "image is copied in 16-byte pieces");

// ---- host side: packing ----------------------------------------------------------------------------------
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// k inside a fragment: element j of lane half h in k-step s
static inline int frag_k(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// internal layer-1 feature index -> standard observation index (tpl_expand_obs order), or -1 for a pad
static inline int std_feature(int k) {
    if (k < 200) return (k % 20) * 10 + (k / 20);
    return k < kObs ? k : -1;
}

}  // namespace tpl

using namespace tpl;

extern "C" size_t tpl_policy_image_bytes(void) { return (size_t)kImageBytes; }

extern "C" int tpl_policy_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                               const float* b3, const float* w4, const float* b4, const float* w5, const float* b5,
                               void* image) {
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4 || !w5 || !b5 || !image)
        return fail_msg(TPL_ERR_ARG, "tpl_policy_pack: null pointer");
    std::vector<uint8_t> img((size_t)kImageBytes, 0);
    uint16_t* p = (uint16_t*)img.data();
    auto pack_layer = [&](int off, const float* w, int in, int ks, bool first) {
        for (int m = 0; m < kMt; ++m)
            for (int s = 0; s < ks; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int r = lane & 31, h = lane >> 5;
                        int k = frag_k(s, h, j);
                        if (first) k = std_feature(k);
                        const float v = (k >= 0 && k < in) ? w[(size_t)(32 * m + r) * in + k] : 0.0f;
                        p[off / 2 + (((m * ks + s) * 64 + lane) * 8 + j)] = bf16_rne(v);
                    }
    };
    pack_layer(kOffW1, w1, kObs, kKs1, true);
    pack_layer(kOffW2, w2, kHidden, kKsH, false);
    pack_layer(kOffW3, w3, kHidden, kKsH, false);
    pack_layer(kOffW4, w4, kHidden, kKsH, false);
    for (int s = 0; s < kKsH; ++s)                          // last layer: 14 rows, stored as 16, two lane halves
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r)
                for (int j = 0; j < 8; ++j) {
                    const float v = r < kOut ? w5[(size_t)r * kHidden + frag_k(s, h, j)] : 0.0f;
                    p[kOffW5 / 2 + (((s * 2 + h) * 16 + r) * 8 + j)] = bf16_rne(v);
                }
    float* bias = (float*)(img.data() + kOffB);
    const float* bs[4] = {b1, b2, b3, b4};
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < kHidden; ++k) bias[l * kHidden + k] = bs[l][k];
    for (int k = 0; k < kOut; ++k) bias[4 * kHidden + k] = b5[k];
    std::memcpy(image, img.data(), (size_t)kImageBytes);
    return TPL_OK;
}

namespace tpl {

// ---- device side -----------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}

// two cell bits -> two packed bf16 values (1.0 = 0x3F80)
__device__ __forceinline__ uint32_t bits_to_bf16x2(uint32_t two_bits) {
    return ((two_bits & 1u) | ((two_bits & 2u) << 15)) * 0x3F80u;
}

struct PolicyArgs {
    const uint4* plane_a;
    const uint4* plane_b;
    int64_t n;
    int32_t L, M;
    const uint4* image;     // packed weights, kImageBytes
    uint8_t* action;        // [n]
    float* logits;          // [n][14] or null
    uint32_t stagger;       // waves 4..7 of a 512-thread workgroup start this many x 1024 cycles late
};

// Two waves that share a SIMD and run the same program fall into lockstep: both in their MFMA loops, then both
// in their VALU epilogues, and the two pipes never overlap.  Delaying the second wave of every SIMD (waves 4-7:
// a workgroup's waves are dealt to the SIMDs round-robin) by a fraction of an iteration puts one wave's VALU
// phases under the other's MFMA phases.
__device__ __forceinline__ void stagger_second_wave(int wave, uint32_t units) {
    if (wave >= 4)
        for (uint32_t k = 0; k < units; ++k) __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles
}

// ReLU on two packed bf16 values: as signed 16-bit integers a negative bf16 (sign bit set, -0 included) is a
// negative number, so max(., 0) clears it and leaves non-negative values untouched -- one v_pk_max_i16 per pair,
// where fmaxf on the f32 accumulators costs two v_max_f32 per VALUE (canonicalise + max).  Rounding to bf16 first
// and clamping second gives the same result as the other order: rounding never changes the sign.
typedef __attribute__((ext_vector_type(2))) short i16x2;
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    const i16x2 zero = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, v), zero));
}

// convert an accumulator tile to bf16, apply ReLU, and hand it on as the two B fragments of the next layer
__device__ __forceinline__ void tile_to_frags(const f32x16& c, bf16x8& f0, bf16x8& f1) {
    uint4 lo, hi;
    lo.x = relu_bf16x2(pack_bf16(c[0], c[1]));
    lo.y = relu_bf16x2(pack_bf16(c[2], c[3]));
    lo.z = relu_bf16x2(pack_bf16(c[4], c[5]));
    lo.w = relu_bf16x2(pack_bf16(c[6], c[7]));
    hi.x = relu_bf16x2(pack_bf16(c[8], c[9]));
    hi.y = relu_bf16x2(pack_bf16(c[10], c[11]));
    hi.z = relu_bf16x2(pack_bf16(c[12], c[13]));
    hi.w = relu_bf16x2(pack_bf16(c[14], c[15]));
    f0 = __builtin_bit_cast(bf16x8, lo);
    f1 = __builtin_bit_cast(bf16x8, hi);
}

// accumulator tile m of a layer starts as the bias of its rows: row = (reg&3) + 8(reg>>2) + 4h
__device__ __forceinline__ f32x16 bias_tile(const float* bias, int m, int h) {
    f32x16 c;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *(const float4*)(bias + 32 * m + 8 * g + 4 * h);
        c[4 * g + 0] = b.x; c[4 * g + 1] = b.y; c[4 * g + 2] = b.z; c[4 * g + 3] = b.w;
    }
    return c;
}

// A fragment q of a layer = (output tile m = q / ks, k-step s = q % ks): consecutive fragments are 1 KiB apart
__device__ __forceinline__ bf16x8 a_frag(const uint8_t* lds, int w_off, int q, int lane) {
    return *(const bf16x8*)(lds + w_off + (q * 64 + lane) * 16);
}

// One layer with 128 outputs: xout^T[128 x boards] = relu(W . xin^T + b), both sides as MFMA fragments in registers.
// Order: output tile m outermost, so that
//   * tile m's epilogue (bf16 convert + ReLU: VALU) sits in program order BEHIND the MFMAs of tile m+1 and runs
//     while the matrix pipe works on them (two accumulator tiles alternate);
//   * the A fragments stream through a four-deep register window: the read of fragment q+4 is issued right behind
//     the MFMA that consumes fragment q, so an LDS read has four MFMAs (128 pipe cycles) to land.
// sched_barrier(0) pins that order; left alone the scheduler hoists every LDS read to the top and spills.
template <int kNt, int kKs>
__device__ __forceinline__ void dense128(const uint8_t* lds, int w_off, const float* bias, int lane, int h,
                                         const bf16x8 (&xin)[kNt][kKs], bf16x8 (&xout)[kNt][kKsH]) {
    f32x16 acc[2][kNt];
    bf16x8 aq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aq[q] = a_frag(lds, w_off, q, lane);
#pragma unroll
    for (int m = 0; m < kMt; ++m) {
        const f32x16 b = bias_tile(bias, m, h);
#pragma unroll
        for (int t = 0; t < kNt; ++t) acc[m & 1][t] = b;
#pragma unroll
        for (int s = 0; s < kKs; ++s) {
            const int q = m * kKs + s;
#pragma unroll
            for (int t = 0; t < kNt; ++t) acc[m & 1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[q & 3], xin[t][s], acc[m & 1][t], 0, 0, 0);
            if (q + 4 < kMt * kKs) aq[q & 3] = a_frag(lds, w_off, q + 4, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (m > 0) {
#pragma unroll
            for (int t = 0; t < kNt; ++t) tile_to_frags(acc[(m - 1) & 1][t], xout[t][2 * (m - 1)], xout[t][2 * (m - 1) + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int t = 0; t < kNt; ++t) tile_to_frags(acc[(kMt - 1) & 1][t], xout[t][2 * (kMt - 1)], xout[t][2 * (kMt - 1) + 1]);
}

// weights -> LDS, eight 16-B loads in flight per thread (a load-store-load-store loop would pay the L2 latency
// forty times over).  Ends with a barrier.
template <int kThreads>
__device__ __forceinline__ void load_image(uint4* s_image, const uint4* image) {
    constexpr int kPieces = kImageBytes / 16;
    for (int base = 0; base < kPieces; base += kThreads * 8) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads + (int)threadIdx.x;
            v[u] = image[i < kPieces ? i : kPieces - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads + (int)threadIdx.x;
            if (i < kPieces) s_image[i] = v[u];
        }
    }
    __syncthreads();
}

// a board -> its cell bit vector in the internal order (bit 20x + y; extras as bits 200..213 and 216) and the
// two numeric features (L_rem, M_rem) as a packed bf16 pair
__device__ __forceinline__ void board_features(const Board& s, int L, int M, uint32_t (&cw)[7], uint32_t& lm) {
    cw[0] = s.c[0] | (s.c[1] << 20);
    cw[1] = (s.c[1] >> 12) | (s.c[2] << 8) | (s.c[3] << 28);
    cw[2] = (s.c[3] >> 4) | (s.c[4] << 16);
    cw[3] = (s.c[4] >> 16) | (s.c[5] << 4) | (s.c[6] << 24);
    cw[4] = (s.c[6] >> 8) | (s.c[7] << 12);
    cw[5] = s.c[8] | (s.c[9] << 20);
    const uint32_t cur = s.window & 7u, nxt = (s.window >> 3) & 7u;
    cw[6] = (s.c[9] >> 12) | ((1u << (8 + cur)) & 0x7F00u) | ((1u << (15 + nxt)) & 0x3F8000u) |
            (s.state != ST_RUNNING ? 1u << 24 : 0u);
    lm = pack_bf16((float)(L - (int)s.lines), (float)(M - (int)s.moves));   // features 214, 215
}

// the five layers for kNt tiles of 32 boards; c[t] ends up holding the 14 logits of board (t, r): outputs
// 4h + {0..3} in c[t][0..3] and 8 + 4h + {0..3} in c[t][4..7]
template <int kNt>
__device__ __forceinline__ void policy_logits(const uint8_t* lds, int lane, int h, int r, const uint32_t (&cw)[kNt][7],
                                              const uint32_t (&lm)[kNt], f32x16 (&c)[kNt]) {
    const float* bias = (const float*)(lds + kOffB);
    // ---- layer 1: 224 (217) -> 128; its B fragments are made from the cell bits, once, up front
    bf16x8 x0[kNt][kKs1];
#pragma unroll
    for (int t = 0; t < kNt; ++t)
#pragma unroll
        for (int s = 0; s < kKs1; ++s) {
            const uint32_t half16 = (cw[t][s >> 1] >> ((s & 1) * 16)) >> (4 * h);
            uint4 q;
            q.x = bits_to_bf16x2(half16);            // k = 16s + 4h + {0,1}
            q.y = bits_to_bf16x2(half16 >> 2);       //                 {2,3}
            q.z = bits_to_bf16x2(half16 >> 8);       // k = 16s + 8 + 4h + {0,1}
            q.w = bits_to_bf16x2(half16 >> 10);
            if (s == 13 && h == 1) q.y = lm[t];      // k = 214, 215: L_rem, M_rem
            x0[t][s] = __builtin_bit_cast(bf16x8, q);
        }
    bf16x8 xa[kNt][kKsH], xb[kNt][kKsH];
    dense128<kNt, kKs1>(lds, kOffW1, bias, lane, h, x0, xa);
    // ---- layers 2-4
    dense128<kNt, kKsH>(lds, kOffW2, bias + 1 * kHidden, lane, h, xa, xb);
    dense128<kNt, kKsH>(lds, kOffW3, bias + 2 * kHidden, lane, h, xb, xa);
    dense128<kNt, kKsH>(lds, kOffW4, bias + 3 * kHidden, lane, h, xa, xb);
    const bf16x8 (&x)[kNt][kKsH] = xb;
    // ---- layer 5: 128 -> 14 (rows 0..13 of one tile; lanes of rows 16..31 re-read rows 0..15, unused)
#pragma unroll
    for (int t = 0; t < kNt; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < 2) b = *(const float4*)(bias + 4 * kHidden + 8 * g + 4 * h);
            c[t][4 * g + 0] = b.x; c[t][4 * g + 1] = b.y; c[t][4 * g + 2] = b.z; c[t][4 * g + 3] = b.w;
        }
#pragma unroll
        for (int s = 0; s < kKsH; ++s) {
            const bf16x8 a = *(const bf16x8*)(lds + kOffW5 + ((s * 2 + h) * 16 + (r & 15)) * 16);
            c[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x[t][s], c[t], 0, 0, 0);
        }
    }
}

// argmax of outputs 0..3 (rotation) and of outputs 4..13 (location), lowest index on ties, NaN never wins; the two
// lane halves of a board exchange their location candidates.  Every lane of the pair returns the action.
__device__ __forceinline__ uint32_t pick_action(const f32x16& c, int h) {
    float rv = -INFINITY; int ri = 0;                    // rotation: all four live on the h = 0 lane
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c[k] > rv) { rv = c[k]; ri = k; }
    float lv = -INFINITY; int li = 99;                   // location candidates of this lane, ascending index
    if (h == 1) {                                        // outputs 4..7 -> loc 0..3, outputs 12, 13 -> loc 8, 9
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c[k] > lv) { lv = c[k]; li = k; }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (c[4 + k] > lv) { lv = c[4 + k]; li = 8 + k; }
    } else {                                             // outputs 8..11 -> loc 4..7
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c[4 + k] > lv) { lv = c[4 + k]; li = 4 + k; }
    }
    const float ov = __shfl_xor(lv, 32);
    const int oi = __shfl_xor(li, 32);
    if (ov > lv || (ov == lv && oi < li)) { lv = ov; li = oi; }
    if (li == 99) li = 0;
    const int partner_ri = __shfl_xor(ri, 32);           // (shuffles stay outside any lane-dependent branch)
    const int rot = h == 0 ? ri : partner_ri;            // the h = 1 lane takes the rotation from its partner
    return (uint32_t)(rot * 10 + li);
}

template <int kNt, int kThreads>
__global__ __launch_bounds__(kThreads, kThreads / 256) void policy_kernel(const PolicyArgs p) {
    __shared__ uint4 s_image[kImageBytes / 16];
    load_image<kThreads>(s_image, p.image);
    const uint8_t* lds = (const uint8_t*)s_image;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int kWaves = kThreads / 64;
    const int64_t tiles = (p.n + 32 * kNt - 1) / (32 * kNt);
    stagger_second_wave(wave, p.stagger);
    for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += (int64_t)gridDim.x * kWaves) {
        uint32_t cw[kNt][7];
        uint32_t lm[kNt];
        bool valid[kNt];
#pragma unroll
        for (int t = 0; t < kNt; ++t) {
            const int64_t b = tile * (32 * kNt) + t * 32 + r;
            valid[t] = b < p.n;
            Board s;
            if (valid[t]) {
                unpack_board(p.plane_a[b], p.plane_b[b], s);
            } else {
#pragma unroll
                for (int c = 0; c < kCols; ++c) s.c[c] = 0;
                s.window = 0x3FFFFFFFu; s.state = 0; s.lines = 0; s.moves = 0; s.episode = 0;
            }
            board_features(s, p.L, p.M, cw[t], lm[t]);
        }
        f32x16 c[kNt];
        policy_logits<kNt>(lds, lane, h, r, cw, lm, c);
#pragma unroll
        for (int t = 0; t < kNt; ++t) {
            const int64_t b = tile * (32 * kNt) + t * 32 + r;
            if (p.logits && valid[t]) {
                float* o = p.logits + b * kOut;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[4 * h + k] = c[t][k];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (8 + 4 * h + k < kOut) o[8 + 4 * h + k] = c[t][4 + k];
            }
            const uint32_t action = pick_action(c[t], h);
            if (h == 0 && valid[t]) p.action[b] = (uint8_t)action;
        }
    }
}

// epsilon-greedy exploration, keyed by (seed, global board, global step): with probability eps_q24 / 2^24 the
// action is replaced by a uniform one in [0, 40)
__device__ __forceinline__ uint32_t explore(uint32_t action, uint64_t seed, uint64_t g, uint32_t step, uint32_t eps_q24) {
    uint32_t u = fmix32((uint32_t)g ^ ((uint32_t)(g >> 32) * 0x9E3779B9u) ^ (uint32_t)seed ^ 0x51ED270Bu);
    u = fmix32(u + step * 0x9E3779B1u + (uint32_t)(seed >> 32));
    return (u >> 8) < eps_q24 ? ((u & 0xFFu) * 40u) >> 8 : action;
}

__global__ __launch_bounds__(kBlock) void explore_kernel(uint8_t* action, int64_t n, uint64_t seed, int64_t global_offset,
                                                        uint32_t step, uint32_t eps_q24) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) action[i] = (uint8_t)explore(action[i], seed, (uint64_t)(global_offset + i), step, eps_q24);
}

// T iterations of (policy -> epsilon-greedy -> step) in ONE launch: weights stay in LDS, boards stay in
// registers, nothing but the trajectory leaves the chip.  Exactly T x (tpl_policy_act, tpl_explore_actions,
// tpl_step).  One wave owns 32 boards; the two lane halves of a board carry identical copies of its state and
// advance it identically, the h = 0 half writes.
struct ActorArgs {
    StepArgs s;
    const uint4* image;
    uint32_t T, step0, eps_q24, stagger;
    uint64_t explore_seed;
    uint8_t* actions;           // [T][n] or null
    float* rewards;             // [T][n] or null
    uint8_t* dones;             // [T][n] or null
    uint4* states_a;            // [T][n] or null: plane-A word of the board BEFORE step t (the observation)
    uint4* states_b;            // [T][n] or null
};

template <bool kAutoReset>
__global__ __launch_bounds__(512, 2) void actor_rollout_kernel(const ActorArgs q) {
    const StepArgs& p = q.s;
    __shared__ uint4 s_image[kImageBytes / 16];
    __shared__ ShapeWord s_shape[32];
    __shared__ uint32_t s_stat[4];
    if (threadIdx.x < 32) s_shape[threadIdx.x] = kShapeTable[threadIdx.x];
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    load_image<512>(s_image, q.image);
    const uint8_t* lds = (const uint8_t*)s_image;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t tiles = (p.n + 31) / 32;
    Tally tally;
    stagger_second_wave(wave, q.stagger);
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < tiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t b = tile * 32 + r;
        const bool valid = b < p.n;
        Board s;
        if (valid) {
            unpack_board(p.plane_a[b], p.plane_b[b], s);
        } else {
#pragma unroll
            for (int c = 0; c < kCols; ++c) s.c[c] = 0;
            s.window = 0x3FFFFFFFu; s.state = ST_LOST_LIMIT; s.lines = 0; s.moves = 0; s.episode = 0;   // frozen filler
        }
        uint32_t cfg = current_config(s, p, (uint32_t)b);
        for (uint32_t t = 0; t < q.T; ++t) {
            if (q.states_a && valid && h == 0) {
                uint4 A, B;
                pack_board(s, A, B);
                q.states_a[(size_t)t * p.n + b] = A;
                q.states_b[(size_t)t * p.n + b] = B;
            }
            uint32_t cw[1][7], lm[1];
            board_features(s, (int)p.L, (int)p.M, cw[0], lm[0]);
            f32x16 c[1];
            policy_logits<1>(lds, lane, h, r, cw, lm, c);
            uint32_t action = pick_action(c[0], h);
            action = explore(action, q.explore_seed, (uint64_t)(p.global_offset + b), q.step0 + t, q.eps_q24);
            const uint32_t rot = action / 10u, loc = action - rot * 10u;
            float reward;
            Tally mine;                                  // only the h = 0 copy of a board counts its episodes
            const bool done = advance_board<kAutoReset>(s, cfg, rot, loc, p, (uint32_t)b, s_shape, reward, mine);
            if (valid && h == 0) {
                tally.episodes += mine.episodes; tally.lines += mine.lines;
                tally.wins += mine.wins; tally.topouts += mine.topouts;
                if (q.actions) q.actions[(size_t)t * p.n + b] = (uint8_t)action;
                if (q.rewards) q.rewards[(size_t)t * p.n + b] = reward;
                if (q.dones) q.dones[(size_t)t * p.n + b] = done ? 1 : 0;
            }
        }
        if (valid && h == 0) {
            uint4 A, B;
            pack_board(s, A, B);
            p.plane_a[b] = A;
            p.plane_b[b] = B;
        }
    }
    flush_tally(tally, s_stat, p.stats);
}

}  // namespace tpl

extern "C" int tpl_policy_act(tpl_env* e, const void* image, uint8_t* action, float* logits, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!image || !action) return fail_msg(TPL_ERR_ARG, "image/action is null");
    if (((uintptr_t)image & 15u) != 0) return fail_msg(TPL_ERR_ARG, "image must be 16-byte aligned");
    DeviceGuard guard(e->device);
    PolicyArgs p{};
    p.plane_a = e->plane_a; p.plane_b = e->plane_b; p.n = e->n; p.L = e->L; p.M = e->M;
    p.image = (const uint4*)image; p.action = action; p.logits = logits; p.stagger = (uint32_t)e->policy_stagger;
    // one resident workgroup per CU (the weights fill its LDS), looping over board tiles.
    // variant 0: 4 waves x 64 boards (one wave per SIMD); variant 1: 8 waves x 32 boards (two per SIMD, so one
    // wave's epilogue overlaps the other's MFMAs)
    if (e->policy_variant == 0) {
        const int64_t groups = ((e->n + 63) / 64 + 3) / 4;
        hipLaunchKernelGGL((policy_kernel<2, 256>), dim3((unsigned)(groups < 256 ? groups : 256)), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        const int64_t groups = ((e->n + 31) / 32 + 7) / 8;
        hipLaunchKernelGGL((policy_kernel<1, 512>), dim3((unsigned)(groups < 256 ? groups : 256)), dim3(512), 0, (hipStream_t)stream, p);
    }
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

extern "C" int tpl_explore_actions(tpl_env* e, uint8_t* action, float epsilon, uint64_t seed, uint32_t step, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!action) return fail_msg(TPL_ERR_ARG, "action is null");
    if (!(epsilon >= 0.0f && epsilon <= 1.0f)) return fail_msg(TPL_ERR_ARG, "epsilon must be in [0, 1]");
    DeviceGuard guard(e->device);
    const uint32_t eps_q24 = (uint32_t)(epsilon * 16777216.0f);
    hipLaunchKernelGGL(explore_kernel, dim3((unsigned)((e->n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       action, e->n, seed, e->global_offset, step, eps_q24);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

extern "C" int tpl_actor_rollout(tpl_env* e, const void* image, int32_t num_steps, float epsilon, uint64_t seed, uint32_t step0,
                                 uint8_t* actions, float* rewards, uint8_t* dones, void* states_a, void* states_b,
                                 void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!image) return fail_msg(TPL_ERR_ARG, "image is null");
    if (((uintptr_t)image & 15u) != 0) return fail_msg(TPL_ERR_ARG, "image must be 16-byte aligned");
    if (num_steps < 1) return fail_msg(TPL_ERR_ARG, "num_steps must be >= 1");
    if (!(epsilon >= 0.0f && epsilon <= 1.0f)) return fail_msg(TPL_ERR_ARG, "epsilon must be in [0, 1]");
    if ((states_a == nullptr) != (states_b == nullptr)) return fail_msg(TPL_ERR_ARG, "states_a and states_b go together");
    if (e->auto_reset && e->pool.n_cfg == 0) return fail_msg(TPL_ERR_STATE, "auto_reset needs tpl_load_configs first");
    DeviceGuard guard(e->device);
    ActorArgs q{};
    q.s = make_args(e);
    q.image = (const uint4*)image; q.T = (uint32_t)num_steps; q.step0 = step0;
    q.eps_q24 = (uint32_t)(epsilon * 16777216.0f); q.explore_seed = seed; q.stagger = (uint32_t)e->policy_stagger;
    q.actions = actions; q.rewards = rewards; q.dones = dones; q.states_a = (uint4*)states_a; q.states_b = (uint4*)states_b;
    const int64_t groups = ((e->n + 31) / 32 + 7) / 8;
    const dim3 grid((unsigned)(groups < 256 ? groups : 256)), block(512);
    if (e->auto_reset) hipLaunchKernelGGL(actor_rollout_kernel<true>, grid, block, 0, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(actor_rollout_kernel<false>, grid, block, 0, (hipStream_t)stream, q);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}
