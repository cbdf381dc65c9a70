// policy_mlp.hip -- obs -> MLP -> action in ONE kernel, and T x (policy, explore, step) in one launch
// (BASELINE configs[4]).
//
// Policy: Model(217, 14) of the reference (model/model.py:9-20, model/train.py:26): Linear 217-128, 128-128,
// 128-128, 128-128, 128-14 with ReLU between them.  This is the only GEMM-shaped work on the path, so it runs on
// the matrix cores: v_mfma_f32_16x16x32_bf16, bf16 operands, f32 accumulation, activations rounded to bf16
// between layers (what a bf16 torch module does).
//
// Everything is computed TRANSPOSED: boards run along the MFMA's N dimension (the lane), features along M/K:
//   H_{l+1}^T [out x boards] = W_{l+1} [out x in] . H_l^T [in x boards]
// so that the accumulator tiles of one layer ARE the B operand of the next (their column is on the lane, their
// rows are the next product's k) -- no LDS round trip, no lane movement.  All weights (158 KB of bf16 + 2 KB of
// f32 biases) stay resident in the CU's 160 KB LDS, pre-packed on the host lane-major in the permuted k order the
// hand-off produces, so an A fragment is one conflict-free ds_read_b128; a 512-thread workgroup loads them once
// and loops over board tiles.  The observation is never materialised: each lane turns its board's 32-B state into
// the layer-1 B fragments directly (a 0/1 feature is the single bf16 bit 0x4000 = 2.0 with halved weights, so a
// fragment register is one shift and one mask of the feature word).
// Internal feature order of layer 1 (the packer permutes W1's columns, callers keep tpl_expand_obs' order):
// k = 20x + y for the cell in row y, column x -- how the state stores the board -- then the 17 extras at 200..216.
//
// (The same kernels were also built on v_mfma_f32_32x32x16_bf16; both shapes take 33-34 us per 262,144 boards.
// This one is kept because the 14-row head is one 16-row tile and the file has a single geometry.)
//
// Geometry of a wave (this file; kNt = 4): 64 boards = FOUR N tiles of 16, one board per lane.  Lane l = (c = l & 15,
// g = l >> 4) owns board (t = g, c) -- board 16 t + c of the wave's tile of 64.  For the matrix products every lane needs the
// features of all four boards of its column c (one per N tile); the other three come from lanes 16 t + c with cross-lane
// moves (column_features), and of the four tiles' logits the lane keeps its own board's action (own_action).  Every A
// fragment and bias read feeds four MFMAs.  (The float32 and split kernels -- policy_f32.hip, policy_split.hip -- keep the
// earlier geometry: 32 boards = two N tiles per wave, lanes g and g ^ 1 holding copies of board (t = g >> 1, c).)
//
// Fragment maps (cdna_hip_programming.md section 3): A: lane holds A[row c][k = 8g + j]; B: B[k = 8g + j][col c];
// C/D: D[row = 4g + reg][col c], reg 0..3.  A layer's output tile m (16 rows) therefore leaves rows 16m + 4g + reg
// on lane (c, g); two consecutive tiles give the eight values a B fragment of the next layer needs, in the
// permuted order k = 32s + 16(j >> 2) + 4g + (j & 3) -- which is the order the weights are pre-packed in.
#include "tpl_internal.h"
#include "tpl_policy.h"
#include "tpl_step.h"

#include <cstring>
#include <vector>

namespace tpl {
namespace p16 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) short i16x2;

constexpr int kKs1 = 7;        // k-steps of 32 in layer 1: 224 = 217 padded
constexpr int kKsH = 4;        // k-steps of a hidden layer
constexpr int kMt = 8;         // 16-row output tiles of a 128-wide layer

constexpr int kOffW1 = 0;
constexpr int kOffW2 = kOffW1 + kMt * kKs1 * 1024;             // 57344
constexpr int kOffW3 = kOffW2 + kMt * kKsH * 1024;
constexpr int kOffW4 = kOffW3 + kMt * kKsH * 1024;
constexpr int kOffW5 = kOffW4 + kMt * kKsH * 1024;             // 155648
constexpr int kOffB = kOffW5 + kKsH * 1024;                    // 159744
constexpr int kImageBytes = kOffB + (4 * kHidden + 16) * 4;    // 161856
static_assert(kImageBytes <= 160 * 1024 - 512, "policy image + shape table must fit the CU's LDS");

}  // namespace p16
}  // namespace tpl

using namespace tpl;
using namespace tpl::p16;

extern "C" size_t tpl_policy_image_bytes(void) { return (size_t)kImageBytes; }

extern "C" int tpl_policy_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                               const float* b3, const float* w4, const float* b4, const float* w5, const float* b5,
                               void* image) {
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4 || !w5 || !b5 || !image)
        return fail_msg(TPL_ERR_ARG, "tpl_policy_pack: null pointer");
    std::vector<uint8_t> img((size_t)kImageBytes, 0);
    uint16_t* p = (uint16_t*)img.data();
    auto pack_layer = [&](int off, const float* w, int rows, int in, int mt, int ks, bool first) {
        for (int m = 0; m < mt; ++m)
            for (int s = 0; s < ks; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int c = lane & 15, g = lane >> 4;
                        int k = first ? frag_k1(s, g, j) : frag_k(s, g, j);
                        // a 0/1 feature enters layer 1 as the single bf16 bit 0x4000 = 2.0, so its weights are halved
                        // (exact: a power of two); the two counters 214, 215 enter as numbers
                        const float scale = (first && k != 214 && k != 215) ? 0.5f : 1.0f;
                        if (first) k = std_feature(k);
                        const int row = 16 * m + c;
                        const float v = (k >= 0 && k < in && row < rows) ? scale * w[(size_t)row * in + k] : 0.0f;
                        p[off / 2 + (((m * ks + s) * 64 + lane) * 8 + j)] = bf16_rne(v);
                    }
    };
    pack_layer(kOffW1, w1, kHidden, kObs, kMt, kKs1, true);
    pack_layer(kOffW2, w2, kHidden, kHidden, kMt, kKsH, false);
    pack_layer(kOffW3, w3, kHidden, kHidden, kMt, kKsH, false);
    pack_layer(kOffW4, w4, kHidden, kHidden, kMt, kKsH, false);
    pack_layer(kOffW5, w5, kOut, kHidden, 1, kKsH, false);
    float* bias = (float*)(img.data() + kOffB);
    const float* bs[4] = {b1, b2, b3, b4};
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < kHidden; ++k) bias[l * kHidden + k] = bs[l][k];
    for (int k = 0; k < kOut; ++k) bias[4 * kHidden + k] = b5[k];
    std::memcpy(image, img.data(), (size_t)kImageBytes);
    return TPL_OK;
}

namespace tpl {
namespace p16 {

// ReLU on two packed bf16 values: a negative bf16 is a negative int16, so one v_pk_max_i16 clears it
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    const i16x2 zero = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, v), zero));
}

// weights -> LDS by LDS-DMA (`global_load_lds_dwordx4`): one instruction of a wave moves 64 consecutive 16-byte
// pieces from memory straight into LDS at a wave-uniform base.  No registers are held, so the whole image is in
// flight at once (the register form -- load every piece of a thread, then store them -- measured 0.8 us slower on
// the 35-us policy kernel).  Tried and dropped: starting layer 1 of the first tile while W2..W5 are still arriving
// (two LDS arrays, a counted wait, a second barrier after layer 1) -- 1.5 us SLOWER.
template <int kThreads>
__device__ __forceinline__ void load_image(uint4* s_image, const uint4* image) {
    typedef __attribute__((address_space(1))) const void global_ptr;
    typedef __attribute__((address_space(3))) void lds_ptr;
    constexpr int kPieces = kImageBytes / 16;
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    for (int chunk = wave; chunk * 64 < kPieces; chunk += kThreads / 64) {
        const int i = chunk * 64 + lane;
        if (i < kPieces)                                        // lane k of the instruction lands at base + 16 k
            __builtin_amdgcn_global_load_lds((global_ptr*)(image + i), (lds_ptr*)(s_image + chunk * 64), 16, 0, 0);
    }
    __syncthreads();                                            // (its fence waits for the transfers)
}

// Where a lane reads the image from.  A ds_read carries a 16-bit byte offset, the image is 158 KB: read as
// `lds + constant + 16 lane` every fragment past the first 64 KB gets an address register of its own -- ninety-six of them,
// all loop-invariant, all live across the whole kernel (that, not the arithmetic, was what filled the register file and put
// two values in scratch).  So the lane keeps THREE bases, 64 KB apart, made opaque so that they are not folded back into
// one, and every read is base[offset >> 16] + (offset & 0xFFFF) with the low part in the instruction.
// (The bases are 32-bit LDS addresses and stay typed as such: through a generic pointer the reads would become flat loads.)
typedef __attribute__((address_space(3))) const uint8_t lds_byte;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
struct LdsBases {
    lds_byte* frag[3];          // image + 16 lane + 65536 i: A fragments
    lds_byte* bias;             // image + 131072 + 16 (lane >> 4): the f32 biases sit in the third segment
};
__device__ __forceinline__ LdsBases lds_bases(const uint8_t* lds, int lane) {
    LdsBases b;
    lds_byte* image = (lds_byte*)lds;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        b.frag[i] = image + 65536 * i + 16 * lane;
        asm volatile("" : "+v"(b.frag[i]));
    }
    b.bias = image + 131072 + 16 * (lane >> 4);
    asm volatile("" : "+v"(b.bias));
    return b;
}
__device__ __forceinline__ bf16x8 a_frag(const LdsBases& at, int w_off, int q) {
    const int off = w_off + q * 1024;                           // a compile-time constant wherever this is called
    return *(lds_bf16x8*)(at.frag[off >> 16] + (off & 0xFFFF));
}
// bias[k .. k + 3], k = 4 (lane >> 4) + first: `first` floats into the bias block
__device__ __forceinline__ float4 bias4(const LdsBases& at, int first) {
    const f32x4 v = *(lds_f32x4*)(at.bias + (kOffB - 131072) + 4 * first);
    return make_float4(v[0], v[1], v[2], v[3]);
}

constexpr int kNt = 4;          // N tiles of 16 boards per wave: 64 boards, one per lane

// One layer with 16*kTiles outputs on the wave's four N tiles.  Output tile m outermost; the A fragments stream through a
// four-deep register window (the read of fragment q+4 is issued right behind the four MFMAs that consume fragment q);
// tile m's epilogue sits behind the MFMAs of tile m+1.  xout[t][s] collects tiles 2s and 2s+1.
template <int kTiles, int kKs, bool kRelu>
__device__ __forceinline__ void dense(const LdsBases& at, int w_off, int bias_first, const uint4 (&xin)[kNt][kKs],
                                      uint4 (&xout)[kNt][kTiles / 2 > 0 ? kTiles / 2 : 1], f32x4 (&last)[kNt]) {
    f32x4 acc[2][kNt];
    bf16x8 aq[4];
#pragma unroll
    for (int q = 0; q < 4 && q < kTiles * kKs; ++q) aq[q] = a_frag(at, w_off, q);
    // the bias of tile m+1 is read while tile m is being multiplied: read just in time, the accumulator's initial
    // value would stall the first MFMA of every tile for one LDS latency
    float4 bnext = bias4(at, bias_first);
    const auto hand_off = [&](int pm) {          // f32 accumulators of output tile pm -> ReLU -> the next layer's bf16 B values
#pragma unroll
        for (int t = 0; t < kNt; ++t) {
            const uint32_t lo = relu_bf16x2(pack_bf16(acc[pm & 1][t][0], acc[pm & 1][t][1]));
            const uint32_t hi = relu_bf16x2(pack_bf16(acc[pm & 1][t][2], acc[pm & 1][t][3]));
            if (pm & 1) { xout[t][pm >> 1].z = lo; xout[t][pm >> 1].w = hi; }
            else { xout[t][pm >> 1].x = lo; xout[t][pm >> 1].y = hi; }
        }
    };
#pragma unroll
    for (int m = 0; m < kTiles; ++m) {
        const float4 b = bnext;
        if (m + 1 < kTiles) bnext = bias4(at, bias_first + 16 * (m + 1));
#pragma unroll
        for (int t = 0; t < kNt; ++t) { acc[m & 1][t][0] = b.x; acc[m & 1][t][1] = b.y; acc[m & 1][t][2] = b.z; acc[m & 1][t][3] = b.w; }
#pragma unroll
        for (int s = 0; s < kKs; ++s) {
            const int q = m * kKs + s;
#pragma unroll
            for (int t = 0; t < kNt; ++t)
                acc[m & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[q & 3], __builtin_bit_cast(bf16x8, xin[t][s]), acc[m & 1][t], 0, 0, 0);
            if (q + 4 < kTiles * kKs) aq[q & 3] = a_frag(at, w_off, q + 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kRelu && m > 0) {
            hand_off(m - 1);
            // (tried: sched_group_barrier patterns that issue four hand-off instructions and a fragment read behind every
            // k-step's four MFMAs -- the schedule came out as asked and the kernel 3 % slower: profiles/NOTES.md, round 4)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (kRelu) hand_off(kTiles - 1);
    else {
#pragma unroll
        for (int t = 0; t < kNt; ++t) last[t] = acc[(kTiles - 1) & 1][t];
    }
}

// Layer 1: its B fragments are made from the boards' feature bits, sixteen registers per k-step for the four tiles -- all
// seven k-steps at once would be 112 -- so here the K-STEPS run outermost and the accumulators of all eight output tiles stay
// live (128 registers), each B fragment made once and used by eight MFMAs.  Register i of lane group g holds bits 4g + i and
// 16 + 4g + i of feature word s as the bf16 pattern 0x4000 (= 2.0; the packer halves the weights): one shift, one mask.
__device__ __forceinline__ void dense_first(const LdsBases& at, int g, const uint32_t (&fb)[kNt][8], uint4 (&xout)[kNt][kMt / 2]) {
    f32x4 acc[kMt][kNt];
#pragma unroll
    for (int m = 0; m < kMt; ++m) {
        const float4 b = bias4(at, 16 * m);
#pragma unroll
        for (int t = 0; t < kNt; ++t) { acc[m][t][0] = b.x; acc[m][t][1] = b.y; acc[m][t][2] = b.z; acc[m][t][3] = b.w; }
    }
    bf16x8 aq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aq[q] = a_frag(at, kOffW1, (q % kMt) * kKs1 + q / kMt);          // order of use: (s, m), m fastest
#pragma unroll
    for (int s = 0; s < kKs1; ++s) {
        uint4 x[kNt];
#pragma unroll
        for (int t = 0; t < kNt; ++t) {
            const uint32_t u = fb[t][s] >> (4 * g);
            x[t].x = (u << 14) & 0x40004000u;
            x[t].y = (u << 13) & 0x40004000u;
            x[t].z = (u << 12) & 0x40004000u;
            x[t].w = (u << 11) & 0x40004000u;
            if (s == 6 && g == 1) {                      // bits 22, 23 of word 6 = features 214, 215: L_rem, M_rem
                x[t].z |= fb[t][7] << 16;
                x[t].w |= fb[t][7] & 0xFFFF0000u;
            }
        }
#pragma unroll
        for (int m = 0; m < kMt; ++m) {
            const int q = s * kMt + m;                   // position in the order of use
#pragma unroll
            for (int t = 0; t < kNt; ++t)
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[q & 3], __builtin_bit_cast(bf16x8, x[t]), acc[m][t], 0, 0, 0);
            if (q + 4 < kKs1 * kMt) aq[q & 3] = a_frag(at, kOffW1, ((q + 4) % kMt) * kKs1 + (q + 4) / kMt);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int m = 0; m < kMt; ++m)
#pragma unroll
        for (int t = 0; t < kNt; ++t) {
            const uint32_t lo = relu_bf16x2(pack_bf16(acc[m][t][0], acc[m][t][1]));
            const uint32_t hi = relu_bf16x2(pack_bf16(acc[m][t][2], acc[m][t][3]));
            if (m & 1) { xout[t][m >> 1].z = lo; xout[t][m >> 1].w = hi; }
            else { xout[t][m >> 1].x = lo; xout[t][m >> 1].y = hi; }
        }
}

// five layers for the wave's four N tiles; fb[t] = features of board (t, c).  Returns the logits tiles: lane (c, g)
// holds outputs 4g + reg of board (t, c) in logits[t][reg].
__device__ __forceinline__ void policy_logits(const LdsBases& at, int g, const uint32_t (&fb)[kNt][8], f32x4 (&logits)[kNt]) {
    uint4 xa[kNt][kKsH], xb[kNt][kKsH];
    f32x4 unused[kNt];
    dense_first(at, g, fb, xa);
    dense<kMt, kKsH, true>(at, kOffW2, 1 * kHidden, xa, xb, unused);
    dense<kMt, kKsH, true>(at, kOffW3, 2 * kHidden, xb, xa, unused);
    dense<kMt, kKsH, true>(at, kOffW4, 3 * kHidden, xa, xb, unused);
    uint4 none[kNt][1];
    dense<1, kKsH, false>(at, kOffW5, 4 * kHidden, xb, none, logits);
}

// the features of all four boards of the lane's column c (board (t, c) lives on lane 16 t + c), and the lane's own action
// out of the four tiles' logits
__device__ __forceinline__ void column_features(const uint32_t (&own)[8], int c, uint32_t (&fb)[kNt][8]) {
#pragma unroll
    for (int t = 0; t < kNt; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) fb[t][k] = __shfl(own[k], 16 * t + c);
}
__device__ __forceinline__ uint32_t own_action(const f32x4 (&lg)[kNt], int g, int lane) {
    uint32_t action = 0;
#pragma unroll
    for (int t = 0; t < kNt; ++t) {
        const uint32_t a = pick_action(lg[t], g, lane);          // on all four lanes of column c: the action of board (t, c)
        action = g == t ? a : action;
    }
    return action;
}

struct PolicyArgs {
    const uint4* plane_a;
    const uint4* plane_b;
    int64_t n;
    int32_t L, M;
    const uint4* image;
    uint8_t* action;
    float* logits;
    unsigned long long* diag;   // diagnostic build only
};

// two waves per SIMD.  (Twelve waves -- three per SIMD, the kernel fits 168 registers -- measured 6 % slower; halving
// the A-fragment LDS reads, as a timing experiment, gained 3 %: the kernel is bound by neither occupancy nor LDS
// bandwidth but by the matrix pipe at the clock the chip holds.)
constexpr int kPolicyWaves = 8;
__global__ __launch_bounds__(64 * kPolicyWaves) void policy_kernel(const PolicyArgs p) {
    __shared__ uint4 s_image[kImageBytes / 16];
    const uint8_t* lds = (const uint8_t*)s_image;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const LdsBases at = lds_bases(lds, lane);
#ifdef TPL_DIAG_CLOCK
    const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int64_t tiles = (p.n + 63) / 64;
    const int64_t tile0 = (int64_t)blockIdx.x * kPolicyWaves + wave, tile_step = (int64_t)gridDim.x * kPolicyWaves;
    // the first tile's boards are requested ahead of the weights and arrive under their transfer
    uint4 A, B;                                                   // the board of the tile about to be computed
    {
        const int64_t b0 = tile0 * 64 + lane;
        const int64_t j = b0 < p.n ? b0 : p.n - 1;                // a lane past the end holds the last real board
        A = p.plane_a[j];
        B = p.plane_b[j];
    }
    load_image<64 * kPolicyWaves>(s_image, p.image);
#ifdef TPL_DIAG_CLOCK
    // diagnostic build only (tools/policy_clock.py): shader-clock and 100 MHz real-time stamps around the tile loop
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int64_t tile = tile0; tile < tiles; tile += tile_step) {
        const int64_t b = tile * 64 + lane;
        const bool valid = b < p.n;
        uint32_t fb[kNt][8];
        {
            Board s;
            unpack_board(A, B, s);
            uint32_t own[8];
            board_features(s, p.L, p.M, own);
            column_features(own, c, fb);
        }
        // the next tile's boards are requested now and used a whole tile of matrix work later (no per-lane branch:
        // the loads land in the registers the loop carries)
        if (tile + tile_step < tiles) {
            const int64_t bn = b + tile_step * 64;
            const int64_t jn = bn < p.n ? bn : p.n - 1;
            A = p.plane_a[jn];
            B = p.plane_b[jn];
        }
        f32x4 lg[kNt];
        policy_logits(at, g, fb, lg);
        const uint32_t action = own_action(lg, g, lane);
        if (p.logits) {
            // rows 4g + reg of board (t, c) live on lane (c, g), for every t
#pragma unroll
            for (int t = 0; t < kNt; ++t) {
                const int64_t bt = tile * 64 + t * 16 + c;
                if (bt < p.n) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (4 * g + k < kOut) p.logits[bt * kOut + 4 * g + k] = lg[t][k];
                }
            }
        }
        if (valid) p.action[b] = (uint8_t)action;
    }
#ifdef TPL_DIAG_CLOCK
    if (lane == 0 && p.diag) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = p.diag + 4 * ((size_t)blockIdx.x * kPolicyWaves + wave);
        d[0] = t1 - t0;
        d[1] = r1 - r0;
        d[2] = r0 - r_entry;                                      // entry -> weights in LDS
        d[3] = r_entry;                                           // when this wave started (100 MHz, free running)
    }
#endif
}

__global__ __launch_bounds__(kBlock) void explore_kernel(uint8_t* action, int64_t n, uint64_t seed, int64_t global_offset,
                                                        uint32_t step, uint32_t eps_q24) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) action[i] = (uint8_t)explore(action[i], seed, (uint64_t)(global_offset + i), step, eps_q24);
}

// T iterations of (policy -> epsilon-greedy -> step) in ONE launch: weights stay in LDS, boards stay in
// registers, nothing but the trajectory leaves the chip.  Exactly T x (tpl_policy_act, tpl_explore_actions,
// tpl_step).  A wave holds 64 boards, one per lane (a tile spans two of the 32-board clock groups).
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
    const int c = lane & 15, g = lane >> 4;
    const LdsBases at = lds_bases(lds, lane);
    const int64_t tiles = (p.n + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < tiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t b = tile * 64 + lane;
        const bool valid = b < p.n;
        // Through the matrix part a board lives as its two packed state words (8 registers instead of 16), its pool entry is
        // worked out again at every step (one hash) and finished episodes go straight to the block's counters in LDS: the
        // four tiles' activations need the registers.
        uint4 A, B;
        unsigned long long clock = 0;
        if (valid) {
            A = p.plane_a[b];
            B = p.plane_b[b];
            clock = p.clock[b >> kClockShift];
        } else {
            Board filler;                                         // frozen: never moves, never resets
#pragma unroll
            for (int k = 0; k < kCols; ++k) filler.c[k] = 0;
            filler.window = 0xFFFFFFFFu; filler.window_hi = 0xFu; filler.state = ST_LOST_LIMIT; filler.lines = 0; filler.moves = 0; filler.slot = 0;
            pack_board(filler, A, B);
        }
        for (uint32_t t = 0; t < q.T; ++t) {
            if (q.states_a && valid) {
                q.states_a[(size_t)t * p.n + b] = A;
                q.states_b[(size_t)t * p.n + b] = B;
            }
            f32x4 lg[kNt];
            {
                uint32_t own[8], fb[kNt][8];
                {
                    Board s;
                    unpack_board(A, B, s);
                    board_features(s, (int)p.L, (int)p.M, own);
                }
                column_features(own, c, fb);
                policy_logits(at, g, fb, lg);
            }
            uint32_t action = own_action(lg, g, lane);
            action = explore(action, q.explore_seed, (uint64_t)(p.global_offset + b), q.step0 + t, q.eps_q24);
            uint32_t rot, loc;
            split_small_action(action, rot, loc);
            float reward;
            Tally mine;
            Board s;
            unpack_board(A, B, s);
            uint32_t cfg = current_config(s, p, (uint32_t)b, clock + t);
            const bool done = advance_board<kAutoReset>(s, cfg, rot, loc, p, (uint32_t)b, clock + t, s_shape, reward, mine);
            pack_board(s, A, B);
            if (valid) {
                if (mine.episodes) {
                    atomicAdd(&s_stat[0], mine.episodes);
                    if (mine.lines) atomicAdd(&s_stat[1], mine.lines);
                    if (mine.wins) atomicAdd(&s_stat[2], mine.wins);
                    if (mine.topouts) atomicAdd(&s_stat[3], mine.topouts);
                }
                if (q.actions) q.actions[(size_t)t * p.n + b] = (uint8_t)action;
                if (q.rewards) q.rewards[(size_t)t * p.n + b] = reward;
                if (q.dones) q.dones[(size_t)t * p.n + b] = done ? 1 : 0;
            }
        }
        if (valid) {
            p.plane_a[b] = A;
            p.plane_b[b] = B;
            if ((b & (kClockGroup - 1)) == 0) p.clock[b >> kClockShift] = clock + q.T;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) {                                        // as flush_tally: one sharded 64-bit atomic per counter
        const uint32_t v = s_stat[threadIdx.x];
        if (v) atomicAdd(&p.stats[(size_t)(blockIdx.x % kStatShards) * kStatStride + threadIdx.x], (unsigned long long)v);
    }
}

}  // namespace p16
}  // namespace tpl

extern "C" int tpl_policy_act(tpl_env* e, const void* image, uint8_t* action, float* logits, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!image || !action) return fail_msg(TPL_ERR_ARG, "image/action is null");
    if (((uintptr_t)image & 15u) != 0) return fail_msg(TPL_ERR_ARG, "image must be 16-byte aligned");
    DeviceGuard guard(e->device);
    PolicyArgs p{};
    p.plane_a = e->plane_a; p.plane_b = e->plane_b; p.n = e->n; p.L = e->L; p.M = e->M;
    p.image = (const uint4*)image; p.action = action; p.logits = logits;
#ifdef TPL_DIAG_CLOCK
    p.diag = (unsigned long long*)logits;    // the diagnostic build writes its stamps where the logits would go
    p.logits = nullptr;
#endif
    // one resident workgroup per CU (the weights fill its LDS), eight waves of 64 boards, looping over board tiles
    const int64_t groups = ((e->n + 63) / 64 + kPolicyWaves - 1) / kPolicyWaves;
    hipLaunchKernelGGL(policy_kernel, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(64 * kPolicyWaves), 0, (hipStream_t)stream, p);
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
    if (int rc = check_can_advance(e)) return rc;
    DeviceGuard guard(e->device);
    ActorArgs q{};
    q.s = make_args(e);
    q.image = (const uint4*)image; q.T = (uint32_t)num_steps; q.step0 = step0;
    q.eps_q24 = (uint32_t)(epsilon * 16777216.0f); q.explore_seed = seed;
    q.actions = actions; q.rewards = rewards; q.dones = dones; q.states_a = (uint4*)states_a; q.states_b = (uint4*)states_b;
    const int64_t groups = ((e->n + 63) / 64 + 7) / 8;
    const dim3 grid((unsigned)(groups < 256 ? groups : 256)), block(512);
    if (e->auto_reset) hipLaunchKernelGGL(actor_rollout_kernel<true>, grid, block, 0, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(actor_rollout_kernel<false>, grid, block, 0, (hipStream_t)stream, q);
    TPL_HIP(hipGetLastError());
    count_steps(e, num_steps);
    return TPL_OK;
}
