// policy_f32.hip -- obs -> MLP -> action with FLOAT32 operands (the reference's own arithmetic width).
//
// Model(217, 14) of the reference (model/model.py:9-20, model/train.py:26) is a float32 nn.Linear stack.  The fused
// kernel of policy_mlp.hip rounds weights and hidden activations to bf16 (what a bf16 torch module does); this one
// keeps everything in float32 on the matrix cores: v_mfma_f32_16x16x4_f32 -- f32 in, f32 accumulate, bit for bit a
// k-ordered fmaf chain, 64 FLOP/clk/SIMD (157 TFLOP/s dense on the chip, 1/16 of the bf16 rate).
//
// Same transposed scheme as the bf16 kernel (boards on the MFMA's N dimension = the lane), and the hand-off between
// layers is even simpler here: the B operand of this instruction is ONE f32 per lane, B[k = lane >> 4][col = lane & 15],
// and a C/D tile leaves D[row = 4g + reg][col c] on lane (c, g) -- so register `reg` of output tile m IS the B operand
// of the next layer's k-step {16m + 4g + reg : g = 0..3}.  No conversion, no lane movement; the weights are pre-packed
// on the host in that k order.
//
// The f32 weights are 320 KB, twice the CU's LDS, so they stream through it: six chunks per pass (layer 1 in two
// halves, layers 2-4, the head), double-buffered -- chunk j+1 arrives by LDS-DMA while the eight waves of the
// workgroup multiply chunk j, one barrier per chunk.  Every wave needs every chunk for its 32 boards, so a pass costs
// the CU 320 KB of L2 reads per 256 boards: noise beside 2496 MFMAs per wave.
//
// Fragment maps: A: lane holds A[row c][k = g] of the k-step; B: B[k = g][col c]; C/D: D[row = 4g + reg][col c].
// Hidden layers: k-step q = 4m + reg takes k = 16m + 4g + reg on lane group g.  Layer 1 (free to choose, its B values are
// made from the board's bits): k-step q takes internal feature 4q + g.
#include "tpl_internal.h"
#include "tpl_policy.h"
#include "tpl_step.h"

#include <cstring>
#include <vector>

namespace tpl {
namespace pf32 {

using namespace tpl::p16;

constexpr int kKs1 = 56;                   // k-steps of 4 in layer 1: 224 = 217 padded
constexpr int kKsH = 32;                   // k-steps of a hidden layer
constexpr int kMt = 8;                     // 16-row output tiles of a 128-wide layer
constexpr int kStepBytes = 64 * 4;         // one A fragment: a float per lane

// chunks, in the order they are used (and laid out in the image)
constexpr int kChunks = 6;
constexpr int kChunkBytes[kChunks] = {4 * kKs1 * kStepBytes, 4 * kKs1 * kStepBytes, kMt * kKsH * kStepBytes,
                                      kMt * kKsH * kStepBytes, kMt * kKsH * kStepBytes, kKsH * kStepBytes};
constexpr int kChunkOff[kChunks] = {0, 57344, 114688, 180224, 245760, 311296};
constexpr int kOffB = 319488;
constexpr int kImageBytes = kOffB + (4 * kHidden + 16) * 4;          // 321,600
constexpr int kBufBytes = 65536;
static_assert(kChunkOff[5] + kChunkBytes[5] == kOffB && kChunkBytes[2] == kBufBytes, "chunk table");

// A fragments are stored four k-steps to a 16-byte piece per lane: [(tile, q4)][lane][q & 3]
static inline size_t frag_index(int tile, int ks4, int q4, int lane, int r) { return (((size_t)tile * ks4 + q4) * 64 + lane) * 4 + r; }

}  // namespace pf32
}  // namespace tpl

using namespace tpl;
using namespace tpl::pf32;

extern "C" size_t tpl_policy_image_bytes_f32(void) { return (size_t)kImageBytes; }

extern "C" int tpl_policy_pack_f32(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                   const float* b3, const float* w4, const float* b4, const float* w5, const float* b5,
                                   void* image) {
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4 || !w5 || !b5 || !image)
        return fail_msg(TPL_ERR_ARG, "tpl_policy_pack_f32: null pointer");
    std::vector<float> img((size_t)kImageBytes / 4, 0.0f);
    // layer 1: tiles 0-3 in chunk 0, tiles 4-7 in chunk 1; k-step q of lane group g = internal feature 4q + g
    for (int m = 0; m < kMt; ++m)
        for (int q = 0; q < kKs1; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int c = lane & 15, g = lane >> 4;
                const int k = std_feature(4 * q + g);
                const float v = k >= 0 ? w1[(size_t)(16 * m + c) * kObs + k] : 0.0f;
                img[kChunkOff[m >> 2] / 4 + frag_index(m & 3, kKs1 / 4, q >> 2, lane, q & 3)] = v;
            }
    // hidden layers and the head: k-step q = 4 * (input tile) + reg takes k = 16 * (input tile) + 4g + reg
    auto pack_hidden = [&](int off, const float* w, int rows, int tiles) {
        for (int m = 0; m < tiles; ++m)
            for (int q = 0; q < kKsH; ++q)
                for (int lane = 0; lane < 64; ++lane) {
                    const int c = lane & 15, g = lane >> 4;
                    const int k = 16 * (q >> 2) + 4 * g + (q & 3), row = 16 * m + c;
                    img[off / 4 + frag_index(m, kKsH / 4, q >> 2, lane, q & 3)] = row < rows ? w[(size_t)row * kHidden + k] : 0.0f;
                }
    };
    pack_hidden(kChunkOff[2], w2, kHidden, kMt);
    pack_hidden(kChunkOff[3], w3, kHidden, kMt);
    pack_hidden(kChunkOff[4], w4, kHidden, kMt);
    pack_hidden(kChunkOff[5], w5, kOut, 1);
    float* bias = img.data() + kOffB / 4;
    const float* bs[4] = {b1, b2, b3, b4};
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < kHidden; ++k) bias[l * kHidden + k] = bs[l][k];
    for (int k = 0; k < kOut; ++k) bias[4 * kHidden + k] = b5[k];
    std::memcpy(image, img.data(), (size_t)kImageBytes);
    return TPL_OK;
}

namespace tpl {
namespace pf32 {

// `bytes` of the image -> LDS by LDS-DMA, spread over the workgroup's waves (each instruction moves 64 consecutive
// 16-byte pieces to a wave-uniform LDS base).  Completion is the caller's next barrier.
template <int kThreads>
__device__ __forceinline__ void start_chunk(uint4* dst, const uint4* src, int bytes) {
    typedef __attribute__((address_space(1))) const void global_ptr;
    typedef __attribute__((address_space(3))) void lds_ptr;
    const int pieces = bytes / 16;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);      // uniform: the chunk offsets stay scalar
    // The lane's byte offset is made opaque HERE, at every call: the source addresses of a chunk are loop-invariant 64-bit
    // values per lane, and left alone the compiler computes all of them (eight per chunk, six chunks) ahead of the pass loop
    // and keeps them -- twenty-one registers' worth went to scratch in a kernel that has none to spare.
    uint32_t lane_bytes = ((uint32_t)threadIdx.x & 63u) * 16u;
    asm volatile("" : "+v"(lane_bytes));
    const uint8_t* base = (const uint8_t*)src;
    for (int chunk = wave; chunk * 64 < pieces; chunk += kThreads / 64) {
        if (chunk * 64 + (int)(lane_bytes >> 4) < pieces)
            __builtin_amdgcn_global_load_lds((global_ptr*)(base + (size_t)chunk * 1024 + lane_bytes), (lds_ptr*)(dst + chunk * 64), 16, 0, 0);
    }
}

// Where a lane reads a chunk's fragments from: the buffer's LDS address + 16 lane as a typed 32-bit LDS pointer, made opaque
// (which buffer a chunk sits in is known at compile time, and a constant address above 64 KB does not fit a ds_read's offset
// field: policy_split.hip).
typedef __attribute__((address_space(3))) const uint8_t lds_byte;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
__device__ __forceinline__ lds_byte* chunk_base(const uint4* buffer, int lane) {
    lds_byte* w = (lds_byte*)buffer + 16 * lane;
    asm volatile("" : "+v"(w));
    return w;
}
__device__ __forceinline__ float4 frag4(lds_byte* w, int piece) {           // piece = index of the 1-KB fragment row in the chunk
    const f32x4 v = *(lds_f32x4*)(w + piece * 1024);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// kTiles output tiles of one layer from the chunk at `w`: acc = bias; acc += W[tile rows][k-step] x xin[k-step].
// xin[t][q4] holds the four B values of k-steps 4 q4 .. 4 q4 + 3 for N tile t (for a hidden layer: the previous layer's
// output tile q4 as it left the matrix core).
template <int kTiles, int kKs4, bool kRelu>
__device__ __forceinline__ void dense(lds_byte* w, const float* bias, int lane, int g, const f32x4 (&xin)[2][kKs4],
                                      f32x4* xout0, f32x4* xout1) {
    // the A fragments run one step ahead of the multiplies that use them (2.5 % on the kernel, same box)
    float4 a_next = frag4(w, 0);
#pragma unroll
    for (int m = 0; m < kTiles; ++m) {
        const float4 b = *(const float4*)(bias + 16 * m + 4 * g);
        f32x4 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][0] = b.x; acc[t][1] = b.y; acc[t][2] = b.z; acc[t][3] = b.w; }
#pragma unroll
        for (int q4 = 0; q4 < kKs4; ++q4) {
            const float4 a = a_next;
            if (m * kKs4 + q4 + 1 < kTiles * kKs4)
                a_next = frag4(w, m * kKs4 + q4 + 1);     // one ds_read_b128
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], xin[t][q4][r], acc[t], 0, 0, 0);
            // (left alone the scheduler hoists dozens of fragment reads ahead of the multiplies and runs out of registers)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (kRelu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.0f);
            }
        }
        xout0[m] = acc[0];
        xout1[m] = acc[1];
    }
}

// kTiles output tiles of LAYER 1 from the chunk at `w`.  The B values are made from the boards' feature bits as they are
// needed (k-step q of lane group g = bit 4 (q & 7) + g of feature word q >> 3; the two counters -- features 214, 215 =
// k-step 53, groups 2 and 3 -- enter as numbers), so the k-steps run outermost and the tiles' accumulators stay live.
template <int kTiles>
__device__ __forceinline__ void dense_first(lds_byte* w, const float* bias, int lane, int g, const uint32_t (&fb)[2][8],
                                            f32x4* xout0, f32x4* xout1) {
    f32x4 acc[2][kTiles];
#pragma unroll
    for (int m = 0; m < kTiles; ++m) {
        const float4 b = *(const float4*)(bias + 16 * m + 4 * g);
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][m][0] = b.x; acc[t][m][1] = b.y; acc[t][m][2] = b.z; acc[t][m][3] = b.w; }
    }
    uint32_t u[2][7];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 7; ++k) u[t][k] = fb[t][k] >> g;
    float4 a_next = frag4(w, 0);          // fragments run one step ahead of their multiplies
#pragma unroll
    for (int q4 = 0; q4 < kKs1 / 4; ++q4) {
        float x[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 4 * q4 + r;
                float v = (float)((u[t][q >> 3] >> (4 * (q & 7))) & 1u);
                if (q == 53) {
                    const float lrem = __uint_as_float(fb[t][7] << 16), mrem = __uint_as_float(fb[t][7] & 0xFFFF0000u);
                    v = g == 2 ? lrem : g == 3 ? mrem : v;
                }
                x[t][r] = v;
            }
#pragma unroll
        for (int m = 0; m < kTiles; ++m) {
            const float4 a = a_next;
            {
                const int mn = m + 1 < kTiles ? m + 1 : 0, qn = m + 1 < kTiles ? q4 : q4 + 1;
                if (qn < kKs1 / 4) a_next = frag4(w, mn * (kKs1 / 4) + qn);
            }
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], x[t][r], acc[t][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int m = 0; m < kTiles; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[0][m][r] = fmaxf(acc[0][m][r], 0.0f); acc[1][m][r] = fmaxf(acc[1][m][r], 0.0f); }
        xout0[m] = acc[0][m];
        xout1[m] = acc[1][m];
    }
}

// Between the two halves of layer 1 the feature words are made opaque: both halves turn the same bits into the same 112
// float B values, and left alone the compiler computes them once and keeps them for the second half -- a hundred registers
// held through a quarter of the pass, in a kernel whose two activation sets already take 128.  Recomputing costs two
// vector instructions per value.
__device__ __forceinline__ void forget_derived(uint32_t (&fb)[2][8]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(fb[t][k]));
}

struct PolicyF32Args {
    const uint4* plane_a;
    const uint4* plane_b;
    int64_t n;
    int32_t L, M;
    const uint4* image;
    uint8_t* action;
    float* logits;
};

constexpr int kWaves = 8;

__global__ __launch_bounds__(64 * kWaves) void policy_f32_kernel(const PolicyF32Args p) {
    __shared__ uint4 s_buf[2][kBufBytes / 16];
    __shared__ float s_bias[4 * kHidden + 16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // (a scalar)
    const int c = lane & 15, g = lane >> 4;
    const int64_t tiles = (p.n + 31) / 32;
    const int64_t tile_step = (int64_t)gridDim.x * kWaves;
    // every wave of the workgroup makes the same number of passes (the chunk barriers need all of them)
    const int64_t passes = (tiles - (int64_t)blockIdx.x * kWaves + tile_step - 1) / tile_step;
    const uint8_t* image = (const uint8_t*)p.image;

    for (int k = threadIdx.x; k < 4 * kHidden + 16; k += 64 * kWaves) s_bias[k] = ((const float*)(image + kOffB))[k];
    start_chunk<64 * kWaves>(s_buf[0], (const uint4*)(image + kChunkOff[0]), kChunkBytes[0]);
    __syncthreads();                                             // (its fence waits for the transfers)

    int buf = 0;
    for (int64_t pass = 0; pass < passes; ++pass) {
        const int64_t tile = (int64_t)blockIdx.x * kWaves + wave + pass * tile_step;
        const int64_t b = tile * 32 + (g >> 1) * 16 + c;
        const bool valid = b < p.n;
        const int64_t j = valid ? b : p.n - 1;                    // a lane past the end holds the last real board
        Board s;
        unpack_board(p.plane_a[j], p.plane_b[j], s);
        uint32_t own[8], fb[2][8];
        board_features(s, p.L, p.M, own);
        both_features(own, g, fb);

        f32x4 xa[2][kMt], xb[2][kMt];
        // chunk 0: layer 1, tiles 0-3
        start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[1]), kChunkBytes[1]);
        dense_first<4>(chunk_base(s_buf[buf], lane), s_bias, lane, g, fb, &xa[0][0], &xa[1][0]);
        __syncthreads(); buf ^= 1;
        forget_derived(fb);
        // chunk 1: layer 1, tiles 4-7
        start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[2]), kChunkBytes[2]);
        dense_first<4>(chunk_base(s_buf[buf], lane), s_bias + 64, lane, g, fb, &xa[0][4], &xa[1][4]);
        __syncthreads(); buf ^= 1;
        // chunks 2-4: the hidden layers
        start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[3]), kChunkBytes[3]);
        dense<kMt, kMt, true>(chunk_base(s_buf[buf], lane), s_bias + 1 * kHidden, lane, g, xa, &xb[0][0], &xb[1][0]);
        __syncthreads(); buf ^= 1;
        start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[4]), kChunkBytes[4]);
        dense<kMt, kMt, true>(chunk_base(s_buf[buf], lane), s_bias + 2 * kHidden, lane, g, xb, &xa[0][0], &xa[1][0]);
        __syncthreads(); buf ^= 1;
        start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[5]), kChunkBytes[5]);
        dense<kMt, kMt, true>(chunk_base(s_buf[buf], lane), s_bias + 3 * kHidden, lane, g, xa, &xb[0][0], &xb[1][0]);
        __syncthreads(); buf ^= 1;
        // chunk 5: the head; the next pass's first chunk arrives under it
        if (pass + 1 < passes) start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[0]), kChunkBytes[0]);
        f32x4 lg[2];
        dense<1, kMt, false>(chunk_base(s_buf[buf], lane), s_bias + 4 * kHidden, lane, g, xb, &lg[0], &lg[1]);
        __syncthreads(); buf ^= 1;

        const uint32_t act0 = pick_action(lg[0], g, lane), act1 = pick_action(lg[1], g, lane);
        const uint32_t action = (g >> 1) ? act1 : act0;
        if (p.logits) {
            // rows 4g + reg of board (t, c) live on lane (c, g) for both t
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t bt = tile * 32 + t * 16 + c;
                if (bt < p.n) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (4 * g + k < kOut) p.logits[bt * kOut + 4 * g + k] = lg[t][k];
                }
            }
        }
        if (valid && (g & 1) == 0) p.action[b] = (uint8_t)action;
    }
}

// T iterations of (float32 policy -> epsilon-greedy -> step) in ONE launch: the reference's arithmetic width
// (model/model.py:9-20) as a multi-step loop.  The boards of a wave's tile stay in registers for the T steps; the 320 KB of
// weights stream through the two LDS buffers once per step (L2 traffic: 82 MB per step over the chip, nothing beside
// 2496 MFMAs per wave-step), the eight waves of the workgroup in lockstep through the six chunks as in policy_f32_kernel.
// Exactly T x (tpl_policy_act_f32, tpl_explore_actions, tpl_step).  Between the policy and the move a board lives as its two
// packed state words (8 registers instead of 16): the matrix part needs every register it can get.
template <bool kAutoReset>
__global__ __launch_bounds__(64 * kWaves) void actor_rollout_f32_kernel(const ActorArgs q) {
    const StepArgs& p = q.s;
    __shared__ uint4 s_buf[2][kBufBytes / 16];
    __shared__ float s_bias[4 * kHidden + 16];
    __shared__ ShapeWord s_shape[32];
    __shared__ uint32_t s_stat[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // (a scalar)
    const int c = lane & 15, g = lane >> 4;
    const bool writer = (g & 1) == 0;
    const int64_t tiles = (p.n + 31) / 32;
    const int64_t tile_step = (int64_t)gridDim.x * kWaves;
    // every wave of the workgroup makes the same number of passes (the chunk barriers need all of them)
    const int64_t passes = (tiles - (int64_t)blockIdx.x * kWaves + tile_step - 1) / tile_step;
    const uint8_t* image = (const uint8_t*)q.image;

    if (threadIdx.x < 32) s_shape[threadIdx.x] = kShapeTable[threadIdx.x];
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    for (int k = threadIdx.x; k < 4 * kHidden + 16; k += 64 * kWaves) s_bias[k] = ((const float*)(image + kOffB))[k];
    start_chunk<64 * kWaves>(s_buf[0], (const uint4*)(image + kChunkOff[0]), kChunkBytes[0]);
    __syncthreads();                                             // (its fence waits for the transfers)

    int buf = 0;
    for (int64_t pass = 0; pass < passes; ++pass) {
        const int64_t tile = (int64_t)blockIdx.x * kWaves + wave + pass * tile_step;
        const int64_t b = tile * 32 + (g >> 1) * 16 + c;
        const bool valid = b < p.n;
        uint4 A, B;                                               // the board between the steps' moves, packed
        if (valid) {
            A = p.plane_a[b];
            B = p.plane_b[b];
        } else {
            Board filler;                                         // frozen: never moves, never resets
#pragma unroll
            for (int k = 0; k < kCols; ++k) filler.c[k] = 0;
            filler.window = 0xFFFFFFFFu; filler.window_hi = 0xFu; filler.state = ST_LOST_LIMIT; filler.lines = 0; filler.moves = 0; filler.slot = 0;
            pack_board(filler, A, B);
        }
        // the tile's 32 boards are one clock group (wave-uniform: kept in scalar registers)
        unsigned long long clock = tile < tiles ? p.clock[tile] : 0ULL;
        clock = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(clock >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)clock);
        for (uint32_t t = 0; t < q.T; ++t) {
            if (q.states_a && valid && writer) {
                q.states_a[(size_t)t * p.n + b] = A;
                q.states_b[(size_t)t * p.n + b] = B;
            }
            f32x4 lg[2];
            {
                uint32_t own[8], fb[2][8];
                {
                    Board s;
                    unpack_board(A, B, s);
                    board_features(s, (int)p.L, (int)p.M, own);
                }
                both_features(own, g, fb);
                f32x4 xa[2][kMt], xb[2][kMt];
                start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[1]), kChunkBytes[1]);
                dense_first<4>(chunk_base(s_buf[buf], lane), s_bias, lane, g, fb, &xa[0][0], &xa[1][0]);
                __syncthreads(); buf ^= 1;
                forget_derived(fb);
                start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[2]), kChunkBytes[2]);
                dense_first<4>(chunk_base(s_buf[buf], lane), s_bias + 64, lane, g, fb, &xa[0][4], &xa[1][4]);
                __syncthreads(); buf ^= 1;
                start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[3]), kChunkBytes[3]);
                dense<kMt, kMt, true>(chunk_base(s_buf[buf], lane), s_bias + 1 * kHidden, lane, g, xa, &xb[0][0], &xb[1][0]);
                __syncthreads(); buf ^= 1;
                start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[4]), kChunkBytes[4]);
                dense<kMt, kMt, true>(chunk_base(s_buf[buf], lane), s_bias + 2 * kHidden, lane, g, xb, &xa[0][0], &xa[1][0]);
                __syncthreads(); buf ^= 1;
                start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[5]), kChunkBytes[5]);
                dense<kMt, kMt, true>(chunk_base(s_buf[buf], lane), s_bias + 3 * kHidden, lane, g, xa, &xb[0][0], &xb[1][0]);
                __syncthreads(); buf ^= 1;
                // the head; the next step's (or pass's) first chunk arrives under it
                if (t + 1 < q.T || pass + 1 < passes)
                    start_chunk<64 * kWaves>(s_buf[buf ^ 1], (const uint4*)(image + kChunkOff[0]), kChunkBytes[0]);
                dense<1, kMt, false>(chunk_base(s_buf[buf], lane), s_bias + 4 * kHidden, lane, g, xb, &lg[0], &lg[1]);
                __syncthreads(); buf ^= 1;
            }
            const uint32_t act0 = pick_action(lg[0], g, lane), act1 = pick_action(lg[1], g, lane);
            uint32_t action = (g >> 1) ? act1 : act0;
            action = explore(action, q.explore_seed, (uint64_t)(p.global_offset + b), q.step0 + t, q.eps_q24);
            uint32_t rot, loc;
            split_small_action(action, rot, loc);
            float reward;
            Tally mine;
            Board s;
            unpack_board(A, B, s);
            // the pool entry of the board's episode is worked out again at every step (one hash) rather than carried through
            // the matrix part
            uint32_t cfg = current_config(s, p, (uint32_t)b, clock + t);
            const bool done = advance_board<kAutoReset>(s, cfg, rot, loc, p, (uint32_t)b, clock + t, s_shape, reward, mine);
            pack_board(s, A, B);
            if (valid && writer) {
                // finished episodes go straight to the block's counters in LDS (a board finishes every thirtieth step or
                // so): four registers fewer to carry through the matrix part than a per-lane tally
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
        if (valid && writer) {
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

}  // namespace pf32
}  // namespace tpl

extern "C" int tpl_policy_act_f32(tpl_env* e, const void* image, uint8_t* action, float* logits, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!image || !action) return fail_msg(TPL_ERR_ARG, "image/action is null");
    if (((uintptr_t)image & 15u) != 0) return fail_msg(TPL_ERR_ARG, "image must be 16-byte aligned");
    DeviceGuard guard(e->device);
    PolicyF32Args p{};
    p.plane_a = e->plane_a; p.plane_b = e->plane_b; p.n = e->n; p.L = e->L; p.M = e->M;
    p.image = (const uint4*)image; p.action = action; p.logits = logits;
    // one resident workgroup per CU, eight waves of 32 boards, looping over board tiles
    const int64_t groups = ((e->n + 31) / 32 + kWaves - 1) / kWaves;
    hipLaunchKernelGGL(policy_f32_kernel, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(64 * kWaves), 0, (hipStream_t)stream, p);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

extern "C" int tpl_actor_rollout_f32(tpl_env* e, const void* image, int32_t num_steps, float epsilon, uint64_t seed, uint32_t step0,
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
    const int64_t groups = ((e->n + 31) / 32 + kWaves - 1) / kWaves;
    const dim3 grid((unsigned)(groups < 256 ? groups : 256)), block(64 * kWaves);
    if (e->auto_reset) hipLaunchKernelGGL(actor_rollout_f32_kernel<true>, grid, block, 0, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(actor_rollout_f32_kernel<false>, grid, block, 0, (hipStream_t)stream, q);
    TPL_HIP(hipGetLastError());
    count_steps(e, num_steps);
    return TPL_OK;
}
