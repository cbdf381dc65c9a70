// policy_split.hip -- obs -> MLP -> action at FLOAT32 ACCURACY on the bf16 matrix pipe.
//
// Model(217, 14) of the reference (model/model.py:9-20) is a float32 nn.Linear stack.  policy_f32.hip computes it with float32
// MFMAs (v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD, one sixteenth of the bf16 rate); gfx950 has no xf32 instruction in
// between.  This kernel takes the other road: every float32 number is the sum of three bf16 numbers,
//     x = x_h + x_l + x_ll,   x_h = bf16(x),  x_l = bf16(x - x_h),  x_ll = bf16(x - x_h - x_l)
// (8 + 8 + 8 = 24 significand bits: exact up to the last place), a product of two bf16 numbers is exact in float32, and the
// MFMA accumulates in float32 -- so
//     w * x  =  w_h x_h + w_h x_l + w_l x_h + w_l x_l + w_h x_ll + w_ll x_h   (+ terms below 2^-24 of the product)
// is six v_mfma_f32_16x16x32_bf16 where the float32 path needs eight v_mfma_f32_16x16x4_f32 of twice the duration each:
// 2.7 x less matrix time.  Layer 1 needs only three (its inputs -- cell bits and two small counters -- are exact in bf16).
// The weights are split on the host; the activations are split as they are used.  Not bit-identical to a float32 FMA chain
// (neither is torch's float32 GEMM to ours): the test is the one the float32 kernel has to pass -- within 2e-5 (1 + max|ref|)
// of a float64 evaluation, where bf16 operands alone are off by three orders of magnitude more.
//
// Geometry: as the other policy kernels -- boards on the MFMA's N dimension, 32 boards (two N tiles) per wave, features
// along M/K in the bf16 kernel's permuted order, so that two consecutive output tiles of a layer, as they leave the matrix
// core (float32), are the eight k-values of one B fragment of the next.  Here the K-STEPS run outermost in every layer: the B
// fragment of a k-step is split into its three bf16 pieces ONCE and meets all three weight planes of all eight output tiles
// there and then (their accumulators stay live, 64 registers; the float32 activations of the previous layer, 64 registers, are
// the only other large thing).  The three weight planes are 474 KB and stream through two 64-KB LDS buffers in ten chunks per
// pass, as in policy_f32.hip: layer 1 plane by plane, a hidden layer as two chunks of two k-steps each, ALL THREE planes of
// those k-steps in the chunk (through round 4 a hidden layer's chunks went plane-wise -- the high plane, then the two low
// ones -- and every activation fragment was split twice, once under each: 2.55 vector instructions per MFMA, a third of
// them that second split).
#include "tpl_internal.h"
#include "tpl_policy.h"
#include "tpl_step.h"

#include <cstring>
#include <vector>

namespace tpl {
namespace psp {

using namespace tpl::p16;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int kKs1 = 7, kKsH = 4, kMt = 8;
constexpr int kPlane1 = kMt * kKs1 * 1024;     // 57,344: one bf16 plane of layer 1's A fragments
constexpr int kPlaneH = kMt * kKsH * 1024;     // 32,768: of a hidden layer
constexpr int kPlane5 = kKsH * 1024;           //  4,096: of the head (one 16-row tile)
// chunks in the order of use: layer 1 plane by plane; a hidden layer as k-steps {0, 1} then {2, 3}, each with its three planes
// (fragment (k-step s, plane i, tile m) of a hidden layer at ((s & 1) * 3 + i) * kMt + m in chunk s >> 1); the head
constexpr int kChunks = 10;
constexpr int kHalfH = 3 * kPlaneH / 2;        // 49,152: two k-steps x three planes x eight tiles
constexpr int kChunkBytes[kChunks] = {kPlane1, kPlane1, kPlane1, kHalfH, kHalfH, kHalfH, kHalfH, kHalfH, kHalfH, 3 * kPlane5};
constexpr int kChunkOff[kChunks] = {0, 57344, 114688, 172032, 221184, 270336, 319488, 368640, 417792, 466944};
constexpr int kOffB = 479232;
constexpr int kImageBytes = kOffB + (4 * kHidden + 16) * 4;
constexpr int kBufBytes = 65536;
static_assert(kChunkOff[9] + kChunkBytes[9] == kOffB, "chunk table");

// the three bf16 pieces of a float (host): RNE at every stage
static inline void split3(float v, uint16_t (&piece)[3]) {
    float r = v;
    for (int i = 0; i < 3; ++i) {
        piece[i] = bf16_rne(r);
        uint32_t u = (uint32_t)piece[i] << 16;
        float back;
        std::memcpy(&back, &u, 4);
        r -= back;
    }
}

}  // namespace psp
}  // namespace tpl

using namespace tpl;
using namespace tpl::psp;

extern "C" size_t tpl_policy_image_bytes_split(void) { return (size_t)kImageBytes; }

extern "C" int tpl_policy_pack_split(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                     const float* b3, const float* w4, const float* b4, const float* w5, const float* b5,
                                     void* image) {
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4 || !w5 || !b5 || !image)
        return fail_msg(TPL_ERR_ARG, "tpl_policy_pack_split: null pointer");
    std::vector<uint8_t> img((size_t)kImageBytes, 0);
    // where(i, m, s) = byte offset of the A fragment (piece i, output tile m, k-step s) of this layer
    auto pack_layer = [&](auto where, const float* w, int rows, int in, int mt, int ks, bool first) {
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
                        uint16_t piece[3];
                        split3(v, piece);
                        for (int i = 0; i < 3; ++i)
                            ((uint16_t*)(img.data() + where(i, m, s)))[lane * 8 + j] = piece[i];
                    }
    };
    pack_layer([](int i, int m, int s) { return kChunkOff[i] + (m * kKs1 + s) * 1024; }, w1, kHidden, kObs, kMt, kKs1, true);
    const float* wh[3] = {w2, w3, w4};
    for (int l = 0; l < 3; ++l)
        pack_layer([l](int i, int m, int s) { return kChunkOff[3 + 2 * l + (s >> 1)] + (((s & 1) * 3 + i) * kMt + m) * 1024; },
                   wh[l], kHidden, kHidden, kMt, kKsH, false);
    pack_layer([](int i, int m, int s) { (void)m; return kChunkOff[9] + i * kPlane5 + s * 1024; }, w5, kOut, kHidden, 1, kKsH, false);
    float* bias = (float*)(img.data() + kOffB);
    const float* bs[4] = {b1, b2, b3, b4};
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < kHidden; ++k) bias[l * kHidden + k] = bs[l][k];
    for (int k = 0; k < kOut; ++k) bias[4 * kHidden + k] = b5[k];
    std::memcpy(image, img.data(), (size_t)kImageBytes);
    return TPL_OK;
}

namespace tpl {
namespace psp {

constexpr int kWaves = 8;

// `bytes` of the image -> LDS by LDS-DMA, spread over the workgroup's waves; completion is the caller's next barrier.  (The
// lane's byte offset is made opaque at every call, so that the 64-bit source addresses are not all computed and kept ahead of
// the pass loop: policy_f32.hip.)
template <int kThreads>
__device__ __forceinline__ void start_chunk(uint4* dst, const uint8_t* src, int bytes) {
    typedef __attribute__((address_space(1))) const void global_ptr;
    typedef __attribute__((address_space(3))) void lds_ptr;
    const int pieces = bytes / 16;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);      // uniform: the chunk offsets stay scalar
    uint32_t lane_bytes = ((uint32_t)threadIdx.x & 63u) * 16u;
    asm volatile("" : "+v"(lane_bytes));
#ifdef TPL_X_SPLIT_NO_STREAM
    if (src != nullptr) return;        /* TIMING EXPERIMENT: no transfers at all (results are garbage) */
#endif
    for (int chunk = wave; chunk * 64 < pieces; chunk += kThreads / 64) {
        if (chunk * 64 + (int)(lane_bytes >> 4) < pieces)
            __builtin_amdgcn_global_load_lds((global_ptr*)(src + (size_t)chunk * 1024 + lane_bytes), (lds_ptr*)(dst + chunk * 64), 16, 0, 0);
    }
}

// Where a lane reads a chunk's fragments from: the buffer's LDS address + 16 lane, as a typed 32-bit LDS pointer made OPAQUE.
// Which of the two buffers a chunk sits in is known at compile time (ten chunks a pass, the buffers alternate), so left
// alone every fragment address is a constant + 16 lane -- and for the second buffer a constant above 64 KB, which does not fit a
// ds_read's 16-bit offset: a register per fragment, dozens of them, in scratch.  From an opaque base every fragment of a
// chunk is base + (an offset below 64 KB) in the instruction.
typedef __attribute__((address_space(3))) const uint8_t lds_byte;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
__device__ __forceinline__ lds_byte* chunk_base(const uint4* buffer, int lane) {
    lds_byte* w = (lds_byte*)buffer + 16 * lane;
    asm volatile("" : "+v"(w));
    return w;
}
__device__ __forceinline__ bf16x8 frag(lds_byte* w, int plane_off, int q) { return *(lds_bf16x8*)(w + plane_off + q * 1024); }
__device__ __forceinline__ f32x4 mfma(bf16x8 a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// float32 pair -> the packed bf16 pair nearest to it, and what is left of the pair after taking it away (exact): the two
// subtractions as ONE packed instruction (v_pk_add_f32 on a register pair)
__device__ __forceinline__ uint32_t take_bf16(f32x2& v) {
    const uint32_t packed = pack_bf16(v[0], v[1]);
    const f32x2 back = {__uint_as_float(packed << 16), __uint_as_float(packed & 0xFFFF0000u)};
    v -= back;
    return packed;
}

// The B fragment of k-step s of a hidden layer -- the eight float32 values (tile 2s, registers 0..3; tile 2s+1, registers
// 0..3) a lane holds of board (t, c) -- as its three bf16 pieces.
__device__ __forceinline__ void split_fragment(const f32x4& lo_tile, const f32x4& hi_tile, uint4& xh, uint4& xl, uint4& xll) {
    f32x2 v[4] = {{lo_tile[0], lo_tile[1]}, {lo_tile[2], lo_tile[3]}, {hi_tile[0], hi_tile[1]}, {hi_tile[2], hi_tile[3]}};
    uint32_t* h = (uint32_t*)&xh; uint32_t* l = (uint32_t*)&xl; uint32_t* ll = (uint32_t*)&xll;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = take_bf16(v[i]);
        l[i] = take_bf16(v[i]);
        ll[i] = pack_bf16(v[i][0], v[i][1]);
    }
}
__device__ __forceinline__ uint4 high_fragment(const f32x4& lo_tile, const f32x4& hi_tile) {
    return make_uint4(pack_bf16(lo_tile[0], lo_tile[1]), pack_bf16(lo_tile[2], lo_tile[3]), pack_bf16(hi_tile[0], hi_tile[1]),
                      pack_bf16(hi_tile[2], hi_tile[3]));
}

// layer-1 B fragment of k-step s for board features f (exact in bf16: policy_mlp.hip)
__device__ __forceinline__ uint4 first_fragment(const uint32_t (&f)[8], int s, int g) {
    const uint32_t u = f[s] >> (4 * g);
    uint4 q;
    q.x = (u << 14) & 0x40004000u;
    q.y = (u << 13) & 0x40004000u;
    q.z = (u << 12) & 0x40004000u;
    q.w = (u << 11) & 0x40004000u;
    if (s == 6 && g == 1) {                      // bits 22, 23 of word 6 = features 214, 215: L_rem, M_rem
        q.z |= f[7] << 16;
        q.w |= f[7] & 0xFFFF0000u;
    }
    return q;
}

struct PolicySplitArgs {
    const uint4* plane_a;
    const uint4* plane_b;
    int64_t n;
    int32_t L, M;
    const uint8_t* image;
    uint8_t* action;
    float* logits;
};

__device__ __forceinline__ void set_bias(f32x4 (&acc)[kMt][2], const float* bias, int g) {
#pragma unroll
    for (int m = 0; m < kMt; ++m) {
        const float4 b = *(const float4*)(bias + 16 * m + 4 * g);
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[m][t][0] = b.x; acc[m][t][1] = b.y; acc[m][t][2] = b.z; acc[m][t][3] = b.w; }
    }
}
__device__ __forceinline__ void relu(f32x4 (&acc)[kMt][2]) {
#pragma unroll
    for (int m = 0; m < kMt; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][t][r] = fmaxf(acc[m][t][r], 0.0f);
}

// The five layers for the wave's two N tiles of boards whose feature words are fb.  On entry chunk 0 is in s_buf[buf] (every
// wave of the workgroup calls this together: the chunk barriers are inside); on exit `buf` names the buffer chunk 0 is arriving
// in -- if `more`, i.e. if there is another call to come.
__device__ __forceinline__ void split_logits(uint4 (&s_buf)[2][kBufBytes / 16], const float* s_bias, const uint8_t* image, int& buf,
                                             bool more, int lane, int g, const uint32_t (&fb)[2][8], f32x4 (&lg)[2]) {
#define TPL_NEXT_CHUNK(next) start_chunk<64 * kWaves>(s_buf[buf ^ 1], image + kChunkOff[next], kChunkBytes[next])
#define TPL_CHUNK_DONE() do { __syncthreads(); buf ^= 1; } while (0)
        f32x4 acc[kMt][2], x[kMt][2];
        // ---- layer 1: three planes, the inputs exact
        set_bias(acc, s_bias, g);
#pragma unroll
        for (int plane = 0; plane < 3; ++plane) {
            TPL_NEXT_CHUNK(plane + 1);
            lds_byte* w = chunk_base(s_buf[buf], lane);
#pragma unroll
            for (int s = 0; s < kKs1; ++s) {
                const uint4 x0 = first_fragment(fb[0], s, g), x1 = first_fragment(fb[1], s, g);
#pragma unroll
                for (int m = 0; m < kMt; ++m) {
                    const bf16x8 a = frag(w, 0, m * kKs1 + s);
                    acc[m][0] = mfma(a, x0, acc[m][0]);
                    acc[m][1] = mfma(a, x1, acc[m][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TPL_CHUNK_DONE();
        }
        // ---- hidden layers
#pragma unroll
        for (int layer = 0; layer < 3; ++layer) {
            relu(acc);
#pragma unroll
            for (int m = 0; m < kMt; ++m) { x[m][0] = acc[m][0]; x[m][1] = acc[m][1]; }
            set_bias(acc, s_bias + (layer + 1) * kHidden, g);
            // two chunks of two k-steps each; a k-step's fragment is split once and multiplied with all three planes
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                TPL_NEXT_CHUNK(4 + 2 * layer + half);
                lds_byte* w = chunk_base(s_buf[buf], lane);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int s = 2 * half + sl;
                    uint4 xh[2], xl[2], xll[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) split_fragment(x[2 * s][t], x[2 * s + 1][t], xh[t], xl[t], xll[t]);
                    // two output tiles at a time, term by term: four independent accumulators between two multiplies into
                    // the same one (several in a row into one accumulator wait for each other); smallest terms first
#pragma unroll
                    for (int m = 0; m < kMt; m += 2) {
                        bf16x8 ah[2], al[2], all[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            ah[i] = frag(w, 0, (sl * 3 + 0) * kMt + m + i);
                            al[i] = frag(w, 0, (sl * 3 + 1) * kMt + m + i);
                            all[i] = frag(w, 0, (sl * 3 + 2) * kMt + m + i);
                        }
#define TPL_TERM(A, X)                                                                          \
                        _Pragma("unroll") for (int i = 0; i < 2; ++i)                          \
                            _Pragma("unroll") for (int t = 0; t < 2; ++t) acc[m + i][t] = mfma(A[i], X[t], acc[m + i][t]);
                        TPL_TERM(all, xh)
                        TPL_TERM(ah, xll)
                        TPL_TERM(al, xl)
                        TPL_TERM(al, xh)
                        TPL_TERM(ah, xl)
                        TPL_TERM(ah, xh)
#undef TPL_TERM
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                TPL_CHUNK_DONE();
            }
        }
        // ---- the head: one output tile, all three planes in one chunk; the next pass's first chunk arrives under it
        relu(acc);
        {
            if (more) TPL_NEXT_CHUNK(0);
            lds_byte* w = chunk_base(s_buf[buf], lane);
            const float4 bb = *(const float4*)(s_bias + 4 * kHidden + 4 * g);
#pragma unroll
            for (int t = 0; t < 2; ++t) { lg[t][0] = bb.x; lg[t][1] = bb.y; lg[t][2] = bb.z; lg[t][3] = bb.w; }
#pragma unroll
            for (int s = 0; s < kKsH; ++s) {
                const bf16x8 ah = frag(w, 0, s), al = frag(w, kPlane5, s), all = frag(w, 2 * kPlane5, s);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    uint4 xh, xl, xll;
                    split_fragment(acc[2 * s][t], acc[2 * s + 1][t], xh, xl, xll);
                    lg[t] = mfma(all, xh, lg[t]);
                    lg[t] = mfma(ah, xll, lg[t]);
                    lg[t] = mfma(al, xl, lg[t]);
                    lg[t] = mfma(al, xh, lg[t]);
                    lg[t] = mfma(ah, xl, lg[t]);
                    lg[t] = mfma(ah, xh, lg[t]);
                }
            }
            TPL_CHUNK_DONE();
        }
#undef TPL_NEXT_CHUNK
#undef TPL_CHUNK_DONE
}

__global__ __launch_bounds__(64 * kWaves) void policy_split_kernel(const PolicySplitArgs p) {
    __shared__ uint4 s_buf[2][kBufBytes / 16];
    __shared__ float s_bias[4 * kHidden + 16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // (a scalar)
    const int c = lane & 15, g = lane >> 4;
    const int64_t tiles = (p.n + 31) / 32;
    const int64_t tile_step = (int64_t)gridDim.x * kWaves;
    // every wave of the workgroup makes the same number of passes (the chunk barriers need all of them)
    const int64_t passes = (tiles - (int64_t)blockIdx.x * kWaves + tile_step - 1) / tile_step;
    const uint8_t* image = p.image;

    for (int k = threadIdx.x; k < 4 * kHidden + 16; k += 64 * kWaves) s_bias[k] = ((const float*)(image + kOffB))[k];
    start_chunk<64 * kWaves>(s_buf[0], image + kChunkOff[0], kChunkBytes[0]);
    __syncthreads();                                             // (its fence waits for the transfers)

    int buf = 0;
    for (int64_t pass = 0; pass < passes; ++pass) {
        const int64_t tile = (int64_t)blockIdx.x * kWaves + wave + pass * tile_step;
        const int64_t b = tile * 32 + (g >> 1) * 16 + c;
        const bool valid = b < p.n;
        const int64_t j = valid ? b : p.n - 1;                    // a lane past the end holds the last real board
        uint32_t fb[2][8];
        {
            Board s;
            unpack_board(p.plane_a[j], p.plane_b[j], s);
            uint32_t own[8];
            board_features(s, p.L, p.M, own);
            both_features(own, g, fb);
        }
        f32x4 lg[2];
        split_logits(s_buf, s_bias, image, buf, pass + 1 < passes, lane, g, fb, lg);
        const uint32_t act0 = pick_action(lg[0], g, lane), act1 = pick_action(lg[1], g, lane);
        const uint32_t action = (g >> 1) ? act1 : act0;
        if (p.logits) {
            int row0 = 4 * g;                                     // opaque here: else the lane's part of the address is worked out
            asm volatile("" : "+v"(row0));                        // before the pass loop and carried through it (in scratch)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t bt = tile * 32 + t * 16 + c;
                if (bt < p.n) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (row0 + k < kOut) p.logits[bt * kOut + row0 + k] = lg[t][k];
                }
            }
        }
        if (valid && (g & 1) == 0) p.action[b] = (uint8_t)action;
    }
}

// T iterations of (split policy -> epsilon-greedy -> step) in ONE launch: float32-grade decisions as a multi-step loop at
// twice the float32 megakernel's rate.  Exactly T x (tpl_policy_act_split, tpl_explore_actions, tpl_step); structure as
// actor_rollout_f32_kernel (policy_f32.hip): boards packed between the moves, the pool entry worked out again at every
// step, episodes tallied in LDS, the weight planes streamed through LDS once per step.
template <bool kAutoReset>
__global__ __launch_bounds__(64 * kWaves) void actor_rollout_split_kernel(const ActorArgs q) {
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
    const int64_t passes = (tiles - (int64_t)blockIdx.x * kWaves + tile_step - 1) / tile_step;
    const uint8_t* image = (const uint8_t*)q.image;

    if (threadIdx.x < 32) s_shape[threadIdx.x] = kShapeTable[threadIdx.x];
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    for (int k = threadIdx.x; k < 4 * kHidden + 16; k += 64 * kWaves) s_bias[k] = ((const float*)(image + kOffB))[k];
    start_chunk<64 * kWaves>(s_buf[0], image + kChunkOff[0], kChunkBytes[0]);
    __syncthreads();

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
                uint32_t fb[2][8];
                {
                    Board s;
                    unpack_board(A, B, s);
                    uint32_t own[8];
                    board_features(s, (int)p.L, (int)p.M, own);
                    both_features(own, g, fb);
                }
                split_logits(s_buf, s_bias, image, buf, t + 1 < q.T || pass + 1 < passes, lane, g, fb, lg);
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
            uint32_t cfg = current_config(s, p, (uint32_t)b, clock + t);
            const bool done = advance_board<kAutoReset>(s, cfg, rot, loc, p, (uint32_t)b, clock + t, s_shape, reward, mine);
            pack_board(s, A, B);
            if (valid && writer) {
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

}  // namespace psp
}  // namespace tpl

extern "C" int tpl_policy_act_split(tpl_env* e, const void* image, uint8_t* action, float* logits, void* stream) {
    if (!e) return fail_msg(TPL_ERR_ARG, "env is null");
    if (!image || !action) return fail_msg(TPL_ERR_ARG, "image/action is null");
    if (((uintptr_t)image & 15u) != 0) return fail_msg(TPL_ERR_ARG, "image must be 16-byte aligned");
    DeviceGuard guard(e->device);
    PolicySplitArgs p{};
    p.plane_a = e->plane_a; p.plane_b = e->plane_b; p.n = e->n; p.L = e->L; p.M = e->M;
    p.image = (const uint8_t*)image; p.action = action; p.logits = logits;
    // one resident workgroup per CU, eight waves of 32 boards, looping over board tiles
    const int64_t groups = ((e->n + 31) / 32 + kWaves - 1) / kWaves;
    hipLaunchKernelGGL(policy_split_kernel, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(64 * kWaves), 0, (hipStream_t)stream, p);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

extern "C" int tpl_actor_rollout_split(tpl_env* e, const void* image, int32_t num_steps, float epsilon, uint64_t seed, uint32_t step0,
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
    if (e->auto_reset) hipLaunchKernelGGL(actor_rollout_split_kernel<true>, grid, block, 0, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(actor_rollout_split_kernel<false>, grid, block, 0, (hipStream_t)stream, q);
    TPL_HIP(hipGetLastError());
    count_steps(e, num_steps);
    return TPL_OK;
}
