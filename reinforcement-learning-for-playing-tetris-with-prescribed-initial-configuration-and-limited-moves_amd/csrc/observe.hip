// observe.hip -- the [N,217] observation (get_state() as a vector: 200 cells row-major, one-hot current piece,
// one-hot next piece, lines left, moves left, terminal flag; Model(217, 14), model/train.py:26) written at the
// memory system's pace.
//
// The output is 868 B (f32) or 434 B (bf16) per board against 32 B read, so the kernel is a store stream, and what
// matters is that every store instruction writes whole lines.  Rows are 217 elements -- never a multiple of 16
// bytes -- but the 64 rows of a wave's boards are one contiguous, 16-byte aligned span of the output, so the wave
// treats it as flat: each lane owns 16-byte chunks of that span and every store instruction writes 1 KiB.
//
//   stage A  each lane turns its own board into one BYTE per feature in LDS (every feature value is an integer in
//            0..254).  Cells leave as whole dwords: the ten column words are first packed four columns to a
//            register (one byte-slice of each), so "bit r of four neighbouring columns" is one shift and one mask.
//   stage B  each lane reads the 8 (bf16) or 4 (f32) bytes of its chunk with one LDS read, converts them
//            (`v_cvt_f32_ubyteN`; a small integer's bf16 is the upper half of its f32) and stores 16 bytes.
//
// LDS rows are 224 bytes apart: 217 features, then a copy of the NEXT board's first seven cells, so that a chunk
// that runs over the end of a row reads straight on.  (Row starts are then 16-byte aligned for stage A; stage B's
// reads are unaligned, which the LDS of gfx950 serves directly.)
#include "tpl_internal.h"
#include "tpl_observe.h"

namespace tpl {
namespace {

using namespace obs;

template <typename T>
__global__ __launch_bounds__(64 * kObsWaves) void observe_kernel(const uint4* plane_a, const uint4* plane_b, int64_t n,
                                                                uint32_t L, uint32_t M, T* out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[kObsWaves][kWaveLds];
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * kObsWaves + (threadIdx.x >> 6)) * 64;     // the wave's first board
    if (base >= n) return;                                                              // wave-uniform
    const int count = (int)((n - base) < 64 ? (n - base) : 64);
    uint8_t* const rows = s_rows[threadIdx.x >> 6];
    int lines_left = 0;
    if (lane < count) {
        Board s;
        unpack_board(plane_a[base + lane], plane_b[base + lane], s);
        lines_left = board_to_bytes(s, L, M, rows, lane);
    }
    store_span<T>(rows, lane, count, base, lines_left, out);
}

// Tetris.board for every board (game/tetris.py:186; the first element of get_state(), :435-436): one byte per cell,
// row-major [n][20][10], 0 / 1 -- the storage of a bool array.  Same two stages as the observation: cells as whole
// dwords into LDS (rows are 200 bytes, so the wave's span is flat in LDS too), then 16-byte non-temporal stores.
__global__ __launch_bounds__(64 * kObsWaves) void cells_kernel(const uint4* plane_a, const uint4* plane_b, int64_t n,
                                                              uint8_t* out) {
    constexpr int kCells = kRows * kCols;                          // 200
    __shared__ __attribute__((aligned(16))) uint8_t s_cells[kObsWaves][64 * kCells];
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * kObsWaves + (threadIdx.x >> 6)) * 64;
    if (base >= n) return;                                                              // wave-uniform
    const int count = (int)((n - base) < 64 ? (n - base) : 64);
    uint8_t* const cells = s_cells[threadIdx.x >> 6];
    if (lane < count) {
        Board s;
        unpack_board(plane_a[base + lane], plane_b[base + lane], s);
        uint32_t d[kCols];
#pragma unroll
        for (int x = 0; x < kCols; ++x) d[x] = s.c[x] >> 1;
        uint32_t* const row32 = (uint32_t*)(cells + lane * kCells);
#define TPL_CELL_SLICE(SL, PAIRS)                                                                                 \
        {                                                                                                         \
            const uint32_t g0 = pack_slice<SL>(s.c[0], s.c[1], s.c[2], s.c[3]);                                   \
            const uint32_t g1 = pack_slice<SL>(s.c[4], s.c[5], s.c[6], s.c[7]);                                   \
            const uint32_t g2 = pack_slice<SL>(s.c[8], s.c[9], d[0], d[1]);                                       \
            const uint32_t g3 = pack_slice<SL>(d[2], d[3], d[4], d[5]);                                           \
            const uint32_t g4 = pack_slice<SL>(d[6], d[7], d[8], d[9]);                                           \
            _Pragma("unroll") for (int k = 0; k < PAIRS; ++k) {                                                   \
                const int u = 2 * k, q = 5 * (4 * SL + k);                                                        \
                row32[q + 0] = (g0 >> u) & 0x01010101u;                                                           \
                row32[q + 1] = (g1 >> u) & 0x01010101u;                                                           \
                row32[q + 2] = (g2 >> u) & 0x01010101u;                                                           \
                row32[q + 3] = (g3 >> u) & 0x01010101u;                                                           \
                row32[q + 4] = (g4 >> u) & 0x01010101u;                                                           \
            }                                                                                                     \
        }
        TPL_CELL_SLICE(0, 4)
        TPL_CELL_SLICE(1, 4)
        TPL_CELL_SLICE(2, 2)
#undef TPL_CELL_SLICE
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int total = count * kCells;                             // bytes in the span; a multiple of 8
    uint8_t* const span = out + base * kCells;
    for (int c = lane; c < total / 16; c += 64) {
        const u32x4 v = *(const u32x4*)(cells + 16 * c);
        __builtin_nontemporal_store(v, (u32x4*)(span + 16 * c));
    }
    if (lane == 0 && (total & 15)) *(uint2*)(span + (total & ~15)) = *(const uint2*)(cells + (total & ~15));
}

}  // namespace

int launch_cells(const uint4* plane_a, const uint4* plane_b, int64_t n, uint8_t* out, hipStream_t stream) {
    const dim3 grid((unsigned)((n + 64 * kObsWaves - 1) / (64 * kObsWaves))), block(64 * kObsWaves);
    hipLaunchKernelGGL(cells_kernel, grid, block, 0, stream, plane_a, plane_b, n, out);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

// 16-byte stores need a 16-byte aligned output; the caller falls back to the element-wise kernel otherwise
bool observe_fast_path(const void* out) { return ((uintptr_t)out & 15u) == 0; }

int launch_observe(const uint4* plane_a, const uint4* plane_b, int64_t n, uint32_t L, uint32_t M, void* out, int32_t dtype,
                   hipStream_t stream) {
    const dim3 grid((unsigned)((n + 64 * kObsWaves - 1) / (64 * kObsWaves))), block(64 * kObsWaves);
    if (dtype == TPL_F32)
        hipLaunchKernelGGL(observe_kernel<float>, grid, block, 0, stream, plane_a, plane_b, n, L, M, (float*)out);
    else
        hipLaunchKernelGGL(observe_kernel<__hip_bfloat16>, grid, block, 0, stream, plane_a, plane_b, n, L, M, (__hip_bfloat16*)out);
    TPL_HIP(hipGetLastError());
    return TPL_OK;
}

}  // namespace tpl
