// tpl_internal.h -- what the translation units of libtetris_piclim.so share: the handle, error reporting, guards.
#pragma once

#include "../../include/tetris_piclim.h"
#include "tpl_device.h"

namespace tpl {

constexpr int kBlock = 256;
constexpr int kStatShards = 256;
constexpr int kStatStride = 16;   // uint64 per shard -> one 128-B line each

struct Pool {
    uint8_t* rec = nullptr;    // [n_cfg] records of 1 << stride_shift bytes: plane-A word, plane-B word, 64-bit piece words 1..
    uint8_t* side = nullptr;   // [n_cfg] 64-byte side records behind them: the same board UNPACKED (tpl_device.h)
    bool side_ready = false;   // the side records are written on first use (tpl_rollout): a caller that only steps never
                               //   pays for them, nor has 64 bytes per configuration pushed through the caches
    int64_t n_cfg = 0;
    void* owned = nullptr;
};

}  // namespace tpl

struct tpl_env {
    int64_t n = 0;
    int32_t L = 0, M = 0, device = 0;
    int64_t global_offset = 0;
    uint64_t seed = 0;
    int32_t auto_reset = 0, assign_mode = 0;
    float r_line = 1.0f, r_win = 0.0f, r_lose = 0.0f;
    int32_t boards_per_lane = 2;        // tuning knobs of the step kernel: boards per lane (1, 2 or 4)
    int32_t block_threads = 256;        //   and threads per block (64, 128, 256 or 512)
    uint4* plane_a = nullptr;
    uint4* plane_b = nullptr;
    unsigned long long* stats = nullptr;// [kStatShards][kStatStride]
    unsigned long long* clock = nullptr;// [padded boards / 32] step clocks (tpl_device.h)
    void* owned = nullptr;
    // Two pool buffers.  A board carries the slot its configuration lives in, so boards that are mid-episode when a new
    // pool arrives finish on the old one; new episodes start from pool[cur_slot].  The other slot may be overwritten
    // once no running board can still refer to it: after a full reset, or M+1 steps after the last swap.
    tpl::Pool pool[2];
    uint32_t stride_shift = 0;
    int32_t cur_slot = 0;
    int64_t steps_since_swap = 0;       // steps ENQUEUED THROUGH THE API since cur_slot last changed (graph replays are
                                        //   not seen: the count errs on the side of refusing a swap)
    bool other_slot_live = false;       // boards may still refer to pool[cur_slot ^ 1]
    bool assign_dirty = false;          // the assignment mode changed under running boards: a full reset is due
#ifdef TPL_DIAG_CLOCK
    unsigned long long* step_diag = nullptr;
#endif
};

namespace tpl {

// sets the calling thread's tpl_last_error() message and returns `code`
int fail_msg(int code, const char* fmt, ...);

#define TPL_HIP(call)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return ::tpl::fail_msg(TPL_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// tetris_piclim.hip: what every board-advancing entry point checks before it launches (auto-reset needs a pool; a changed
// assignment mode needs a full reset first), and the step count behind the pool-swap rule
int check_can_advance(tpl_env* e);
void count_steps(tpl_env* e, int64_t steps);

// carve_generator.hip: host threads worth starting when a caller names none (affinity mask capped by the cgroup CPU quota)
int host_cpu_budget();

// carve_generator.hip: the restart rule tried on a few fixed configurations on the host (verdict kept per (L, M, cutoff)): TPL_OK,
// or TPL_ERR_ARG / TPL_ERR_STATE with the message set when this (L, M) cannot be carved / NONE of the pilots finishes within the
// rule's bound (single configurations that cap are the callers' status[] to report).
// Both generators call it before a batch goes out -- on the device the alternative is a kernel that runs for minutes.
int carve_pilot(int32_t L, int32_t M, int64_t cutoff);

// forward_generator.hip: the rotation count of Tetris.move that shows the same shape as the forward solver's (letter index in
// IJLOSTZ, rotation) -- the two sub-packages of the reference order their rotations differently
int forward_move_rotations(int letter, int rotation);

// observe.hip: the [N,217] observation with 16-byte stores (needs a 16-byte aligned output)
bool observe_fast_path(const void* out);
int launch_observe(const uint4* plane_a, const uint4* plane_b, int64_t n, uint32_t L, uint32_t M, void* out, int32_t dtype,
                   hipStream_t stream);

// observe.hip: Tetris.board of every board as bytes [n][20][10] (needs a 16-byte aligned output)
int launch_cells(const uint4* plane_a, const uint4* plane_b, int64_t n, uint8_t* out, hipStream_t stream);

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace tpl
