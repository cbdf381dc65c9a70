// mem_pattern.hip -- how fast can gfx950 move the step kernel's bytes, as a function of the access structure?
// Stand-alone micro-benchmark (hipcc --offload-arch=gfx950 -O3 tools/mem_pattern.hip -o /tmp/mem_pattern).
// Every variant touches two uint4 planes of N boards in place (32 B read + 32 B written per board); some add the
// step kernel's small streams (1 B action read, 4 B reward + 1 B done written).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    uint4* a; uint4* b; const uint8_t* act; float* reward; uint8_t* done; int64_t n;
};

// the step kernel's structure: one block-strided tile of kBpl boards per lane, all loads, then all stores
template <int kBpl, bool kStreams>
__global__ __launch_bounds__(256) void tile_kernel(Args p) {
    const int64_t base = (int64_t)blockIdx.x * (256 * kBpl) + threadIdx.x;
    uint4 A[kBpl], B[kBpl]; uint32_t act[kBpl];
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        const int64_t i = base + k * 256;
        A[k] = p.a[i]; B[k] = p.b[i];
        act[k] = kStreams ? p.act[i] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kBpl; ++k) {
        const int64_t i = base + k * 256;
        A[k].x ^= act[k] + 1u; B[k].y += 3u;
        if (kStreams) { p.reward[i] = (float)(A[k].x & 3u); p.done[i] = (uint8_t)(B[k].y & 1u); }
        p.a[i] = A[k]; p.b[i] = B[k];
    }
}

// persistent grid: `blocks` blocks walk the tiles with a stride, the next tile's loads issued before this tile's stores
template <bool kStreams>
__global__ __launch_bounds__(256) void stream_kernel(Args p, int64_t tiles) {
    int64_t t = blockIdx.x;
    if (t >= tiles) return;
    int64_t i = t * 256 + threadIdx.x;
    uint4 A = p.a[i], B = p.b[i]; uint32_t act = kStreams ? p.act[i] : 0u;
    for (;;) {
        const int64_t tn = t + gridDim.x;
        uint4 An = A, Bn = B; uint32_t actn = 0;
        const int64_t in = tn * 256 + threadIdx.x;
        if (tn < tiles) { An = p.a[in]; Bn = p.b[in]; actn = kStreams ? p.act[in] : 0u; }
        A.x ^= act + 1u; B.y += 3u;
        if (kStreams) { p.reward[i] = (float)(A.x & 3u); p.done[i] = (uint8_t)(B.y & 1u); }
        p.a[i] = A; p.b[i] = B;
        if (tn >= tiles) break;
        t = tn; i = in; A = An; B = Bn; act = actn;
    }
}


// which of the small streams cost what: kFlags bit0 = action read (1 B), bit1 = reward write (4 B), bit2 = done
// write (1 B, each wave stores 64 B = half a line), bit3 = done write staged through LDS (the block's 256 bytes
// leave as 16-B stores of 16 lanes: whole lines from one instruction)
template <int kFlags>
__global__ __launch_bounds__(256) void flags_kernel(Args p) {
    __shared__ uint8_t s_done[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint4 A = p.a[i], B = p.b[i];
    const uint32_t act = (kFlags & 1) ? p.act[i] : 0u;
    A.x ^= act + 1u; B.y += 3u;
    if (kFlags & 2) p.reward[i] = (float)(A.x & 3u);
    if (kFlags & 4) p.done[i] = (uint8_t)(B.y & 1u);
    if (kFlags & 8) {
        s_done[threadIdx.x] = (uint8_t)(B.y & 1u);
        __syncthreads();
        if (threadIdx.x < 16) ((uint4*)(p.done + (int64_t)blockIdx.x * 256))[threadIdx.x] = ((const uint4*)s_done)[threadIdx.x];
    }
    p.a[i] = A; p.b[i] = B;
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 20; ++r) launch(r);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch(r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int steps = 300;
    Args p{};
    CK(hipMalloc(&p.a, n * 16)); CK(hipMalloc(&p.b, n * 16));
    uint8_t* act; CK(hipMalloc(&act, (size_t)n * steps));
    CK(hipMalloc(&p.reward, n * 4)); CK(hipMalloc(&p.done, n));
    CK(hipMemset(p.a, 1, n * 16)); CK(hipMemset(p.b, 2, n * 16)); CK(hipMemset(act, 3, (size_t)n * steps));
    auto report = [&](const char* name, float us, double bytes) {
        printf("%-44s %7.2f us  %6.2f TB/s\n", name, us, bytes / us / 1e6);
        fflush(stdout);
    };
    struct Case { const char* name; std::function<void(int)> launch; double bytes; };
    std::vector<Case> cases;
    auto rot = [&](int r) { p.act = act + (size_t)(r % steps) * n; };
    cases.push_back({"tile bpl=1 planes only", [&](int) { hipLaunchKernelGGL((tile_kernel<1, false>), dim3(n / 256), dim3(256), 0, 0, p); }, 64.0});
    cases.push_back({"tile bpl=2 planes only", [&](int) { hipLaunchKernelGGL((tile_kernel<2, false>), dim3(n / 512), dim3(256), 0, 0, p); }, 64.0});
    cases.push_back({"tile bpl=2 + streams", [&](int r) { rot(r); hipLaunchKernelGGL((tile_kernel<2, true>), dim3(n / 512), dim3(256), 0, 0, p); }, 70.0});
    cases.push_back({"stream 2048 planes only", [&](int) { hipLaunchKernelGGL((stream_kernel<false>), dim3(2048), dim3(256), 0, 0, p, n / 256); }, 64.0});
    cases.push_back({"stream 2048 + streams", [&](int r) { rot(r); hipLaunchKernelGGL((stream_kernel<true>), dim3(2048), dim3(256), 0, 0, p, n / 256); }, 70.0});
#define FLAGS(F, label, bytes) cases.push_back({label, [&](int r) { rot(r); hipLaunchKernelGGL((flags_kernel<F>), dim3(n / 256), dim3(256), 0, 0, p); }, bytes})
    FLAGS(0, "flags: planes only", 64.0);
    FLAGS(1, "flags: + action read", 65.0);
    FLAGS(2, "flags: + reward write", 68.0);
    FLAGS(4, "flags: + done write (64 B per wave)", 65.0);
    FLAGS(8, "flags: + done write (LDS-staged lines)", 65.0);
    FLAGS(7, "flags: + action + reward + done", 70.0);
    FLAGS(11, "flags: + action + reward + staged done", 70.0);
    cases.push_back({"empty kernel, 1 block (launch period)", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, (int*)nullptr); }, 0.0});
    cases.push_back({"empty kernel, 2048 blocks", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0, (int*)nullptr); }, 0.0});
    for (int round = 0; round < 2; ++round) {
        printf("-- round %d\n", round);
        for (auto& c : cases) report(c.name, time_us(c.launch, steps), c.bytes * n);
    }
    return 0;
}
