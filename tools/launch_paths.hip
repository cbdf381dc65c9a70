// launch_paths.hip -- what one kernel launch costs the HOST thread on this runtime, by the API it goes through, for a kernel
// that takes a 200-byte argument struct by value (as step_kernel does) and does next to nothing on the device:
//   hipLaunchKernelGGL (the <<<>>> path: the kernel is looked up by its host stub on every call),
//   hipModuleLaunchKernel on a hipFunction_t obtained once (hipGetFuncBySymbol), arguments as one buffer,
//   the same through hipExtModuleLaunchKernel.
// Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/launch_paths.hip -o /tmp/launch_paths && /tmp/launch_paths
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { uint64_t w[25]; };   // 200 bytes

__global__ __launch_bounds__(256) void small_kernel(Args a) {
    if (a.w[24] == 0x1234u && threadIdx.x == 999) ((volatile uint64_t*)a.w[0])[0] = a.w[1];
}

template <typename F>
static double per_call_us(F&& f, int n) {
    for (int i = 0; i < 200; ++i) f();
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    const auto t1 = std::chrono::steady_clock::now();
    CK(hipDeviceSynchronize());
    const auto t2 = std::chrono::steady_clock::now();
    printf("   (the device needed %.1f ms more after the last call)\n", std::chrono::duration<double, std::milli>(t2 - t1).count());
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

int main() {
    Args a{};
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int n = 20000, blocks = 512;            // 131,072 threads: the grid of a shard's step
    printf("hipLaunchKernelGGL:               %.2f us per call\n", per_call_us([&] { hipLaunchKernelGGL(small_kernel, dim3(blocks), dim3(256), 0, s, a); }, n));
    hipFunction_t f;
    CK(hipGetFuncBySymbol(&f, (const void*)small_kernel));
    size_t size = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    printf("hipModuleLaunchKernel (extra):    %.2f us per call\n",
           per_call_us([&] { CK(hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, 0, s, nullptr, extra)); }, n));
    printf("hipExtModuleLaunchKernel (extra): %.2f us per call\n",
           per_call_us([&] { CK(hipExtModuleLaunchKernel(f, blocks * 256, 1, 1, 256, 1, 1, 0, s, nullptr, extra, nullptr, nullptr, 0)); }, n));
    void* params[] = {&a};
    printf("hipLaunchKernel (params):         %.2f us per call\n",
           per_call_us([&] { CK(hipLaunchKernel((const void*)small_kernel, dim3(blocks), dim3(256), params, 0, s)); }, n));
    return 0;
}
