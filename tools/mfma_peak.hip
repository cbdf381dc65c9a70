// mfma_peak.hip -- what the matrix pipe delivers when NOTHING else runs: v_mfma_f32_16x16x32_bf16 back to back on eight independent
// accumulators, operands held in registers (no LDS, no other vector instruction in the loop), W waves per SIMD on every SIMD of the
// chip, for about a millisecond and for tens of milliseconds.  It separates two readings of "the policy kernel holds 1.85-1.9 GHz":
// a clock the chip takes under ANY dense matrix load (then 2.5 PFLOP/s, which assumes 2.4 GHz, is not reachable by any kernel),
// or a clock it takes under THIS kernel's mix of matrix, LDS and vector work.
// Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, unsigned long long* clocks) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (float)(threadIdx.x + i)); b[i] = (__bf16)(0.002f * (float)(threadIdx.x * 3 + i)); }
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }   // shader cycles, 100-MHz ticks
}

// the same with v_mfma_f32_32x32x16_bf16 (twice the flops per instruction, four independent accumulators of sixteen registers)
__global__ __launch_bounds__(256) void mfma32_kernel(float* out, int iters, unsigned long long* clocks) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (float)(threadIdx.x + i)); b[i] = (__bf16)(0.002f * (float)(threadIdx.x * 3 + i)); }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) s += acc[k][j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }
}

// the float32 pipe: v_mfma_f32_16x16x4_f32 (2048 flops an instruction), eight independent accumulators
__global__ __launch_bounds__(256) void mfma_f32_kernel(float* out, int iters, unsigned long long* clocks) {
    float a = 0.001f * (float)threadIdx.x, b = 0.002f * (float)(threadIdx.x * 3);
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }
}

int main() {
    float* out; unsigned long long* clocks;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clocks, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int waves_per_simd : {1, 2, 4}) {
        for (int iters : {2000, 60000}) {
            const int blocks = 256 * waves_per_simd;             // blocks of four waves: one wave per SIMD per block
            hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, out, 200, clocks);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clocks);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CK(hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost));
            const double flops = 2.0 * 16 * 16 * 32 * 32.0 * iters * (double)blocks * 4;      // per MFMA x 32 per iteration x waves
            printf("%d wave(s) per SIMD, %6d x 32 MFMAs per wave: %8.3f ms = %7.1f TFLOP/s = %.3f of 2500; shader clock in the loop %.3f GHz\n",
                   waves_per_simd, iters, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2500.0,
                   (double)h[0] / ((double)h[1] * 10.0) );
        }
    }
    for (int waves_per_simd : {1, 2, 4}) {
        const int iters = 60000, blocks = 256 * waves_per_simd;
        hipLaunchKernelGGL(mfma32_kernel, dim3(blocks), dim3(256), 0, 0, out, 200, clocks);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma32_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clocks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CK(hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost));
        const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * (double)blocks * 4;
        printf("32x32x16: %d wave(s) per SIMD, %6d x 16 MFMAs per wave: %8.3f ms = %7.1f TFLOP/s = %.3f of 2500; shader clock in the loop %.3f GHz\n",
               waves_per_simd, iters, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2500.0, (double)h[0] / ((double)h[1] * 10.0));
    }
    for (int waves_per_simd : {1, 2, 4}) {
        const int iters = 60000, blocks = 256 * waves_per_simd;
        hipLaunchKernelGGL(mfma_f32_kernel, dim3(blocks), dim3(256), 0, 0, out, 200, clocks);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_f32_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clocks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CK(hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost));
        const double flops = 2.0 * 16 * 16 * 4 * 32.0 * iters * (double)blocks * 4;
        printf("f32 16x16x4: %d wave(s) per SIMD, %6d x 32 MFMAs per wave: %8.3f ms = %7.1f TFLOP/s = %.3f of 157.3; shader clock in the loop %.3f GHz\n",
               waves_per_simd, iters, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3, (double)h[0] / ((double)h[1] * 10.0));
    }
    return 0;
}
