// glds_probe.hip -- what `global_load_lds_dword` does with its immediate offset and with inactive lanes (stand-alone):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/glds_probe tools/glds_probe.hip && /tmp/glds_probe
// Each lane l points at src[16 l .. 16 l + 15] (a 64-byte record of its own).  Three LDS-DMA instructions:
//   row 0: offset 0, all lanes         row 1: offset 4 (immediate), all lanes         row 2: offset 0, odd lanes only
// The LDS rows are pre-filled with 0xEEEEEEEE; the kernel copies the three rows (and a guard row behind each) out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void global_ptr;
typedef __attribute__((address_space(3))) void lds_ptr;

__global__ void probe(const unsigned* src, unsigned* out) {
    __shared__ unsigned rows[8][64];
    const int l = threadIdx.x;
    for (int r = 0; r < 8; ++r) rows[r][l] = 0xEEEEEEEEu;
    __syncthreads();
    const unsigned* mine = src + 16 * l;
    __builtin_amdgcn_global_load_lds((global_ptr*)mine, (lds_ptr*)&rows[0][0], 4, 0, 0);
    __builtin_amdgcn_global_load_lds((global_ptr*)mine, (lds_ptr*)&rows[2][0], 4, 4, 0);
    if (l & 1) __builtin_amdgcn_global_load_lds((global_ptr*)mine, (lds_ptr*)&rows[4][0], 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int r = 0; r < 8; ++r) out[r * 64 + l] = rows[r][l];
}

int main() {
    std::vector<unsigned> h(64 * 16);
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 16; ++k) h[16 * l + k] = 1000u * l + k;      // record l, word k
    unsigned *src, *out;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, 8 * 64 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
    std::vector<unsigned> o(8 * 64);
    hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < 6; ++r) {
        printf("row %d:", r);
        for (int l = 0; l < 6; ++l) printf(" %08x", o[r * 64 + l]);
        printf(" ... lane 63: %08x\n", o[r * 64 + 63]);
    }
    // interpretation
    printf("offset 0: lane 1 got %u (want 1000: word 0 of record 1)\n", o[0 * 64 + 1]);
    printf("offset 4: row 2 lane 0 = %08x, lane 1 = %08x; row 3 lane 0 = %08x  (source word 1 = 1, 1001; if the LDS side moved too, row 2 is shifted by one lane)\n",
           o[2 * 64 + 0], o[2 * 64 + 1], o[3 * 64 + 0]);
    printf("odd lanes only: row 4 lane 0 = %08x (EEEEEEEE = untouched), lane 1 = %u\n", o[4 * 64 + 0], o[4 * 64 + 1]);
    return 0;
}
