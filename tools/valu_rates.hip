// valu_rates.hip -- issue cost of single gfx950 vector instructions relative to v_add_u32, measured: which of the integer
// instructions the non-MFMA kernels are made of are full rate (one quad-cycle per wave64 instruction) and which are not.
// Stand-alone (hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates).
// Every kernel runs kIter x 32 copies of ONE instruction on eight independent registers per wave, eight waves per SIMD on every
// SIMD of the chip, so the vector ALU is the only thing that can be busy; time / time(v_add_u32) = quad-cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIter = 2000;

// OP(d, a): one instruction with destination register d (also a source, so the copies of one register form a chain) and a
// second source a
#define KERNEL(NAME, DECL, OP)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                                        \
        DECL                                                                                                           \
        for (int it = 0; it < kIter; ++it) {                                                                           \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) }          \
        }                                                                                                              \
        SINK                                                                                                           \
    }

#define DECL32                                                                                                         \
    uint32_t r[8], a = seed + threadIdx.x;                                                                             \
    for (int k = 0; k < 8; ++k) r[k] = seed * (k + 3) + threadIdx.x;
#define SINK                                                                                                           \
    uint32_t acc = 0;                                                                                                  \
    for (int k = 0; k < 8; ++k) acc ^= (uint32_t)r[k];                                                                 \
    if (acc == 0x12345678u) out[threadIdx.x] = acc;

#define A1(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_add_u32, DECL32, A1)
#define A2(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_mul_lo_u32, DECL32, A2)
#define A3(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_mul_hi_u32, DECL32, A3)
#define A4(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_mul_u32_u24, DECL32, A4)
#define A5(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_mad_u32_u24, DECL32, A5)
#define A6(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r[i]));
KERNEL(k_ffbl_b32, DECL32, A6)
#define A7(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_bcnt_u32_b32, DECL32, A7)
#define A8(i) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x78" : "+v"(r[i]) : "v"(a));
KERNEL(k_bitop3_b32, DECL32, A8)
#define A9(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_add3_u32, DECL32, A9)
#define A10(i) asm volatile("v_lshl_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_lshl_or_b32, DECL32, A10)
#define A11(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[i]) : "v"(a));
KERNEL(k_lshlrev_b32_sdwa, DECL32, A11)
#define A12(i) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_min3_u32, DECL32, A12)
#define A13(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(r[i]));
KERNEL(k_bfe_u32, DECL32, A13)
#define A14(i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(r[i]) : "v"(a));
KERNEL(k_alignbit_b32, DECL32, A14)
#define A15(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : );
KERNEL(k_cndmask_b32, DECL32, A15)
#define A16(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_mul_i32_i24, DECL32, A16)
#define A17(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_perm_b32, DECL32, A17)
#define A18(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i]));
KERNEL(k_cvt_f32_u32, DECL32, A18)
#define A19(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_fma_f32, DECL32, A19)
#define A20(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_pk_mul_lo_u16, DECL32, A20)

#define C1(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_and_b32, DECL32, C1)
#define C2(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_xor_b32, DECL32, C2)
#define C3(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]));
KERNEL(k_lshlrev_b32, DECL32, C3)
#define C4(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_sub_u32, DECL32, C4)
#define C5(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_min_u32, DECL32, C5)
#define C6(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_add_f32, DECL32, C6)
#define C7(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_mov_b32, DECL32, C7)
#define C8(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_and_or_b32, DECL32, C8)
#define C9(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r[i]) : "v"(a) : "s10", "s11");
KERNEL(k_cndmask_b32_sgpr_mask, DECL32, C9)
#define C10(i) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1" : : "v"(r[i]), "v"(a) : "s10", "s11");
KERNEL(k_cmp_lt_u32_to_sgprs, DECL32, C10)
#define C11(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_add_u32_e64, DECL32, C11)
#define C12(i) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[i]) : "v"(a));
KERNEL(k_add_u32_sdwa, DECL32, C12)

// v_cndmask_b32 reading VCC: with VCC set to a constant before the loop, and in the 64-bit encoding
#define DECL32_VCC DECL32 asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
#define C13(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
KERNEL(k_cndmask_b32_vcc_set, DECL32_VCC, C13)
#define C14(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
KERNEL(k_cndmask_b32_e64_vcc, DECL32_VCC, C14)
#define C15(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
KERNEL(k_addc_co_u32, DECL32_VCC, C15)
#define C16(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_or_b32, DECL32, C16)
#define C17(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[i]));
KERNEL(k_lshrrev_b32, DECL32, C17)
#define C18(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_max_u32, DECL32, C18)
#define C19(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_mul_f32, DECL32, C19)
#define C20(i) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_xad_u32, DECL32, C20)
#define C21(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_lshl_add_u32, DECL32, C21)
#define C22(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
KERNEL(k_cmp_lt_u32_to_vcc, DECL32, C22)

// one v_cndmask_b32 (VOP2 encoding, VCC implicit) among seven v_mul_u32_u24: is its cost hidden behind the other
// instructions (a wait that other waves fill) or added to them (the pipe is held)?
#define C23(i) if (i == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a)); else asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_one_cndmask_vcc_in_eight, DECL32_VCC, C23)
#define C24(i) if (i == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a)); else asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
KERNEL(k_one_cndmask_e64_in_eight, DECL32_VCC, C24)
// the same VOP2 instruction whose destination is not one of its sources
#define C25(i) asm volatile("v_cndmask_b32 %0, %1, %1, vcc" : "=v"(r[i]) : "v"(a));
KERNEL(k_cndmask_b32_vcc_independent, DECL32_VCC, C25)

// shifts: left and right, by a constant and by a register
#define D1(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[i]));
KERNEL(k_lshlrev_b32_by_3, DECL32, D1)
#define D2(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
KERNEL(k_lshlrev_b32_by_vgpr, DECL32, D2)
#define D3(i) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
KERNEL(k_lshrrev_b32_by_vgpr, DECL32, D3)
#define D4(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r[i]));
KERNEL(k_ashrrev_i32, DECL32, D4)
#define D5(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
KERNEL(k_subrev_like, DECL32, D5)

// 64-bit destinations
#define DECL64                                                                                                         \
    uint64_t r[8]; uint32_t a = seed + threadIdx.x; uint64_t a64 = ((uint64_t)seed << 32) | threadIdx.x;              \
    for (int k = 0; k < 8; ++k) r[k] = (uint64_t)seed * (k + 3) + threadIdx.x;
#define B1(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[i]) : "v"(a64));
KERNEL(k_lshl_add_u64, DECL64 (void)a;, B1)
#define B2(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(r[i]));
KERNEL(k_lshlrev_b64, DECL64 (void)a; (void)a64;, B2)
#define B3(i) asm volatile("v_mov_b64 %0, %1" : "+v"(r[i]) : "v"(a64));
KERNEL(k_mov_b64, DECL64 (void)a;, B3)
#define B4(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(r[i]) : "v"(a) : "vcc");
KERNEL(k_mad_u64_u32, DECL64 (void)a64;, B4)
#define B5(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(*(uint32_t*)&r[i]) : "v"(a));
KERNEL(k_pk_add_u16, DECL64 (void)a64;, B5)

typedef void (*Kern)(uint32_t*, uint32_t);

static float run(Kern k, uint32_t* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;                      // 256 CUs x 8 blocks of four waves: eight waves on every SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 7u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 7u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t* out;
    CK(hipMalloc(&out, 4096));
    struct { const char* name; Kern k; } ks[] = {
        {"v_add_u32", k_add_u32}, {"v_add_u32_e64", k_add_u32_e64}, {"v_add_u32_sdwa", k_add_u32_sdwa}, {"v_sub_u32", k_sub_u32},
        {"v_and_b32", k_and_b32}, {"v_xor_b32", k_xor_b32}, {"v_lshlrev_b32", k_lshlrev_b32}, {"v_min_u32", k_min_u32}, {"v_add_f32", k_add_f32},
        {"v_mov_b32", k_mov_b32}, {"v_and_or_b32", k_and_or_b32}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_b32_sgpr_mask},
        {"v_cmp_lt_u32 -> sgprs", k_cmp_lt_u32_to_sgprs},
        {"v_cmp_lt_u32 -> vcc", k_cmp_lt_u32_to_vcc}, {"v_cndmask_b32 (vcc, set)", k_cndmask_b32_vcc_set}, {"v_cndmask_b32_e64 (vcc)", k_cndmask_b32_e64_vcc},
        {"v_cndmask_b32 (vcc) dest != src", k_cndmask_b32_vcc_independent},
        {"1 cndmask(vcc) + 7 mul_u24", k_one_cndmask_vcc_in_eight}, {"1 cndmask_e64 + 7 mul_u24", k_one_cndmask_e64_in_eight},
        {"v_addc_co_u32 (vcc)", k_addc_co_u32}, {"v_or_b32", k_or_b32}, {"v_lshrrev_b32", k_lshrrev_b32}, {"v_lshlrev_b32 by 3", k_lshlrev_b32_by_3}, {"v_lshlrev_b32 by a register", k_lshlrev_b32_by_vgpr},
        {"v_lshrrev_b32 by a register", k_lshrrev_b32_by_vgpr}, {"v_ashrrev_i32", k_ashrrev_i32}, {"v_sub_u32 (operands swapped)", k_subrev_like}, {"v_max_u32", k_max_u32}, {"v_mul_f32", k_mul_f32},
        {"v_xad_u32", k_xad_u32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mul_i32_i24", k_mul_i32_i24}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_ffbl_b32", k_ffbl_b32}, {"v_bcnt_u32_b32", k_bcnt_u32_b32},
        {"v_bitop3_b32", k_bitop3_b32}, {"v_add3_u32", k_add3_u32}, {"v_lshl_or_b32", k_lshl_or_b32},
        {"v_lshlrev_b32_sdwa", k_lshlrev_b32_sdwa}, {"v_min3_u32", k_min3_u32}, {"v_bfe_u32", k_bfe_u32}, {"v_alignbit_b32", k_alignbit_b32},
        {"v_cndmask_b32 (vcc)", k_cndmask_b32}, {"v_perm_b32", k_perm_b32}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_fma_f32", k_fma_f32},
        {"v_pk_mul_lo_u16", k_pk_mul_lo_u16}, {"v_pk_add_u16", k_pk_add_u16},
        {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshlrev_b64", k_lshlrev_b64}, {"v_mov_b64", k_mov_b64}, {"v_mad_u64_u32", k_mad_u64_u32},
    };
    const float base = run(k_add_u32, out);
    const double instr_per_simd = 8.0 * kIter * 32;          // eight waves per SIMD
    printf("v_add_u32: %.3f ms for %d x 32 instructions on eight waves per SIMD = %.2f GHz if one wave64 instruction takes four cycles\n",
           base, kIter, instr_per_simd * 4.0 / (base * 1e-3) / 1e9);
    for (auto& e : ks) {
        const float ms = run(e.k, out);
        printf("%-28s %8.3f ms  = %5.2f x v_add_u32\n", e.name, ms, ms / base);
    }
    return 0;
}
