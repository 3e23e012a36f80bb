// valu_cost.hip — issue cost of the VALU instructions the trace kernel is made of, on one SIMD of gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o racer-tracer_amd/build/valu_cost tools/microbench/valu_cost.hip
// Every case is a loop of 16 independent copies of one instruction (inline asm, so the compiler neither folds nor
// reorders them), run by W waves per SIMD on ONE CU; the cost printed is SIMD cycles per wave-instruction,
// measured with s_memtime (100 MHz constant clock) scaled by the shader clock the runtime reports.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 2048;

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

// d = 64-bit regs, f = 32-bit regs
#define CASE_D3(NAME, ASM)                                                                          \
    __global__ void NAME(double *out, double seed) {                                                \
        double a[16], b = seed, c = seed * 0.5;                                                     \
        for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;                                 \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                           \
        double s = 0;                                                                               \
        for (int i = 0; i < 16; ++i) s += a[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                             \
    }
#define CASE_F3(NAME, ASM)                                                                          \
    __global__ void NAME(double *out, double seed) {                                                \
        unsigned a[16], b = (unsigned)seed + 12345u, c = (unsigned)seed * 77u + 3u;                 \
        for (int i = 0; i < 16; ++i) a[i] = (unsigned)seed + i + threadIdx.x;                       \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                           \
        unsigned s = 0;                                                                             \
        for (int i = 0; i < 16; ++i) s += a[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                             \
    }
// 64-bit destination, 32-bit sources (v_mad_u64_u32, v_cvt_f64_u32)
#define CASE_DF(NAME, ASM)                                                                          \
    __global__ void NAME(double *out, double seed) {                                                \
        unsigned long long a[16];                                                                   \
        unsigned b = (unsigned)seed + 12345u, c = (unsigned)seed * 77u + 3u;                        \
        for (int i = 0; i < 16; ++i) a[i] = (unsigned)seed + i + threadIdx.x;                       \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                           \
        unsigned long long s = 0;                                                                   \
        for (int i = 0; i < 16; ++i) s += a[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;                                     \
    }

CASE_D3(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
CASE_D3(k_mul_f64, "v_mul_f64 %0, %0, %1")
CASE_D3(k_add_f64, "v_add_f64 %0, %0, %1")
CASE_D3(k_min_f64, "v_min_f64 %0, %0, %1")
CASE_D3(k_rcp_f64, "v_rcp_f64 %0, %0")
CASE_D3(k_rsq_f64, "v_rsq_f64 %0, %0")
CASE_D3(k_sqrt_f64, "v_sqrt_f64 %0, %0")
CASE_D3(k_floor_f64, "v_floor_f64 %0, %0")
CASE_D3(k_fract_f64, "v_fract_f64 %0, %0")
CASE_D3(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
CASE_D3(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
CASE_D3(k_mov_b64, "v_mov_b64 %0, %1")
CASE_D3(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")
CASE_D3(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
CASE_D3(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
CASE_DF(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
CASE_F3(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
CASE_F3(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
CASE_F3(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
CASE_F3(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
CASE_F3(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
CASE_F3(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
CASE_F3(k_xor_b32, "v_xor_b32 %0, %0, %1")
CASE_F3(k_add_u32, "v_add_u32 %0, %0, %1")
CASE_F3(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
CASE_F3(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
CASE_F3(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
CASE_F3(k_cndmask_const, "v_cndmask_b32_e64 %0, 0, 1.0, vcc")
CASE_F3(k_cndmask_b, "v_cndmask_b32 %0, %1, %2, vcc")
CASE_F3(k_cmp_then_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
CASE_F3(k_cndmask_fma_mix, "v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %0, %0, %1, %2")
CASE_F3(k_cmp_u32_vcc, "v_cmp_lt_u32 vcc, %0, %1")
CASE_F3(k_cmp_u32_sgpr, "v_cmp_lt_u32_e64 s[20:21], %0, %1")
CASE_F3(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
CASE_F3(k_and_b32, "v_and_b32 %0, %0, %1")
CASE_F3(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
CASE_F3(k_sub_u32, "v_sub_u32 %0, %0, %1")
CASE_F3(k_mov_b32, "v_mov_b32 %0, %1")
CASE_F3(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
CASE_F3(k_writelane, "v_writelane_b32 %0, s20, 3")
CASE_F3(k_readfirstlane, "v_readfirstlane_b32 s20, %0")
CASE_F3(k_swizzle, "ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)")
CASE_F3(k_permlane_dpp_bcast, "v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf")
CASE_D3(k_max_f64, "v_max_f64 %0, %0, %1")
CASE_D3(k_fmac_f64, "v_fmac_f64 %0, %1, %2")
CASE_D3(k_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
CASE_D3(k_cmp_f64_sgpr, "v_cmp_lt_f64_e64 s[20:21], %0, %1")
CASE_D3(k_cmp_class_f64, "v_cmp_class_f64 vcc, %0, 3")
CASE_D3(k_trunc_f64, "v_trunc_f64 %0, %0")
CASE_D3(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
CASE_D3(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
CASE_D3(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %2")
CASE_D3(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
CASE_F3(k_rcp_f32, "v_rcp_f32 %0, %0")
CASE_F3(k_rsq_f32, "v_rsq_f32 %0, %0")
CASE_F3(k_sqrt_f32, "v_sqrt_f32 %0, %0")
CASE_F3(k_sin_f32, "v_sin_f32 %0, %0")
CASE_F3(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
CASE_F3(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
CASE_F3(k_readlane, "v_readlane_b32 s20, %0, 3")
CASE_F3(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
CASE_F3(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
CASE_F3(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 5")

// 32-bit source -> 64-bit result and back
__global__ void k_cvt_f64_u32(double *out, double seed) {
    double a[16];
    unsigned b = (unsigned)seed + threadIdx.x;
    for (int i = 0; i < 16; ++i) a[i] = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(a[i]) : "v"(b));
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cvt_f32_f64(double *out, double seed) {
    float a[16];
    double b = seed + threadIdx.x;
    for (int i = 0; i < 16; ++i) a[i] = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(b));
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_empty(double *out, double seed) {
    double a = seed;
    for (int it = 0; it < ITERS; ++it) asm volatile("" : "+v"(a));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

struct Case {
    const char *name;
    void (*fn)(double *, double);
    int insts_per_iter;
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const double clock_hz = prop.clockRate * 1e3;
    printf("device %s, %d CUs, shader clock %.0f MHz\n", prop.gcnArchName, prop.multiProcessorCount, clock_hz / 1e6);
    double *out;
    CHECK(hipMalloc(&out, 1024 * sizeof(double)));
    std::vector<Case> cases = {
#define C(n) {#n, n, 16}
        C(k_fma_f64), C(k_mul_f64), C(k_add_f64), C(k_min_f64), C(k_rcp_f64), C(k_rsq_f64), C(k_sqrt_f64), C(k_floor_f64),
        C(k_fract_f64), C(k_ldexp_f64), C(k_cmp_f64), C(k_mov_b64), C(k_lshl_add_u64),
        C(k_pk_fma_f32), C(k_pk_mul_f32), C(k_mad_u64_u32), C(k_fma_f32), C(k_mul_lo_u32), C(k_mul_hi_u32), C(k_mul_u32_u24),
        C(k_mul_hi_u32_u24), C(k_mad_u32_u24), C(k_xor_b32), C(k_add_u32), C(k_add3_u32), C(k_cndmask), C(k_cndmask_sgpr), C(k_cndmask_const), C(k_cndmask_b), {"k_cmp_then_cndmask (2)", k_cmp_then_cndmask, 16},
        {"k_cndmask_fma_mix (2)", k_cndmask_fma_mix, 16}, C(k_cmp_u32_vcc), C(k_cmp_u32_sgpr), C(k_bitop3), C(k_and_b32), C(k_lshrrev_b32),
        C(k_sub_u32), C(k_mov_b32), C(k_mbcnt), C(k_writelane), C(k_readfirstlane), C(k_swizzle), C(k_permlane_dpp_bcast), C(k_max_f64),
        C(k_fmac_f64), C(k_lshlrev_b64), C(k_cmp_f64_sgpr), C(k_cmp_class_f64), C(k_trunc_f64),
        C(k_div_fixup_f64), C(k_div_fmas_f64), C(k_div_scale_f64), C(k_frexp_mant_f64), C(k_rcp_f32),
        C(k_rsq_f32), C(k_sqrt_f32), C(k_sin_f32), C(k_cvt_f32_u32), C(k_cvt_f64_u32), C(k_cvt_f32_f64), C(k_mov_dpp), C(k_readlane),
        C(k_bpermute), C(k_alignbit), C(k_bfe_u32),
#undef C
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-28s %12s %12s %12s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (const Case &c : cases) {
        printf("%-28s", c.name);
        for (int waves_per_simd : {1, 2, 4}) {
            const int threads = 64 * 4 * waves_per_simd; // one block = one CU, 4 SIMDs
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(c.fn, dim3(1), dim3(threads), 0, 0, out, 1.5);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            float base = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_empty, dim3(1), dim3(threads), 0, 0, out, 1.5);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < base) base = ms;
            }
            // SIMD cycles per wave-instruction: time x clock / (instructions issued per SIMD)
            const double per_simd = (double)ITERS * c.insts_per_iter * waves_per_simd;
            printf(" %12.2f", (best - base) * 1e-3 * clock_hz / per_simd);
        }
        printf("\n");
    }
    return 0;
}
