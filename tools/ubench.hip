// ubench.hip — VALU issue-rate microbenchmark for the instructions the trace
// kernel leans on (developer tool; run on the GPU box):
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench.hip && /tmp/ubench
// Every CU runs 8 waves per SIMD of 8 independent dependency chains, so the
// numbers are throughput (cycles per wave64 instruction per SIMD), not latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096

template <int OP> __global__ void k(uint32_t *out, uint32_t seed) {
    uint32_t a[8];
    double f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + threadIdx.x * 977u + i * 131u;
        f[i] = 1.0 + 1e-9 * a[i];
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) { // v_mad_u64_u32 (hi half feeds the chain)
                uint64_t p = (uint64_t)a[i] * 0xD2511F53u + seed;
                a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p;
            } else if (OP == 1) { // v_mul_hi_u32
                a[i] = __umulhi(a[i], 0xD2511F53u);
            } else if (OP == 2) { // v_mul_lo_u32
                a[i] = a[i] * 0xCD9E8D57u;
            } else if (OP == 3) { // v_mul_u32_u24
                a[i] = __umul24(a[i], 0x9E3779u) + 1u;
            } else if (OP == 4) { // v_bitop3 (xor3)
                a[i] = __builtin_amdgcn_bitop3_b32(a[i], seed, a[(i + 1) & 7], 0x96);
            } else if (OP == 5) { // v_fma_f64
                f[i] = fma(f[i], 1.0000001, 1e-7);
            } else if (OP == 6) { // v_cvt_f64_u32 + v_cvt_u32_f64 pair
                a[i] = (uint32_t)((double)a[i] * 0.999);
            } else if (OP == 7) { // v_rcp_f64
                f[i] = __builtin_amdgcn_rcp(f[i]) + 0.5;
            } else if (OP == 8) { // v_sqrt_f64
                f[i] = __builtin_amdgcn_sqrt(f[i]) + 1.0;
            } else if (OP == 9) { // v_xor_b32
                a[i] ^= a[(i + 3) & 7] + 1u;
            } else if (OP == 10) { // v_fma_f32
                float g = __uint_as_float(a[i]);
                g = fmaf(g, 1.0000001f, 1e-7f);
                a[i] = __float_as_uint(g);
            } else if (OP == 11) { // v_add_f64
                f[i] = f[i] + 1e-7;
            } else if (OP == 12) { // v_mul_f64
                f[i] = f[i] * 1.0000001;
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP> double run(const char *name, int insts_per_iter, uint32_t *buf, int cus, double ghz) {
    dim3 grid((unsigned)cus * 8), block(256); // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, buf, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, buf, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_insts_per_simd = 8.0 /*waves*/ * ITERS * 8.0 * insts_per_iter;
    double cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-28s %8.3f ms  -> %6.2f cycles per wave-instruction (assuming %d inst/op, %.2f GHz)\n", name, ms,
           cycles / wave_insts_per_simd, insts_per_iter, ghz);
    return ms;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    double ghz = khz / 1e6;
    uint32_t *buf;
    hipMalloc(&buf, (size_t)cus * 8 * 256 * 4);
    printf("CUs %d, clock %.2f GHz\n", cus, ghz);
    run<9>("v_xor_b32 (+add)", 2, buf, cus, ghz);
    run<4>("v_bitop3_b32", 1, buf, cus, ghz);
    run<10>("v_fma_f32", 1, buf, cus, ghz);
    run<3>("v_mul_u32_u24 (+add)", 2, buf, cus, ghz);
    run<2>("v_mul_lo_u32", 1, buf, cus, ghz);
    run<1>("v_mul_hi_u32", 1, buf, cus, ghz);
    run<0>("v_mad_u64_u32 (+xor)", 2, buf, cus, ghz);
    run<5>("v_fma_f64", 1, buf, cus, ghz);
    run<11>("v_add_f64", 1, buf, cus, ghz);
    run<12>("v_mul_f64", 1, buf, cus, ghz);
    run<6>("cvt u32->f64->u32 (+mul)", 3, buf, cus, ghz);
    run<7>("v_rcp_f64 (+add)", 2, buf, cus, ghz);
    run<8>("v_sqrt_f64 (+add)", 2, buf, cus, ghz);
    hipFree(buf);
    return 0;
}
