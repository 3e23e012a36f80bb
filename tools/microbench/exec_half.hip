// exec_half.hip — does a CDNA4 SIMD skip the half of a wave64 instruction whose 32 lanes are all inactive?
// (If it does, compacting a wave's active lanes into one half would halve the cost of divergent code such as the
// BVH walk, which runs at ~21 of 64 lanes.)  Loops of independent v_fma_f64 / v_fma_f32 under different exec masks,
// 4 waves per SIMD on one CU; prints SIMD cycles per wave-instruction (shader clock from the runtime).
//   hipcc --offload-arch=gfx950 -O2 -o build/exec_half tools/microbench/exec_half.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITERS = 16384;

template <int MODE, bool F64> __global__ void k(double *out, double seed, unsigned long long *ticks) {
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (MODE == 1) on = lane < 32;          // low half
    if (MODE == 2) on = lane >= 32;         // high half
    if (MODE == 3) on = (lane & 1) == 0;    // every other lane
    if (MODE == 4) on = lane == 5;          // one lane
    if (MODE == 5) on = lane < 16;          // a quarter
    if (MODE == 6) on = lane < 21 ;         // 21 lanes, compacted
    if (MODE == 7) on = (lane % 3) == 0;    // 22 lanes, spread
    double a[16];
    float f[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; }
    const double b = seed, c = seed * 0.5;
    const float fb = (float)seed, fc = fb * 0.5f;
    const unsigned long long t0 = wall_clock64();
    if (on) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fb), "v"(fc));
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, bool F64> void run(const char *name, double clock_mhz) {
    double *out;
    unsigned long long *ticks;
    CHECK(hipMalloc((void **)&out, 1024 * sizeof(double)));
    CHECK(hipMalloc((void **)&ticks, sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) { // one block of 1024 threads = 16 waves on one CU = 4 per SIMD
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, F64>), dim3(1), dim3(1024), 0, 0, out, 1.5, ticks);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-34s %s: %.2f SIMD cycles per wave-instruction (4 waves per SIMD, launch overhead included)\n", name,
           F64 ? "v_fma_f64" : "v_fma_f32", best * 1e-3 * clock_mhz * 1e6 / (ITERS * 16.0 * 4.0));
    CHECK(hipFree(out));
    CHECK(hipFree(ticks));
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double mhz = p.clockRate / 1000.0;
    printf("device %s, shader clock %.0f MHz\n", p.gcnArchName, mhz);
#define BOTH(M, N) run<M, true>(N, mhz); run<M, false>(N, mhz);
    BOTH(0, "all 64 lanes")
    BOTH(1, "lanes 0-31 (low half)")
    BOTH(2, "lanes 32-63 (high half)")
    BOTH(3, "even lanes (32, both halves)")
    BOTH(5, "lanes 0-15")
    BOTH(6, "lanes 0-20 (21 lanes, compacted)")
    BOTH(7, "every third lane (22, spread)")
    BOTH(4, "one lane")
    return 0;
}
