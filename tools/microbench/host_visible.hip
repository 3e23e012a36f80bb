// Are a running kernel's stores to pinned host memory visible to the host BEFORE the kernel ends, and in order
// behind a system-scope fence?  (The tile stream of rt_render relies on it: waves write finished pixels into
// the caller-visible pinned buffer and then raise a flag the host polls.)
// Build: hipcc --offload-arch=gfx950 -O2 -o build/host_visible tools/microbench/host_visible.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void writer(double *host_data, volatile unsigned *host_flags, unsigned *dev_count, int n_regions, int per_region,
                       long long spin_ticks) {
    // block b belongs to region b % n_regions; writes 256 doubles, fences, counts; the last block of a region raises its flag
    const int region = blockIdx.x % n_regions;
    // stagger: region r waits r * spin_ticks of the 100 MHz wall clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks * region) {}
    host_data[(size_t)blockIdx.x * 256 + threadIdx.x] = (double)blockIdx.x + 0.5;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(&dev_count[region], 1u);
        if (old == (unsigned)per_region - 1) {
            __threadfence_system();
            host_flags[region] = 1u;
        }
    }
    // keep the kernel alive until long after the last region
    while (wall_clock64() - t0 < spin_ticks * (n_regions + 4)) {}
}

int main() {
    const int n_regions = 10, per_region = 200, blocks = n_regions * per_region;
    double *data;
    unsigned *flags, *count;
    CK(hipHostMalloc((void **)&data, (size_t)blocks * 256 * sizeof(double), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
    CK(hipHostMalloc((void **)&flags, n_regions * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
    CK(hipMalloc((void **)&count, n_regions * sizeof(unsigned)));
    CK(hipMemset(count, 0, n_regions * sizeof(unsigned)));
    for (int i = 0; i < n_regions; ++i) flags[i] = 0;
    for (size_t i = 0; i < (size_t)blocks * 256; ++i) data[i] = -1.0;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const long long ticks = 200000; // 2 ms per region at 100 MHz
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(writer, dim3(blocks), dim3(256), 0, s, data, flags, count, n_regions, per_region, ticks);
    CK(hipGetLastError());
    int bad = 0;
    for (int r = 0; r < n_regions; ++r) {
        while (((volatile unsigned *)flags)[r] == 0) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) { printf("timeout waiting for region %d\n", r); return 2; }
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const bool running = hipStreamQuery(s) == hipErrorNotReady;
        (void)hipGetLastError();
        int wrong = 0;
        for (int b = r; b < blocks; b += n_regions)
            for (int t = 0; t < 256; ++t) wrong += data[(size_t)b * 256 + t] != (double)b + 0.5;
        printf("region %d flag seen at %.2f ms, kernel still running: %d, wrong values: %d\n", r, ms, (int)running, wrong);
        bad += wrong;
    }
    CK(hipStreamSynchronize(s));
    printf("kernel done at %.2f ms; %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
           bad ? "STALE DATA SEEN" : "all data visible behind its flag");
    return bad ? 3 : 0;
}
