// lds_fit.hip — how many 256-thread blocks does a CU of this GPU REALLY hold at once for a given LDS size per block?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor divides 160 KB by the request; the hardware allocates in granules.)
// Every block notes the wall clock when it starts, spins for ~1 ms and notes when it ends; blocks that started before the
// first block ended were resident together.  Build: hipcc --offload-arch=gfx950 -O2 -o build/lds_fit tools/microbench/lds_fit.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k(unsigned long long *start, unsigned long long *end, int spin_ticks) {
    extern __shared__ unsigned char lds[];
    lds[threadIdx.x] = (unsigned char)threadIdx.x; // the allocation is used
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {}
    if (threadIdx.x == 0) {
        start[blockIdx.x] = t0;
        end[blockIdx.x] = wall_clock64() + lds[17];
    }
}
int main(int argc, char **argv) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int per_cu = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = cus * per_cu;
    unsigned long long *ds, *de;
    hipMalloc(&ds, blocks * 8);
    hipMalloc(&de, blocks * 8);
    std::vector<unsigned long long> s(blocks), e(blocks);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int last = -1;
    for (int bytes = 16 * 1024; bytes <= 64 * 1024; bytes += 128) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), bytes, 0, ds, de, 100000 /* 1 ms at 100 MHz */);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%d bytes: launch failed\n", bytes); break; }
        hipMemcpy(s.data(), ds, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(e.data(), de, blocks * 8, hipMemcpyDeviceToHost);
        const unsigned long long first_end = *std::min_element(e.begin(), e.end());
        int together = 0;
        for (int b = 0; b < blocks; ++b) together += s[b] < first_end;
        int api = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k, 256, bytes);
        const int now = together * 100 / cus;
        if (now != last) printf("%6d bytes of LDS per block: %.2f blocks per CU resident together (occupancy API: %d)\n", bytes, together / (double)cus, api);
        last = now;
    }
    return 0;
}
