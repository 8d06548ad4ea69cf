// Throughput of returning device atomics in the binning access pattern: 16k waves, ~6 distinct counters per wave.
// mode 0: one atomic per lane (no aggregation); mode 1: one atomic per (wave, distinct counter) group leader.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int WIDTH>
__global__ __launch_bounds__(256) void k(unsigned long long* fill, unsigned* entries, unsigned qcap, unsigned nbins) {
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned bin = ((wave >> 1) * 3 + lane / 11) % nbins;  // 6 distinct counters per wave, neighbours overlap
    unsigned off;
    if (MODE == 0) {
        if (WIDTH == 64) off = (unsigned)atomicAdd(&fill[bin], 1ull | (1ull << 32));
        else off = atomicAdd(reinterpret_cast<unsigned*>(&fill[bin]), 1u);
    } else {
        const unsigned first = (lane / 11) * 11;  // group leader
        unsigned long long t = 0;
        const unsigned cnt = min(11u, 64u - first);
        if (lane == first) t = atomicAdd(&fill[bin], (unsigned long long)cnt | (1ull << 32));
        off = (unsigned)__shfl((int)(unsigned)t, first) + (lane - first);
    }
    if (off < qcap) entries[(size_t)bin * qcap + off] = wave * 64 + lane;
}

int main() {
    const unsigned nbins = 8160, qcap = 2048, nwaves = 16384;
    unsigned long long* fill; unsigned* entries;
    CK(hipMalloc(&fill, nbins * 8)); CK(hipMalloc(&entries, (size_t)nbins * qcap * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9;
        for (int it = 0; it < 6; it++) {
            CK(hipMemset(fill, 0, nbins * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL((k<0, 64>), dim3(nwaves / 4), dim3(256), 0, 0, fill, entries, qcap, nbins);
            if (mode == 1) hipLaunchKernelGGL((k<1, 64>), dim3(nwaves / 4), dim3(256), 0, 0, fill, entries, qcap, nbins);
            if (mode == 2) hipLaunchKernelGGL((k<0, 32>), dim3(nwaves / 4), dim3(256), 0, 0, fill, entries, qcap, nbins);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it && ms < best) best = ms;
        }
        printf("mode %d (%s): %.1f us for %u lanes\n", mode, mode == 0 ? "per-lane u64" : mode == 1 ? "per-group u64" : "per-lane u32", best * 1e3, nwaves * 64);
    }
    return 0;
}
