// hipStreamWaitValue32 on signal memory as "the previous launch is fully resident": kernel A (stream 0, WG workgroups that each add one to a counter when they START, then
// spin 200 us); stream 1: wait until the counter shows all of A's workgroups, then kernel B, which notes when it started. B must start after A's last workgroup started and
// long before A ends.   hipcc --offload-arch=gfx950 -O2 -o wait_value_probe.bin wait_value_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void kernel_a(unsigned long long clocks, unsigned* counter, unsigned long long* starts, unsigned long long* ends) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { starts[blockIdx.x] = t0; __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) ends[blockIdx.x] = wall_clock64();
}
__global__ void kernel_b(unsigned long long* start) { if (threadIdx.x == 0) start[0] = wall_clock64(); }

int main() {
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long clocks = (unsigned long long)rate_khz * 200 / 1000;
    const int WG = 250;
    unsigned* counter = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&counter, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipMemset(counter, 0, 8);
    unsigned long long *starts, *ends, *bstart;
    hipMalloc((void**)&starts, WG * 8); hipMalloc((void**)&ends, WG * 8); hipMalloc((void**)&bstart, 8);
    hipStream_t s0, s1;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    for (int rep = 0; rep < 4; ++rep) {
        hipDeviceSynchronize();
        const unsigned target = (unsigned)WG * (unsigned)(rep + 1);
        hipLaunchKernelGGL(kernel_a, dim3(WG), dim3(1024), 150 * 1024, s0, clocks, counter, starts, ends);
        e = hipStreamWaitValue32(s1, counter, target, hipStreamWaitValueGte, 0xFFFFFFFFu);
        hipLaunchKernelGGL(kernel_b, dim3(1), dim3(64), 0, s1, bstart);
        const auto a = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        std::vector<unsigned long long> hs(WG), he(WG); unsigned long long hb = 0;
        hipMemcpy(hs.data(), starts, WG * 8, hipMemcpyDeviceToHost); hipMemcpy(he.data(), ends, WG * 8, hipMemcpyDeviceToHost); hipMemcpy(&hb, bstart, 8, hipMemcpyDeviceToHost);
        const unsigned long long first = *std::min_element(hs.begin(), hs.end()), last_start = *std::max_element(hs.begin(), hs.end()), last_end = *std::max_element(he.begin(), he.end());
        const double us = 1e3 / rate_khz;
        printf("rep %d: wait-value %s | A's workgroups start over %.1f us, A ends at %.1f us | B starts at %.1f us (%.1f us after A's last workgroup started)\n", rep, hipGetErrorString(e),
               (last_start - first) * us, (last_end - first) * us, ((double)hb - (double)first) * us, ((double)hb - (double)last_start) * us);
    }
    return 0;
}
