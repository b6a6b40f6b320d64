// What the fixed time between two dependent launches of one stream depends on: an empty kernel (and one that spins 50 us in every workgroup) launched 300 times back to back
// for several grid shapes; the period minus the kernel's own time is the launch-to-launch cost.   hipcc --offload-arch=gfx950 -O2 -o launch_gap_probe.bin launch_gap_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

struct Big { float v[120]; };
__global__ void probe_kernel(unsigned long long clocks, int spin_blocks, unsigned* sink, Big big) {
    extern __shared__ float lds[];
    if (clocks && (int)blockIdx.x < spin_blocks) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(16);
    }
    if (big.v[5] == 123.0f) { lds[threadIdx.x] = 1.0f; sink[0] = (unsigned)lds[0]; }
}

int main() {
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    unsigned* sink; hipMalloc((void**)&sink, 64);
    hipStream_t s; hipStreamCreate(&s);
    hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big big{};
    struct Shape { const char* label; int grid, block; size_t lds; int spin_blocks; } shapes[] = {
        {"1 x 64, no LDS", 1, 64, 0, 1},
        {"250 x 1024, no LDS", 250, 1024, 0, 250},
        {"250 x 1024, 150 KB LDS", 250, 1024, 150 * 1024, 250},
        {"486 x 1024, 150 KB LDS (236 of them exit at once)", 486, 1024, 150 * 1024, 250},
        {"256 x 1024, 150 KB LDS (6 exit at once)", 256, 1024, 150 * 1024, 250},
        {"250 x 512, 150 KB LDS", 250, 512, 150 * 1024, 250},
        {"250 x 1024, 64 KB LDS", 250, 1024, 64 * 1024, 250},
    };
    for (const Shape& sh : shapes)
        for (int spin_us : {0, 50}) {
            const unsigned long long clocks = (unsigned long long)rate_khz * spin_us / 1000;
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(probe_kernel, dim3(sh.grid), dim3(sh.block), sh.lds, s, clocks, sh.spin_blocks, sink, big);
                hipStreamSynchronize(s);
                const auto a = std::chrono::steady_clock::now();
                for (int k = 0; k < 300; ++k) hipLaunchKernelGGL(probe_kernel, dim3(sh.grid), dim3(sh.block), sh.lds, s, clocks, sh.spin_blocks, sink, big);
                hipStreamSynchronize(s);
                best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / 300.0);
            }
            printf("%-52s kernel body %2d us: period %6.2f us -> launch-to-launch cost %6.2f us\n", sh.label, spin_us, best, best - spin_us);
        }
    // ... and what timing every launch costs: an event recorded in front of and behind it on the stream (two marker packets), or the same two events handed to
    // hipExtLaunchKernel (they ride on the kernel's own completion signal)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned long long clocks50 = (unsigned long long)rate_khz * 50 / 1000;
    int spin_blocks = 250;
    void* args[] = {(void*)&clocks50, (void*)&spin_blocks, (void*)&sink, (void*)&big};
    for (int mode = 0; mode < 4; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            const auto a = std::chrono::steady_clock::now();
            for (int k = 0; k < 300; ++k) {
                if (mode == 1 || mode == 2) hipEventRecord(e0, s);
                if (mode == 3) hipExtLaunchKernel((const void*)probe_kernel, dim3(486), dim3(1024), args, 150 * 1024, s, e0, e1, 0);
                else hipLaunchKernel((const void*)probe_kernel, dim3(486), dim3(1024), args, 150 * 1024, s);
                if (mode == 2) hipEventRecord(e1, s);
            }
            hipStreamSynchronize(s);
            best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / 300.0);
        }
        float ms = 0; if (mode >= 2) hipEventElapsedTime(&ms, e0, e1);
        const char* label[] = {"plain launches", "hipEventRecord in front of every launch", "hipEventRecord in front of and behind every launch", "hipExtLaunchKernel(start event, stop event)"};
        printf("486 x 1024, 150 KB LDS, 50 us body, %-52s period %6.2f us -> launch-to-launch cost %6.2f us (events say %.1f us)\n", label[mode], best, best - 50, ms * 1e3);
    }
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
