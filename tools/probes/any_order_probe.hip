// Does a kernel launched with hipExtAnyOrderLaunch start before the previous kernel of the same stream has ended on this device? (hip_ext.h says the flag "is not supported
// on AMD GFX9xx boards".) Two one-workgroup kernels that each spin for ~200 us: back to back on one stream they take 400 us unless the second may start early; the same
// pair on two streams is the control that does overlap.   hipcc --offload-arch=gfx950 -O2 -o any_order_probe any_order_probe.hip && ./any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin_kernel(unsigned long long clocks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64();
}

int main() {
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long clocks = (unsigned long long)rate_khz * 200 / 1000;  // 200 us
    unsigned long long* out = nullptr;
    hipMalloc((void**)&out, 64 * sizeof(unsigned long long));
    hipStream_t s0, s1;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    void* args[] = {(void*)&clocks, (void*)&out};
    auto run = [&](const char* label, int mode) {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipDeviceSynchronize();
            const auto a = std::chrono::steady_clock::now();
            for (int k = 0; k < 4; ++k) {
                if (mode == 0) hipLaunchKernel((const void*)spin_kernel, dim3(1), dim3(64), args, 0, s0);
                else if (mode == 1) hipExtLaunchKernel((const void*)spin_kernel, dim3(1), dim3(64), args, 0, s0, nullptr, nullptr, k == 0 ? 0 : hipExtAnyOrderLaunch);
                else hipLaunchKernel((const void*)spin_kernel, dim3(1), dim3(64), args, 0, (k & 1) ? s1 : s0);
            }
            hipDeviceSynchronize();
            best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
        }
        printf("%-64s four 200 us kernels in %7.1f us\n", label, best);
    };
    run("one stream, ordinary launches:", 0);
    run("one stream, hipExtAnyOrderLaunch on the second to fourth:", 1);
    run("two streams, alternating (two chains of two):", 2);
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
