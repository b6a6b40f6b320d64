// Does instruction fetch cost the same on every box of the pool? 64 distinct straight-line code bodies of ~16 KiB each (1 MiB of code, far beyond the 64 KiB
// instruction cache two CUs share); every wave walks through them starting at a different one, so the CU's waves keep missing, as cluster_kernel's 16 waves in
// 16 different constraint types do. Reported: s_memtime cycles per FMA executed (1.0x = cache-resident loop of the same instructions).
// Build: hipcc --offload-arch=gfx950 -O1 -o icache_probe.bin icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__device__ __noinline__ float body(float a, float b) {
#pragma unroll
    for (int i = 0; i < 1000; ++i) { a = __builtin_fmaf(a, b, (float)(K + 1) * 0.001f + (float)i); }  // distinct immediates: no folding across bodies
    return a;
}
#define CASE(K) case K: a = body<K>(a, b); break;
__device__ float dispatch(int k, float a, float b) {
    switch (k & 63) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
        CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29) CASE(30) CASE(31)
        CASE(32) CASE(33) CASE(34) CASE(35) CASE(36) CASE(37) CASE(38) CASE(39) CASE(40) CASE(41) CASE(42) CASE(43) CASE(44) CASE(45) CASE(46) CASE(47)
        CASE(48) CASE(49) CASE(50) CASE(51) CASE(52) CASE(53) CASE(54) CASE(55) CASE(56) CASE(57) CASE(58) CASE(59) CASE(60) CASE(61) CASE(62) CASE(63)
    }
    return a;
}
__global__ void walk(float* out, int rounds, int spread, unsigned long long* cycles) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) a = dispatch(spread ? wave * 4 + r : 0, a, b);  // spread 0: every wave runs body 0 again and again (cache-resident)
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spread : {0, 1, 0, 1}) {
        const int rounds = 200;
        hipEventRecord(e0); hipLaunchKernelGGL(walk, dim3(256), dim3(1024), 0, 0, out, rounds, spread, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double c = 0; for (auto v : h) c += v; c /= 256;
        printf("%s: wall %7.3f ms, %.2f cycles per FMA per wave (16 waves/CU)\n", spread ? "64 bodies, waves spread over them" : "one body, cache-resident      ", ms, c / (rounds * 1000.0));
    }
    return 0;
}
