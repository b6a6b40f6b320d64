// Which clock do the pool's two box classes differ in? A fixed amount of dependent ALU work per wave, at 1 and 16 waves per CU, timed three ways:
// wall clock (HIP events), s_memtime (what __builtin_readcyclecounter reads, what bepuhip_get_cluster_cycles reports) and s_memrealtime (constant 100 MHz).
// If the same work takes more wall time on one class while s_memtime / wall stays the same, s_memtime is not the clock the ALUs run at there.
// A second kernel does dependent global loads (pointer chase through 256 MiB) to read the memory latency the same way.
// Build: hipcc --offload-arch=gfx950 -O2 -o clock_probe.bin clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void alu_kernel(float* out, int iters, unsigned long long* stamps) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; a = a * b - 0.5f; a = a * b + 0.25f; a = a * b - 0.25f; }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
__global__ void chase_kernel(const unsigned* next, unsigned n, int hops, unsigned* out, unsigned long long* stamps) {
    unsigned p = (blockIdx.x * 977u) % n;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < hops; ++i) p = next[p];
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x] = p;
    stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0;
}
__global__ void stream_kernel(const float4* __restrict__ src, size_t n, float* out) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void lds_kernel(float* out, int iters) {
    __shared__ float4 tile[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tile[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0; int idx = threadIdx.x;
    for (int i = 0; i < iters; ++i) { const float4 v = tile[idx & 4095]; acc += v.x; idx = idx * 5 + 1 + (int)v.y; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    float* out; unsigned long long* stamps;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&stamps, 4096 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int threads : {64, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(alu_kernel, dim3(256), dim3(threads), 0, 0, out, iters, stamps); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(512); hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
            cyc /= 256; rt /= 256;
            printf("alu  %4d threads/CU: wall %8.3f ms, s_memtime %10.0f cycles (%.3f GHz vs wall), s_memrealtime %8.0f ticks (%.1f MHz), %.2f cycles per dependent FMA\n", threads, ms, cyc,
                   cyc / (ms * 1e6), rt, rt / (ms * 1e3), cyc / (4.0 * iters));
        }
    }
    for (size_t links : {(size_t)64 << 10, (size_t)256 << 10, (size_t)4 << 20, (size_t)64 << 20}) {  // 256 KiB (L2), 1 MiB (L2), 16 MiB (MALL), 256 MiB (MALL / HBM)
        const size_t n = links;
        std::vector<unsigned> perm(n); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(5); for (size_t i = n - 1; i > 0; --i) std::swap(perm[i], perm[rng() % (i + 1)]);
        std::vector<unsigned> next(n); for (size_t i = 0; i + 1 < n; ++i) next[perm[i]] = perm[i + 1]; next[perm[n - 1]] = perm[0];
        unsigned* d_next; unsigned* d_out; hipMalloc(&d_next, n * 4); hipMalloc(&d_out, 4096 * 4); hipMemcpy(d_next, next.data(), n * 4, hipMemcpyHostToDevice);
        for (int blocks : {1, 256}) {
            hipLaunchKernelGGL(chase_kernel, dim3(blocks), dim3(1), 0, 0, d_next, (unsigned)n, 20000, d_out, stamps);  // warm
            hipEventRecord(e0); hipLaunchKernelGGL(chase_kernel, dim3(blocks), dim3(1), 0, 0, d_next, (unsigned)n, 20000, d_out, stamps); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), stamps, 2 * blocks * 8, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
            printf("chase %6zu KiB, %4d chains: %.0f s_memtime cycles per hop, %.0f ns per hop\n", n * 4 / 1024, blocks, cyc / blocks / 20000, rt / blocks / 20000 * 10.0);
        }
        hipFree(d_next); hipFree(d_out);
    }
    for (size_t mib : {128, 1024, 8192}) {
        const size_t n = mib * (1u << 20) / 16;
        float4* buf; float4* dst; if (hipMalloc(&buf, n * 16) != hipSuccess || hipMalloc(&dst, n * 16) != hipSuccess) { printf("alloc %zu MiB failed\n", mib); break; }
        hipMemset(buf, 0, n * 16);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, 0, buf, n, out);
        hipEventRecord(e0); for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, 0, buf, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read  %5zu MiB: %.2f TB/s\n", mib, 4.0 * n * 16 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, buf, dst, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("copy  %5zu MiB: %.2f TB/s (read + write)\n", mib, 8.0 * n * 16 / (ms * 1e-3) / 1e12);
        hipFree(buf); hipFree(dst);
    }
    hipEventRecord(e0); hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(1024), 0, 0, out, 100000); hipEventRecord(e1); hipEventSynchronize(e1);
    { float ms; hipEventElapsedTime(&ms, e0, e1); printf("lds   dependent 16-byte reads, 16 waves/CU: %.3f ms for 100000 per lane\n", ms); }
    return 0;
}
