// Measures what a device-wide barrier inside one persistent kernel costs on MI355X (8 XCDs, one L2 each): the question behind a
// "whole step in one launch" variant of the launch-per-batch schedule. Developer probe, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// One arrive per workgroup, everybody spins on the counter; `work` floats per thread are written before and read (from another workgroup's
// range) after each barrier so that the release/acquire cache maintenance has something to do, as velocities would.
template <bool FENCES>
__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, float* data, int rounds, int work, unsigned long long* cycles, unsigned* bad) {
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    float acc = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < work; ++k) data[((size_t)wg * work + k) * 256 + tid] = (float)(r + 1);
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, FENCES ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nwg;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { atomicAdd(bad, 1u); break; }
            }
        }
        __syncthreads();
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int other = (wg + nwg / 2 + 1) % nwg;  // a workgroup on (most likely) another XCD
        for (int k = 0; k < work; ++k) {
            const float v = data[((size_t)other * work + k) * 256 + tid];
            if (FENCES && v != (float)(r + 1)) atomicAdd(bad + 1, 1u);
            acc += v;
        }
    }
    if (tid == 0) cycles[wg] = __builtin_readcyclecounter() - t0;
    if (acc == -1.0f) data[0] = acc;
}

template <bool FENCES>
static int run(int nwg, int rounds, int work) {
    unsigned *counter, *bad; float* data; unsigned long long* cycles;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&data, (size_t)nwg * work * 256 * 4 + 1024)); CHECK(hipMalloc(&cycles, nwg * 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9f;
    unsigned hbad[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(bad, 0, 8));
        void* args[] = {&counter, &data, &rounds, &work, &cycles, &bad};
        CHECK(hipEventRecord(a));
        CHECK(hipLaunchCooperativeKernel((const void*)barrier_loop<FENCES>, dim3(nwg), dim3(256), args, 0, 0));
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        CHECK(hipMemcpy(hbad, bad, 8, hipMemcpyDeviceToHost));
    }
    printf("fences=%d workgroups=%4d work=%2d floats/thread: %.2f us per round (%d rounds)  spin-timeouts=%u stale-reads=%u\n", (int)FENCES, nwg, work, best * 1e3f / rounds, rounds, hbad[0], hbad[1]);
    hipFree(counter); hipFree(bad); hipFree(data); hipFree(cycles);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, cooperative launch %d\n", p.name, p.multiProcessorCount, p.cooperativeLaunch);
    for (int nwg : {64, 256, 512})
        for (int work : {0, 6}) {
            if (run<false>(nwg, 2000, work)) return 1;
            if (run<true>(nwg, 2000, work)) return 1;
        }
    return 0;
}
