// What does a resident frame's traffic cost over the host link, piece by piece? (frame_through_abi_ms 2.9 ms for 10.3 MB in + 15.4 MB out.)
// Host buffers are malloc'ed and hipHostRegister'ed, as the reference's BufferPool blocks are by the shim. Timed with HIP events on one stream, 20 repetitions each:
//   H2D  one hipMemcpyAsync of 10 MB | the same bytes as 8 and as 96 pieces | a kernel reading the registered host memory directly (zero copy) into HBM
//   D2H  one linear hipMemcpyAsync of 15.4 MB | hipMemcpy2DAsync 64 of every 128 bytes x 240,000 rows (what bepuhip_get_poses_and_velocities_async does) |
//        a kernel writing 64 of every 128 bytes straight into the registered host memory | a kernel packing into a device buffer + one linear copy
//   and the cost of an empty kernel launch / an empty event pair on this box, for scale.
// Developer probe, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o pcie_frame_probe.bin pcie_frame_probe.hip && ./pcie_frame_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel() {}
__global__ void copy_in_kernel(const float4* __restrict__ host, float4* __restrict__ dev, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dev[i] = host[i];
}
// 64 of every 128 bytes: lane quartets move one body's four float4
__global__ void poses_out_kernel(const float4* __restrict__ dev, float4* __restrict__ host, size_t bodies) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bodies * 4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t body = i >> 2, part = i & 3;
        host[body * 8 + part] = dev[body * 8 + part];
    }
}
__global__ void poses_pack_kernel(const float4* __restrict__ dev, float4* __restrict__ packed, size_t bodies) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bodies * 4; i += (size_t)gridDim.x * blockDim.x) packed[i] = dev[(i >> 2) * 8 + (i & 3)];
}

template <class F>
static float timed(hipStream_t s, int reps, F&& f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();  // untimed
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

int main() {
    const size_t in_bytes = 10300000 / 16 * 16, bodies = 240001, out_bytes = bodies * 64;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    char* h_in = (char*)aligned_alloc(4096, in_bytes + 4096);
    char* h_bodies = (char*)aligned_alloc(4096, bodies * 128 + 4096);
    memset(h_in, 1, in_bytes); memset(h_bodies, 2, bodies * 128);
    CHECK(hipHostRegister(h_in, in_bytes, hipHostRegisterDefault));
    CHECK(hipHostRegister(h_bodies, bodies * 128, hipHostRegisterDefault));
    void *m_in = nullptr, *m_bodies = nullptr;
    CHECK(hipHostGetDevicePointer(&m_in, h_in, 0));
    CHECK(hipHostGetDevicePointer(&m_bodies, h_bodies, 0));
    char *d_in, *d_bodies, *d_packed;
    CHECK(hipMalloc((void**)&d_in, in_bytes)); CHECK(hipMalloc((void**)&d_bodies, bodies * 128)); CHECK(hipMalloc((void**)&d_packed, out_bytes));
    CHECK(hipMemset(d_bodies, 3, bodies * 128));
    const int reps = 20;
    printf("empty kernel launch            %8.4f ms\n", timed(s, 200, [&] { empty_kernel<<<1, 64, 0, s>>>(); }));
    printf("H2D %5.1f MB, 1 hipMemcpyAsync   %8.4f ms\n", in_bytes / 1e6, timed(s, reps, [&] { hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s); }));
    for (int pieces : {8, 96}) {
        const size_t piece = in_bytes / pieces / 16 * 16;
        printf("H2D the same as %3d pieces      %8.4f ms\n", pieces, timed(s, reps, [&] { for (int p = 0; p < pieces; ++p) hipMemcpyAsync(d_in + p * piece, h_in + p * piece, piece, hipMemcpyHostToDevice, s); }));
    }
    for (int blocks : {256, 1024, 4096})
        printf("H2D zero-copy kernel, %4d wg   %8.4f ms\n", blocks, timed(s, reps, [&] { copy_in_kernel<<<blocks, 256, 0, s>>>((const float4*)m_in, (float4*)d_in, in_bytes / 16); }));
    printf("D2H %5.1f MB linear             %8.4f ms\n", out_bytes / 1e6, timed(s, reps, [&] { hipMemcpyAsync(h_bodies, d_packed, out_bytes, hipMemcpyDeviceToHost, s); }));
    printf("D2H 2D 64 of 128 x %zu rows  %8.4f ms\n", bodies, timed(s, reps, [&] { hipMemcpy2DAsync(h_bodies, 128, d_bodies, 128, 64, bodies, hipMemcpyDeviceToHost, s); }));
    printf("D2H whole records %5.1f MB      %8.4f ms\n", bodies * 128 / 1e6, timed(s, reps, [&] { hipMemcpyAsync(h_bodies, d_bodies, bodies * 128, hipMemcpyDeviceToHost, s); }));
    for (int blocks : {256, 1024, 4096})
        printf("D2H zero-copy kernel, %4d wg   %8.4f ms\n", blocks, timed(s, reps, [&] { poses_out_kernel<<<blocks, 256, 0, s>>>((const float4*)d_bodies, (float4*)m_bodies, bodies); }));
    printf("D2H pack kernel + linear copy   %8.4f ms\n", timed(s, reps, [&] { poses_pack_kernel<<<1024, 256, 0, s>>>((const float4*)d_bodies, (float4*)d_packed, bodies); hipMemcpyAsync(h_bodies, d_packed, out_bytes, hipMemcpyDeviceToHost, s); }));
    // in and out at once on two streams (full duplex?)
    hipStream_t s2;
    CHECK(hipStreamCreate(&s2));
    printf("H2D 10 MB (stream 1) || D2H 15 MB linear (stream 2): %8.4f ms\n", timed(s, reps, [&] {
        hipMemcpyAsync(h_bodies, d_packed, out_bytes, hipMemcpyDeviceToHost, s2);
        hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s2);
    }));
    for (size_t i = 0; i < 4; ++i) if (h_bodies[i] != 3) { printf("zero-copy write did not land\n"); break; }
    CHECK(hipStreamSynchronize(s));
    printf("done\n");
    return 0;
}
