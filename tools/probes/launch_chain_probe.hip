// What does ONE launch of the launch-per-batch schedule cost on MI355X, and what would a "push" layout buy? Developer probe, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o launch_chain_probe.bin launch_chain_probe.hip && ./launch_chain_probe.bin
// A hipGraph of 85 dependent kernel nodes (the pile's launch count per frame), ~49k lanes each (one 64-lane workgroup per 64 constraints), for:
//   empty      nothing but the launch
//   stream     ONE memory round trip: coalesced rows in (35 floats per lane), 7 floats out
//   gather     TWO dependent round trips: rows + 2 body references in, then 2 x 64 B gathered from a 128-byte body record, 2 x 32 B scattered back  (today's batch_kernel)
//   push       ONE round trip: rows + velocities/inertia delivered into the constraint's own rows by its predecessors (61 floats per lane in), 2 x 24 B scattered to
//              the successors' rows (the layout this probe is meant to price)
// each with and without a serial chain of `alu` dependent float operations in between (a Contact4 solve is ~800 wave64 VALU instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kRows = 35;      // Contact4: 26 prestep + 7 accumulated + 2 references
constexpr int kPushRows = 26;  // 2 x (6 velocity + 7 inertia) delivered by predecessors / the integration pass

__device__ __forceinline__ float chain(float x, int alu) {
    if (alu < 0) alu = alu == -1 ? 0 : -alu;
#pragma unroll 8
    for (int k = 0; k < alu; ++k) x = x * 1.0000001f + 0.25f;  // dependent, not contracted (-ffp-contract=off)
    return x;
}

__global__ __launch_bounds__(64) void k_empty(int n) {}

__global__ __launch_bounds__(64) void k_stream(const float* __restrict__ rows, float* __restrict__ out, int n, int stride, int alu) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float s = 0;
#pragma unroll
    for (int f = 0; f < kRows; ++f) s += rows[(size_t)f * stride + i];
    s = chain(s, alu);
#pragma unroll
    for (int f = 0; f < 7; ++f) out[(size_t)f * stride + i] = s + f;
}

__global__ __launch_bounds__(64) void k_gather(const float* __restrict__ rows, const int* __restrict__ refs, float4* bodies, float* __restrict__ out, int n, int stride, int alu) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int ra = refs[i], rb = refs[stride + i];
    float s = 0;
#pragma unroll
    for (int f = 0; f < kRows - 2; ++f) s += rows[(size_t)f * stride + i];
    float4* A = bodies + (size_t)ra * 8;
    float4* B = bodies + (size_t)rb * 8;
    float4 a2 = A[2], a3 = A[3], a6 = A[6], a7 = A[7], b2 = B[2], b3 = B[3], b6 = B[6], b7 = B[7];
    s += a2.x + a3.y + a6.z + a7.w + b2.x + b3.y + b6.z + b7.w;
    s = chain(s, alu);
#pragma unroll
    for (int f = 0; f < 7; ++f) out[(size_t)f * stride + i] = s + f;
    if (alu >= 0) {
        A[2] = make_float4(s, a2.y, a2.z, a2.w); A[3] = make_float4(s, a3.y, a3.z, a3.w);
        B[2] = make_float4(s, b2.y, b2.z, b2.w); B[3] = make_float4(s, b3.y, b3.z, b3.w);
    }
}

// gather, velocities written back with non-temporal (MODE 1) or write-through system-scope (MODE 2) stores; rows read non-temporally in MODE 3
template <int MODE>
__global__ __launch_bounds__(64) void k_gather_st(const float* __restrict__ rows, const int* __restrict__ refs, float4* bodies, float* __restrict__ out, int n, int stride, int alu) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int ra = refs[i], rb = refs[stride + i];
    float s = 0;
#pragma unroll
    for (int f = 0; f < kRows - 2; ++f) s += MODE == 3 ? __builtin_nontemporal_load(&rows[(size_t)f * stride + i]) : rows[(size_t)f * stride + i];
    float4* A = bodies + (size_t)ra * 8;
    float4* B = bodies + (size_t)rb * 8;
    float4 a2 = A[2], a3 = A[3], a6 = A[6], a7 = A[7], b2 = B[2], b3 = B[3], b6 = B[6], b7 = B[7];
    s += a2.x + a3.y + a6.z + a7.w + b2.x + b3.y + b6.z + b7.w;
    s = chain(s, alu);
#pragma unroll
    for (int f = 0; f < 7; ++f) {
        if (MODE == 3) __builtin_nontemporal_store(s + f, &out[(size_t)f * stride + i]); else out[(size_t)f * stride + i] = s + f;
    }
    const float4 va = make_float4(s, a2.y, a2.z, a2.w), wa = make_float4(s, a3.y, a3.z, a3.w), vb = make_float4(s, b2.y, b2.z, b2.w), wb = make_float4(s, b3.y, b3.z, b3.w);
    if (MODE == 1 || MODE == 3) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4{va.x, va.y, va.z, va.w}, (v4*)&A[2]); __builtin_nontemporal_store(v4{wa.x, wa.y, wa.z, wa.w}, (v4*)&A[3]);
        __builtin_nontemporal_store(v4{vb.x, vb.y, vb.z, vb.w}, (v4*)&B[2]); __builtin_nontemporal_store(v4{wb.x, wb.y, wb.z, wb.w}, (v4*)&B[3]);
    } else {
        { typedef float v4 __attribute__((ext_vector_type(4))); v4 t = {va.x, va.y, va.z, va.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&A[2]), "v"(t) : "memory"); }
        { typedef float v4 __attribute__((ext_vector_type(4))); v4 t = {wa.x, wa.y, wa.z, wa.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&A[3]), "v"(t) : "memory"); }
        { typedef float v4 __attribute__((ext_vector_type(4))); v4 t = {vb.x, vb.y, vb.z, vb.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&B[2]), "v"(t) : "memory"); }
        { typedef float v4 __attribute__((ext_vector_type(4))); v4 t = {wb.x, wb.y, wb.z, wb.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&B[3]), "v"(t) : "memory"); }
    }
}

// gather with a compact 64-byte solver record per body: [lin.xyz ang.x][ang.yz i0 i1][i2 i3 i4 i5][i6 - - -]; velocity written back as the first two float4
__global__ __launch_bounds__(64) void k_gather64(const float* __restrict__ rows, const int* __restrict__ refs, float4* bodies, float* __restrict__ out, int n, int stride, int alu, int write) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int ra = refs[i], rb = refs[stride + i];
    float s = 0;
#pragma unroll
    for (int f = 0; f < kRows - 2; ++f) s += rows[(size_t)f * stride + i];
    float4* A = bodies + (size_t)ra * 4;
    float4* B = bodies + (size_t)rb * 4;
    float4 a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], b0 = B[0], b1 = B[1], b2 = B[2], b3 = B[3];
    s += a0.x + a1.y + a2.z + a3.w + b0.x + b1.y + b2.z + b3.w;
    s = chain(s, alu);
#pragma unroll
    for (int f = 0; f < 7; ++f) out[(size_t)f * stride + i] = s + f;
    if (write) {
        A[0] = make_float4(s, a0.y, a0.z, a0.w); A[1] = make_float4(s, a1.y, a1.z, a1.w);
        B[0] = make_float4(s, b0.y, b0.z, b0.w); B[1] = make_float4(s, b1.y, b1.z, b1.w);
    }
}

__global__ __launch_bounds__(64) void k_push(const float* __restrict__ rows, const float* __restrict__ inbox, const int* __restrict__ succ, float* next_inbox, float* __restrict__ out,
                                              int n, int stride, int alu) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int sa = succ[i], sb = succ[stride + i];  // row index (in the NEXT batch's inbox) of the constraints that touch my bodies next
    float s = 0;
#pragma unroll
    for (int f = 0; f < kRows - 2; ++f) s += rows[(size_t)f * stride + i];
#pragma unroll
    for (int f = 0; f < kPushRows; ++f) s += inbox[(size_t)f * stride + i];
    s = chain(s, alu);
#pragma unroll
    for (int f = 0; f < 7; ++f) out[(size_t)f * stride + i] = s + f;
#pragma unroll
    for (int f = 0; f < 6; ++f) { next_inbox[(size_t)f * stride + sa] = s + f; next_inbox[(size_t)(13 + f) * stride + sb] = s - f; }
}

int main() {
    const int batches = 6, per_batch = 49285, launches = 85;
    const int stride = ((per_batch + 63) / 64) * 64, blocks = stride / 64, bodies_n = 100000;
    float *rows, *out, *inbox; int *refs, *succ; float4* bodies;
    CHECK(hipMalloc(&rows, (size_t)batches * kRows * stride * 4));
    CHECK(hipMalloc(&out, (size_t)batches * 7 * stride * 4));
    CHECK(hipMalloc(&inbox, (size_t)batches * kPushRows * stride * 4));
    CHECK(hipMalloc(&refs, (size_t)batches * 2 * stride * 4));
    CHECK(hipMalloc(&succ, (size_t)batches * 2 * stride * 4));
    CHECK(hipMalloc(&bodies, (size_t)bodies_n * 128));
    CHECK(hipMemset(rows, 0, (size_t)batches * kRows * stride * 4));
    CHECK(hipMemset(inbox, 0, (size_t)batches * kPushRows * stride * 4));
    CHECK(hipMemset(bodies, 0, (size_t)bodies_n * 128));
    std::vector<int> h((size_t)batches * 2 * stride), hs((size_t)batches * 2 * stride);
    srand(5);
    for (int b = 0; b < batches; ++b) {  // a batch references each body at most once: a random permutation of the bodies, two per constraint
        std::vector<int> perm(bodies_n);
        for (int i = 0; i < bodies_n; ++i) perm[i] = i;
        for (int i = bodies_n - 1; i > 0; --i) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        std::vector<int> sp(2 * per_batch);
        for (int i = 0; i < 2 * per_batch; ++i) sp[i] = i % per_batch;
        for (int i = 2 * per_batch - 1; i > 0; --i) { int j = rand() % (i + 1); int t = sp[i]; sp[i] = sp[j]; sp[j] = t; }
        for (int i = 0; i < stride; ++i)
            for (int k = 0; k < 2; ++k) {
                h[((size_t)b * 2 + k) * stride + i] = i < per_batch ? perm[2 * i + k] : 0;
                hs[((size_t)b * 2 + k) * stride + i] = i < per_batch ? sp[2 * i + k] : 0;
            }
    }
    CHECK(hipMemcpy(refs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(succ, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    hipStream_t stream; CHECK(hipStreamCreate(&stream));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[] = {"empty", "stream", "gather", "push", "gath64", "gath64R", "gathR", "gathNT", "gathWT", "gathNT3"};
    for (int alu : {0, 800})
        for (int variant = 0; variant < 10; ++variant) {
            if (variant == 0 && alu != 0) continue;
            hipGraph_t graph; hipGraphExec_t exec;
            CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < launches; ++l) {
                const int b = l % batches, nb = (l + 1) % batches;
                const float* r = rows + (size_t)b * kRows * stride;
                float* o = out + (size_t)b * 7 * stride;
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(64), 0, stream, per_batch);
                else if (variant == 1) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(64), 0, stream, r, o, per_batch, stride, alu);
                else if (variant == 2) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(64), 0, stream, r, refs + (size_t)b * 2 * stride, bodies, o, per_batch, stride, alu);
                else if (variant == 4 || variant == 5) hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(64), 0, stream, r, refs + (size_t)b * 2 * stride, bodies, o, per_batch, stride, alu, variant == 4 ? 1 : 0);
                else if (variant == 6) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(64), 0, stream, r, refs + (size_t)b * 2 * stride, bodies, o, per_batch, stride, alu == 0 ? -1 : -alu);
                else if (variant == 7) hipLaunchKernelGGL(k_gather_st<1>, dim3(blocks), dim3(64), 0, stream, r, refs + (size_t)b * 2 * stride, bodies, o, per_batch, stride, alu);
                else if (variant == 8) hipLaunchKernelGGL(k_gather_st<2>, dim3(blocks), dim3(64), 0, stream, r, refs + (size_t)b * 2 * stride, bodies, o, per_batch, stride, alu);
                else if (variant == 9) hipLaunchKernelGGL(k_gather_st<3>, dim3(blocks), dim3(64), 0, stream, r, refs + (size_t)b * 2 * stride, bodies, o, per_batch, stride, alu);
                else hipLaunchKernelGGL(k_push, dim3(blocks), dim3(64), 0, stream, r, inbox + (size_t)b * kPushRows * stride, succ + (size_t)b * 2 * stride,
                                        inbox + (size_t)nb * kPushRows * stride, o, per_batch, stride, alu);
            }
            CHECK(hipStreamEndCapture(stream, &graph));
            CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            float best = 1e9f;
            for (int rep = 0; rep < 12; ++rep) {
                CHECK(hipEventRecord(e0, stream));
                CHECK(hipGraphLaunch(exec, stream));
                CHECK(hipEventRecord(e1, stream));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("alu=%4d %-7s: %7.3f ms per %d launches = %6.2f us per launch\n", alu, names[variant], best, launches, best * 1e3f / launches);
            hipGraphExecDestroy(exec); hipGraphDestroy(graph);
        }
    return 0;
}
