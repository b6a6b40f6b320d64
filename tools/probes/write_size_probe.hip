// What does rocprofv3's WRITE_SIZE count for the stores a split plan's shared-body records are written with? (MI355X_MICROARCH.md: "WRITE_SIZE [is] uncalibrated:
// calibrate on a known byte count in your own access pattern".) cluster_kernel writes a record as two 16-byte `global_store_dwordx4 ... sc1` per lane, every lane to a
// different body's record; bench.py reports 351 MB of WRITE_SIZE per step of the pile against ~83 MB of rows and bodies. Five kernels, each writing a known number of
// bytes in one pattern; run under `rocprofv3 --pmc WRITE_SIZE` (and FETCH_SIZE in a second run: a partial-line write-through may read) and compare with the numbers printed.
// The second half asks the same of FETCH_SIZE and the polls (round 4, profiles/r04_s18_access_size_probe.txt): a coalesced 16-byte-per-lane stream reports half its bytes
// (the x2 of MI355X_MICROARCH.md), a record poll reports 64 bytes raw whether a lane loads its record's halves alone or a lane pair shares them — so the x2 that
// bench.py applies to all of FETCH_SIZE counts every poll as 128 bytes (it moves 64, and needs 32).
// Developer probe, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o write_size_probe.bin write_size_probe.hip && rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- ./write_size_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned scatter(unsigned i, unsigned mask) {  // a bijection of [0, mask]: neighbouring lanes end up far apart
    unsigned x = i * 0x9E3779B1u;
    x ^= x >> 15;
    return (x * 0x85EBCA77u) & mask;  // (odd multipliers and xor-shifts are bijections modulo 2^32; the mask keeps the low bits, which stay a bijection only approximately — records may repeat, bytes written do not change)
}
// the product's pattern: lane -> its own 32-byte record, two 16-byte agent-scope stores
__global__ void records_two_stores_sc1(float4* table, unsigned mask) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    float4* p = table + (size_t)scatter(i, mask) * 2;
    f4 a = {1.0f, 2.0f, 3.0f, (float)i}, b = {4.0f, 5.0f, 6.0f, (float)i};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(p), "v"(a), "v"(b) : "memory");
}
// the same records through ordinary (write-back) stores
__global__ void records_two_stores_plain(float4* table, unsigned mask) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    float4* p = table + (size_t)scatter(i, mask) * 2;
    f4 a = {1.0f, 2.0f, 3.0f, (float)i}, b = {4.0f, 5.0f, 6.0f, (float)i};
    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16" ::"v"(p), "v"(a), "v"(b) : "memory");
}
// only the first half of every record: one scattered 16-byte agent-scope store per lane
__global__ void records_one_store_sc1(float4* table, unsigned mask) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    float4* p = table + (size_t)scatter(i, mask) * 2;
    f4 a = {1.0f, 2.0f, 3.0f, (float)i};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(a) : "memory");
}
// lane pairs: lanes 2k and 2k + 1 write the two halves of ONE record with one instruction (32 contiguous bytes per pair); every lane still stores 16 bytes
__global__ void records_lane_pairs_sc1(float4* table, unsigned mask) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    float4* p = table + (size_t)scatter(i >> 1, mask) * 2 + (i & 1u);
    f4 a = {1.0f, 2.0f, 3.0f, (float)i};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(a) : "memory");
}
// a coalesced stream: lane i writes bytes [16 i, 16 i + 16), agent scope
__global__ void stream_sc1(float4* table, unsigned mask) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    f4 a = {1.0f, 2.0f, 3.0f, (float)i};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(table + (i & (2u * mask + 1u))), "v"(a) : "memory");
}
// ... and with ordinary stores
__global__ void stream_plain(float4* table, unsigned mask) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    table[i & (2u * mask + 1u)] = make_float4(1.0f, 2.0f, 3.0f, (float)i);
}

// ---- the same question for FETCH_SIZE and the polls: every lane reads its own 32-byte record (the sum is written so that nothing is optimised away) ----
__device__ inline void load16(const float4* p, f4& a) { asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(p) : "memory"); }
__global__ void read_records_two_loads_sc1(const float4* table, unsigned mask, float* sink) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const float4* p = table + (size_t)scatter(i, mask) * 2;
    f4 a, b;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    if (a.x + b.x == 12345.0f) sink[0] = 1.0f;
}
__global__ void read_records_lane_pairs_sc1(const float4* table, unsigned mask, float* sink) {  // lanes 2k, 2k + 1: the two halves of one record, 16 bytes per lane
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    f4 a;
    load16(table + (size_t)scatter(i >> 1, mask) * 2 + (i & 1u), a);
    if (a.x == 12345.0f) sink[0] = 1.0f;
}
__global__ void read_records_one_load_sc1(const float4* table, unsigned mask, float* sink) {  // a lone 16-byte load per lane
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    f4 a;
    load16(table + (size_t)scatter(i, mask) * 2, a);
    if (a.x == 12345.0f) sink[0] = 1.0f;
}
__global__ void read_stream_sc1(const float4* table, unsigned mask, float* sink) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    f4 a;
    load16(table + (i & (2u * mask + 1u)), a);
    if (a.x == 12345.0f) sink[0] = 1.0f;
}
__global__ void read_stream_plain(const float4* table, unsigned mask, float* sink) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 a = table[i & (2u * mask + 1u)];
    if (a.x == 12345.0f) sink[0] = 1.0f;
}

int main() {
    const unsigned records = 1u << 22;  // 4 Mi records of 32 bytes: 128 MiB
    float4* table;
    CHECK(hipMalloc(&table, (size_t)records * 32));
    CHECK(hipMemset(table, 0, (size_t)records * 32));
    CHECK(hipDeviceSynchronize());
    const unsigned lanes = 1u << 21;  // 2 Mi lanes per kernel
    const dim3 grid(lanes / 256), block(256);
    records_two_stores_sc1<<<grid, block>>>(table, records - 1); CHECK(hipDeviceSynchronize());
    records_two_stores_plain<<<grid, block>>>(table, records - 1); CHECK(hipDeviceSynchronize());
    records_one_store_sc1<<<grid, block>>>(table, records - 1); CHECK(hipDeviceSynchronize());
    records_lane_pairs_sc1<<<grid, block>>>(table, records - 1); CHECK(hipDeviceSynchronize());
    stream_sc1<<<grid, block>>>(table, records - 1); CHECK(hipDeviceSynchronize());
    stream_plain<<<grid, block>>>(table, records - 1); CHECK(hipDeviceSynchronize());
    float* sink; CHECK(hipMalloc(&sink, 4));
    read_records_two_loads_sc1<<<grid, block>>>(table, records - 1, sink); CHECK(hipDeviceSynchronize());
    read_records_lane_pairs_sc1<<<grid, block>>>(table, records - 1, sink); CHECK(hipDeviceSynchronize());
    read_records_one_load_sc1<<<grid, block>>>(table, records - 1, sink); CHECK(hipDeviceSynchronize());
    read_stream_sc1<<<grid, block>>>(table, records - 1, sink); CHECK(hipDeviceSynchronize());
    read_stream_plain<<<grid, block>>>(table, records - 1, sink); CHECK(hipDeviceSynchronize());
    printf("bytes loaded: read_records_two_loads_sc1 %u, read_records_lane_pairs_sc1 %u, read_records_one_load_sc1 %u, read_stream_* %u\n", lanes * 32, lanes * 16, lanes * 16, lanes * 16);
    printf("lanes per kernel %u; bytes stored: records_two_stores_* %u, records_one_store_sc1 %u, records_lane_pairs_sc1 %u, stream_* %u\n", lanes, lanes * 32, lanes * 16, lanes * 16, lanes * 16);
    hipFree(table);
    return 0;
}
