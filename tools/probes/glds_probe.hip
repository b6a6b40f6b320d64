// Probe: does LDS-DMA (global_load_lds_dword, M0 = destination base) reach LDS addresses above 64 KiB on gfx950, and does `s_waitcnt vmcnt(0)` by the
// issuing wave make the data visible to its own ds_read? Each wave of a 512-thread workgroup copies 4 rows of 64 dwords into its slot at a given LDS byte
// offset (asm form and builtin form), reads them back with ds_read and writes them out. Build: hipcc --offload-arch=gfx950 -O2 -o glds_probe.bin glds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds_dword(const unsigned* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <bool BUILTIN>
__global__ __launch_bounds__(512) void probe(const unsigned* src, unsigned* dst, unsigned base_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned slot = base_offset + (unsigned)wave * 1024u;  // 4 rows x 256 B per wave
    for (int r = 0; r < 4; ++r) {
        const unsigned* g = src + ((size_t)blockIdx.x * 8 + wave) * 256 + r * 64 + lane;
        if (BUILTIN) {
            __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)lds + __builtin_amdgcn_readfirstlane(slot + r * 256)), 4, 0, 0);
        } else {
            glds_dword(g, (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) char*)lds + (unsigned)__builtin_amdgcn_readfirstlane(slot + r * 256));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int r = 0; r < 4; ++r) dst[((size_t)blockIdx.x * 8 + wave) * 256 + r * 64 + lane] = lds[(slot + r * 256) / 4 + lane];
}

int main() {
    const int blocks = 64, n = blocks * 8 * 256;
    std::vector<unsigned> h(n), out(n);
    for (int i = 0; i < n; ++i) h[i] = 0x9E3779B9u * (unsigned)(i + 1);
    unsigned *d_src, *d_dst;
    hipMalloc(&d_src, n * 4); hipMalloc(&d_dst, n * 4);
    hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad_total = 0;
    for (int form = 0; form < 2; ++form)
        for (unsigned off : {0u, 32768u, 60000u & ~255u, 65536u, 100000u & ~255u, 150u * 1024u}) {
            hipMemset(d_dst, 0, n * 4);
            if (form) hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(512), 160 * 1024, 0, d_src, d_dst, off);
            else hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(512), 160 * 1024, 0, d_src, d_dst, off);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(out.data(), d_dst, n * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < n; ++i) bad += out[i] != h[i];
            printf("%s form, LDS offset %6u: %s (%d mismatches) %s\n", form ? "builtin" : "asm", off, bad ? "FAIL" : "ok", bad, e == hipSuccess ? "" : hipGetErrorString(e));
            bad_total += bad;
        }
    return bad_total ? 1 : 0;
}
