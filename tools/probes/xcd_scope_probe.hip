// Two questions behind an XCD-aware split plan, asked of the hardware before anything is built on the answers:
//  1. Which XCD does workgroup b of a grid run on? (Assumed: b mod 8, the static round-robin of the dispatcher — for plain and cooperative launches, for the workgroup
//     shapes cluster_kernel uses: 512 / 1024 threads with ~150 KB of LDS, one workgroup per CU.)
//  2. What does a shared-body record hand-off cost between two workgroups of the SAME XCD when the record only moves at workgroup scope (sc0: bypass the CU's L1, meet in
//     the XCD's L2) instead of agent scope (sc1: through to the memory side, what the records use today) — and does the reader always see the writer's record?
//     The record is the product's: {xyz, n} {xyz, n}, two 16-byte accesses, n = event number in both halves; a reader polls until both halves carry its number.
// Developer probe, not part of the product; every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_scope_probe xcd_scope_probe.hip && ./xcd_scope_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void where_am_i(unsigned* xcc) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) { lds[0] = 1.0f; xcc[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }  // HW_REG_XCC_ID[3:0]
}

template <int SCOPE>  // 0: sc0 (workgroup scope bits), 1: sc1 (agent), 2: plain (negative control)
__device__ inline void load_pair(const float4* p, f4& a, f4& b) {
    if (SCOPE == 0) asm volatile("global_load_dwordx4 %0, %2, off sc0\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    else if (SCOPE == 1) asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
template <int SCOPE>
__device__ inline void store_pair(float4* p, f4 a, f4 b) {
    if (SCOPE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0" ::"v"(p), "v"(a), "v"(b) : "memory");
    else if (SCOPE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(p), "v"(a), "v"(b) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16" ::"v"(p), "v"(a), "v"(b) : "memory");
}

// Workgroups form pairs (a, a + delta). A pair plays ping-pong on 64 records (one per lane, 128 bytes apart like the product's per-body record blocks when `stride` is
// 32 float4, or packed 32 bytes apart when 2): lane l of A writes record l with number 2r + 1, B polls for it, checks the payload, writes 2r + 2, A polls for that.
template <int SCOPE>
__global__ __launch_bounds__(64) void pingpong(float4* records, unsigned* ctrl, int rounds, int delta, int active, int stride) {
    const int lane = threadIdx.x, wg = blockIdx.x;
    const int group = wg / (2 * delta), within = wg % (2 * delta);
    const bool is_a = within < delta;
    const int pair = group * delta + (is_a ? within : within - delta);
    if (pair >= active) return;
    float4* rec = records + ((size_t)pair * 64 + lane) * stride;
    unsigned bad_payload = 0, timeouts = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned mine = 2u * r + (is_a ? 1u : 2u), theirs = 2u * r + (is_a ? 2u : 1u);
        if (is_a) {
            const float n = __uint_as_float(mine);
            store_pair<SCOPE>(rec, f4{(float)mine, 1.0f, 2.0f, n}, f4{3.0f, 4.0f, (float)mine, n});
        }
        unsigned spins = 0;
        f4 l, w;
        for (;;) {
            load_pair<SCOPE>(rec, l, w);
            if (__float_as_uint(l.w) == __float_as_uint(w.w) && __float_as_uint(l.w) >= theirs) break;
            if (++spins > (1u << 18)) { ++timeouts; break; }
        }
        if (__builtin_amdgcn_ballot_w64(spins > (1u << 18)) != 0) break;
        bad_payload += (l.x != (float)theirs) || (w.z != (float)theirs);
        if (!is_a) {
            const float n = __uint_as_float(mine);
            store_pair<SCOPE>(rec, f4{(float)mine, 1.0f, 2.0f, n}, f4{3.0f, 4.0f, (float)mine, n});
        }
    }
    if (bad_payload) atomicAdd(ctrl + 1, bad_payload);
    if (timeouts) atomicAdd(ctrl, timeouts);
}

static int mapping(int blocks, int threads, size_t lds, bool cooperative) {
    unsigned* xcc;
    CHECK(hipMalloc(&xcc, blocks * 4));
    CHECK(hipMemset(xcc, 0xFF, blocks * 4));
    CHECK(hipFuncSetAttribute((const void*)where_am_i, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int wrong_total = 0;
    for (int rep = 0; rep < 5; ++rep) {
        void* args[] = {(void*)&xcc};
        if (cooperative) CHECK(hipLaunchCooperativeKernel((const void*)where_am_i, dim3(blocks), dim3(threads), args, (unsigned)lds, 0));
        else CHECK(hipLaunchKernel((const void*)where_am_i, dim3(blocks), dim3(threads), args, lds, 0));
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned> h(blocks);
        CHECK(hipMemcpy(h.data(), xcc, blocks * 4, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int b = 0; b < blocks; ++b) wrong += h[b] != (unsigned)(b % 8);
        wrong_total += wrong;
        if (rep == 0) {
            printf("  %s launch, %4d workgroups x %4d threads, %6zu B LDS: first sixteen XCC ids", cooperative ? "cooperative" : "plain      ", blocks, threads, lds);
            for (int b = 0; b < 16 && b < blocks; ++b) printf(" %u", h[b]);
            printf("\n");
        }
    }
    printf("  %s launch, %4d workgroups x %4d threads: workgroups NOT on XCD (id mod 8) over 5 launches: %d\n", cooperative ? "cooperative" : "plain      ", blocks, threads, wrong_total);
    hipFree(xcc);
    return 0;
}

static const char* kScope[] = {"sc0 (workgroup scope bits: meet in the XCD's L2)", "sc1 (agent scope: today's records)            ", "plain (negative control)                        "};

static int handoff(int scope, int delta, int active, int stride, int rounds) {
    const int nwg = 256;
    float4* records; unsigned* ctrl;
    const size_t bytes = (size_t)nwg * 64 * stride * 16;
    CHECK(hipMalloc(&records, bytes)); CHECK(hipMalloc(&ctrl, 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9f;
    unsigned total[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(records, 0, bytes)); CHECK(hipMemset(ctrl, 0, 8));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        if (scope == 0) pingpong<0><<<nwg, 64>>>(records, ctrl, rounds, delta, active, stride);
        else if (scope == 1) pingpong<1><<<nwg, 64>>>(records, ctrl, rounds, delta, active, stride);
        else pingpong<2><<<nwg, 64>>>(records, ctrl, rounds, delta, active, stride);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        unsigned h[2]; CHECK(hipMemcpy(h, ctrl, 8, hipMemcpyDeviceToHost));
        total[0] += h[0]; total[1] += h[1];
        best = ms < best ? ms : best;
    }
    printf("  %s pair %s, %3d pairs, records %3d B apart: %6.2f us per hand-off, time-outs %u, wrong payloads %u\n", kScope[scope], delta == 1 ? "across XCDs  " : "inside an XCD",
           active, stride * 16, best * 1e3f / rounds / 2, total[0], total[1]);
    fflush(stdout);
    hipFree(records); hipFree(ctrl); hipEventDestroy(a); hipEventDestroy(b);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n1. workgroup -> XCD\n", p.name, p.multiProcessorCount);
    for (int coop = 0; coop < 2; ++coop)
        for (int threads : {1024, 512})
            for (int blocks : {248, 256, 97})
                if (mapping(blocks, threads, 150 * 1024, coop != 0)) return 1;
    if (mapping(2000, 1024, 150 * 1024, false)) return 1;  // more workgroups than CUs: the later rounds too?
    printf("2. record hand-off\n");
    const int rounds = 3000;
    for (int stride : {8, 2})
        for (int active : {1, 128})
            for (int delta : {8, 1})
                for (int scope : {0, 1, 2})
                    if (handoff(scope, delta, active, stride, rounds)) return 1;
    return 0;
}
