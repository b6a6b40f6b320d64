// What does it cost on MI355X to hand a few body records from one workgroup to another INSIDE a running kernel, through global memory, when the
// two workgroups sit on different XCDs (one L2 each)? This is the unit step of a dataflow schedule for islands too large for one workgroup's
// LDS: producer writes velocities + publishes a flag, consumer polls the flag + gathers the velocities. The probe measures the
// latency of that hand-off and counts stale reads for several ways of making the data visible:
//   fence   plain stores, agent-scope release fence (L2 write-back), flag; consumer: flag, agent-scope acquire fence (L2 invalidate), plain loads
//   agent   every data dword moved by a relaxed agent-scope atomic load/store (sc1), s_waitcnt vmcnt(0) before the flag, no fences
//   system  the same at system scope (sc0 sc1)
//   plain   plain loads/stores + s_waitcnt; only coherent when the ALLOCATION is (fine-grained / uncached memory); on coarse memory it is the
//           negative control that shows stale reads
// Developer probe, not part of the product; every spin is bounded, so a protocol that does not work reports time-outs instead of hanging.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_handoff_probe xcd_handoff_probe.hip && ./xcd_handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Flavour { kFence = 0, kAgent = 1, kSystem = 2, kPlain = 3 };
constexpr int kPacketDwords = 4;  // per lane: 16 B; a wave moves 1 KiB = 8 body records of 128 B per hand-off

template <int F> __device__ inline void put(unsigned* p, unsigned v) {
    if (F == kAgent) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (F == kSystem) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");  // no scope bits (a volatile access would carry sc0 sc1)
}
template <int F> __device__ inline unsigned get(unsigned* p) {
    if (F == kAgent) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (F == kSystem) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned r;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ inline void drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// Returns false on time-out or abort. The flag itself is always polled with an agent-scope atomic.
__device__ inline bool wait_flag(unsigned* flag, unsigned want, unsigned* abort_word, unsigned* bad) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if ((++spins & 1023u) == 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > (1u << 21)) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (threadIdx.x == 0) atomicAdd(bad, 1u);
                return false;
            }
        }
    }
    return true;
}

// Workgroups (one wave each) form pairs (a, a + delta); `delta` = 1 pairs neighbours in dispatch order (different XCDs under the round-robin
// workgroup placement), `delta` = 8 pairs workgroups of the same XCD. Only the first `active` pairs play; the rest exit at once.
template <int F>
__global__ __launch_bounds__(64) void handoff(unsigned* packets, unsigned* flags, unsigned* scratch, unsigned* ctrl, unsigned* xcc, int rounds, int delta, int active, int dirty) {
    const int lane = threadIdx.x, wg = blockIdx.x;
    if (lane == 0) xcc[wg] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;  // HW_REG_XCC_ID[3:0]
    const int group = wg / (2 * delta), within = wg % (2 * delta);
    const bool is_a = within < delta;
    const int pair = group * delta + (is_a ? within : within - delta);
    if (pair >= active) return;
    unsigned* mine = packets + ((size_t)pair * 2 + (is_a ? 0 : 1)) * 64 * kPacketDwords + lane * kPacketDwords;
    unsigned* theirs = packets + ((size_t)pair * 2 + (is_a ? 1 : 0)) * 64 * kPacketDwords + lane * kPacketDwords;
    unsigned* my_flag = flags + ((size_t)pair * 2 + (is_a ? 0 : 1)) * 32;     // 128 B apart
    unsigned* their_flag = flags + ((size_t)pair * 2 + (is_a ? 1 : 0)) * 32;
    unsigned* priv = scratch + (size_t)wg * 64 * 64 + lane;                   // private lines that stay dirty in this XCD's L2
    unsigned* abort_word = ctrl, *bad = ctrl + 1;
    unsigned stale = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (!is_a) {  // B consumes first
            if (!wait_flag(their_flag, (unsigned)r, abort_word, bad)) break;
            if (F == kFence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int k = 0; k < kPacketDwords; ++k) stale += get<F>(theirs + k) != (unsigned)r;
        }
        for (int k = 0; k < dirty; ++k) priv[(size_t)k * 64] = (unsigned)r;    // accumulated impulses and the like: plain, private
        for (int k = 0; k < kPacketDwords; ++k) put<F>(mine + k, (unsigned)r);
        if (F == kFence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else drain();
        if (lane == 0) __hip_atomic_store(my_flag, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (is_a) {
            if (!wait_flag(their_flag, (unsigned)r, abort_word, bad)) break;
            if (F == kFence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int k = 0; k < kPacketDwords; ++k) stale += get<F>(theirs + k) != (unsigned)r;
        }
    }
    if (stale) atomicAdd(bad + 1, stale);
}

enum Alloc { kCoarse = 0, kFine = 1, kUncached = 2 };
static const char* kFlavourName[] = {"fence ", "agent ", "system", "plain "};
static const char* kAllocName[] = {"coarse  ", "fine    ", "uncached"};

static int alloc_shared(void** p, size_t bytes, int kind) {
    if (kind == kCoarse) CHECK(hipMalloc(p, bytes));
    else CHECK(hipExtMallocWithFlags(p, bytes, kind == kFine ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    CHECK(hipMemset(*p, 0, bytes));
    return 0;
}

static int run(int flavour, int kind, int delta, int active, int dirty, int rounds) {
    const int nwg = 256;
    unsigned *packets, *flags, *scratch, *ctrl, *xcc;  // ctrl: [0] abort word, [1] time-outs, [2] stale dwords
    const size_t packet_bytes = (size_t)nwg * 64 * kPacketDwords * 4, flag_bytes = (size_t)nwg * 32 * 4;
    if (alloc_shared((void**)&packets, packet_bytes, kind)) return 1;
    if (alloc_shared((void**)&flags, flag_bytes, kind)) return 1;
    CHECK(hipMalloc(&scratch, (size_t)nwg * 64 * 64 * 4)); CHECK(hipMalloc(&ctrl, 16)); CHECK(hipMalloc(&xcc, nwg * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9f;
    unsigned total[3] = {0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(packets, 0, packet_bytes)); CHECK(hipMemset(flags, 0, flag_bytes)); CHECK(hipMemset(ctrl, 0, 16));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        switch (flavour) {
            case kFence: handoff<kFence><<<nwg, 64>>>(packets, flags, scratch, ctrl, xcc, rounds, delta, active, dirty); break;
            case kAgent: handoff<kAgent><<<nwg, 64>>>(packets, flags, scratch, ctrl, xcc, rounds, delta, active, dirty); break;
            case kSystem: handoff<kSystem><<<nwg, 64>>>(packets, flags, scratch, ctrl, xcc, rounds, delta, active, dirty); break;
            default: handoff<kPlain><<<nwg, 64>>>(packets, flags, scratch, ctrl, xcc, rounds, delta, active, dirty); break;
        }
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        unsigned h[3]; CHECK(hipMemcpy(h, ctrl, 12, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) total[i] += h[i];
        best = ms < best ? ms : best;
    }
    std::vector<unsigned> hx(nwg); CHECK(hipMemcpy(hx.data(), xcc, nwg * 4, hipMemcpyDeviceToHost));
    printf("%s on %s memory, pair %s (xcc %u -> %u), %3d pairs, %2d dirty dwords/lane: %6.2f us per hand-off  time-outs=%u stale-dwords=%u\n", kFlavourName[flavour],
           kAllocName[kind], delta == 1 ? "across XCDs" : "inside an XCD", hx[0], hx[delta], active, dirty, best * 1e3f / rounds / 2, total[1], total[2]);
    fflush(stdout);
    hipFree(packets); hipFree(flags); hipFree(scratch); hipFree(ctrl); hipFree(xcc);
    hipEventDestroy(a); hipEventDestroy(b);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int rounds = 2000;
    struct { int flavour, kind; } combos[] = {{kFence, kCoarse}, {kAgent, kCoarse}, {kSystem, kCoarse}, {kPlain, kCoarse}, {kPlain, kFine}, {kAgent, kFine}, {kPlain, kUncached}};
    for (int delta : {1, 8})
        for (int active : {1, 128})
            for (int dirty : {0, 32})
                for (auto& cb : combos)
                    if (run(cb.flavour, cb.kind, delta, active, dirty, rounds)) return 1;
    return 0;
}
