// Do shared-body records survive being moved as lane pairs? (bepu_cluster_kernel.h: a record is {xyz, n} {xyz, n}; a reader accepts it when both halves carry the
// same event number n >= the one it waits for.) Waves play ping-pong on 64 records each — across workgroups, or two waves of one workgroup (same CU, same L1) — in the
// four combinations of lone / paired loads and stores, and every accepted record's payload is checked against the number it was accepted with: a payload that does
// not belong to its number means the equal-numbers test can be fooled (a half torn below 16 bytes, or halves of two events with one number).
// Developer probe, not part of the product; every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o pair_pingpong_probe.bin pair_pingpong_probe.hip && ./pair_pingpong_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swap_neighbour(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float swap_neighbour(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ void load_two(const float4* p, const float4* q, f4& a, f4& b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p), "v"(q) : "memory");
}
// NOP = false: as the product's asm stores were until round 4 — the instruction after the statement may overwrite the store's data registers before the store has
// read them (a store of more than 64 bits wants two wait states on gfx940+, and the compiler's hazard recognizer does not look inside asm statements).
template <bool NOP>
__device__ __forceinline__ void store_one(float4* p, f4 a) {
    if (NOP) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(a) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(a) : "memory");
}

__device__ __forceinline__ f4 half_of(unsigned n, int which) {  // what event n writes into half `which` of a record
    const float b = (float)(n & 0xFFFFu) + 0.125f * which;
    return f4{b, b + 1.0f, b + 2.0f, __uint_as_float(n)};
}
template <bool PAIRED, bool NOP>
__device__ __forceinline__ void write_record(float4* rec, unsigned n, bool odd) {
    if (PAIRED) {  // first the record of the pair's even lane, then the odd lane's: an even lane writes first halves, an odd lane second halves
        float4* mine = rec;
        const long long theirs_bits = ((long long)swap_neighbour((int)((unsigned long long)mine >> 32)) << 32) | (unsigned)swap_neighbour((int)(unsigned long long)mine);
        float4* theirs = (float4*)theirs_bits;
        const unsigned their_n = (unsigned)swap_neighbour((int)n);
        store_one<NOP>((odd ? theirs : mine) + (odd ? 1 : 0), odd ? half_of(their_n, 1) : half_of(n, 0));
        store_one<NOP>((odd ? mine : theirs) + (odd ? 1 : 0), odd ? half_of(n, 1) : half_of(their_n, 0));
    } else {
        store_one<NOP>(rec, half_of(n, 0));
        store_one<NOP>(rec + 1, half_of(n, 1));
    }
}
template <bool PAIRED>
__device__ __forceinline__ void read_record(const float4* rec, bool odd, f4& l, f4& w) {
    if (PAIRED) {
        const float4* mine = rec;
        const long long theirs_bits = ((long long)swap_neighbour((int)((unsigned long long)mine >> 32)) << 32) | (unsigned)swap_neighbour((int)(unsigned long long)mine);
        const float4* theirs = (const float4*)theirs_bits;
        f4 x1, x2;
        load_two((odd ? theirs : mine) + (odd ? 1 : 0), (odd ? mine : theirs) + (odd ? 1 : 0), x1, x2);
        const f4 send = odd ? x1 : x2;
        const f4 got = {swap_neighbour(send.x), swap_neighbour(send.y), swap_neighbour(send.z), swap_neighbour(send.w)};
        l = odd ? got : x1; w = odd ? x2 : got;
    } else load_two(rec, rec + 1, l, w);
}

// Wave `2 g` and wave `2 g + 1` of the launch are partners (SAME_WG: the two waves of one workgroup; else workgroups g and g + delta of a group of 2 delta).
// Lane l owns record l of the pair; lanes advance at their own pace (a lane whose partner is late keeps polling while its neighbours move on — as in the product).
template <bool PL, bool PS, bool SAME_WG, bool NOP>
__global__ __launch_bounds__(128) void pingpong(float4* records, unsigned* ctrl, int rounds, int delta) {
    const int lane = threadIdx.x & 63;
    int pair; bool is_a;
    if (SAME_WG) { pair = blockIdx.x; is_a = threadIdx.x < 64; }
    else { const int group = blockIdx.x / (2 * delta), within = blockIdx.x % (2 * delta); is_a = within < delta; pair = group * delta + (is_a ? within : within - delta); if (threadIdx.x >= 64) return; }
    const bool odd = (lane & 1) != 0;
    // records scattered like bodies: lane l's record sits 32 x hash(l) bytes into the pair's area (its upper half: scratch records)
    float4* rec = records + (size_t)pair * 8192 + (size_t)((lane * 37 + 11) & 2047) * 2;
    unsigned bad = 0, timeouts = 0;
    unsigned mine = is_a ? 1u : 2u, theirs = is_a ? 2u : 1u;  // A writes odd numbers, B even ones: A writes 2r + 1 and waits for 2r + 2; B waits for 2r + 1, then writes 2r + 2
    unsigned done = 0;        // rounds this lane has completed
    int phase = is_a ? 0 : 1; // 0: this lane's turn to write, 1: waiting for the partner's number
    unsigned spins = 0;
    for (;;) {
        const bool active = done < (unsigned)rounds;
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        const bool write_now = active && phase == 0;
        if (__builtin_amdgcn_ballot_w64(write_now) != 0) {
            if (PS) { float4* target = write_now ? rec : records + (size_t)pair * 8192 + 4096 + (is_a ? 0 : 2048) + lane * 2; write_record<true, NOP>(target, mine, odd); }  // (a lane with nothing to write serves its neighbour; its own goes to a scratch record)
            else if (write_now) write_record<false, NOP>(rec, mine, odd);
            if (write_now) { phase = 1; if (!is_a) { ++done; mine += 2u; theirs += 2u; } }
        }
        f4 l, w;
        read_record<PL>(rec, odd, l, w);
        const unsigned nl = __float_as_uint(l.w), nw = __float_as_uint(w.w);
        if (active && phase == 1 && done < (unsigned)rounds && nl == nw && nl >= theirs) {
            const f4 e0 = half_of(nl, 0), e1 = half_of(nl, 1);
            bad += (l.x != e0.x) || (l.y != e0.y) || (l.z != e0.z) || (w.x != e1.x) || (w.y != e1.y) || (w.z != e1.z) || nl != theirs;
            phase = 0; spins = 0;
            if (is_a) { ++done; mine += 2u; theirs += 2u; }
        }
        if (++spins > (1u << 16)) { if (active) ++timeouts; done = rounds; }
    }
    if (bad) atomicAdd(ctrl + 1, bad);
    if (timeouts) atomicAdd(ctrl, timeouts);
}

template <bool PL, bool PS, bool SAME_WG, bool NOP = true>
static int run(const char* what, int delta, int rounds) {
    const int blocks = SAME_WG ? 248 : 496;
    float4* records; unsigned* ctrl;
    const size_t bytes = (size_t)(blocks + 4) * 8192 * sizeof(float4);
    CHECK(hipMalloc(&records, bytes)); CHECK(hipMalloc(&ctrl, 8));
    CHECK(hipMemset(records, 0, bytes)); CHECK(hipMemset(ctrl, 0, 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    pingpong<PL, PS, SAME_WG, NOP><<<blocks, 128>>>(records, ctrl, rounds, delta);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned h[2]; CHECK(hipMemcpy(h, ctrl, 8, hipMemcpyDeviceToHost));
    printf("  %-26s loads %-6s stores %-6s%s: %7.2f us per hand-off, lanes timed out %u, accepted records with a wrong payload or number %u\n", what, PL ? "paired" : "lone", PS ? "paired" : "lone", NOP ? "" : " (no s_nop behind the asm stores)",
           ms * 1e3f / rounds / 2, h[0], h[1]);
    fflush(stdout);
    hipFree(records); hipFree(ctrl);
    return 0;
}

int main() {
    const int rounds = 4000;
    if (run<false, false, false, false>("across XCDs", 1, rounds)) return 1;
    if (run<true, true, false, false>("across XCDs", 1, rounds)) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        if (run<false, false, false>("across XCDs", 1, rounds)) return 1;
        if (run<true, false, false>("across XCDs", 1, rounds)) return 1;
        if (run<false, true, false>("across XCDs", 1, rounds)) return 1;
        if (run<true, true, false>("across XCDs", 1, rounds)) return 1;
        if (run<false, false, false>("inside an XCD", 8, rounds)) return 1;
        if (run<true, true, false>("inside an XCD", 8, rounds)) return 1;
        if (run<false, false, true>("two waves of a workgroup", 1, rounds)) return 1;
        if (run<true, false, true>("two waves of a workgroup", 1, rounds)) return 1;
        if (run<false, true, true>("two waves of a workgroup", 1, rounds)) return 1;
        if (run<true, true, true>("two waves of a workgroup", 1, rounds)) return 1;
    }
    return 0;
}
