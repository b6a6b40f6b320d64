// Batch colouring on the device. The reference colours incrementally on the host: Solver.Add walks the batches and takes the first one none of the
// constraint's dynamic bodies is in (BepuPhysics/Solver.cs:984-1014 FindCandidateBatch, :1182-1199), and BatchCompressor later moves constraints down into
// earlier batches that have become free (BepuPhysics/BatchCompressor.cs:233). Bulk equivalent for a whole constraint list: every constraint bids for its bodies
// with a priority; a constraint that holds the highest uncoloured bid on ALL of its dynamic bodies takes the lowest batch none of them is in yet. With
// priority = insertion order this reproduces the reference's first-fit result exactly (a constraint is coloured only after every earlier constraint it
// conflicts with), in as many rounds as the longest chain of conflicts; with priority = body degree (largest first) it is the classic parallel greedy
// colouring. Kinematic references never conflict (Solver.cs:1002: only dynamic bodies are added to a batch's referenced handles).
#pragma once

namespace {

constexpr int kColourBodies = 4;  // references per constraint in the flattened list (-1 = unused)

__global__ void colour_degree_kernel(const int* __restrict__ refs, int count, unsigned* degree) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    for (int k = 0; k < kColourBodies; ++k) {
        const int r = refs[(size_t)i * kColourBodies + k];
        if (r >= 0 && (unsigned)r < kDynamicLimit) atomicAdd(&degree[r], 1u);
    }
}
// order 0: earlier constraints first (the reference's insertion order); order 1: constraints on the busiest bodies first, ties by insertion order.
__global__ void colour_priority_kernel(const int* __restrict__ refs, int count, const unsigned* __restrict__ degree, int order, unsigned long long* priority) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned long long key = 0;
    if (order == 1)
        for (int k = 0; k < kColourBodies; ++k) {
            const int r = refs[(size_t)i * kColourBodies + k];
            if (r >= 0 && (unsigned)r < kDynamicLimit) key = max(key, (unsigned long long)degree[r]);
        }
    priority[i] = (key << 32) | (unsigned)(count - i);  // unique, never zero
}
__global__ void colour_bid_kernel(const int* __restrict__ refs, int count, const int* __restrict__ colour, const unsigned long long* __restrict__ priority,
                                  unsigned long long* best) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || colour[i] >= 0) return;
    for (int k = 0; k < kColourBodies; ++k) {
        const int r = refs[(size_t)i * kColourBodies + k];
        if (r >= 0 && (unsigned)r < kDynamicLimit) atomicMax(&best[r], priority[i]);
    }
}
// A winner is alone on each of its bodies, so the masks it updates are touched by nobody else in this round.
__global__ void colour_pick_kernel(const int* __restrict__ refs, int count, int* colour, const unsigned long long* __restrict__ priority,
                                   const unsigned long long* __restrict__ best, unsigned long long* used, int fallback, unsigned* remaining) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || colour[i] >= 0) return;
    unsigned long long taken = 0;
    bool wins = true;
    for (int k = 0; k < kColourBodies; ++k) {
        const int r = refs[(size_t)i * kColourBodies + k];
        if (r < 0 || (unsigned)r >= kDynamicLimit) continue;
        wins &= best[r] == priority[i];
        taken |= used[r];
    }
    if (!wins) { atomicAdd(remaining, 1u); return; }
    if (fallback < 64) taken |= ~0ull << fallback;  // batches at and beyond the fallback threshold are not synchronized batches
    const int c = taken == ~0ull ? fallback : __ffsll((long long)~taken) - 1;
    colour[i] = c;
    if (c < fallback)
        for (int k = 0; k < kColourBodies; ++k) {
            const int r = refs[(size_t)i * kColourBodies + k];
            if (r >= 0 && (unsigned)r < kDynamicLimit) used[r] |= 1ull << c;
        }
}

}  // namespace
