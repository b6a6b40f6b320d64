// Island-per-workgroup schedule: cluster_kernel and its work-item dataflow.
#pragma once

#include "bepu_kernels_common.h"
#include "bepu_batch_kernels.h"  // the per-body integration functions shared with the launch-per-batch schedule

namespace {

// ------------------------------------------------------------------------------------------------
// Cluster path: islands (connected components of the constraint graph through dynamic bodies) are independent, so a workgroup
// that owns whole islands can run EVERY stage of EVERY substep for them without leaving the CU: the islands' bodies live in
// LDS for the whole frame (Bodies_GatherScatter's gather/scatter becomes ds_read_b128/ds_write_b128 on a per-workgroup body table)
// and the 100+ dependent kernel boundaries of the launch-per-batch schedule disappear. HBM sees each body twice per frame (load,
// write back) plus the constraint stream.
//
// Inside a pass (one WarmStart or one Solve sweep over the batches) the waves do not meet at a barrier per batch. The host splits
// every cluster's constraints into work items (<= 64 consecutive constraints of one type batch) sorted by batch, and records for
// each item its predecessors: the items that last touched any of its dynamic bodies. Waves claim items in that order from an LDS
// counter, issue the item's global loads (body references, prestep, accumulated impulses), THEN wait on the predecessors' LDS
// completion flags, gather, solve, scatter, and publish their own flag. The per-body order of constraint application is exactly
// the host's batch order (hence every result bit is unchanged), the memory latency of item t+1 hides under the math of item t
// running on another wave, and a heavy constraint type only delays the items that really depend on it.
// Deadlock freedom: items are claimed in a topological order and a wave holds one item at a time, so the earliest unfinished
// item always has all its predecessors finished.
//
// LDS: [6 or 8 planes of float4 x ncap body slots: the fields of the BodyDynamics record the sweeps touch, then (if the cluster leaves room) the local inertia,
// one plane per 16-byte field][work items][flags, counters].
// Slot numbering is rotated by the host inside every group of 16 (slot = (i & ~15) | ((i + (i >> 4)) & 15)) so that the regular
// "same joint of consecutive ragdolls" access pattern (lane stride = island size) spreads over all 16 bank slots of ds_read_b128.
// ------------------------------------------------------------------------------------------------

#ifndef BEPU_VARIANT_NT
#define BEPU_VARIANT_NT 0
#endif
#ifndef BEPU_VARIANT_CONSERVING
#define BEPU_VARIANT_CONSERVING 0
#endif
// Per translation unit: the momentum-conserving angular integration modes (PoseIntegrator.cs:193-253) compiled in. The hot variants stay as they are; a solve that asks
// for such a mode launches the unit's twin that carries this code (the integration phase's angular step, and substep 0's second transformation of the bodies the
// reference's conditionally integrating bundles transform twice — one bit per body slot of a constraint, set by the host, see build_requirk_lists).
constexpr bool kConserving = BEPU_VARIANT_CONSERVING != 0;
#ifndef BEPU_VARIANT_PASS
#define BEPU_VARIANT_PASS 0
#endif
// Per translation unit: ONE sweep per launch — all batches of one warm start or one velocity iteration, velocities from and back to HBM — for the exchanged solves of
// a scene split across GPUs (bepuhip_solve_exchanged / bepuhip_solve_lattice in the per-pass mode), which trade boundary velocities between the passes. Integration,
// the incremental contact update and the final pass stay the launch-per-batch schedule's kernels there; this unit replaces its batch-after-batch launches.
constexpr bool kPass = BEPU_VARIANT_PASS != 0;
constexpr bool kRowsNonTemporal = BEPU_VARIANT_NT != 0;  // one- and two-body constraint rows loaded with the non-temporal hint (they stream: 70 MB per pass, no reuse)
#ifndef BEPU_VARIANT_CONTACTS
#define BEPU_VARIANT_CONTACTS 0
#endif
// Per translation unit (round 6): the TYPE SET the unit is compiled for. The reference registers one TypeProcessor per constraint type and a batch only ever runs the
// processors of the types it holds (DefaultTypes.cs:18-63, TypeBatch dispatch in Solver_Solve.cs); a cluster_kernel holds the code of every type of its family, and a
// scene made of convex contact manifolds alone (Contact1-4, one- and two-body: type ids 0-7 — box stacks, piles, BASELINE.json configs[0] and configs[1]) dragged the
// eight hot joint types along: code it never executes, in an instruction cache it does not fit, under a register budget the joints set. The contacts family carries the
// eight manifolds (typed and merged items) and nothing else; the launcher picks it when no other type id is present (bepu_host_state.h, cluster_kernel_variant).
constexpr bool kContactsOnly = BEPU_VARIANT_CONTACTS != 0;
#ifndef BEPU_VARIANT_TYPE_MASK
#define BEPU_VARIANT_TYPE_MASK 0xFFFFFFFFFFFFFFFFull
#endif
// Per translation unit (round 6, last session): bit t set = the unit carries the code of constraint type id t (all of its family by default). A unit compiled for the exact
// type set of a scene — bepuhip_specialise_units: hipcc at run time, in the background, cached on disk — differs from its family's unit only in the switch cases it leaves
// out: same bits by construction. Why bother: the headline scene (sixteen hot types) runs 9 % slower on the all-44 unit than on the hot one (0.1697 against 0.1555 ms,
// profiles/r06_s22_ab_family_headline.txt) — the extra types' code costs the shared types registers and scheduling even when it never runs.
constexpr unsigned long long kTypeMask = BEPU_VARIANT_TYPE_MASK;
constexpr bool type_compiled(int id) { return id >= 0 && id < 64 && ((kTypeMask >> id) & 1ull) != 0; }

typedef __attribute__((address_space(1))) float gfloat;  // global
typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(3))) unsigned lds_u32;  // LDS: ds_read/ds_write, lgkmcnt only (a generic pointer would poll with flat loads and drag vmcnt in)

struct ClusterShared {
    float4* planes;        // [cp.planes][ncap]
    int ncap;
    ClusterItem* items;
    volatile lds_u32* flags;  // per item: epoch of the last completed pass
    int item_count;           // the cluster's work items per pass
    int* lbib;                // batch -> first item of the cluster (batch_count + 1 entries)
    int fallback_batch;       // index of the sequential fallback batch, -1 if the scene has none
    lds_u32* counter;         // item claim counter, monotonic
    int batch_count;
    unsigned* status;      // global: [0] != 0 when a wait ran out of patience (a scheduling bug, never expected); [1..7] first offender
    // SHARED plans only (split islands): slot -> body index | flags (LDS copy), the global shared-body tables, and where the step is:
    const int* slot_body;
    SharedTables st;
    unsigned events;       // integration events every shared body has seen so far (substep index + 1 during the sweeps of a substep)
    unsigned passes;       // passes (warm starts + velocity iterations) completed before the current one, over the whole step
    int code_touch;        // see touch_code_ahead
    unsigned jitter;       // schedule fuzzing (BEPUHIP_DEBUG_JITTER, 0 = off): see jitter_nap
    unsigned scratch_row;  // LDS byte address of the 256-byte row that swallows the code-touch reads
    // kConserving units only:
    int substep;             // the substep the sweeps belong to
    int angular_mode;        // 1 ConserveMomentum, 2 ConserveMomentumWithGyroscopicTorque
    float substep_dt;
    int plane_count;         // 8: the local inverse inertia is in planes 6, 7; 6: it stays in HBM
    const int* slot_table;   // slot -> body index | flags (global)
    const float4* bodies;
};

template <int ACCESS>
__device__ __forceinline__ void load_body_lds(const ClusterShared& sh, int lref, DBody& b) {
    const float4* base = sh.planes + (lref & kRefMask);
    const int n = sh.ncap;
    if (ACCESS & kOri) { float4 q = base[0]; b.ori = {q.x, q.y, q.z, q.w}; } else b.ori = {0, 0, 0, 0};
    if (ACCESS & kPos) { float4 p = base[n]; b.pos = {p.x, p.y, p.z}; } else b.pos = {0, 0, 0};
    if (ACCESS & kLin) { float4 l = base[2 * n]; b.vel.lin = {l.x, l.y, l.z}; b.linw = l.w; } else { b.vel.lin = {0, 0, 0}; b.linw = 0; }
    if (ACCESS & kAng) { float4 a = base[3 * n]; b.vel.ang = {a.x, a.y, a.z}; b.angw = a.w; } else { b.vel.ang = {0, 0, 0}; b.angw = 0; }
    if (ACCESS & kInertia) {
        float4 i0 = base[4 * n], i1 = base[5 * n];
        b.inertia.t = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
        b.inertia.invMass = i1.z;
    } else { b.inertia.t = {0, 0, 0, 0, 0, 0}; b.inertia.invMass = 0; }
}
template <int ACCESS>
__device__ __forceinline__ void load_velocity_lds(const ClusterShared& sh, int lref, DBody& b) {
    const float4* base = sh.planes + (lref & kRefMask);
    const int n = sh.ncap;
    if (ACCESS & kLin) { float4 l = base[2 * n]; b.vel.lin = {l.x, l.y, l.z}; b.linw = l.w; }
    if (ACCESS & kAng) { float4 a = base[3 * n]; b.vel.ang = {a.x, a.y, a.z}; b.angw = a.w; }
}
template <int ACCESS>
__device__ __forceinline__ void store_velocity_lds(const ClusterShared& sh, int lref, const DBody& b) {
    if ((unsigned)lref >= kDynamicLimit) return;
    float4* base = sh.planes + lref;
    if (ACCESS & kLin) base[2 * sh.ncap] = make_float4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, b.linw);
    if (ACCESS & kAng) base[3 * sh.ncap] = make_float4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, b.angw);
}


// Substep 0 of the conserving modes: the body's angular velocity goes through the mode's transformation once more right before this constraint's warm start
// (momentum_requirk_kernel of the launch-per-batch schedule, applied by the lane that holds the body; TypeProcessor.cs:1264-1281).
constexpr unsigned kLrefRequirk = 1u << 14;   // whole-island plans: in the 16-bit local reference (the bit split plans use for "shared")
constexpr unsigned kRankRequirk = 1u << 18;   // split plans: in the rank word
// Out of line (round 4): the transformation runs for a handful of lanes in the warm start of substep 0, but inlined into the gate of every constraint type it cost every
// sweep of every substep its registers (hot 1024-thread unit: 226 spilled VGPRs against 84 without the conserving code). Everything it needs travels by value.
__device__ __noinline__ V3 requirk_core(const float4* planes, int ncap, int plane_count, const int* slot_table, const float4* bodies, int angular_mode, float substep_dt, int slot, V3 ang) {
    const float4 q4 = planes[slot];
    float4 i0, i1;
    if (plane_count == kAllPlanes) { i0 = planes[6 * ncap + slot]; i1 = planes[7 * ncap + slot]; }
    else { const int body = slot_table[slot] & kSlotBodyMask; i0 = bodies[(size_t)body * 8 + 4]; i1 = bodies[(size_t)body * 8 + 5]; }
    const Q ori = {q4.x, q4.y, q4.z, q4.w};
    const Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
    if (angular_mode == 1) {
        const Sym3 world = rotateInverseInertia(local, ori);
        const Q previousOrientation = integrateOrientation(ori, ang, substep_dt * -0.5f);
        return integrateAngularVelocityConserveMomentum(previousOrientation, local, world, ang);
    }
    return integrateAngularVelocityConserveMomentumWithGyroscopicTorque(ori, local, ang, substep_dt);
}
__device__ __forceinline__ V3 requirk_angular_velocity(const ClusterShared& sh, int lref, V3 ang) {
    return requirk_core(sh.planes, sh.ncap, sh.plane_count, sh.slot_table, sh.bodies, sh.angular_mode, sh.substep_dt, lref & kRefMask, ang);
}

// ... on a private body: in place in its LDS slot, whatever part of the velocity the constraint type itself reads (the lane owns the body once its predecessors are done).
__device__ __forceinline__ void requirk_in_lds(const ClusterShared& sh, int lref) {
    float4* angular = sh.planes + 3 * sh.ncap + (lref & kRefMask);
    const float4 a4 = *angular;
    const V3 ang = requirk_angular_velocity(sh, lref, {a4.x, a4.y, a4.z});
    *angular = make_float4(ang.x, ang.y, ang.z, a4.w);
}

// Local body references travel as 16-bit halves (slot | kinematic << 15); the gather / scatter helpers take the 32-bit form (slot | kinematic << 30).
__device__ __forceinline__ int unpack_local_ref(unsigned half) { return (int)((half & 0x3FFFu) | ((half & 0x8000u) << 15)); }  // bit 14: kLrefShared

struct ItemHeader {  // wave-uniform copy of the fields the constraint code needs (SGPRs)
    int type_id, count, stride, start, batch, npred, nxpred, overflow, xoverflow;
    unsigned lrefs_off, prestep_off, accum_off;
};

__device__ __forceinline__ ItemHeader read_item(const ClusterItem* it) {
    ItemHeader h;
    h.type_id = __builtin_amdgcn_readfirstlane(it->type_id);
    h.count = __builtin_amdgcn_readfirstlane(it->count);
    h.stride = __builtin_amdgcn_readfirstlane(it->stride);
    h.start = __builtin_amdgcn_readfirstlane(it->start);
    h.lrefs_off = __builtin_amdgcn_readfirstlane(it->lrefs_off);
    h.prestep_off = __builtin_amdgcn_readfirstlane(it->prestep_off);
    h.accum_off = __builtin_amdgcn_readfirstlane(it->accum_off);
    const int packed = __builtin_amdgcn_readfirstlane(it->batch_npred);
    h.batch = packed & 0xFFFF; h.npred = (packed >> 16) & 0xF; h.nxpred = (packed >> 20) & 0xF; h.overflow = (packed >> 24) & 1; h.xoverflow = (packed >> 25) & 1;
    return h;
}

// The lane id from v_mbcnt instead of threadIdx.x & 63: nothing the compiler could have kept live (or spilled) from earlier — for addresses that are only needed
// behind a gate (ClusterGate's late impulses).
__device__ __forceinline__ int fresh_lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// ... and as an asm statement, which the compiler can neither hoist nor share: the builtins are pure, and the 168-VGPR split unit evaluated them once at kernel entry,
// spilled the result and reloaded it from scratch in front of the merged items' flag publish — a memory round trip between a finished item and its waiters.
__device__ __forceinline__ int opaque_lane_id() {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

// The lane id the gates use. Split units take the opaque form: their waits sit behind thousands of clocks of shared-body code, and a lane id carried from kernel entry is
// what the 168-VGPR unit spilled and reloaded in front of a poll. The whole-island units keep the plain form (their code is validated as it is).
#if defined(BEPU_VARIANT_SHARED) && BEPU_VARIANT_SHARED
__device__ __forceinline__ int gate_lane_id() { return opaque_lane_id(); }
#else
__device__ __forceinline__ int gate_lane_id() { return (int)(threadIdx.x & 63); }
#endif

// Wave-level claim / publish as single opaque instructions sequences: one lane (exec = 1) touches the LDS word, the result is wave-uniform.
// Written as inline asm so that the compiler sees no lane-0 branch next to the loop back-edge (it otherwise threads the "lane == 0"
// publish of one iteration into the "lane == 0" claim of the next and builds a divergent loop around convergent operations).
__device__ __forceinline__ unsigned lds_address(const volatile lds_u32* p) { return (unsigned)(__SIZE_TYPE__)p; }
__device__ __forceinline__ unsigned claim_next(lds_u32* counter) {
    unsigned ret;
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "ds_add_rtn_u32 %[r], %[a], %[one]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [r] "=&v"(ret), [sv] "=&s"(saved)
        : [a] "v"(lds_address(counter)), [one] "v"(1u)
        : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)ret);
}
// The wave's LDS velocity stores must have landed before the flag does: LDS executes a wave's instructions in order, the explicit
// wait makes that independent of the pipeline's internals. (Round 4 measured the publish WITHOUT the wait, same box, two runs each: 0.1589 / 0.1596 ms with it,
// 0.1592 / 0.1593 without on the bench scene, 0.4177 / 0.4167 on the pile — nothing to gain; profiles/r04_s6_ab_publish_nowait.txt.)
__device__ __forceinline__ void publish_item(volatile lds_u32* flag, unsigned epoch) {
    unsigned long long saved;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "ds_write_b32 %[fa], %[e]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sv] "=&s"(saved)
        : [fa] "v"(lds_address(flag)), [e] "v"(epoch)
        : "memory");
}

// Every spin is bounded: a wait that runs out of patience (~0.1 s) records itself in the status words and lets the wave continue, so a
// scheduling bug turns into an error code from bepuhip_sync instead of a hung GPU.
constexpr unsigned kSpinLimit = 1u << 21;
__device__ __noinline__ void report_stall(unsigned* status, unsigned claims, int kind, int k, int what, unsigned want, unsigned seen) {
    if ((threadIdx.x & 63) == 0 && atomicCAS(status, 0u, 1u) == 0u) {
        status[1] = blockIdx.x; status[2] = (unsigned)kind; status[3] = (unsigned)k; status[4] = (unsigned)what;
        status[5] = want; status[6] = seen; status[7] = claims;
    }
}
// Schedule fuzzing (a debugging aid, off unless BEPUHIP_DEBUG_JITTER names a seed): results must not depend on which wave runs which item when, and the waits of this
// file are what guarantees it. A missing or wrong wait shows only under unusual timing — the overflow-wait race of round 3 needed a cold instruction cache, 0.2 % of the
// runs of a warm process. With a seed every wave naps a pseudo-random time (0 - 31 x 512 clocks; one item in eight sixteen times longer) before an item waits for its
// predecessors and again before it publishes: items start and finish in orders no natural run produces, deterministically per (seed, cluster, item, pass). The fuzzers
// and the regression tests run with it; one wave-uniform compare per call site when it is off. Round 5: the cross-workgroup record protocol naps too — before a wave
// starts polling its shared bodies' records (acquire_shared / acquire_shared_many) and before it publishes a record (release_shared): hand-offs between clusters then
// happen in orders and at distances in time no natural run produces, which is what a missing wait or a torn record would need to show.
__device__ __forceinline__ void jitter_nap(const ClusterShared& sh, unsigned salt) {
    if (sh.jitter == 0u) return;
    unsigned x = sh.jitter ^ (salt * 0x9E3779B9u) ^ ((unsigned)blockIdx.x * 0x85EBCA6Bu);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    x = (unsigned)__builtin_amdgcn_readfirstlane((int)x);
    const unsigned naps = (x & 31u) * (((x >> 5) & 7u) == 0u ? 16u : 1u);
    for (unsigned n = 0; n < naps; ++n) __builtin_amdgcn_s_sleep(8);
}
// ---- shared bodies (split islands): agent-scope traffic, see SharedTables ----
// The loads wait for their data inside the asm statement (the compiler does not count an asm load in vmcnt); the stores are followed by wait_vm() where
// an event counter is about to announce them.
typedef float agent_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_agent_pair(const float4* p, float4& a, float4& b) {  // p[0], p[1]
    agent_f4 x, y;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(p) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w);
}
__device__ __forceinline__ float4 load_agent_f4(const float4* p) {
    agent_f4 x;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    return make_float4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void load_agent_pose_inertia(const float4* body, float4& q, float4& pos, float4& w0, float4& w1) {  // planes 0, 1, 6, 7 of a body record
    agent_f4 a, b, c, d;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:96 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:112 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(body) : "memory");
    q = make_float4(a.x, a.y, a.z, a.w); pos = make_float4(b.x, b.y, b.z, b.w); w0 = make_float4(c.x, c.y, c.z, c.w); w1 = make_float4(d.x, d.y, d.z, d.w);
}
// The compiler's hazard recognizer does not look inside an asm statement. A VMEM store of more than 64 bits reads its data registers after it has issued: gfx940+
// wants two wait states before a VALU instruction may overwrite them (LLVM's GCNHazardRecognizer inserts them behind its own stores), so the asm stores carry an
// `s_nop 1` themselves. Without it the instruction after the statement can change the record's first word before the store has read it: round 4's paired records
// happened to be followed, one scalar instruction later, by a `v_cndmask v0, 0, 1` — the crowd test read 0.0 / 1.0 velocities out of records
// (tools/probes/pair_pingpong_probe.hip shows the same with a store followed by a `v_add_f32` on its first data register: a quarter of all accepted records wrong).
__device__ __forceinline__ void store_agent_f4(float4* p, float4 v) {
    agent_f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// ... and the same record into the other devices' copies of the table (SharedTables.peer): system scope, the data leaves over the fabric.
__device__ __forceinline__ void store_system_f4(float4* p, float4 v) {
    agent_f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned load_seq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent_pair(float4* p, float4 a, float4 b) {
    agent_f4 x = {a.x, a.y, a.z, a.w}, y = {b.x, b.y, b.z, b.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1" ::"v"(p), "v"(x), "v"(y) : "memory");
}
// Records travel as lane pairs. A lone 16-byte agent-scope access is a request of its own on the memory side (rocprofv3 tallies 64 bytes for it; two per record and
// lane: tools/probes/write_size_probe.hip, profiles/r04_s12_write_size_probe.txt), while two neighbouring lanes that touch the two halves of ONE record with one
// instruction make one 32-byte request. So lanes 2k and 2k + 1 serve each other: the first instruction moves the record of the pair's even lane (the even lane its
// first half, the odd lane its second half), the second instruction the odd lane's record, and the halves change lanes through DPP (quad_perm [1,0,3,2]; every lane
// of the wave is enabled wherever these are called: inactive lanes carry data, not a cleared EXEC bit).
__device__ __forceinline__ int swap_neighbour(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float swap_neighbour(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ void load_agent_two(const float4* p, const float4* q, float4& a, float4& b) {  // *p and *q, one round trip
    agent_f4 x, y;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(p), "v"(q) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w);
}
__device__ __forceinline__ void load_agent_four(const float4* p, const float4* q, const float4* r, const float4* t, float4& a, float4& b, float4& c, float4& d) {
    agent_f4 x, y, z, u;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(u) : "v"(p), "v"(q), "v"(r), "v"(t) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w); c = make_float4(z.x, z.y, z.z, z.w); d = make_float4(u.x, u.y, u.z, u.w);
}
// A shared body's record: {linear xyz, event number} {angular xyz, event number}. The event number (the count of events of this step that have happened on
// the body, see SharedTables) travels IN the record: the reader's poll returns the velocity together with the news that it is the one it waits for — one
// memory round trip per hand-off instead of three (poll a counter, fetch the velocity; store, drain, bump the counter). Every event rewrites both halves
// with the same number, so a read that mixes two events shows unequal numbers and is polled again.
// Hand-offs that stay inside a cluster skip the record: when the application before this one on the body (rank - 1 of the same pass) ran in THIS cluster, its
// velocity is in the cluster's LDS slot of the body (home or ghost slot) and its item is among this item's predecessors — the lane behaves like one on a private
// body; when the application after this one runs here too, the velocity goes to the LDS slot instead of the record. The host marks both cases in the rank word
// (kRankPredLocal / kRankSuccLocal). Event numbers are absolute, so the records that are still written carry the numbers their readers wait for. The first and the
// last application of a pass always use the record (the end-of-substep readers and the next pass's first application poll it).
constexpr unsigned kRankPredLocal = 1u << 16, kRankSuccLocal = 1u << 17;
struct SharedRef {  // one body slot of one lane's constraint: the body (-1: not a shared body), the event number its record must carry before this application, and how the
    int body; unsigned number;  // velocity arrives (poll: from the record) and leaves (publish: into the record; else through the LDS slot)
    bool poll, publish;
    __device__ __forceinline__ bool shared() const { return body >= 0; }
};
template <bool END_OF_SUBSTEP>  // END_OF_SUBSTEP: the number every application of the passes so far has happened (what the incremental contact update waits for)
__device__ __forceinline__ SharedRef make_shared_ref(const ClusterShared& sh, unsigned half, unsigned srank, bool active) {
    SharedRef r;
    const bool shared = active && (half & kLrefShared) != 0 && (half & 0x8000u) == 0;
    r.body = shared ? (sh.slot_body[half & 0x3FFFu] & kSlotBodyMask) : -1;
    r.number = sh.st.base + sh.events + ((srank >> 8) & 0xFFu) * sh.passes + (END_OF_SUBSTEP ? 0u : (srank & 0xFFu));  // rank | degree << 8
    // (END_OF_SUBSTEP in a launch's first substep — a chained step's later launch — has no record to poll: every slot, ghost copies included, was staged from HBM with
    // the velocity the previous launch's home cluster wrote back)
    r.poll = shared && (END_OF_SUBSTEP ? sh.events > 0u : !(srank & kRankPredLocal));
    r.publish = shared && !(srank & kRankSuccLocal);
    return r;
}
// Two records per body: substep s works on record s & 1, and what reads the END of substep s - 1 (the incremental contact update, the pose integration of the
// home cluster and of every cluster holding a ghost copy) reads record (s - 1) & 1, which nobody rewrites before substep s + 1's integration — and that
// waits for every application of substep s, each of which comes after its own cluster's reads. So readers never have to check in anywhere.
__device__ __forceinline__ float4* shared_record(const SharedTables& st, int body, unsigned substep) { return st.vel + ((size_t)body * 2 + (substep & 1u)) * 2; }
// A record half (16 bytes at `at`, a pointer into the device's own table) into every copy of the table: the own one with an agent-scope store, the peers' at the same offset.
__device__ __forceinline__ void publish_record_f4(const SharedTables& st, float4* at, float4 v) {
    store_agent_f4(at, v);
    const int peers = __builtin_amdgcn_readfirstlane(st.peers);
    for (int q = 0; q < peers; ++q) store_system_f4(st.peer[q] + (at - st.vel), v);
}
__device__ __forceinline__ void publish_record_pair(const SharedTables& st, float4* at, float4 a, float4 b) {
    store_agent_pair(at, a, b);
    const int peers = __builtin_amdgcn_readfirstlane(st.peers);
    for (int q = 0; q < peers; ++q) { store_system_f4(st.peer[q] + (at - st.vel), a); store_system_f4(st.peer[q] + (at - st.vel) + 1, b); }
}
// Per-lane: poll the records of up to two shared bodies until each shows event number >= want (both halves equal), leaving their velocities in A / B.
// Lanes without a shared body pass at once. Bounded like every other wait of this kernel.
// Where the two accesses of a lane go when the pair serves `body` (this lane's) and the neighbour's: first the record of the pair's even lane, then the odd lane's;
// an even lane touches first halves, an odd lane second halves.
__device__ __forceinline__ void pair_addresses(const SharedTables& st, int body, unsigned substep, bool odd, float4*& first, float4*& second) {
    const int theirs = swap_neighbour(body);
    first = shared_record(st, odd ? theirs : body, substep) + (odd ? 1 : 0);
    second = shared_record(st, odd ? body : theirs, substep) + (odd ? 1 : 0);
}
// ... and what came back: the lane's own record is the half it loaded itself plus the half its neighbour loaded for it.
__device__ __forceinline__ void pair_route(bool odd, const float4& first, const float4& second, float4& l, float4& w) {
    const float4 send = odd ? first : second;
    const float4 got = make_float4(swap_neighbour(send.x), swap_neighbour(send.y), swap_neighbour(send.z), swap_neighbour(send.w));
    l = odd ? got : first; w = odd ? second : got;
}
template <bool TWO>
__device__ __forceinline__ void acquire_shared(const ClusterShared& sh, const SharedRef& ra, DBody& A, const SharedRef& rb, DBody& B, int kind, int k) {
    bool need_a = ra.poll, need_b = TWO && rb.poll;
    if (__builtin_amdgcn_ballot_w64(need_a || need_b) == 0) return;
    jitter_nap(sh, (unsigned)k * 2u + 0x51u + sh.passes * 0x2545F491u);
    const bool odd = (threadIdx.x & 1u) != 0u;
    const unsigned want_a = ra.number, want_b = rb.number;
    const unsigned record = sh.events - 1u;  // during the sweeps of substep s events == s + 1; the incremental update of substep s runs while events is still s: the record of substep s - 1
    unsigned spins = 0;
    for (;;) {
        // (a lane that has what it waits for keeps loading for its neighbour; with nothing to load for either, record 0: one request for all such lanes)
        float4 *a1, *a2, *b1, *b2, xa1, xa2, xb1, xb2, l, w;
        pair_addresses(sh.st, need_a ? ra.body : 0, record, odd, a1, a2);
        if (TWO) {
            pair_addresses(sh.st, need_b ? rb.body : 0, record, odd, b1, b2);
            load_agent_four(a1, a2, b1, b2, xa1, xa2, xb1, xb2);
        } else load_agent_two(a1, a2, xa1, xa2);
        pair_route(odd, xa1, xa2, l, w);
        if (need_a && __float_as_uint(l.w) == __float_as_uint(w.w) && __float_as_uint(l.w) >= want_a) { A.vel.lin = {l.x, l.y, l.z}; A.vel.ang = {w.x, w.y, w.z}; need_a = false; }
        if (TWO) {
            pair_route(odd, xb1, xb2, l, w);
            if (need_b && __float_as_uint(l.w) == __float_as_uint(w.w) && __float_as_uint(l.w) >= want_b) { B.vel.lin = {l.x, l.y, l.z}; B.vel.ang = {w.x, w.y, w.z}; need_b = false; }
        }
        const unsigned long long late = __builtin_amdgcn_ballot_w64(need_a || need_b);
        if (late == 0) break;
        for (int nap = 0; nap < sh.st.poll_sleep; ++nap) __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) {
            const int first = (int)__builtin_ctzll(late);
            report_stall(sh.status, *sh.counter, kind, k, __builtin_amdgcn_readlane(need_a ? ra.body : rb.body, first), __builtin_amdgcn_readlane((int)(need_a ? want_a : want_b), first), 0u);
            break;
        }
        if ((spins & 1023u) == 0 && __hip_atomic_load(sh.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // somebody already gave up
    }
}
__device__ __forceinline__ void release_shared(const ClusterShared& sh, const SharedRef& r, const DBody& b) {
    if (__builtin_amdgcn_ballot_w64(r.publish) == 0) return;
    jitter_nap(sh, (unsigned)__builtin_amdgcn_readfirstlane((int)r.number) * 0x9E3779B1u + 0xA7u);
    const bool odd = (threadIdx.x & 1u) != 0u;
    const int mine = r.publish ? r.body : -1, theirs = swap_neighbour(mine);
    const float n = __uint_as_float(r.number + 1u), their_n = swap_neighbour(n);
    // an even lane hands the angular half of its record to its odd neighbour, an odd lane the linear half of its record to its even neighbour
    const float gx = swap_neighbour(odd ? b.vel.lin.x : b.vel.ang.x), gy = swap_neighbour(odd ? b.vel.lin.y : b.vel.ang.y), gz = swap_neighbour(odd ? b.vel.lin.z : b.vel.ang.z);
    const int first_body = odd ? theirs : mine, second_body = odd ? mine : theirs;  // the record of the pair's even lane, then the odd lane's
    const float4 first = odd ? make_float4(gx, gy, gz, their_n) : make_float4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, n);
    const float4 second = odd ? make_float4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, n) : make_float4(gx, gy, gz, their_n);
    if (first_body >= 0) publish_record_f4(sh.st, shared_record(sh.st, first_body, sh.events - 1u) + (odd ? 1 : 0), first);
    if (second_body >= 0) publish_record_f4(sh.st, shared_record(sh.st, second_body, sh.events - 1u) + (odd ? 1 : 0), second);
}
// One lane, one record (integration phases): wait for event number >= want, return the velocity.
__device__ __forceinline__ void acquire_shared_one(const SharedTables& st, unsigned* status, int body, unsigned substep, unsigned want, float4& l, float4& w, int kind, int slot) {
    unsigned spins = 0;
    for (;;) {
        load_agent_pair(shared_record(st, body, substep), l, w);
        if (__float_as_uint(l.w) == __float_as_uint(w.w) && __float_as_uint(l.w) >= want) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpinLimit) { if (atomicCAS(status, 0u, 1u) == 0u) { status[1] = blockIdx.x; status[2] = (unsigned)kind; status[3] = (unsigned)slot; status[4] = (unsigned)body; status[5] = want; status[6] = __float_as_uint(l.w); status[7] = 0; } break; }
        if ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    }
}

// Every item [0, count) of the cluster has completed pass `want` (its flag holds the epoch of the last pass it completed): lane l watches item base + l.
__device__ __forceinline__ void wait_items(const ClusterShared& sh, int count, unsigned want, int kind, int k) {
    const int lane = gate_lane_id();
    for (int base = 0; base < count; base += 64) {
        const bool mine = base + lane < count;
        const volatile lds_u32* word = sh.flags + (mine ? base + lane : k);
        const unsigned need = mine ? want : 0u;
        unsigned spins = 0;
        for (;;) {
            const unsigned seen = *word;
            const unsigned long long late = __builtin_amdgcn_ballot_w64(seen < need);
            if (late == 0) break;
            if (++spins > kSpinLimit) {
                const int first = (int)__builtin_ctzll(late);
                report_stall(sh.status, *sh.counter, kind, k, base + first, want, __builtin_amdgcn_readlane((int)seen, first));
                break;
            }
            if ((spins & 4095u) == 0 && __hip_atomic_load(sh.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // somebody already gave up
        }
    }
}
// Block until every predecessor of the item has published: same-pass predecessors must have finished `epoch`; with CROSS (a Solve item: the warm start
// pass before it is not separated by a barrier) the last touchers of the bodies this item touches first must have finished `epoch - 1`.
template <bool CROSS>
__device__ __forceinline__ void wait_predecessors(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h, int k, unsigned epoch) {
    // All listed predecessors are polled at once: lane q < 6 watches same-pass predecessor q, lane 6 + q cross-pass predecessor q (pred[] and xpred[] are
    // adjacent in the item), every other lane a word that always passes. One LDS round trip after the last of them publishes, the wave is through —
    // polled one after the other, each already finished predecessor would still cost its own round trip on the cluster's critical path.
    {
        const int lane = gate_lane_id();
        const unsigned short listed = (&it->pred[0])[lane < 2 * kMaxPreds ? lane : 0];
        const bool same = lane < h.npred;
        const bool cross = CROSS && lane >= kMaxPreds && lane < kMaxPreds + h.nxpred;
        const int idx = (same || cross) ? (int)listed : k;
        const unsigned want = same ? epoch : (cross ? epoch - 1 : 0u);
        const volatile lds_u32* word = sh.flags + idx;
        unsigned spins = 0;
        for (;;) {
            const unsigned seen = *word;
            const unsigned long long late = __builtin_amdgcn_ballot_w64(seen < want);
            if (late == 0) break;  // polled back to back: pollers run at the lowest priority, and a sleep only adds to the detection latency on the chain
            if (++spins > kSpinLimit) {
                const int first = (int)__builtin_ctzll(late);
                report_stall(sh.status, *sh.counter, first < kMaxPreds ? 1 : 3, k, __builtin_amdgcn_readlane(idx, first), __builtin_amdgcn_readlane((int)want, first), __builtin_amdgcn_readlane((int)seen, first));
                break;
            }
            if ((spins & 4095u) == 0 && __hip_atomic_load(sh.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // somebody already gave up
        }
    }
    // More predecessors than the item records: wait for every item of every earlier batch (same pass), resp. for every item of the previous pass — on the items' own
    // flags, 64 at a time. (Until round 4 these two waits counted publishes per batch — "epoch x items of the batch" — and were wrong inside a fused sweep: there the
    // warm start (epoch e) and the first velocity iteration (e + 1) run concurrently, a Solve item of batch b may publish before a WarmStart item of batch b has, and
    // the count then reaches e x n_b with a warm-start item still outstanding; a cross-overflow Solve item of batch 0 started on a body whose last warm-start
    // application had not happened. Found by tools/fuzz_device.py seed 81 ordinal 91, 12 % of the runs of a cold process, 0.2 % of a warm one: it takes a wave that is
    // slow on its first pass through a heavy type's code. The flags say which PASS an item has completed, so they cannot be confused.)
    // An item of the sequential fallback batch may have predecessors in its own batch (round 4: the fallback batch runs the island schedule): it waits for every
    // item before it — items are claimed in a topological order, so that is a superset of its predecessors. (Doing the same for every batch serialises the overflow
    // items of a batch behind each other: the pile's 64-lane contact items all overflow, 0.395 -> 0.51 ms/step, measured and taken back.)
    if (h.overflow) wait_items(sh, h.batch == sh.fallback_batch ? k : (int)__builtin_amdgcn_readfirstlane(sh.lbib[h.batch]), epoch, 2, k);
    if (CROSS && h.xoverflow) wait_items(sh, sh.item_count, epoch - 1, 4, k);
    asm volatile("" ::: "memory");  // nothing below may be hoisted above the polls
}

// LDS-DMA: every lane's dword at `gsrc` lands at LDS byte address lds_dst + 4 * lane (lds_dst wave-uniform), no VGPR involved; counted in vmcnt like any load.
// hipcc does not know about it: whoever reads the destination waits (wait_vm) first.
__device__ __forceinline__ void glds_dword(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Instruction fetch is what separates the pool's two box classes: a work item runs 4-10 KB of straight-line code once, and an instruction-cache
// miss that also misses L2 (the constraint rows stream through it) costs several times more on the slow class. L2 is shared by code and data, so a wave can fetch
// its own upcoming code as DATA: one LDS-DMA read per 8 KB span, lane l reading the 128-byte line l of the span that starts at the current PC, issued right
// behind the item's row loads. The bytes go to a scratch row nobody reads; what matters is that the lines are on their way into L2 — all at once — before the
// instruction fetcher asks for them one after the other. Off (0 spans) unless ClusterParams.code_touch says otherwise (BEPUHIP_CODE_TOUCH).
// The read never leaves the unit's code: the bytes behind the last cluster_kernel of a translation unit are code_pad_kernel (bepu_cluster_variant.inc), kCodeTouchMaxSpans
// spans of s_nop that are never executed — the last switch cases of the last kernel are closer than a span to the end of the kernels themselves.
// bepuphysics2_amd/build.py checks the layout in every built unit (the pad is the unit's last function and at least that long); the host clamps the span counts.
__device__ __forceinline__ void touch_code_span(const ClusterShared& sh, int lane, int spans) {
    unsigned long long pc;
    asm volatile("s_getpc_b64 %0" : "=s"(pc));
    for (int span = 0; span < spans; ++span) glds_dword((const char*)pc + (size_t)span * 8192 + (size_t)lane * 128, (unsigned)__builtin_amdgcn_readfirstlane((int)sh.scratch_row));
}
__device__ __forceinline__ void touch_code_ahead(const ClusterShared& sh, int lane) {
    if (sh.code_touch == 0) return;
    touch_code_span(sh, lane, sh.code_touch);
}

struct ItemStamps { unsigned long long loaded, pre_gate, post_gate; };  // trace builds only

// Types whose Solve fetches its tangent and twist impulses at the gate (F::lateImpulses: Contact<N, true>, N >= 2, in the 128-VGPR units).
template <class F, class = void> struct LateImpulses { static constexpr bool value = false; };
template <class F> struct LateImpulses<F, std::void_t<decltype(F::lateImpulses)>> { static constexpr bool value = F::lateImpulses; };

// The gate the cluster path hands to the constraint functions: wait for the item's predecessors, then gather the velocities.
template <int ACC_A, int ACC_B, int BODIES, bool CROSS, bool TRACE, bool SHARED, int LATE_TWIST_ROW = -1>
struct ClusterGate {
    static constexpr bool kPin = true;  // the constraint pins its velocity-independent values before calling: they are computed while the predecessors still run
    static constexpr bool kLateImpulses = LATE_TWIST_ROW >= 0;  // the type's tangent (rows 0, 1) and twist (row LATE_TWIST_ROW) impulses are fetched here, not with the rows
    const ClusterShared& sh; const ClusterItem* it; const ItemHeader& h; int k; unsigned epoch; int ra, rb; DBody& A; DBody& B; ItemStamps& stamps;
    const SharedRef& sa; const SharedRef& sb;
    bool requirk_a, requirk_b;  // kConserving units, warm start of substep 0
    const unsigned* slab; float* late;  // kLateImpulses: the constraint slab, and where the three values go
    __device__ __forceinline__ float lateImpulse(int which) const { return late[which]; }
    __device__ __forceinline__ void operator()(BodyVel&, BodyVel&) const {
        if constexpr (kLateImpulses) {  // asked for now, used behind the penetration rows: the round trip runs under the wait below
            const int lane = fresh_lane_id();
            const gfloat* rows = (const gfloat*)(slab + h.accum_off) + (h.start + (lane < h.count ? lane : h.count - 1));
            late[0] = kRowsNonTemporal ? __builtin_nontemporal_load(&rows[0]) : rows[0];
            late[1] = kRowsNonTemporal ? __builtin_nontemporal_load(&rows[(size_t)h.stride]) : rows[(size_t)h.stride];
            late[2] = kRowsNonTemporal ? __builtin_nontemporal_load(&rows[(size_t)LATE_TWIST_ROW * h.stride]) : rows[(size_t)LATE_TWIST_ROW * h.stride];
        }
        if (TRACE) stamps.pre_gate = __builtin_readcyclecounter();
        jitter_nap(sh, (unsigned)k * 2u + epoch * 0x632BE5ABu);
        wait_predecessors<CROSS>(sh, it, h, k, epoch);
        __builtin_amdgcn_s_setprio(3);  // from here to the publish the item is on its bodies' critical path: issue ahead of waves still preparing theirs
        if constexpr (kConserving && !CROSS) {
            if (requirk_a && !sa.shared()) requirk_in_lds(sh, ra);
            if (BODIES == 2 && requirk_b && !sb.shared()) requirk_in_lds(sh, rb);
        }
        load_velocity_lds<ACC_A>(sh, ra, A);
        if (BODIES == 2) load_velocity_lds<ACC_B>(sh, rb, B);
        if constexpr (SHARED) {
            // A shared body's velocity travels whole: a type that reads only the angular half still hands the linear half on (into a record, or to the next local
            // application), so a lane that takes the body from the LDS slot (its predecessor ran in this cluster) takes both halves.
            if (sa.shared() && !sa.poll) load_velocity_lds<kLin | kAng>(sh, ra, A);
            if (BODIES == 2 && sb.shared() && !sb.poll) load_velocity_lds<kLin | kAng>(sh, rb, B);
            // bodies other clusters also touch: our turn comes when the body's record carries this application's event number (the velocity comes with it)
            acquire_shared<BODIES == 2>(sh, sa, A, sb, B, 6, k);
        }
        if constexpr (kConserving && !CROSS && SHARED) {  // a shared body's velocity is whole in A / B by now (from its record, or from the LDS slot) and leaves whole
            if (requirk_a && sa.shared()) A.vel.ang = requirk_angular_velocity(sh, ra, A.vel.ang);
            if (BODIES == 2 && requirk_b && sb.shared()) B.vel.ang = requirk_angular_velocity(sh, rb, B.vel.ang);
        }
        if (TRACE) stamps.post_gate = __builtin_readcyclecounter();
    }
};


// acquire_shared for the N bodies of a three- or four-body constraint.
template <int N>
__device__ __forceinline__ void acquire_shared_many(const ClusterShared& sh, const SharedRef* s, DBody* b, int kind, int k) {
    bool need[N];
    bool any = false;
    _Pragma("unroll") for (int j = 0; j < N; ++j) { need[j] = s[j].poll; any |= need[j]; }
    if (__builtin_amdgcn_ballot_w64(any) == 0) return;
    jitter_nap(sh, (unsigned)k * 2u + 0x53u + sh.passes * 0x2545F491u);
    unsigned spins = 0;
    for (;;) {
        any = false;
        _Pragma("unroll") for (int j = 0; j < N; ++j) {
            if (need[j]) {
                float4 l, w;
                load_agent_pair(shared_record(sh.st, s[j].body, sh.events - 1u), l, w);
                if (__float_as_uint(l.w) == __float_as_uint(w.w) && __float_as_uint(l.w) >= s[j].number) { b[j].vel.lin = {l.x, l.y, l.z}; b[j].vel.ang = {w.x, w.y, w.z}; need[j] = false; }
            }
            any |= need[j];
        }
        const unsigned long long late = __builtin_amdgcn_ballot_w64(any);
        if (late == 0) break;
        for (int nap = 0; nap < sh.st.poll_sleep; ++nap) __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) {
            int body = -1; unsigned want = 0u;
            _Pragma("unroll") for (int j = 0; j < N; ++j) if (need[j]) { body = s[j].body; want = s[j].number; }
            const int first = (int)__builtin_ctzll(late);
            report_stall(sh.status, *sh.counter, kind, k, __builtin_amdgcn_readlane(body, first), __builtin_amdgcn_readlane((int)want, first), 0u);
            break;
        }
        if ((spins & 1023u) == 0 && __hip_atomic_load(sh.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // somebody already gave up
    }
}
// The gate of three- and four-body constraints: same wait, then the velocities of all N bodies (SHARED: shared bodies as in ClusterGate — from the LDS slot when the
// application before this one ran in this cluster, from the body's record otherwise).
template <int ACCESS, int N, bool CROSS, bool SHARED>
struct ClusterGateMany {
    static constexpr bool kPin = true;
    const ClusterShared& sh; const ClusterItem* it; const ItemHeader& h; int k; unsigned epoch; const int* refs; DBody* b; const SharedRef* s; const bool* requirk;
    __device__ __forceinline__ void many(BodyVel* vel) const {
        jitter_nap(sh, (unsigned)k * 2u + epoch * 0x632BE5ABu);
        wait_predecessors<CROSS>(sh, it, h, k, epoch);
        __builtin_amdgcn_s_setprio(3);
        if constexpr (kConserving && !CROSS) {
            _Pragma("unroll") for (int j = 0; j < N; ++j) if (requirk[j] && !s[j].shared()) requirk_in_lds(sh, refs[j]);
        }
        _Pragma("unroll") for (int j = 0; j < N; ++j) load_velocity_lds<ACCESS>(sh, refs[j], b[j]);
        if constexpr (SHARED) {
            _Pragma("unroll") for (int j = 0; j < N; ++j) if (s[j].shared() && !s[j].poll) load_velocity_lds<kLin | kAng>(sh, refs[j], b[j]);
            acquire_shared_many<N>(sh, s, b, 6, k);
        }
        if constexpr (kConserving && !CROSS && SHARED) {
            _Pragma("unroll") for (int j = 0; j < N; ++j) if (requirk[j] && s[j].shared()) b[j].vel.ang = requirk_angular_velocity(sh, refs[j], b[j].vel.ang);
        }
        _Pragma("unroll") for (int j = 0; j < N; ++j) vel[j] = b[j].vel;
    }
};
template <class F, int STAGE, bool SHARED>
__device__ __forceinline__ void run_cluster_constraint_many(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h, int k, int lane, unsigned epoch,
                                                            unsigned* __restrict__ slab, float dt, float inv_dt) {
    constexpr int N = F::bodies;
    const bool active = lane < h.count;
    const int i = h.start + (active ? lane : h.count - 1), stride = h.stride;
    const gint* lrefs = (const gint*)(slab + h.lrefs_off);
    gfloat* prestep = (gfloat*)(slab + h.prestep_off);
    gfloat* accum = (gfloat*)(slab + h.accum_off);
    float p[F::prestepFloats], a[F::impulseFloats];
    int refs[N];
    SharedRef s[N];
    bool requirk[N];
    const bool first_warm_start = kConserving && STAGE == kStageWarmStart && sh.substep == 0;
    _Pragma("unroll") for (int j = 0; j < N; j += 2) {
        const unsigned w = (unsigned)lrefs[(size_t)(j / 2) * stride + i];
        refs[j] = unpack_local_ref(w & 0xFFFFu);
        if (j + 1 < N) refs[j + 1] = unpack_local_ref(w >> 16);
        s[j] = SharedRef{-1, 0u, false, false};
        if (j + 1 < N) s[j + 1] = SharedRef{-1, 0u, false, false};
        requirk[j] = !SHARED && first_warm_start && active && (w & kLrefRequirk) != 0 && (w & 0x8000u) == 0;
        if (j + 1 < N) requirk[j + 1] = !SHARED && first_warm_start && active && ((w >> 16) & kLrefRequirk) != 0 && (w & 0x80000000u) == 0;
        if constexpr (SHARED) {  // rank | degree << 8 | hand-off flags of this application on each shared body: the rows right behind the local references
            const gint* srank = lrefs + (size_t)((N + 1) / 2) * stride;
            const unsigned rank_j = (unsigned)srank[(size_t)j * stride + i];
            s[j] = make_shared_ref<false>(sh, w & 0xFFFFu, rank_j, active);
            requirk[j] = first_warm_start && active && (rank_j & kRankRequirk) != 0;
            if (j + 1 < N) {
                const unsigned rank_j1 = (unsigned)srank[(size_t)(j + 1) * stride + i];
                s[j + 1] = make_shared_ref<false>(sh, w >> 16, rank_j1, active);
                requirk[j + 1] = first_warm_start && active && (rank_j1 & kRankRequirk) != 0;
            }
        }
    }
    _Pragma("unroll") for (int f = 0; f < F::prestepFloats; ++f) p[f] = prestep[(size_t)f * stride + i];
    _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) a[f] = accum[(size_t)f * stride + i];
    DBody b[N];
    V3 pos[N]; float inverseMass[N]; BodyVel vel[N];
    _Pragma("unroll") for (int j = 0; j < N; ++j) {
        load_body_lds<F::access & ~(kLin | kAng)>(sh, refs[j], b[j]);
        pos[j] = b[j].pos; inverseMass[j] = b[j].inertia.invMass; vel[j] = b[j].vel;
    }
    ClusterGateMany<F::access, N, STAGE == kStageSolve, SHARED> gate{sh, it, h, k, epoch, refs, b, s, requirk};
    if (STAGE == kStageWarmStart) F::warmStartN(pos, inverseMass, p, a, vel, gate);
    else F::solveN(pos, inverseMass, dt, inv_dt, p, a, vel, gate);
    _Pragma("unroll") for (int j = 0; j < N; ++j) {
        b[j].vel = vel[j];
        store_velocity_lds<F::access>(sh, (active && !s[j].shared()) ? refs[j] : -1, b[j]);
        if constexpr (SHARED) {  // as in run_cluster_constraint: to the LDS slot when the next application runs here, into the record otherwise
            store_velocity_lds<kLin | kAng>(sh, (s[j].shared() && !s[j].publish) ? refs[j] : -1, b[j]);
            release_shared(sh, s[j], b[j]);
        }
    }
    jitter_nap(sh, (unsigned)k * 2u + 1u + epoch * 0x632BE5ABu);
    publish_item(sh.flags + k, epoch);
    __builtin_amdgcn_s_setprio(0);
    if (STAGE == kStageSolve && active) { _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) accum[(size_t)f * stride + i] = a[f]; }
}

#ifndef BEPU_ITEM_INLINE
#define BEPU_ITEM_INLINE __forceinline__
#endif
template <class F, int STAGE, bool TRACE, bool SHARED>
__device__ BEPU_ITEM_INLINE void run_cluster_constraint(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h, int k, int lane, unsigned epoch,
                                                       unsigned* __restrict__ slab, float dt, float inv_dt, ItemStamps& stamps) {
    // Lanes beyond the item's count mirror its last constraint and never store: the whole body runs with a full exec mask,
    // which keeps the control flow around the (wave-uniform) waits trivially structured.
    const bool active = lane < h.count;
    const int i = h.start + (active ? lane : h.count - 1), stride = h.stride;
    const gint* lrefs = (const gint*)(slab + h.lrefs_off);
    gfloat* prestep = (gfloat*)(slab + h.prestep_off);
    gfloat* accum = (gfloat*)(slab + h.accum_off);
    float p[F::prestepFloats];
    float a[F::impulseFloats];
    // issue the item's global loads first: their latency hides under the velocity-independent work and the wait for the predecessors
    const unsigned both = kRowsNonTemporal ? (unsigned)__builtin_nontemporal_load(&lrefs[i]) : (unsigned)lrefs[i];  // two 16-bit local references per word
    unsigned rank_a = 0u, rank_b = 0u;  // SHARED: rank | degree << 8 of this application on each shared body (rows right behind the local references)
    if constexpr (SHARED) {
        const gint* srank = lrefs + (size_t)((F::bodies + 1) / 2) * stride;
        rank_a = (unsigned)srank[i];
        if (F::bodies == 2) rank_b = (unsigned)srank[(size_t)stride + i];
    }
    constexpr bool late = LateImpulses<F>::value && STAGE == kStageSolve && !SHARED;  // tangent and twist impulses at the gate (ClusterGate): rows 0, 1 and the last one
    auto with_the_rows = [](int f) { return !late || (f >= 2 && f < F::impulseFloats - 1); };
    if (late) { a[0] = a[1] = a[F::impulseFloats - 1] = 0.0f; }
    if (kRowsNonTemporal) {  // per translation unit (BEPU_VARIANT_NT): the constraint rows are read once per pass; see the note on box classes in DESIGN.md 5
        _Pragma("unroll") for (int f = 0; f < F::prestepFloats; ++f) p[f] = __builtin_nontemporal_load(&prestep[(size_t)f * stride + i]);
        if (STAGE != kStageIncremental) { _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) if (with_the_rows(f)) a[f] = __builtin_nontemporal_load(&accum[(size_t)f * stride + i]); }
    } else {
        _Pragma("unroll") for (int f = 0; f < F::prestepFloats; ++f) p[f] = prestep[(size_t)f * stride + i];
        if (STAGE != kStageIncremental) { _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) if (with_the_rows(f)) a[f] = accum[(size_t)f * stride + i]; }
    }
    if (STAGE != kStageIncremental) touch_code_ahead(sh, lane);
    const int ra = unpack_local_ref(both & 0xFFFFu);
    const int rb = (F::bodies == 2) ? unpack_local_ref(both >> 16) : -1;
    SharedRef sa = {-1, 0u, false, false}, sb = {-1, 0u, false, false};
    if constexpr (SHARED) {
        sa = make_shared_ref<STAGE == kStageIncremental>(sh, both & 0xFFFFu, rank_a, active);
        if (F::bodies == 2) sb = make_shared_ref<STAGE == kStageIncremental>(sh, both >> 16, rank_b, active);
    }
    DBody A, B;
    if (STAGE == kStageIncremental) {  // reads velocities, writes only this constraint's depths: no ordering inside the stage
        load_body_lds<kAccessOnlyVelocity>(sh, ra, A);
        if (F::bodies == 2) load_body_lds<kAccessOnlyVelocity>(sh, rb, B); else load_body_lds<0>(sh, 0, B);
        if constexpr (SHARED) {
            // a shared body's end-of-substep velocity is in its record once every application of the previous substep has happened (rank 0 of the pass that
            // would come next)
            acquire_shared<F::bodies == 2>(sh, sa, A, sb, B, 8, k);
        }
        F::incrementalUpdate(dt, A.vel, B.vel, p);
        if constexpr (F::incremental) {
            if (active) {
                _Pragma("unroll") for (int cidx = 0; cidx < F::contacts; ++cidx) {
                    if (kRowsNonTemporal) __builtin_nontemporal_store(p[F::depthRow(cidx)], &prestep[(size_t)F::depthRow(cidx) * stride + i]);
                    else prestep[(size_t)F::depthRow(cidx) * stride + i] = p[F::depthRow(cidx)];
                }
            }
        }
        return;
    }
    constexpr int accA = (STAGE == kStageWarmStart) ? F::wsA : F::svA;
    constexpr int accB = (STAGE == kStageWarmStart) ? F::wsB : F::svB;
    // Poses and inertias only change in the integration phase (a barrier away): gather them and let the constraint do all its
    // velocity-independent work (jacobians, effective mass, bias) BEFORE waiting for the predecessors; the gate then waits and
    // gathers the velocities, so only the corrective-impulse tail of the constraint sits on the cluster's critical path.
    load_body_lds<accA & ~(kLin | kAng)>(sh, ra, A);
    if (F::bodies == 2) load_body_lds<accB & ~(kLin | kAng)>(sh, rb, B); else load_body_lds<0>(sh, 0, B);
    if (TRACE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps.loaded = __builtin_readcyclecounter(); }
    bool requirk_a = false, requirk_b = false;
    if constexpr (kConserving && STAGE == kStageWarmStart) {
        if (sh.substep == 0 && active) {
            requirk_a = SHARED ? (rank_a & kRankRequirk) != 0 : ((both & kLrefRequirk) != 0 && (both & 0x8000u) == 0);
            requirk_b = F::bodies == 2 && (SHARED ? (rank_b & kRankRequirk) != 0 : (((both >> 16) & kLrefRequirk) != 0 && (both & 0x80000000u) == 0));
        }
    }
    float late_impulses[3] = {0.0f, 0.0f, 0.0f};
    ClusterGate<accA, accB, F::bodies, STAGE == kStageSolve, TRACE, SHARED, late ? F::impulseFloats - 1 : -1> gate{sh, it, h, k, epoch, ra, rb, A, B, stamps, sa, sb, requirk_a, requirk_b, slab, late_impulses};
    if (STAGE == kStageWarmStart) F::warmStart(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, p, a, A.vel, B.vel, gate);
    else F::solve(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, dt, inv_dt, p, a, A.vel, B.vel, gate);
    // -1: never stored (same rule as kinematic / empty references). A shared body's velocity goes to the LDS slot only when the next application on it runs in this
    // cluster — then both halves, whatever the type's access filter (the slot has to hold what a record would).
    store_velocity_lds<accA>(sh, (active && !sa.shared()) ? ra : -1, A);
    if (F::bodies == 2) store_velocity_lds<accB>(sh, (active && !sb.shared()) ? rb : -1, B);
    if constexpr (SHARED) {
        store_velocity_lds<kLin | kAng>(sh, (sa.shared() && !sa.publish) ? ra : -1, A);
        if (F::bodies == 2) store_velocity_lds<kLin | kAng>(sh, (sb.shared() && !sb.publish) ? rb : -1, B);
    }
    if constexpr (SHARED) {
        release_shared(sh, sa, A);  // velocity and "event done" in one record: the next application on the body polls exactly this
        if (F::bodies == 2) release_shared(sh, sb, B);
    }
    jitter_nap(sh, (unsigned)k * 2u + 1u + epoch * 0x632BE5ABu);
    publish_item(sh.flags + k, epoch);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (late) {  // ... from wave-uniform values and the lane id again: the row pointer is not carried through the tail either
        const int lane_now = fresh_lane_id();
        if (lane_now < h.count) {
            gfloat* rows = (gfloat*)(slab + h.accum_off) + (h.start + lane_now);
            _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) { if (kRowsNonTemporal) __builtin_nontemporal_store(a[f], &rows[(size_t)f * h.stride]); else rows[(size_t)f * h.stride] = a[f]; }
        }
    } else if (STAGE == kStageSolve && active) {  // off the critical path: nothing reads the impulses before the next pass (a barrier away)
        _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) { if (kRowsNonTemporal) __builtin_nontemporal_store(a[f], &accum[(size_t)f * stride + i]); else accum[(size_t)f * stride + i] = a[f]; }
    }
}

// ---- merged manifold items (round 5; split plans only) ----
// A pile cluster's batch holds four or five convex manifold type batches (Contact1..4, one or two bodies) with 10 - 60 constraints each: typed work items run at
// 13 - 64 lanes, and a wave spends the same time on an item whatever its lane count — wave time is what a split cluster runs out of. The planner
// therefore GROUPS typed items of one batch and one family (two-body Contact1..4, or the one-body four) whose lane counts sum to at most 64: the group's first item
// (the leader, ClusterItem.shape bits 24-25 = members behind it) is followed by its members (bit 26) in the item array. Nothing else about the items changes — every
// member keeps its rows, its predecessor lists and its flag, so structural updates rebuild the lists as before. The wave that claims the leader runs ALL the group's
// constraints as one item: lane l takes its rows from the typed item its lane range falls in, with that type's contact count (ContactFused, per-lane count); it waits
// for the predecessors of every item of the group and publishes every item's flag. A wave that claims a member goes on to its next claim.
constexpr int kTraceFusedType = 0x80;  // the type column of a fused group in the cluster trace (| 1: two bodies)

// wait_predecessors for the up to four items of a group at once: lanes [12 s, 12 s + 12) watch the listed predecessors of item s (pred[] then xpred[]).
// What the wave needs from the items' headers comes by value: `npred` / `nxpred` are the counts of the item THIS lane's twelve belong to (0 beyond the group).
template <bool CROSS>
__device__ __forceinline__ void wait_predecessors_fused(const ClusterShared& sh, const ClusterItem* it, int members, int npred, int nxpred, bool overflow, bool xoverflow, int batch, int k, unsigned epoch) {
    {
        const int lane = gate_lane_id();
        const int s = (lane >= 12) + (lane >= 24) + (lane >= 36), q = lane - 12 * s;
        const bool present = lane < 48 && s <= members;
        const unsigned short listed = (&(it + (present ? s : 0))->pred[0])[present ? q : 0];
        const bool same = present && q < npred;
        const bool cross = CROSS && present && q >= kMaxPreds && q < kMaxPreds + nxpred;
        const int idx = (same || cross) ? (int)listed : k;
        const unsigned want = same ? epoch : (cross ? epoch - 1 : 0u);
        const volatile lds_u32* word = sh.flags + idx;
        unsigned spins = 0;
        for (;;) {
            const unsigned seen = *word;
            const unsigned long long late = __builtin_amdgcn_ballot_w64(seen < want);
            if (late == 0) break;
            if (++spins > kSpinLimit) {
                const int first = (int)__builtin_ctzll(late);
                report_stall(sh.status, *sh.counter, 12, k, __builtin_amdgcn_readlane(idx, first), __builtin_amdgcn_readlane((int)want, first), __builtin_amdgcn_readlane((int)seen, first));
                break;
            }
            if ((spins & 4095u) == 0 && __hip_atomic_load(sh.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
    }
    // overflow waits as in wait_predecessors, once for the group (its items share a batch; the sequential fallback batch is never grouped)
    if (overflow) wait_items(sh, (int)__builtin_amdgcn_readfirstlane(sh.lbib[batch]), epoch, 2, k);
    if (CROSS && xoverflow) wait_items(sh, sh.item_count, epoch - 1, 4, k);
    asm volatile("" ::: "memory");
}
// publish_item for `n` consecutive items: lane s < n writes flag s.
__device__ __forceinline__ void publish_items(volatile lds_u32* flag, int n, unsigned epoch) {
    unsigned long long saved;
    const unsigned long long mask = (1ull << n) - 1ull;
    const unsigned address = lds_address(flag) + 4u * (unsigned)opaque_lane_id();
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "ds_write_b32 %[fa], %[e]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sv] "=&s"(saved)
        : [fa] "v"(address), [e] "v"(epoch), [m] "s"(mask)
        : "memory");
}
template <int ACC_A, int ACC_B, int BODIES, bool CROSS, bool TRACE, bool SHARED>
struct FusedGate {
    static constexpr bool kPin = true;
    const ClusterShared& sh; const ClusterItem* it; int members, npred, nxpred; bool overflow, xoverflow; int batch; int k; unsigned epoch; int ra, rb; DBody& A; DBody& B; ItemStamps& stamps;
    const SharedRef& sa; const SharedRef& sb;
    bool requirk_a, requirk_b;
    __device__ __forceinline__ void operator()(BodyVel&, BodyVel&) const {
        if (TRACE) stamps.pre_gate = __builtin_readcyclecounter();
        jitter_nap(sh, (unsigned)k * 2u + epoch * 0x632BE5ABu);
        wait_predecessors_fused<CROSS>(sh, it, members, npred, nxpred, overflow, xoverflow, batch, k, epoch);
        __builtin_amdgcn_s_setprio(3);
        if constexpr (kConserving && !CROSS) {
            if (requirk_a && !sa.shared()) requirk_in_lds(sh, ra);
            if (BODIES == 2 && requirk_b && !sb.shared()) requirk_in_lds(sh, rb);
        }
        load_velocity_lds<ACC_A>(sh, ra, A);
        if (BODIES == 2) load_velocity_lds<ACC_B>(sh, rb, B);
        if constexpr (SHARED) {
            if (sa.shared() && !sa.poll) load_velocity_lds<kLin | kAng>(sh, ra, A);
            if (BODIES == 2 && sb.shared() && !sb.poll) load_velocity_lds<kLin | kAng>(sh, rb, B);
            acquire_shared<BODIES == 2>(sh, sa, A, sb, B, 6, k);
        }
        if constexpr (kConserving && !CROSS && SHARED) {
            if (requirk_a && sa.shared()) A.vel.ang = requirk_angular_velocity(sh, ra, A.vel.ang);
            if (BODIES == 2 && requirk_b && sb.shared()) B.vel.ang = requirk_angular_velocity(sh, rb, B.vel.ang);
        }
        if (TRACE) stamps.post_gate = __builtin_readcyclecounter();
    }
};

__device__ __forceinline__ float load_row(const gfloat* p) { return kRowsNonTemporal ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ int load_row(const gint* p) { return kRowsNonTemporal ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void store_row(gfloat* p, float v) { if (kRowsNonTemporal) __builtin_nontemporal_store(v, p); else *p = v; }

// The group's work item: run_cluster_constraint with per-lane rows. `h0` is the leader's header, `members` the items behind it (1..3).
template <bool TWO, int STAGE, bool TRACE, bool SHARED>
__device__ __forceinline__ int run_cluster_fused(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h0, int members, int k, int lane, unsigned epoch,
                                                  unsigned* __restrict__ slab, float dt, float inv_dt, ItemStamps& stamps) {
    using F = ContactFused<TWO>;
    // the members' headers (wave-uniform; an absent member reads the leader's and counts no lanes)
    const ItemHeader h1 = read_item(it + (members >= 1 ? 1 : 0)), h2 = read_item(it + (members >= 2 ? 2 : 0)), h3 = read_item(it + (members >= 3 ? 3 : 0));
    const int c1 = members >= 1 ? h1.count : 0, c2 = members >= 2 ? h2.count : 0, c3 = members >= 3 ? h3.count : 0;
    const int off1 = h0.count, off2 = off1 + c1, off3 = off2 + c2, total = off3 + c3;
    const bool active = lane < total;
    const int l = active ? lane : total - 1;  // lanes beyond the group mirror its last constraint and never store
    const int sub = (l >= off1) + (l >= off2) + (l >= off3);
#define BEPU_PICK(field) (sub == 0 ? h0.field : (sub == 1 ? h1.field : (sub == 2 ? h2.field : h3.field)))
    const int stride = BEPU_PICK(stride);
    const int i = BEPU_PICK(start) + (l - (sub == 0 ? 0 : (sub == 1 ? off1 : (sub == 2 ? off2 : off3))));
    const int count = (BEPU_PICK(type_id) & 3) + 1;  // one-body Contact N is type N - 1, two-body Contact N type 3 + N
    const gint* lrefs = (const gint*)(slab + BEPU_PICK(lrefs_off)) + i;
    gfloat* prestep = (gfloat*)(slab + BEPU_PICK(prestep_off)) + i;
    gfloat* accum = (gfloat*)(slab + BEPU_PICK(accum_off)) + i;
#undef BEPU_PICK
    // for the gate: the predecessor counts of the item whose lists this lane's twelve watch, the group's overflow flags
    const int watch = (lane >= 12) + (lane >= 24) + (lane >= 36);
    const int watch_npred = watch == 0 ? h0.npred : (watch == 1 ? h1.npred : (watch == 2 ? h2.npred : h3.npred));
    const int watch_nxpred = watch == 0 ? h0.nxpred : (watch == 1 ? h1.nxpred : (watch == 2 ? h2.nxpred : h3.nxpred));
    const bool group_overflow = h0.overflow != 0 || (members >= 1 && h1.overflow != 0) || (members >= 2 && h2.overflow != 0) || (members >= 3 && h3.overflow != 0);
    const bool group_xoverflow = h0.xoverflow != 0 || (members >= 1 && h1.xoverflow != 0) || (members >= 2 && h2.xoverflow != 0) || (members >= 3 && h3.xoverflow != 0);
    float p[F::prestepFloats], a[F::impulseFloats];
    const unsigned both = (unsigned)load_row(lrefs);
    unsigned rank_a = 0u, rank_b = 0u;
    if constexpr (SHARED) {
        const gint* srank = lrefs + (size_t)((F::bodies + 1) / 2) * stride;
        rank_a = (unsigned)srank[0];
        if (F::bodies == 2) rank_b = (unsigned)srank[(size_t)stride];
    }
    // the lane's rows in Contact4's layout: contact c at p[4c..4c+3] (absent contacts: zero, never used), the common block (its rows follow the lane's own contacts)
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {
        _Pragma("unroll") for (int r = 0; r < 4; ++r) p[4 * c + r] = 0.0f;
        if (c < count) { _Pragma("unroll") for (int r = 0; r < 4; ++r) p[4 * c + r] = load_row(&prestep[(size_t)(4 * c + r) * stride]); }
    }
    {
        const gfloat* common = prestep + (size_t)(4 * count) * stride;
        _Pragma("unroll") for (int f = 0; f < F::commonFloats; ++f) p[16 + f] = load_row(&common[(size_t)f * stride]);
    }
    if (STAGE != kStageIncremental) {
        a[0] = load_row(&accum[0]); a[1] = load_row(&accum[(size_t)stride]);
        _Pragma("unroll") for (int c = 0; c < 4; ++c) { a[2 + c] = 0.0f; if (c < count) a[2 + c] = load_row(&accum[(size_t)(2 + c) * stride]); }
        a[6] = load_row(&accum[(size_t)(2 + count) * stride]);
    }
    if (STAGE != kStageIncremental) touch_code_ahead(sh, lane);
    const int ra = unpack_local_ref(both & 0xFFFFu);
    const int rb = (F::bodies == 2) ? unpack_local_ref(both >> 16) : -1;
    SharedRef sa = {-1, 0u, false, false}, sb = {-1, 0u, false, false};
    if constexpr (SHARED) {
        sa = make_shared_ref<STAGE == kStageIncremental>(sh, both & 0xFFFFu, rank_a, active);
        if (F::bodies == 2) sb = make_shared_ref<STAGE == kStageIncremental>(sh, both >> 16, rank_b, active);
    }
    DBody A, B;
    if (STAGE == kStageIncremental) {
        load_body_lds<kAccessOnlyVelocity>(sh, ra, A);
        if (F::bodies == 2) load_body_lds<kAccessOnlyVelocity>(sh, rb, B); else load_body_lds<0>(sh, 0, B);
        if constexpr (SHARED) acquire_shared<F::bodies == 2>(sh, sa, A, sb, B, 8, k);
        F::incrementalUpdate(dt, A.vel, B.vel, p, count);
        if (active) {
            _Pragma("unroll") for (int c = 0; c < 4; ++c) { if (c < count) store_row(&prestep[(size_t)(4 * c + 3) * stride], p[4 * c + 3]); }
        }
        return total;
    }
    constexpr int accA = (STAGE == kStageWarmStart) ? F::wsA : F::svA;
    constexpr int accB = (STAGE == kStageWarmStart) ? F::wsB : F::svB;
    load_body_lds<accA & ~(kLin | kAng)>(sh, ra, A);
    if (F::bodies == 2) load_body_lds<accB & ~(kLin | kAng)>(sh, rb, B); else load_body_lds<0>(sh, 0, B);
    if (TRACE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps.loaded = __builtin_readcyclecounter(); }
    bool requirk_a = false, requirk_b = false;
    if constexpr (kConserving && STAGE == kStageWarmStart) {
        if (sh.substep == 0 && active) {
            requirk_a = SHARED ? (rank_a & kRankRequirk) != 0 : ((both & kLrefRequirk) != 0 && (both & 0x8000u) == 0);
            requirk_b = F::bodies == 2 && (SHARED ? (rank_b & kRankRequirk) != 0 : (((both >> 16) & kLrefRequirk) != 0 && (both & 0x80000000u) == 0));
        }
    }
    FusedGate<accA, accB, F::bodies, STAGE == kStageSolve, TRACE, SHARED> gate{sh, it, members, watch_npred, watch_nxpred, group_overflow, group_xoverflow, h0.batch, k, epoch, ra, rb, A, B, stamps, sa, sb, requirk_a, requirk_b};
    if (STAGE == kStageWarmStart) F::warmStart(A.inertia, B.inertia, p, a, count, A.vel, B.vel, gate);
    else F::solve(A.inertia, B.inertia, dt, inv_dt, p, a, count, A.vel, B.vel, gate);
    store_velocity_lds<accA>(sh, (active && !sa.shared()) ? ra : -1, A);
    if (F::bodies == 2) store_velocity_lds<accB>(sh, (active && !sb.shared()) ? rb : -1, B);
    if constexpr (SHARED) {
        store_velocity_lds<kLin | kAng>(sh, (sa.shared() && !sa.publish) ? ra : -1, A);
        if (F::bodies == 2) store_velocity_lds<kLin | kAng>(sh, (sb.shared() && !sb.publish) ? rb : -1, B);
        release_shared(sh, sa, A);
        if (F::bodies == 2) release_shared(sh, sb, B);
    }
    jitter_nap(sh, (unsigned)k * 2u + 1u + epoch * 0x632BE5ABu);
    publish_items(sh.flags + k, members + 1, epoch);
    __builtin_amdgcn_s_setprio(0);
    if (STAGE == kStageSolve && active) {
        store_row(&accum[0], a[0]); store_row(&accum[(size_t)stride], a[1]);
        _Pragma("unroll") for (int c = 0; c < 4; ++c) { if (c < count) store_row(&accum[(size_t)(2 + c) * stride], a[2 + c]); }
        store_row(&accum[(size_t)(2 + count) * stride], a[6]);
    }
    return total;
}

using DC1O = Contact<1, false>; using DC2O = Contact<2, false>; using DC3O = Contact<3, false>; using DC4O = Contact<4, false>;
using DC1T = Contact<1, true>; using DC2T = Contact<2, true>; using DC3T = Contact<3, true>; using DC4T = Contact<4, true>;

template <int STAGE, bool TRACE, bool WIDE, bool SHARED>
__device__ __forceinline__ void run_cluster_item(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h, int k, int lane, unsigned epoch,
                                                 unsigned* __restrict__ slab, float dt, float inv_dt, ItemStamps& stamps) {
#define BEPU_CASE(ID, F) case ID: if constexpr (type_compiled(ID)) run_cluster_constraint<F, STAGE, TRACE, SHARED>(sh, it, h, k, lane, epoch, slab, dt, inv_dt, stamps); break;
    switch (h.type_id) {
        BEPU_CASE(kContact1OneBody, DC1O) BEPU_CASE(kContact2OneBody, DC2O) BEPU_CASE(kContact3OneBody, DC3O) BEPU_CASE(kContact4OneBody, DC4O)
        BEPU_CASE(kContact1, DC1T) BEPU_CASE(kContact2, DC2T) BEPU_CASE(kContact3, DC3T) BEPU_CASE(kContact4, DC4T)
        default:
            if constexpr (WIDE) {
                bool nonconvex = true;
                switch (h.type_id) {
                    BD_NONCONVEX_CONTACT_TYPES(BEPU_CASE)
                    default: nonconvex = false; break;
                }
                if (nonconvex) break;
            }
            if constexpr (STAGE != kStageIncremental && !kContactsOnly) {  // only contacts need incremental updates (RequiresIncrementalSubstepUpdates)
                switch (h.type_id) {
                    BD_HOT_JOINT_TYPES(BEPU_CASE)
                    default:
                        if constexpr (WIDE) {  // SURVEY 8(f) types live in a second kernel variant: scenes made of the sixteen hot-path types keep the leaner one
                            switch (h.type_id) {
                                BD_WIDENED_JOINT_TYPES(BEPU_CASE)
#define BEPU_CASE_MANY(ID, F) case ID: if constexpr (type_compiled(ID)) run_cluster_constraint_many<F, STAGE, SHARED>(sh, it, h, k, lane, epoch, slab, dt, inv_dt); break;
                                BD_MANY_BODY_TYPES(BEPU_CASE_MANY)
#undef BEPU_CASE_MANY
                                default: break;
                            }
                        }
                        break;
                }
            }
            break;
    }
#undef BEPU_CASE
}

// A sweep over the cluster's batches (Solver_Solve.cs:1447-1476 for the cluster's islands): the items of a WarmStart pass (epoch `epoch`) followed,
// when `solve_items` > 0, by the items of the first velocity iteration (epoch + 1) in ONE claim sequence. No barrier separates the two: a Solve item
// waits for its same-pass predecessors and, for the bodies it is the first to touch, for their last toucher of the warm start (cross-pass
// predecessors), so the head of the iteration runs while the tail of the warm start's dependency chain is still draining. The warm start does
// not write accumulated impulses, hence nothing the iteration loads from HBM is in flight. STAGE0 = kStageSolve with solve_items = 0 runs a
// later iteration on its own (a barrier precedes it: its impulses were stored by the previous one).
template <int STAGE0, bool TRACE, bool WIDE, bool SHARED>
__device__ __forceinline__ void run_cluster_sweep(ClusterShared& sh, int item_count, int solve_items, int kernel_lane, int wave, unsigned epoch, unsigned claim_base,
                                                  unsigned* __restrict__ slab, float dt, float inv_dt, unsigned long long* trace) {
    const unsigned pass_base = sh.passes;
    for (;;) {
        const int v = (int)(claim_next(sh.counter) - claim_base);
        if (v >= item_count + solve_items) break;
        // Split units: the lane id is made afresh for every item. Carried from kernel entry it is a value the 168-VGPR unit keeps in scratch and reloads at the top of
        // every item — a memory round trip before the item's row loads are even issued (one scratch_load in every type's path in front of the gate, read in the
        // disassembly; every item of a plan that is short of wave time pays it).
        const int lane = SHARED ? gate_lane_id() : kernel_lane;
        const bool second = v >= item_count;
        const int k = second ? v - item_count : v;
        const unsigned item_epoch = second ? epoch + 1 : epoch;
        const ClusterItem* it = sh.items + k;
        const ItemHeader h = read_item(it);
        unsigned long long t0 = 0;
        if (TRACE) t0 = __builtin_readcyclecounter();
        ItemStamps stamps = {0, 0, 0};
        if constexpr (SHARED) sh.passes = pass_base + (second ? 1u : 0u);  // wave-private copy: which pass of the step this item belongs to
        int traced_type = h.type_id, traced_count = h.count;
        bool typed = true;
        if constexpr (SHARED) {  // merged manifold items: the leader's wave runs the group, a member is not an item of its own
            const int fuse = __builtin_amdgcn_readfirstlane(it->shape) >> kItemFuseShift;
            if (fuse & kItemFuseMember) continue;
            if (fuse & 3) {
                typed = false;
                const bool two = h.type_id >= kContact1;
                traced_type = kTraceFusedType | (two ? 1 : 0);
                if (STAGE0 == kStageWarmStart && !second) {
                    if (two) traced_count = run_cluster_fused<true, kStageWarmStart, TRACE, SHARED>(sh, it, h, fuse & 3, k, lane, item_epoch, slab, dt, inv_dt, stamps);
                    else traced_count = run_cluster_fused<false, kStageWarmStart, TRACE, SHARED>(sh, it, h, fuse & 3, k, lane, item_epoch, slab, dt, inv_dt, stamps);
                } else {
                    if (two) traced_count = run_cluster_fused<true, kStageSolve, TRACE, SHARED>(sh, it, h, fuse & 3, k, lane, item_epoch, slab, dt, inv_dt, stamps);
                    else traced_count = run_cluster_fused<false, kStageSolve, TRACE, SHARED>(sh, it, h, fuse & 3, k, lane, item_epoch, slab, dt, inv_dt, stamps);
                }
            }
        }
        if (typed) {
            if (STAGE0 == kStageWarmStart && !second) run_cluster_item<kStageWarmStart, TRACE, WIDE, SHARED>(sh, it, h, k, lane, item_epoch, slab, dt, inv_dt, stamps);
            else run_cluster_item<kStageSolve, TRACE, WIDE, SHARED>(sh, it, h, k, lane, item_epoch, slab, dt, inv_dt, stamps);
        }
        if (TRACE && trace && blockIdx.x == 0 && lane == 0 && item_epoch - 1 < (unsigned)kClusterTracePasses) {  // iteration counts are unbounded: never write past the buffer
            unsigned long long* rec = trace + ((size_t)(item_epoch - 1) * item_count + k) * 8;
            rec[4] = stamps.loaded; rec[5] = stamps.pre_gate; rec[6] = stamps.post_gate; rec[7] = 0;
            rec[0] = t0; rec[1] = __builtin_readcyclecounter();
            rec[2] = (unsigned long long)wave | ((unsigned long long)traced_type << 8) | ((unsigned long long)h.batch << 16) | ((unsigned long long)((STAGE0 == kStageWarmStart && !second) ? kStageWarmStart : kStageSolve) << 32);
            rec[3] = (unsigned long long)traced_count;
        }
    }
    if constexpr (SHARED) sh.passes = pass_base;
}

template <int THREADS, bool TRACE, bool WIDE, bool SHARED>
__global__ __launch_bounds__(THREADS) void cluster_kernel(const ClusterDesc* __restrict__ clusters, const ClusterItem* __restrict__ items,
                                                                   const int* __restrict__ batch_item_begin, const int* __restrict__ cluster_bodies,
                                                                   float4* bodies, unsigned* __restrict__ slab, ClusterParams cp, int ncap, int max_items,
                                                                   unsigned long long* trace, unsigned* status, unsigned long long* cycles, TailParams tp, SharedTables shared_tables) {
    if ((int)blockIdx.x + tp.block_offset >= tp.cluster_count) {
        // ---- not a cluster: the bodies no cluster owns (IntegrateBundlesAfterSubstepping for unconstrained bodies), and, in the last workgroup, the
        // constrained kinematic bodies. Clusters stage private copies of the kinematic bodies they reference from HBM when they start, so those are
        // advanced only after every cluster has reported its staging done (clusters are dispatched before this workgroup: it cannot starve them; with
        // tp.block_offset > 0 these workgroups are a launch of their own behind the clusters' on the same stream).
        const int t = (int)blockIdx.x + tp.block_offset - tp.cluster_count;
        if (t < tp.body_blocks) {
            if (!tp.final_launch) return;  // IntegrateAfterSubstepping belongs to the step's last launch
            const int i = t * (int)blockDim.x + (int)threadIdx.x;
            if (i < tp.body_count) {
                const unsigned f = tp.flags[i];
                if (!(f & (kFlagClustered | kFlagClusterKinematic)))
                    final_integrate_body(bodies, f, i, tp.dt, tp.substep_dt, tp.substep_count, tp.allow_substeps_for_unconstrained, tp.integrate_velocity_for_kinematics, tp.final_sp);
            }
            return;
        }
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(tp.staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tp.cluster_count) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > kSpinLimit) { report_stall(status, 0u, 5, t, tp.cluster_count, (unsigned)tp.cluster_count, __hip_atomic_load(tp.staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); break; }
            }
            __hip_atomic_store(tp.staged, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch (launches of one context are stream-ordered)
        }
        __syncthreads();
        for (int j = (int)threadIdx.x; j < tp.kin_count; j += (int)blockDim.x) {
            const int index = tp.kinlist[j] & kRefMask;
            kinematic_substeps_body(bodies, index, tp.launch_substeps, tp.integrate_velocity_for_kinematics, cp.sp, tp.substep_base);
            if (tp.final_launch) final_integrate_body(bodies, tp.flags[index], index, tp.dt, tp.substep_dt, tp.substep_count, tp.allow_substeps_for_unconstrained, tp.integrate_velocity_for_kinematics, tp.final_sp);
        }
        return;
    }
    const unsigned long long kernel_t0 = __builtin_readcyclecounter();
    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    ClusterShared sh;
    sh.planes = lds;
    sh.ncap = ncap;
    sh.items = reinterpret_cast<ClusterItem*>(lds + cp.planes * ncap);
    unsigned* words = reinterpret_cast<unsigned*>(lds + cp.planes * ncap + max_items * (int)(sizeof(ClusterItem) / 16));
    sh.flags = (volatile lds_u32*)words;
    sh.lbib = reinterpret_cast<int*>(words + max_items);
    sh.counter = (lds_u32*)(words + max_items + kClusterBatchTable + 1);
    sh.fallback_batch = cp.fallback_batch;
    sh.status = status;
    sh.batch_count = cp.batch_count;
    sh.st = shared_tables; sh.events = 0; sh.passes = 0;
    int* slot_body_lds = reinterpret_cast<int*>(words + ((cluster_sync_words(max_items) + 3) / 4) * 4);  // SHARED plans: behind the sync words
    sh.slot_body = slot_body_lds;
    sh.code_touch = cp.code_touch; sh.jitter = cp.jitter;
    sh.substep = 0; sh.angular_mode = cp.sp.angular_mode; sh.substep_dt = cp.sp.dt; sh.plane_count = cp.planes; sh.bodies = bodies;
    // The slot -> body table in LDS: always on split plans; on whole-island plans when the launcher found the room (ClusterParams.slot_table_in_lds, round 6) — every
    // integration phase and the write-back start from a body's slot entry, and read from global memory that is a full memory latency in front of each of them
    // (profiles/r06_s36_*: the pose half of the integration takes 8 k clocks whether sixteen or eight waves run it).
    const bool slot_table = SHARED || cp.slot_table_in_lds != 0;
    sh.scratch_row = lds_address((const volatile lds_u32*)lds) + (unsigned)cluster_lds_core_bytes(cp.planes, ncap, max_items, slot_table);
    const ClusterDesc cd = clusters[blockIdx.x];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    const float dt = cp.sp.dt, inv_dt = cp.sp.inv_dt;
    const int* slots = cluster_bodies + cd.body_begin;  // slot -> body index (bit 30: kinematic, private read-only copy; -1: unused slot)
    sh.slot_table = slots;
    // ---- stage the cluster in LDS: bodies (one plane per 16-byte field), work items, batch -> item ranges; clear the sync words ----
    for (int j = tid; j < cd.slot_count * cp.planes; j += blockDim.x) {
        const int slot = j / cp.planes, v = j - slot * cp.planes;
        const int g = slots[slot];
        lds[v * ncap + slot] = g >= 0 ? bodies[(size_t)(g & kSlotBodyMask) * 8 + (v < 4 ? v : (v < 6 ? v + 2 : v - 2))] : make_float4(0, 0, 0, 0);  // planes 4, 5 = record fields 6, 7 (world inertia); planes 6, 7 (if present) = fields 4, 5 (local inertia)
    }
    {
        const int4* src = reinterpret_cast<const int4*>(items + cd.item_begin);
        int4* dst = reinterpret_cast<int4*>(sh.items);
        for (int j = tid; j < cd.item_count * (int)(sizeof(ClusterItem) / 16); j += blockDim.x) dst[j] = src[j];
    }
    for (int j = tid; j < max_items; j += blockDim.x) words[j] = 0;  // flags
    sh.item_count = cd.item_count;
    for (int j = tid; j <= cp.batch_count; j += blockDim.x) sh.lbib[j] = batch_item_begin[cd.batch_item_offset + j] - cd.item_begin;
    if (tid == 0) *sh.counter = 0;
    if (slot_table) { for (int j = tid; j < cd.slot_count; j += blockDim.x) slot_body_lds[j] = slots[j]; }
    auto slot_entry = [&](int j) { return slot_table ? slot_body_lds[j] : slots[j]; };
    __syncthreads();
    if (tid == 0 && tp.kin_count > 0) __hip_atomic_fetch_add(tp.staged, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // this cluster's copies of kinematic bodies are in LDS

    if constexpr (kPass) {
        // The sweep behaves like the one pass of a one-substep step: a shared body's home publishes the velocity the launch found in HBM as "integration done"
        // (event base + 1), the applications number on from there, the home waits for all of them before it writes the body back.
        if constexpr (SHARED) {
            for (int j = tid; j < cd.slot_count; j += blockDim.x) {
                const int g = slots[j];
                if (g < 0 || !(g & kSlotSharedHome)) continue;
                const float4 l4 = lds[2 * ncap + j], a4 = lds[3 * ncap + j];
                const float number = __uint_as_float(shared_tables.base + 1u);
                publish_record_pair(shared_tables, shared_record(shared_tables, g & kSlotBodyMask, 0u), make_float4(l4.x, l4.y, l4.z, number), make_float4(a4.x, a4.y, a4.z, number));
            }
            sh.events = 1u;
        }
        sh.substep = cp.pass_substep;
        __syncthreads();
        if (cp.pass_stage == kStageWarmStart) run_cluster_sweep<kStageWarmStart, false, WIDE, SHARED>(sh, cd.item_count, 0, lane, wave, 1u, 0u, slab, dt, inv_dt, nullptr);
        else run_cluster_sweep<kStageSolve, false, WIDE, SHARED>(sh, cd.item_count, 0, lane, wave, 1u, 0u, slab, dt, inv_dt, nullptr);
        sh.passes = 1u;
        __syncthreads();
        for (int j = tid; j < cd.slot_count; j += blockDim.x) {
            int g = slots[j];
            if (SHARED && g >= 0 && (g & kSlotGhost)) continue;
            const bool home = SHARED && g >= 0 && (g & kSlotSharedHome) != 0;
            if (home) g &= kSlotBodyMask;
            if ((unsigned)g >= kDynamicLimit) continue;
            float4 l4 = lds[2 * ncap + j], a4 = lds[3 * ncap + j];
            if (home) {
                const float lw = l4.w, aw = a4.w;
                acquire_shared_one(shared_tables, status, g, 0u, shared_tables.base + 1u + (shared_tables.info[g] & 0xFFu), l4, a4, 11, j);
                l4.w = lw; a4.w = aw;
            }
            float4* gb = bodies + (size_t)g * 8;
            gb[2] = l4; gb[3] = a4;
        }
        return;
    }
    unsigned epoch = 0, claim_base = 0;
    for (int s = 0; s < cp.substeps; ++s) {
        const int gs = cp.substep_base + s;  // the substep's index in the STEP (a chained step: this launch starts at substep_base); `s` counts this launch's substeps (records, events)
        if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(&sh.status[10], (unsigned)gs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (trace builds: where the time between two sweeps goes — pass slot kClusterTracePasses - 16 + s holds, per wave, the clock at the top of the substep,
        // behind the incremental contact update's barrier and behind the integration's barrier; tools/cluster_trace.py prints them)
        unsigned long long* phase = (TRACE && trace && blockIdx.x == 0 && lane == 0 && s < 16 && nwaves * 4 <= cd.item_count * 8) ? trace + (size_t)(kClusterTracePasses - 16 + s) * cd.item_count * 8 + wave * 4 : nullptr;
        if (phase) phase[0] = __builtin_readcyclecounter();
        // Solver_Solve.cs:1427-1439: contact depths advance with the pre-integration velocities — the incremental update of the contact items, one item per wave at a time.
        auto incremental_items = [&]() {
            for (int k = wave; k < cd.item_count; k += nwaves) {
                const ClusterItem* it = sh.items + k;
                const ItemHeader h = read_item(it);
                if (!isContactType(h.type_id)) continue;
                ItemStamps stamps = {0, 0, 0};
                const int lane = SHARED ? gate_lane_id() : (tid & 63);  // (split units: afresh per item, see run_cluster_sweep)
                if constexpr (SHARED) {
                    const int fuse = __builtin_amdgcn_readfirstlane(it->shape) >> kItemFuseShift;
                    if (fuse & kItemFuseMember) continue;
                    if (fuse & 3) {
                        if (h.type_id >= kContact1) run_cluster_fused<true, kStageIncremental, false, SHARED>(sh, it, h, fuse & 3, k, lane, 0u, slab, dt, inv_dt, stamps);
                        else run_cluster_fused<false, kStageIncremental, false, SHARED>(sh, it, h, fuse & 3, k, lane, 0u, slab, dt, inv_dt, stamps);
                        continue;
                    }
                }
                run_cluster_item<kStageIncremental, false, WIDE, SHARED>(sh, it, h, k, lane, 0u, slab, dt, inv_dt, stamps);
            }
        };
        // Integration of every constrained body of the cluster (TypeProcessor.cs:1204-1283, PoseIntegrator.cs:451-535):
        // substep 0 velocity only, later substeps pose then velocity; world inverse inertia refreshed either way.
        // PART 0: all of it. PART 1: the pose half (pose, world inverse inertia; a shared body's end-of-substep velocity moves from its record into the home's LDS slot).
        // PART 2: the velocity half (IntegrateVelocity, the home's "integration done" record). Same arithmetic per body in the same order whichever way it is cut.
        auto integrate_bodies = [&](auto part_tag) {
            constexpr int PART = decltype(part_tag)::value;
            for (int j0 = tid; j0 < cd.slot_count; j0 += blockDim.x) {
                // (opaque: the planes' LDS addresses of slot j are otherwise computed once per launch, in front of the substep loop, and kept in scratch by the 128-VGPR
                // units — every integration phase then starts with a chain of scratch loads, each a memory round trip; recomputed here they cost a few integer instructions)
                int j = j0;
                asm volatile("" : "+v"(j));
                const int g = slot_entry(j);
                if (g < 0) continue;
                const bool ghost = SHARED && (g & kSlotGhost) != 0;  // another cluster owns the body; this one keeps its pose and world inertia current by the same arithmetic
                if (PART == 2 && ghost) continue;
                float4* r = lds + j;
                const bool home = SHARED && (g & kSlotSharedHome) != 0;
                const int body = g & kSlotBodyMask;
                const bool dynamic = (unsigned)(g & ~(kSlotSharedHome | kSlotGhost)) < kDynamicLimit;
                if (PART == 2 && !dynamic && !cp.integrate_velocity_for_kinematics) continue;
                float4 q4 = make_float4(0, 0, 0, 1), p4 = r[ncap], l4 = r[2 * ncap], a4 = r[3 * ncap];
                if (PART != 2) q4 = r[0];
                unsigned applications = 0;  // shared bodies: applications per pass
                if (home || ghost) applications = shared_tables.info[body] & 0xFFu;
                if (PART != 2 && (home || ghost) && s > 0) {  // the velocity the last substep ended with: in last substep's record once every application on the body has happened
                    const float lw = l4.w, aw = a4.w;
                    acquire_shared_one(shared_tables, status, body, (unsigned)s - 1u, shared_tables.base + (unsigned)s + applications * sh.passes, l4, a4, 9, j);
                    l4.w = lw; a4.w = aw;  // the record's fourth lanes carry the event number; the body's own padding stays what it was
                    if (PART == 1 && home) { r[2 * ncap] = l4; r[3 * ncap] = a4; }  // (the velocity half finds it there)
                }
                Q ori = {q4.x, q4.y, q4.z, q4.w};
                V3 pos = {p4.x, p4.y, p4.z};
                BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
                if (PART != 2 && gs > 0) {
                    pos = add(pos, scale(vel.lin, dt));
                    ori = integrateOrientation(ori, vel.ang, dt * 0.5f);
                    r[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
                    r[ncap] = make_float4(pos.x, pos.y, pos.z, p4.w);
                }
                if (dynamic) {
                    if constexpr (PART != 2) {
                        // local inverse inertia and mass: constant over the step; in LDS when the cluster left room for the two planes, else read where needed
                        const float4 i0 = cp.planes == kAllPlanes ? r[6 * ncap] : bodies[(size_t)body * 8 + 4], i1 = cp.planes == kAllPlanes ? r[7 * ncap] : bodies[(size_t)body * 8 + 5];
                        Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
                        Sym3 world = rotateInverseInertia(local, ori);
                        r[4 * ncap] = make_float4(world.xx, world.yx, world.yy, world.zx);
                        r[5 * ncap] = make_float4(world.zy, world.zz, i1.z, r[5 * ncap].w);
                        if constexpr (kConserving) {  // substep_integrate_dynamic's angular step (TypeProcessor.cs:1224-1238, 1264-1272); a ghost's velocity belongs to its home
                            static_assert(PART == 0, "the conserving modes integrate a body in one piece (the angular step reads the orientation the substep started from)");
                            if (!ghost) {
                                const Q before = {q4.x, q4.y, q4.z, q4.w};  // substep > 0: the orientation the step started from; substep 0: "integrating backwards" from the current one
                                if (cp.sp.angular_mode == 1)
                                    vel.ang = integrateAngularVelocityConserveMomentum(gs > 0 ? before : integrateOrientation(ori, vel.ang, dt * -0.5f), local, world, vel.ang);
                                else if (cp.sp.angular_mode == 2)
                                    vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(ori, local, vel.ang, dt);
                            }
                        }
                    }
                    if constexpr (PART != 1) {
                        if (!ghost) {
                            velocity_callback(cp.sp, vel, pos, body);
                            r[2 * ncap] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
                            r[3 * ncap] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
                        }
                        if (home) {  // this substep's record: the integrated velocity, and "integration done" as the event number
                            const float number = __uint_as_float(shared_tables.base + (unsigned)s + 1u + applications * sh.passes);
                            publish_record_pair(shared_tables, shared_record(shared_tables, body, (unsigned)s), make_float4(vel.lin.x, vel.lin.y, vel.lin.z, number), make_float4(vel.ang.x, vel.ang.y, vel.ang.z, number));
                        }
                    }
                } else if (PART != 1 && cp.integrate_velocity_for_kinematics) {  // kinematic: private copy, same arithmetic as the global kinematic pass
                    velocity_callback(cp.sp, vel, pos, body);
                    r[2 * ncap] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
                    r[3 * ncap] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
                }
            }
        };
        // Between two substeps (round 6): the contact items' incremental update waits for memory (its rows: 2 - 4 k clocks per item, profiles/r06_s35_*), the pose half of
        // the integration is arithmetic (about 500 instructions per body), and neither touches what the other writes — the update reads velocities and writes depth rows,
        // the pose half reads velocities and writes poses and inertias. So half the waves of every SIMD take their bodies' pose half first and their contact items second,
        // the other half the other way round: one's arithmetic under the other's loads. The velocity half (IntegrateVelocity: a few instructions) waits behind a barrier
        // for every update to have read the velocities it is about to change. The conserving modes' angular step needs the body in one piece: old order there.
        const bool halves = !kConserving && gs > 0 && cp.split_integration != 0;  // (wave-uniform, the same for every wave of the workgroup: the barriers below are taken by all or none)
        const bool pose_first = halves && ((wave >> 2) & 1) != 0;  // waves w, w + 4, ... share a SIMD: every SIMD gets both kinds
        if (pose_first) integrate_bodies(std::integral_constant<int, kConserving ? 0 : 1>{});
        if (gs > 0) {
            incremental_items();
            if (phase) phase[3] = __builtin_readcyclecounter();  // this wave's incremental items are done
        }
        if (halves && !pose_first) integrate_bodies(std::integral_constant<int, kConserving ? 0 : 1>{});
        if (gs > 0) __syncthreads();
        if (phase) phase[1] = __builtin_readcyclecounter();
        if (halves) integrate_bodies(std::integral_constant<int, kConserving ? 0 : 2>{});
        else integrate_bodies(std::integral_constant<int, 0>{});
        if constexpr (SHARED) sh.events = (unsigned)s + 1u;
        sh.substep = gs;
        __syncthreads();
        if (phase) phase[2] = __builtin_readcyclecounter();
        ++epoch;
        const int fused = cp.iters[s] > 0 ? cd.item_count : 0;  // the first velocity iteration rides in the warm start's claim sequence
        run_cluster_sweep<kStageWarmStart, TRACE, WIDE, SHARED>(sh, cd.item_count, fused, lane, wave, epoch, claim_base, slab, dt, inv_dt, trace);
        claim_base += cd.item_count + fused + nwaves;  // every wave makes exactly one failing claim per sweep
        sh.passes += fused ? 2u : 1u;
        if (fused) ++epoch;
        __syncthreads();
        for (int iter = 1; iter < cp.iters[s]; ++iter) {
            ++epoch;
            run_cluster_sweep<kStageSolve, TRACE, WIDE, SHARED>(sh, cd.item_count, 0, lane, wave, epoch, claim_base, slab, dt, inv_dt, trace);
            claim_base += cd.item_count + nwaves;
            sh.passes += 1u;
            __syncthreads();
        }
    }
    // Trailing pose integration of constrained bodies (PoseIntegrator.cs:684-691) and write-back.
    for (int j = tid; j < cd.slot_count; j += blockDim.x) {
        int g = slot_entry(j);
        if (SHARED && g >= 0 && (g & kSlotGhost)) continue;  // written back by its home cluster
        const bool home = SHARED && g >= 0 && (g & kSlotSharedHome) != 0;
        if (home) g &= kSlotBodyMask;
        if ((unsigned)g >= kDynamicLimit) continue;  // unused slot, or kinematic (advanced in global memory by this launch's kinematic workgroup)
        const float4* r = lds + j;
        float4 q4 = r[0], p4 = r[ncap], l4 = r[2 * ncap], a4 = r[3 * ncap];
        if (home) {  // the last applications on a shared body may belong to other clusters: wait for the step's full event count, then take its velocity
            const unsigned want = shared_tables.base + (unsigned)cp.substeps + (shared_tables.info[g] & 0xFFu) * sh.passes;
            const float lw = l4.w, aw = a4.w;
            acquire_shared_one(shared_tables, status, g, (unsigned)cp.substeps - 1u, want, l4, a4, 11, j);
            l4.w = lw; a4.w = aw;
        }
        Q ori = {q4.x, q4.y, q4.z, q4.w};
        V3 pos = {p4.x, p4.y, p4.z};
        V3 lin = {l4.x, l4.y, l4.z}, ang = {a4.x, a4.y, a4.z};
        if (cp.final_launch) {  // (a chained step's earlier launches leave the pose of their last substep: the next launch's first substep integrates it on)
            ori = integrateOrientation(ori, ang, dt * 0.5f);
            pos = add(pos, scale(lin, dt));
        }
        float4* gb = bodies + (size_t)g * 8;
        gb[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
        gb[1] = make_float4(pos.x, pos.y, pos.z, p4.w);
        gb[2] = l4;
        gb[3] = a4;
        gb[6] = r[4 * ncap];
        gb[7] = r[5 * ncap];
    }
    if (tid == 0) cycles[blockIdx.x] = __builtin_readcyclecounter() - kernel_t0;  // shader clocks this cluster took: a clock-frequency-independent measure
}

}  // namespace
