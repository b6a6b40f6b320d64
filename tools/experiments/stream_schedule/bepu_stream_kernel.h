// Persistent launch-per-batch schedule ("stream schedule"): the launch sequence of enqueue_solve (Solver_Solve.cs:1415-1479 + PoseIntegrator.cs:707-726)
// executed by ONE cooperative launch of resident wavefronts. Every launch of the sequence becomes a "hop"; the device-wide ordering between two
// hops, which the launch-per-batch schedule gets from kernel boundaries (8.7 us per dependent launch on a 100k-box pile), comes from an arrival
// counter per hop instead.
//
// What makes that possible on a part with eight L2s (tools/probes/xcd_handoff_probe.hip, profiles/r01_xcd_handoff_probe.txt): body records that cross
// workgroups are only ever moved with agent-scope (sc1) loads and stores, which bypass the per-CU L1 and are coherent across the XCDs without any
// release/acquire fence (no L2 write-back / invalidate: 1.5 us per hand-off instead of 4.8-5.3 us). Per-constraint data (prestep, accumulated
// impulses) never crosses wavefronts inside a launch: block `vb` of batch `b` is run by the same wavefront in every hop, so plain cached accesses
// are coherent for it by construction. Opt-in (BEPUHIP_FLAG_STREAM): bit-exact, 10 % slower than the graph replay on the pile.
#pragma once

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct BodyPlanes { f4 ori, pos, lin, ang, w0, w1; };  // planes 0,1,2,3,6,7 of the 128-byte body record
__device__ __forceinline__ f4 make_f4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }

// Velocities of the two bodies of a constraint, fetched by the lane that owns it. One asm statement for the loads AND the wait, so that the compiler
// can never touch a destination register before its data has landed.
__device__ __forceinline__ void sc1_load_velocity2(const float4* baseA, const float4* baseB, f4& linA, f4& angA, f4& linB, f4& angB) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off offset:32 sc1\n\t"
        "global_load_dwordx4 %1, %4, off offset:48 sc1\n\t"
        "global_load_dwordx4 %2, %5, off offset:32 sc1\n\t"
        "global_load_dwordx4 %3, %5, off offset:48 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(linA), "=&v"(angA), "=&v"(linB), "=&v"(angB)
        : "v"(baseA), "v"(baseB)
        : "memory");
}
// (the s_nop covers the wait state a wide VMEM store needs before its data registers may be rewritten; the compiler cannot see inside the asm)
__device__ __forceinline__ void sc1_store(float4* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

// ---- cooperative gather: whole 128-byte records, eight lanes per record, staged through LDS ----
// A lane that fetches the planes of its own bodies sends one 16-byte request per plane: 8-12 uncached requests per constraint, and the request rate
// of the fabric, not its bandwidth, is what a hop then waits for. Here eight consecutive lanes fetch the eight planes of ONE record (a single
// 128-byte request), the wavefront's 64 or 128 records land in its LDS slice plane-major, and every lane reads the planes its constraint needs from there.
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f4 lds_f4;
constexpr int kStagePlaneStride = 129;                    // 128 record slots + one: the eight planes a lane group writes fall into different banks
constexpr int kStageWaveF4 = 8 * kStagePlaneStride;       // one wavefront's slice, in 16-byte units
struct StreamCtx {
    __amdgpu_buffer_rsrc_t rsrc;  // the body array as a raw buffer: 128-bit loads with the sc1 bit, compiler-managed waits, out-of-range reads return 0
    lds_f4* stage;                // this wavefront's LDS slice
    int lane;
};
constexpr int kAuxSc1 = 16;  // cache-policy operand of the raw buffer builtins on gfx94x/gfx950: bit 0 sc0, bit 1 nt, bit 4 sc1

// Byte offsets of the requests this lane sends for a block of constraints (slot j < 64: body A of lane j, slot 64 + j: body B of lane j).
template <int NB>
__device__ __forceinline__ void stage_offsets(int refA, int refB, int lane, int (&off)[NB * 8]) {
    const int sub = lane >> 3, plane = lane & 7;
    _Pragma("unroll") for (int r = 0; r < NB * 8; ++r) {
        const int ref = __shfl(r < 8 ? refA : refB, (r & 7) * 8 + sub);
        off[r] = (ref & kRefMask) * 128 + plane * 16;
    }
}
template <int ROUNDS>
__device__ __forceinline__ void stage_records(const StreamCtx& cx, const int (&off)[ROUNDS]) {
    i4 v[ROUNDS];
    _Pragma("unroll") for (int r = 0; r < ROUNDS; ++r) v[r] = __builtin_amdgcn_raw_buffer_load_b128(cx.rsrc, off[r], 0, kAuxSc1);
    const int sub = cx.lane >> 3, plane = cx.lane & 7;
    _Pragma("unroll") for (int r = 0; r < ROUNDS; ++r) {
        const f4 t = {__int_as_float(v[r].x), __int_as_float(v[r].y), __int_as_float(v[r].z), __int_as_float(v[r].w)};
        cx.stage[plane * kStagePlaneStride + r * 8 + sub] = t;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slice is written before any lane of this wavefront reads it (LDS serves a wavefront in order)
    __builtin_amdgcn_wave_barrier();
}
template <int ACCESS>
__device__ __forceinline__ void staged_planes(const StreamCtx& cx, int slot, BodyPlanes& r) {  // only the planes load_body<ACCESS> would gather
    const f4 zero = make_f4(0, 0, 0, 0);
    r.ori = (ACCESS & kOri) ? cx.stage[0 * kStagePlaneStride + slot] : zero;
    r.pos = (ACCESS & kPos) ? cx.stage[1 * kStagePlaneStride + slot] : zero;
    r.lin = (ACCESS & kLin) ? cx.stage[2 * kStagePlaneStride + slot] : zero;
    r.ang = (ACCESS & kAng) ? cx.stage[3 * kStagePlaneStride + slot] : zero;
    r.w0 = (ACCESS & kInertia) ? cx.stage[6 * kStagePlaneStride + slot] : zero;
    r.w1 = (ACCESS & kInertia) ? cx.stage[7 * kStagePlaneStride + slot] : zero;
}

template <int ACCESS>
__device__ __forceinline__ void planes_to_body(const BodyPlanes& r, DBody& b) {  // the same field selection as load_body<ACCESS>
    if (ACCESS & kOri) b.ori = {r.ori.x, r.ori.y, r.ori.z, r.ori.w}; else b.ori = {0, 0, 0, 0};
    if (ACCESS & kPos) b.pos = {r.pos.x, r.pos.y, r.pos.z}; else b.pos = {0, 0, 0};
    if (ACCESS & kLin) { b.vel.lin = {r.lin.x, r.lin.y, r.lin.z}; b.linw = r.lin.w; } else { b.vel.lin = {0, 0, 0}; b.linw = 0; }
    if (ACCESS & kAng) { b.vel.ang = {r.ang.x, r.ang.y, r.ang.z}; b.angw = r.ang.w; } else { b.vel.ang = {0, 0, 0}; b.angw = 0; }
    if (ACCESS & kInertia) { b.inertia.t = {r.w0.x, r.w0.y, r.w0.z, r.w0.w, r.w1.x, r.w1.y}; b.inertia.invMass = r.w1.z; }
    else { b.inertia.t = {0, 0, 0, 0, 0, 0}; b.inertia.invMass = 0; }
}
template <int ACCESS>
__device__ __forceinline__ void sc1_store_velocity(float4* bodies, int ref, const DBody& b) {  // ScatterVelocities: never for kinematic references
    if ((unsigned)ref >= kDynamicLimit) return;
    float4* base = bodies + (size_t)ref * 8;
    if (ACCESS & kLin) sc1_store(base + 2, make_f4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, b.linw));
    if (ACCESS & kAng) sc1_store(base + 3, make_f4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, b.angw));
}

// ---- hop ordering ----
constexpr int kHopLanes = 16;  // arrival counter of a hop = 16 dwords (one 64-byte line): arrivals spread over them, a waiter reads the line once per poll
struct StreamSync {
    unsigned* counters;  // [hops][16], zeroed before the launch
    unsigned* status;    // host-visible watchdog words (shared with the island schedule): [0] != 0 = stalled
    unsigned long long* trace;  // diagnostics (bepuhip_set_cluster_trace): per traced wavefront and hop, four 100 MHz stamps; null = off
};
constexpr int kStreamTraceWaves = 3, kStreamTraceHops = 1024;

__device__ __forceinline__ unsigned wave_sum16(unsigned v) {  // sum over lanes 0..15, uniform result
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// Blocks until every block of hop `hop` has arrived. False = watchdog (another wavefront never arrived): the caller unwinds.
__device__ __forceinline__ bool hop_wait(const StreamSync& sy, int hop, unsigned expected, int lane) {
    if (hop < 0) return true;
    unsigned* line = sy.counters + (size_t)hop * kHopLanes;
    // (Polls share a memory channel with the arrivals they wait for. The poller is wavefront 0, which reaches this point after the velocity-independent
    // half of its own block: late enough that pacing the polls, or a pause before the first one, measured no better.)
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = lane < kHopLanes ? __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (wave_sum16(v) >= expected) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(sy.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;  // somebody already gave up
            if (spins > kSpinLimit) { report_stall(sy.status, 0u, 6, hop, 0, expected, wave_sum16(v)); return false; }  // kind 6: a hop of the stream schedule
        }
    }
}
// One arrival per workgroup, after its wavefronts have drained their stores and met at the workgroup barrier.
__device__ __forceinline__ void hop_arrive(const StreamSync& sy, int hop, unsigned blocks_done, int wg) {
    __hip_atomic_fetch_add(sy.counters + (size_t)hop * kHopLanes + (wg & (kHopLanes - 1)), blocks_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool type_is_incremental(int type_id) { return type_id >= kContact1OneBody && type_id <= kContact4; }  // the convex contact ids are contiguous

// ---- constraint blocks ----
// The gate is called once per block, by all 64 lanes, after the block's own rows (references, prestep, accumulated impulses: never shared) have been
// requested and before the first body access: it is where the wavefront waits for the previous hop.
// The gate handed INTO a constraint function (the island schedule's idea, bepu_cluster_kernel.h): poses, inertias and everything the function derives
// from them and from its prestep rows are computed before it is called; it waits for the previous hop and fetches the velocities, nothing else.
// Legal whenever the previous hop was not an integration hop (poses and inertias only change there).
template <int ACC_A, int ACC_B, int BODIES, class WAIT>
struct StreamGate {
    static constexpr bool kPin = true;
    const float4* baseA; const float4* baseB; DBody& A; DBody& B; WAIT& wait; bool& ok;
    __device__ __forceinline__ void operator()(BodyVel& vA, BodyVel& vB) const {
        ok = wait();
        f4 la, aa, lb, ab;
        sc1_load_velocity2(baseA, baseB, la, aa, lb, ab);
        if (ACC_A & kLin) { vA.lin = {la.x, la.y, la.z}; A.linw = la.w; }
        if (ACC_A & kAng) { vA.ang = {aa.x, aa.y, aa.z}; A.angw = aa.w; }
        if (BODIES == 2) {
            if (ACC_B & kLin) { vB.lin = {lb.x, lb.y, lb.z}; B.linw = lb.w; }
            if (ACC_B & kAng) { vB.ang = {ab.x, ab.y, ab.z}; B.angw = ab.w; }
        }
    }
};

// PRE: the block may read poses and inertias before the hop wait (see StreamGate).
template <class F, int STAGE, bool PRE, class GATE>
__device__ __forceinline__ bool stream_constraint(const DevTypeBatch& tb, int i, float4* bodies, const StreamCtx& cx, float dt, float inv_dt, GATE&& gate) {
    const int stride = tb.stride;
    const bool valid = i < tb.count;
    const int row = valid ? i : 0;
    const int refA = tb.refs[row];
    const int refB = (F::bodies == 2) ? tb.refs[stride + row] : refA;
    float p[F::prestepFloats];
    _Pragma("unroll") for (int f = 0; f < F::prestepFloats; ++f) p[f] = tb.prestep[(size_t)f * stride + row];
    float a[F::impulseFloats];
    if (STAGE != kStageIncremental) { _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) a[f] = tb.accum[(size_t)f * stride + row]; }
    const float4* baseA = bodies + (size_t)(refA & kRefMask) * 8;
    const float4* baseB = bodies + (size_t)(refB & kRefMask) * 8;
    int off[F::bodies * 8];
    if (STAGE != kStageIncremental) stage_offsets<F::bodies>(refA, refB, cx.lane, off);
    DBody A, B;
    if constexpr (PRE && STAGE != kStageIncremental) {
        constexpr int accA = (STAGE == kStageWarmStart) ? F::wsA : F::svA;
        constexpr int accB = (STAGE == kStageWarmStart) ? F::wsB : F::svB;
        constexpr int kStatic = ~(kLin | kAng);
        stage_records<F::bodies * 8>(cx, off);  // the velocity planes that come along may be stale: they are not read
        BodyPlanes ra, rb;
        staged_planes<accA & kStatic>(cx, cx.lane, ra);
        planes_to_body<accA & kStatic>(ra, A);
        if (F::bodies == 2) { staged_planes<accB & kStatic>(cx, 64 + cx.lane, rb); planes_to_body<accB & kStatic>(rb, B); } else planes_to_body<0>(ra, B);
        bool ok = true;
        StreamGate<accA, accB, F::bodies, std::remove_reference_t<GATE>> inner{baseA, baseB, A, B, gate, ok};
        // lanes past the end of the type batch run the function on row 0 with a full exec mask (the wait inside is wave-wide) and store nothing
        if (STAGE == kStageWarmStart) F::warmStart(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, p, a, A.vel, B.vel, inner);
        else F::solve(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, dt, inv_dt, p, a, A.vel, B.vel, inner);
        if (!ok) return false;
        if (!valid) return true;
        if (STAGE == kStageSolve) { _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) tb.accum[(size_t)f * stride + i] = a[f]; }
        sc1_store_velocity<accA>(bodies, refA, A);
        if (F::bodies == 2) sc1_store_velocity<accB>(bodies, refB, B);
        return true;
    }
    if (!gate()) return false;
    if (STAGE == kStageIncremental) {
        if (!valid) return true;
        if constexpr (F::incremental) {
            f4 la, aa, lb, ab;
            sc1_load_velocity2(baseA, baseB, la, aa, lb, ab);
            A.vel = {{la.x, la.y, la.z}, {aa.x, aa.y, aa.z}};
            if (F::bodies == 2) B.vel = {{lb.x, lb.y, lb.z}, {ab.x, ab.y, ab.z}}; else B.vel = {{0, 0, 0}, {0, 0, 0}};
            F::incrementalUpdate(dt, A.vel, B.vel, p);
            _Pragma("unroll") for (int cidx = 0; cidx < F::contacts; ++cidx) tb.prestep[(size_t)F::depthRow(cidx) * stride + i] = p[F::depthRow(cidx)];
        }
        return true;
    }
    constexpr int accA = (STAGE == kStageWarmStart) ? F::wsA : F::svA;
    constexpr int accB = (STAGE == kStageWarmStart) ? F::wsB : F::svB;
    stage_records<F::bodies * 8>(cx, off);  // every lane fetches for its lane group, valid constraint or not
    if (!valid) return true;
    BodyPlanes ra, rb;
    staged_planes<accA>(cx, cx.lane, ra);
    planes_to_body<accA>(ra, A);
    if (F::bodies == 2) { staged_planes<accB>(cx, 64 + cx.lane, rb); planes_to_body<accB>(rb, B); } else planes_to_body<0>(ra, B);
    if (STAGE == kStageWarmStart) {
        F::warmStart(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, p, a, A.vel, B.vel, NoGate{});
    } else {
        F::solve(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, dt, inv_dt, p, a, A.vel, B.vel, NoGate{});
        _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) tb.accum[(size_t)f * stride + i] = a[f];
    }
    sc1_store_velocity<accA>(bodies, refA, A);
    if (F::bodies == 2) sc1_store_velocity<accB>(bodies, refB, B);
    return true;
}

template <int STAGE, bool PRE, class GATE>
__device__ __forceinline__ bool stream_constraint_block(const DevTypeBatch& tb, int i, float4* bodies, const StreamCtx& cx, float dt, float inv_dt, GATE&& gate) {
    switch (tb.type_id) {
        case kContact1OneBody: return stream_constraint<Contact<1, false>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        case kContact2OneBody: return stream_constraint<Contact<2, false>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        case kContact3OneBody: return stream_constraint<Contact<3, false>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        case kContact4OneBody: return stream_constraint<Contact<4, false>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        case kContact1: return stream_constraint<Contact<1, true>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        case kContact2: return stream_constraint<Contact<2, true>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        case kContact3: return stream_constraint<Contact<3, true>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        case kContact4: return stream_constraint<Contact<4, true>, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
        default: break;
    }
    if (STAGE != kStageIncremental) {
        switch (tb.type_id) {
#define X(ID, T) case ID: return stream_constraint<T, STAGE, PRE>(tb, i, bodies, cx, dt, inv_dt, gate);
            BD_HOT_JOINT_TYPES(X)
#undef X
            default: break;
        }
    }
    return gate();  // a type without work in this stage still takes part in the hop
}

// ---- body blocks: substep_integrate_kernel / final_integrate_kernel, one lane per body, on sc1 accesses ----
// The 64 records of a body block are contiguous: eight coalesced 1 KiB requests per wavefront bring them into the LDS slice (slot = body - first).
__device__ __forceinline__ void stage_body_block(const StreamCtx& cx, int first_body) {
    int off[8];
    _Pragma("unroll") for (int r = 0; r < 8; ++r) off[r] = (first_body + r * 8 + (cx.lane >> 3)) * 128 + (cx.lane & 7) * 16;
    stage_records<8>(cx, off);
}
__device__ __forceinline__ void stream_integrate_body(float4* bodies, const StreamCtx& cx, unsigned f, int i, int integrate_pose, int integrate_velocity_for_kinematics,
                                                      const StepParams& sp) {
    if (!(f & (kFlagDynamicConstrained | kFlagConstrainedKinematic))) return;
    float4* base = bodies + (size_t)i * 8;
    const f4 l0 = cx.stage[4 * kStagePlaneStride + cx.lane], l1 = cx.stage[5 * kStagePlaneStride + cx.lane];  // local inverse inertia: never written
    const float4 i0 = make_float4(l0.x, l0.y, l0.z, l0.w), i1 = make_float4(l1.x, l1.y, l1.z, l1.w);
    BodyPlanes r;
    staged_planes<kAccessAll>(cx, cx.lane, r);
    BodyRegs b = {{r.ori.x, r.ori.y, r.ori.z, r.ori.w}, {r.pos.x, r.pos.y, r.pos.z}, {{r.lin.x, r.lin.y, r.lin.z}, {r.ang.x, r.ang.y, r.ang.z}}};
    if (f & kFlagDynamicConstrained) {
        const Sym3 world = substep_integrate_dynamic(b, i0, i1, integrate_pose, sp);
        if (integrate_pose) {
            sc1_store(base + 0, make_f4(b.ori.x, b.ori.y, b.ori.z, b.ori.w));
            sc1_store(base + 1, make_f4(b.pos.x, b.pos.y, b.pos.z, r.pos.w));
        }
        sc1_store(base + 2, make_f4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, r.lin.w));
        sc1_store(base + 3, make_f4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, r.ang.w));
        sc1_store(base + 6, make_f4(world.xx, world.yx, world.yy, world.zx));
        sc1_store(base + 7, make_f4(world.zy, world.zz, i1.z, r.w1.w));
    } else {
        substep_integrate_kinematic(b, integrate_pose, integrate_velocity_for_kinematics, sp);
        if (integrate_pose) {
            sc1_store(base + 0, make_f4(b.ori.x, b.ori.y, b.ori.z, b.ori.w));
            sc1_store(base + 1, make_f4(b.pos.x, b.pos.y, b.pos.z, r.pos.w));
        }
        if (integrate_velocity_for_kinematics) {
            sc1_store(base + 2, make_f4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, r.lin.w));
            sc1_store(base + 3, make_f4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, r.ang.w));
        }
    }
}
__device__ __forceinline__ void stream_final_body(float4* bodies, const StreamCtx& cx, unsigned f, int i, float dt, float substep_dt, int substep_count,
                                                  int allow_substeps_for_unconstrained, int integrate_velocity_for_kinematics, const StepParams& sp) {
    float4* base = bodies + (size_t)i * 8;
    const f4 l0 = cx.stage[4 * kStagePlaneStride + cx.lane], l1 = cx.stage[5 * kStagePlaneStride + cx.lane];
    const float4 i0 = make_float4(l0.x, l0.y, l0.z, l0.w), i1 = make_float4(l1.x, l1.y, l1.z, l1.w);
    BodyPlanes r;
    staged_planes<kAccessAll>(cx, cx.lane, r);
    BodyRegs b = {{r.ori.x, r.ori.y, r.ori.z, r.ori.w}, {r.pos.x, r.pos.y, r.pos.z}, {{r.lin.x, r.lin.y, r.lin.z}, {r.ang.x, r.ang.y, r.ang.z}}};
    const bool velocity_written = final_integrate_regs(b, f, i0, i1, dt, substep_dt, substep_count, allow_substeps_for_unconstrained, integrate_velocity_for_kinematics, sp);
    if (velocity_written) {
        sc1_store(base + 2, make_f4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, r.lin.w));
        sc1_store(base + 3, make_f4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, r.ang.w));
    }
    sc1_store(base + 0, make_f4(b.ori.x, b.ori.y, b.ori.z, b.ori.w));
    sc1_store(base + 1, make_f4(b.pos.x, b.pos.y, b.pos.z, r.pos.w));
}

constexpr int kMaxStreamSubsteps = 16;
struct StreamParams {
    int substeps, batch_count, body_count, integrate_velocity_for_kinematics;
    int iters[kMaxStreamSubsteps];
    int has_incremental;
    float frame_dt, substep_dt, inv_substep_dt;
    int allow_substeps_for_unconstrained;
    StepParams sp, final_sp;
};

// Every workgroup resident (cooperative launch), kStreamWaves wavefronts each. All wavefronts walk the same hop sequence; a hop's blocks are dealt
// round-robin over the workgroups first (block vb -> workgroup vb % NWG, wavefront (vb / NWG) % kStreamWaves: the same in every hop of a batch), so
// a hop of up to NWG * kStreamWaves blocks costs every wavefront one block. Only wavefront 0 of a workgroup polls the arrival counter; the others wait
// for it at the workgroup barrier, which keeps the number of pollers at one per CU (2048 single-wave pollers saturate the fabric).
constexpr int kStreamWaves = 8;
__global__ __launch_bounds__(64 * kStreamWaves) void stream_kernel(const DevTypeBatch* __restrict__ tbs, const int* __restrict__ batch_begin, const int* __restrict__ batch_blocks,
                                                                   float4* bodies, const unsigned* __restrict__ flags, StreamSync sy, StreamParams P) {
    __shared__ int s_alive;
    extern __shared__ __attribute__((aligned(16))) f4 s_stage[];  // kStreamWaves slices of kStageWaveF4 (host passes the size)
    const int wg = blockIdx.x, NWG = gridDim.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    StreamCtx cx;
    cx.rsrc = __builtin_amdgcn_make_buffer_rsrc(bodies, 0, P.body_count * 128, 0x00020000);  // gfx9 raw-buffer descriptor word 3 (32-bit data format)
    cx.stage = (lds_f4*)s_stage + wv * kStageWaveF4;
    cx.lane = lane;
    const int fw = wg + NWG * wv, TW = NWG * kStreamWaves;  // this wavefront's place in the deal, and the deal's period
    // traced workgroups (wavefront 0 of each): the first, one in the middle, the last; stamps: reached the hop, passed the gate, finished the blocks, arrived
    const int traced = (sy.trace == nullptr || wv != 0) ? -1 : (wg == 0 ? 0 : (wg == NWG / 2 ? 1 : (wg == NWG - 1 ? 2 : -1)));
    auto stamp = [&](int h, int k) {
        if (traced >= 0 && lane == 0 && h < kStreamTraceHops) sy.trace[((size_t)traced * kStreamTraceHops + h) * 4 + k] = wall_clock64();
    };
    int hop = 0;               // index of the hop being run; it may start once hop - 1 is complete
    bool after_body_hop = true;  // the previous hop integrated bodies (or nothing ran yet): poses and inertias may not be read before the wait
    unsigned prev_blocks = 0;  // arrivals that complete hop - 1
    bool alive = true;

    // Once per hop and wavefront of a workgroup that owns blocks of the hop: wavefront 0 waits for the previous hop, the others for wavefront 0.
    auto gate_once = [&](bool& waited) -> bool {
        if (waited) return alive;
        waited = true;
        if (wv == 0) {
            const bool ok = hop_wait(sy, hop - 1, prev_blocks, lane);
            if (lane == 0) s_alive = ok ? 1 : 0;
        }
        __syncthreads();
        alive = s_alive != 0;
        stamp(hop, 1);
        return alive;
    };
    auto arrive = [&](unsigned wg_blocks) {
        stamp(hop, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every sc1 store of this wavefront has been acknowledged = is visible device-wide
        __syncthreads();                                  // ... and of the other wavefronts of the workgroup (also fences s_alive for the next hop)
        if (wv == 0 && lane == 0) hop_arrive(sy, hop, wg_blocks, wg);
        stamp(hop, 3);
    };
    auto owned = [&](int blocks) -> unsigned { return blocks > wg ? (unsigned)((blocks - wg + NWG - 1) / NWG) : 0u; };  // blocks of a grid this workgroup runs
    // (Dealing consecutive batches to consecutive wavefronts, so that the hop over all batches spreads out, was measured and dropped: wavefront 0 then
    // owns no block in most hops, starts polling at once, and 256 such pollers delay the arrivals they wait for: 0.86 -> 1.14 ms per step.)

    // (batch b, block vb) -> type batch and first constraint, as batch_kernel resolves blockIdx
    auto locate = [&](int b, int vb, int& i) -> DevTypeBatch {
        const int t0 = batch_begin[b], t1 = batch_begin[b + 1];
        int t = t0;
        for (int k = t0 + 1; k < t1; ++k)
            if (vb >= tbs[k].block_begin) t = k;
        const DevTypeBatch tb = tbs[t];
        i = (vb - tb.block_begin) * kBlock + lane;
        return tb;
    };
    auto constraint_hop = [&](auto stage_tag, int b_first, int b_last) {  // the blocks of batches [b_first, b_last] as one hop
        constexpr int STAGE = decltype(stage_tag)::value;
        unsigned total = 0, wg_blocks = 0;
        for (int b = b_first; b <= b_last; ++b) { total += (unsigned)batch_blocks[b]; wg_blocks += owned(batch_blocks[b]); }
        if (total == 0) return;
        if (wg_blocks > 0) {
            bool waited = false;
            stamp(hop, 0);
            auto gate = [&]() -> bool { return gate_once(waited); };
            for (int b = b_first; b <= b_last && alive; ++b) {
                const int blocks = batch_blocks[b];
                for (int vb = fw; vb < blocks && alive; vb += TW) {
                    int i;
                    const DevTypeBatch tb = locate(b, vb, i);
                    if (STAGE == kStageIncremental && !type_is_incremental(tb.type_id)) continue;
                    if (STAGE == kStageSolve) stream_constraint_block<STAGE, true>(tb, i, bodies, cx, P.substep_dt, P.inv_substep_dt, gate);  // never follows an integration hop
                    else if (STAGE == kStageWarmStart && !after_body_hop) stream_constraint_block<STAGE, true>(tb, i, bodies, cx, P.substep_dt, P.inv_substep_dt, gate);
                    else stream_constraint_block<STAGE, false>(tb, i, bodies, cx, P.substep_dt, P.inv_substep_dt, gate);
                }
            }
            gate();  // wavefronts without a block in this hop meet the others at the barrier all the same
            if (alive) arrive(wg_blocks);
        }
        prev_blocks = total;
        after_body_hop = false;
        ++hop;
    };
    auto body_hop = [&](auto&& per_body) {
        const int blocks = (P.body_count + 63) / 64;
        if (blocks == 0) return;
        const unsigned wg_blocks = owned(blocks);
        if (wg_blocks > 0) {
            bool waited = false;
            stamp(hop, 0);
            for (int vb = fw; vb < blocks && alive; vb += TW) {
                const int i = vb * 64 + lane;
                const unsigned f = i < P.body_count ? flags[i] : 0u;
                if (!gate_once(waited)) break;
                stage_body_block(cx, vb * 64);
                if (i < P.body_count) per_body(i, f);
            }
            gate_once(waited);
            if (alive) arrive(wg_blocks);
        }
        prev_blocks = (unsigned)blocks;
        after_body_hop = true;
        ++hop;
    };

    // `alive` is uniform over the workgroup after every gate, so all its wavefronts leave the loops (and skip the barriers) together.
    for (int s = 0; s < P.substeps && alive; ++s) {
        if (s > 0 && P.has_incremental) constraint_hop(std::integral_constant<int, kStageIncremental>{}, 0, P.batch_count - 1);
        if (alive) body_hop([&](int i, unsigned f) { stream_integrate_body(bodies, cx, f, i, s > 0 ? 1 : 0, P.integrate_velocity_for_kinematics, P.sp); });
        for (int b = 0; b < P.batch_count && alive; ++b) constraint_hop(std::integral_constant<int, kStageWarmStart>{}, b, b);
        for (int it = 0; it < P.iters[s] && alive; ++it)
            for (int b = 0; b < P.batch_count && alive; ++b) constraint_hop(std::integral_constant<int, kStageSolve>{}, b, b);
    }
    if (alive)
        body_hop([&](int i, unsigned f) {
            stream_final_body(bodies, cx, f, i, P.frame_dt, P.substep_dt, P.substeps, P.allow_substeps_for_unconstrained, P.integrate_velocity_for_kinematics, P.final_sp);
        });
}

}  // namespace
