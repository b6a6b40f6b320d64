// Launch-per-batch schedule and the per-body kernels: batch_kernel<stage>, constrained-set marking, kinematic / substep /
// final integration, the boundary exchange of a scene split across GPUs, and the ranged update / read-back transposes.
#pragma once

#include "bepu_kernels_common.h"

namespace {

// Three- and four-body constraints (ThreeBodyTypeProcessor.cs / FourBodyTypeProcessor.cs): the same gather / function / scatter over F::bodies slots.
template <class F, int STAGE>
__device__ __forceinline__ void run_constraint_many(const DevTypeBatch& tb, int i, float4* bodies, float dt, float inv_dt) {
    if (STAGE == kStageIncremental) return;
    constexpr int N = F::bodies;
    const int stride = tb.stride;
    int refs[N];
    _Pragma("unroll") for (int k = 0; k < N; ++k) refs[k] = tb.refs[(size_t)k * stride + i];
    float p[F::prestepFloats], a[F::impulseFloats];
    _Pragma("unroll") for (int f = 0; f < F::prestepFloats; ++f) p[f] = tb.prestep[(size_t)f * stride + i];
    _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) a[f] = tb.accum[(size_t)f * stride + i];
    DBody b[N];
    V3 pos[N]; float inverseMass[N]; BodyVel vel[N];
    _Pragma("unroll") for (int k = 0; k < N; ++k) {
        load_body<F::access>(bodies, refs[k], b[k]);
        pos[k] = b[k].pos; inverseMass[k] = b[k].inertia.invMass; vel[k] = b[k].vel;
    }
    if (STAGE == kStageWarmStart) {
        F::warmStartN(pos, inverseMass, p, a, vel, NoGate{});
    } else {
        F::solveN(pos, inverseMass, dt, inv_dt, p, a, vel, NoGate{});
        _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) tb.accum[(size_t)f * stride + i] = a[f];
    }
    _Pragma("unroll") for (int k = 0; k < N; ++k) { b[k].vel = vel[k]; store_velocity<F::access>(bodies, refs[k], b[k]); }
}

template <class F, int STAGE>
__device__ __forceinline__ void run_constraint(const DevTypeBatch& tb, int i, float4* bodies, float dt, float inv_dt) {
    const int stride = tb.stride;
    const int refA = tb.refs[i];
    const int refB = (F::bodies == 2) ? tb.refs[stride + i] : -1;
    float p[F::prestepFloats];
    _Pragma("unroll") for (int f = 0; f < F::prestepFloats; ++f) p[f] = tb.prestep[(size_t)f * stride + i];
    DBody A, B;
    if (STAGE == kStageIncremental) {
        load_body<kAccessOnlyVelocity>(bodies, refA, A);
        if (F::bodies == 2) load_body<kAccessOnlyVelocity>(bodies, refB, B); else load_body<0>(bodies, 0, B);
        F::incrementalUpdate(dt, A.vel, B.vel, p);
        // Only the contact depths change (PenetrationLimit.cs:42): prestep rows F::depthRow(c), c < contact count.
        if constexpr (F::incremental) {
            _Pragma("unroll") for (int cidx = 0; cidx < F::contacts; ++cidx) tb.prestep[(size_t)F::depthRow(cidx) * stride + i] = p[F::depthRow(cidx)];
        }
        return;
    }
    float a[F::impulseFloats];
    _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) a[f] = tb.accum[(size_t)f * stride + i];
    constexpr int accA = (STAGE == kStageWarmStart) ? F::wsA : F::svA;
    constexpr int accB = (STAGE == kStageWarmStart) ? F::wsB : F::svB;
    load_body<accA>(bodies, refA, A);
    if (F::bodies == 2) load_body<accB>(bodies, refB, B); else load_body<0>(bodies, 0, B);
    if (STAGE == kStageWarmStart) {
        F::warmStart(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, p, a, A.vel, B.vel, NoGate{});
    } else {
        F::solve(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, dt, inv_dt, p, a, A.vel, B.vel, NoGate{});
        _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) tb.accum[(size_t)f * stride + i] = a[f];
    }
    store_velocity<accA>(bodies, refA, A);
    if (F::bodies == 2) store_velocity<accB>(bodies, refB, B);
}

// One grid per (batch, stage): the block index selects the type batch, the type id (wave-uniform) selects the function.
// Graph colouring guarantees that no dynamic body is referenced twice inside a batch (Solver.cs:1046-1051), so no two lanes of
// the grid write the same body and results do not depend on lane order.
// Register budget (round 5): two waves per SIMD are asked for instead of three — the Solve instantiation then has 218 VGPRs and no scratch (at three:
// 168 VGPRs, 55 spilled, 192 B of scratch, four rounds running). A launch of this schedule is a few hundred lanes per CU (the pile: 50,000 constraints over 256 CUs), bound
// by its four dependent memory round trips, so occupancy above two waves buys nothing and the scratch reloads sat inside the dependent chain.
template <int STAGE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void batch_kernel(const DevTypeBatch* __restrict__ tbs, int tb_begin, int tb_count, float4* bodies, float dt, float inv_dt) {
    const int b = blockIdx.x;
    int t = tb_begin;
    for (int k = 1; k < tb_count; ++k)
        if (b >= tbs[tb_begin + k].block_begin) t = tb_begin + k;
    const DevTypeBatch tb = tbs[t];
    int i = (b - tb.block_begin) * kBlock + threadIdx.x;
    if (i >= tb.count) return;
    if (tb.indices) i = tb.indices[i];
    switch (tb.type_id) {
        case kContact1OneBody: run_constraint<Contact<1, false>, STAGE>(tb, i, bodies, dt, inv_dt); break;
        case kContact2OneBody: run_constraint<Contact<2, false>, STAGE>(tb, i, bodies, dt, inv_dt); break;
        case kContact3OneBody: run_constraint<Contact<3, false>, STAGE>(tb, i, bodies, dt, inv_dt); break;
        case kContact4OneBody: run_constraint<Contact<4, false>, STAGE>(tb, i, bodies, dt, inv_dt); break;
        case kContact1: run_constraint<Contact<1, true>, STAGE>(tb, i, bodies, dt, inv_dt); break;
        case kContact2: run_constraint<Contact<2, true>, STAGE>(tb, i, bodies, dt, inv_dt); break;
        case kContact3: run_constraint<Contact<3, true>, STAGE>(tb, i, bodies, dt, inv_dt); break;
        case kContact4: run_constraint<Contact<4, true>, STAGE>(tb, i, bodies, dt, inv_dt); break;
#define X(ID, T) case ID: run_constraint<T, STAGE>(tb, i, bodies, dt, inv_dt); break;
        BD_NONCONVEX_CONTACT_TYPES(X)
#undef X
        default: break;
    }
    if (STAGE == kStageIncremental) return;  // only contacts need incremental updates (RequiresIncrementalSubstepUpdates)
    switch (tb.type_id) {
#define X(ID, T) case ID: run_constraint<T, STAGE>(tb, i, bodies, dt, inv_dt); break;
        BD_JOINT_TYPES(X)
#undef X
#define X(ID, T) case ID: run_constraint_many<T, STAGE>(tb, i, bodies, dt, inv_dt); break;
        BD_MANY_BODY_TYPES(X)
#undef X
        default: break;
    }
}

// Body flag bits (per body index).
enum { kFlagConstrained = 1, kFlagDynamicConstrained = 2, kFlagConstrainedKinematic = 4, kFlagClustered = 8 /* dynamic body owned by a cluster_kernel workgroup */,
       kFlagClusterKinematic = 16 /* constrained kinematic body of the island schedule: advanced by cluster_kernel's kinematic block */ };

// Device-side equivalent of the merged constrained-body set of PrepareConstraintIntegrationResponsibilities
// (Solver_Solve.cs:1198-1207,1378-1381): every body referenced as dynamic gets integration inside the solver.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy. A group of type batches per launch: refs_off in words from the slab's base,
// blocks of 256 constraints numbered through the group; a dynamic reference -> integrated inside the solver, a kinematic one -> member of Solver.ConstrainedKinematicHandles (Solver.cs:68)
constexpr int kMarkGroupEntries = 96;
struct MarkGroup {
    struct Entry { unsigned long long refs_off; int extent, stride, bodies, block_begin; };
    int count, blocks;
    Entry entries[kMarkGroupEntries];
};
__global__ void mark_constrained_group_kernel(const int* __restrict__ slab, const MarkGroup group, unsigned* flags) {
    int entry = 0;
    while (entry + 1 < group.count && group.entries[entry + 1].block_begin <= (int)blockIdx.x) ++entry;
    const MarkGroup::Entry e = group.entries[entry];
    const int i = ((int)blockIdx.x - e.block_begin) * (int)blockDim.x + (int)threadIdx.x;
    if (i >= e.extent) return;
    const int* refs = slab + e.refs_off;
    for (int k = 0; k < e.bodies; ++k) {
        const int ref = refs[(size_t)k * e.stride + i];
        if (ref < 0) continue;
        const unsigned bits = kFlagConstrained | (((unsigned)ref < kDynamicLimit) ? kFlagDynamicConstrained : kFlagConstrainedKinematic);
        atomicOr(&flags[ref & kRefMask], bits);
    }
}
#endif
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ void mark_indices_kernel(const int* __restrict__ indices, int count, unsigned* flags, unsigned bits) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) atomicOr(&flags[indices[i] & kRefMask], bits);
}
#endif

// IPoseIntegratorCallbacks.IntegrateVelocity (PoseIntegrator.cs:91-93) for the models that cross the ABI (bepuhip_velocity_model); `position` and `body` are the
// callback's position and bodyIndices arguments. The caller masks the result where the reference does.
__device__ __forceinline__ void velocity_callback(const StepParams& sp, BodyVel& v, const V3& position, int body) {
    if (sp.velocity_model == 0) {  // Demos/DemoCallbacks.cs:100-109
        V3 g = {sp.gx, sp.gy, sp.gz};
        v.lin = scale(add(v.lin, g), sp.lin_damp);
        v.ang = scale(v.ang, sp.ang_damp);
    } else if (sp.velocity_model == 1) {  // Demos/Demos/PerBodyGravityDemo.cs:87: velocity.Linear.Y += new Vector<float>(gravityValues) * dt
        v.lin.y = v.lin.y + sp.body_gravity[body] * sp.callback_dt;
    } else {  // Demos/Demos/PlanetDemo.cs:44-46: offset = position - centre; velocity.Linear -= gravityDt * offset / max(1, distance^3)  (Vector3Wide.cs:357-365: "/" multiplies by 1 / scalar)
        const V3 offset = sub(position, V3{sp.cx, sp.cy, sp.cz});
        const float distance = sqrtf(offset.x * offset.x + offset.y * offset.y + offset.z * offset.z);
        const V3 scaled = {offset.x * sp.radial, offset.y * sp.radial, offset.z * sp.radial};
        const float inverse = 1.0f / vmax(1.0f, distance * distance * distance);
        v.lin = sub(v.lin, V3{scaled.x * inverse, scaled.y * inverse, scaled.z * inverse});
    }
}

// Cluster path only: advance the constrained kinematic bodies in global memory through the in-solver substeps
// (PoseIntegrator.cs:451-535 applied substep_count times: substep 0 velocity only, later substeps pose then velocity).
// (the island schedule's kinematic workgroup: all substeps of a constrained kinematic body at once, PoseIntegrator.cs:451-535)
__device__ __forceinline__ void kinematic_substeps_body(float4* bodies, int index, int substeps, int integrate_velocity_for_kinematics, const StepParams& sp, int substep_base = 0) {
    float4* base = bodies + (size_t)(index & kRefMask) * 8;
    float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3];
    Q ori = {q4.x, q4.y, q4.z, q4.w};
    V3 pos = {p4.x, p4.y, p4.z};
    BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
    for (int s = 0; s < substeps; ++s) {
        if (substep_base + s > 0) {  // (a chained step's later launches start past the step's first substep)
            pos = add(pos, scale(vel.lin, sp.dt));
            ori = integrateOrientation(ori, vel.ang, sp.dt * 0.5f);
        }
        if (integrate_velocity_for_kinematics) velocity_callback(sp, vel, pos, index & kRefMask);
    }
    base[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
    base[1] = make_float4(pos.x, pos.y, pos.z, p4.w);
    if (integrate_velocity_for_kinematics) {
        base[2] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
        base[3] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
    }
}

// The per-body state the integration functions work on, in registers (the kernels differ only in how they move it to and from memory).
struct BodyRegs { Q ori; V3 pos; BodyVel vel; };

// IntegratePoseAndVelocity (TypeProcessor.cs:1204-1248) / IntegrateVelocity (:1251-1283) of one constrained dynamic body; returns its refreshed world
// inverse inertia. substep 0: velocity only; substep > 0: pose, then velocity.
__device__ __forceinline__ Sym3 substep_integrate_dynamic(BodyRegs& b, const float4& i0, const float4& i1, int integrate_pose, const StepParams& sp, int body) {
    const Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
    Sym3 world;
    if (integrate_pose) {
        b.pos = add(b.pos, scale(b.vel.lin, sp.dt));                    // :1217
        const Q previousOrientation = b.ori;
        b.ori = integrateOrientation(b.ori, b.vel.ang, sp.dt * 0.5f);   // :1240
        world = rotateInverseInertia(local, b.ori);                     // :1242
        if (sp.angular_mode == 1) b.vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, local, world, b.vel.ang);                // :1224-1231
        else if (sp.angular_mode == 2) b.vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(b.ori, local, b.vel.ang, sp.dt);   // :1232-1238
    } else {
        world = rotateInverseInertia(local, b.ori);                     // :1262
        if (sp.angular_mode == 1) {
            const Q previousOrientation = integrateOrientation(b.ori, b.vel.ang, sp.dt * -0.5f);  // :1266 "integrating backwards"
            b.vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, local, world, b.vel.ang);
        } else if (sp.angular_mode == 2) {
            b.vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(b.ori, local, b.vel.ang, sp.dt);
        }
    }
    velocity_callback(sp, b.vel, b.pos, body);                          // :1244 / :1273-1281
    return world;
}
// The kinematic prepass (PoseIntegrator.cs:451-535) of one constrained kinematic body.
__device__ __forceinline__ void substep_integrate_kinematic(BodyRegs& b, int integrate_pose, int integrate_velocity_for_kinematics, const StepParams& sp, int body) {
    if (integrate_pose) {                                               // :519-523
        b.pos = add(b.pos, scale(b.vel.lin, sp.dt));
        b.ori = integrateOrientation(b.ori, b.vel.ang, sp.dt * 0.5f);
    }
    if (integrate_velocity_for_kinematics) velocity_callback(sp, b.vel, b.pos, body);  // :524-529, :481-485
}

// Per-substep integration of every constrained body — the work the reference fuses into the first-touching constraint's
// warm start (TypeProcessor.cs:1204-1283) plus the kinematic prepass (PoseIntegrator.cs:451-535). World inverse inertia is refreshed either way.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void substep_integrate_kernel(float4* bodies, const unsigned* __restrict__ flags, int count, int integrate_pose,
                                                                 int integrate_velocity_for_kinematics, int skip_clustered, StepParams sp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned f = flags[i];
    float4* base = bodies + (size_t)i * 8;
    if (skip_clustered && (f & kFlagClustered)) return;  // integrated in LDS by the owning cluster_kernel workgroup
    if (!(f & (kFlagDynamicConstrained | kFlagConstrainedKinematic))) return;
    const float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3];
    BodyRegs b = {{q4.x, q4.y, q4.z, q4.w}, {p4.x, p4.y, p4.z}, {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}}};
    if (f & kFlagDynamicConstrained) {
        const float4 i0 = base[4], i1 = base[5];
        const Sym3 world = substep_integrate_dynamic(b, i0, i1, integrate_pose, sp, i);
        if (integrate_pose) {
            base[0] = make_float4(b.ori.x, b.ori.y, b.ori.z, b.ori.w);
            base[1] = make_float4(b.pos.x, b.pos.y, b.pos.z, p4.w);
        }
        base[2] = make_float4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, l4.w);
        base[3] = make_float4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, a4.w);
        base[6] = make_float4(world.xx, world.yx, world.yy, world.zx);
        base[7] = make_float4(world.zy, world.zz, i1.z, base[7].w);
    } else {
        substep_integrate_kinematic(b, integrate_pose, integrate_velocity_for_kinematics, sp, i);
        if (integrate_pose) {
            base[0] = make_float4(b.ori.x, b.ori.y, b.ori.z, b.ori.w);
            base[1] = make_float4(b.pos.x, b.pos.y, b.pos.z, p4.w);
        }
        if (integrate_velocity_for_kinematics) {
            base[2] = make_float4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, l4.w);
            base[3] = make_float4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, a4.w);
        }
    }
}
#endif

// Substep 0, conserving modes only: the reference transforms the angular velocity of EVERY lane of a conditionally integrating bundle before it saves
// the "previous velocity" it later restores non-integrating lanes to (TypeProcessor.cs:1264-1281), so a body that was integrated by an earlier batch
// is transformed once more when it shares a bundle (slot-wise) with a body that is integrated there. The host lists those bodies per batch
// (bundle membership depends on the host's bundle width); this kernel runs before the batch's warm start. Bodies within a batch are distinct.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ void momentum_requirk_kernel(float4* bodies, const int* __restrict__ indices, int count, StepParams sp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float4* base = bodies + (size_t)indices[i] * 8;
    const float4 q4 = base[0], a4 = base[3], i0 = base[4], i1 = base[5];
    const Q ori = {q4.x, q4.y, q4.z, q4.w};
    V3 ang = {a4.x, a4.y, a4.z};
    const Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
    if (sp.angular_mode == 1) {
        const Sym3 world = rotateInverseInertia(local, ori);
        const Q previousOrientation = integrateOrientation(ori, ang, sp.dt * -0.5f);
        ang = integrateAngularVelocityConserveMomentum(previousOrientation, local, world, ang);
    } else {
        ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(ori, local, ang, sp.dt);
    }
    base[3] = make_float4(ang.x, ang.y, ang.z, a4.w);
}
#endif

// PoseIntegrator.IntegrateBundlesAfterSubstepping (PoseIntegrator.cs:537-693) of one body, on registers. True = the velocity changed too.
__device__ __forceinline__ bool final_integrate_regs(BodyRegs& b, unsigned body_flags, const float4& i0, const float4& i1, float dt, float substep_dt, int substep_count,
                                                     int allow_substeps_for_unconstrained, int integrate_velocity_for_kinematics, const StepParams& sp, int body) {
    const bool unconstrained = !(body_flags & kFlagConstrained);
    const float effective_dt = allow_substeps_for_unconstrained ? substep_dt : (unconstrained ? dt : substep_dt);  // :591-599
    const float half_dt = effective_dt * 0.5f;
    if (!unconstrained) {
        b.ori = integrateOrientation(b.ori, b.vel.ang, half_dt);   // :684-691
        b.pos = add(b.pos, scale(b.vel.lin, effective_dt));
        return false;
    }
    const bool is_kinematic = i0.x == 0 && i0.y == 0 && i0.z == 0 && i0.w == 0 && i1.x == 0 && i1.y == 0 && i1.z == 0;  // Bodies.cs:326-349
    const bool velocity_mask = integrate_velocity_for_kinematics ? true : !is_kinematic;                                // :604-616
    const int steps = allow_substeps_for_unconstrained ? substep_count : 1;
    for (int s = 0; s < steps; ++s) {
        if (velocity_mask) velocity_callback(sp, b.vel, b.pos, body);   // velocity -> pose for unconstrained bodies (:634-667)
        b.pos = add(b.pos, scale(b.vel.lin, effective_dt));
        if (sp.angular_mode == 1) {                        // :649-655
            const Q previousOrientation = b.ori;
            b.ori = integrateOrientation(b.ori, b.vel.ang, half_dt);
            const Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
            b.vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, local, rotateInverseInertia(local, b.ori), b.vel.ang);
        } else if (sp.angular_mode == 2) {                 // :656-660
            b.ori = integrateOrientation(b.ori, b.vel.ang, half_dt);
            const Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
            b.vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(b.ori, local, b.vel.ang, effective_dt);
        } else {
            b.ori = integrateOrientation(b.ori, b.vel.ang, half_dt);
        }
    }
    return velocity_mask;
}
__device__ __forceinline__ void final_integrate_body(float4* bodies, unsigned body_flags, int i, float dt, float substep_dt, int substep_count,
                                                     int allow_substeps_for_unconstrained, int integrate_velocity_for_kinematics, const StepParams& sp) {
    float4* base = bodies + (size_t)i * 8;
    const float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3];
    float4 i0 = make_float4(0, 0, 0, 0), i1 = i0;
    if (!(body_flags & kFlagConstrained)) { i0 = base[4]; i1 = base[5]; }
    BodyRegs b = {{q4.x, q4.y, q4.z, q4.w}, {p4.x, p4.y, p4.z}, {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}}};
    if (final_integrate_regs(b, body_flags, i0, i1, dt, substep_dt, substep_count, allow_substeps_for_unconstrained, integrate_velocity_for_kinematics, sp, i)) {
        base[2] = make_float4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, l4.w);
        base[3] = make_float4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, a4.w);
    }
    base[0] = make_float4(b.ori.x, b.ori.y, b.ori.z, b.ori.w);
    base[1] = make_float4(b.pos.x, b.pos.y, b.pos.z, p4.w);
}
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void final_integrate_kernel(float4* bodies, const unsigned* __restrict__ flags, int count, float dt, float substep_dt, int substep_count,
                                                               int allow_substeps_for_unconstrained, int integrate_velocity_for_kinematics, int skip_clustered, StepParams sp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned f = flags[i];
    if (skip_clustered && (f & kFlagClustered)) return;  // final pose already written by the owning cluster_kernel workgroup
    final_integrate_body(bodies, f, i, dt, substep_dt, substep_count, allow_substeps_for_unconstrained, integrate_velocity_for_kinematics, sp);
}
#endif


// PoseIntegrator.PredictBoundingBoxes (PoseIntegrator.cs:307-370), one lane per body: the stage before collision detection, on the bodies the solver left in HBM.
// The reference walks the bodies in bundles of Vector<float>.Count and calls the velocity callback on a whole bundle as soon as one of its lanes is to be integrated
// (:337-338); the demo callbacks ignore the mask and nothing masks afterwards, so a kinematic body is predicted with gravity and damping applied exactly when its
// bundle also holds a body that integrates. A wave covers 64 consecutive bodies = whole bundles (4, 8 or 16 wide), so the bundle's "any" is a slice of a ballot.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void predict_bounds_kernel(const float4* __restrict__ bodies, int count, CollidableIn* collidables, int keep_activity,
                                                              PredictedBounds* __restrict__ out, float dt, int integrate_velocity_for_kinematics, StepParams sp, ShapeTables tables,
                                                              int bundle_width, int2* heavy_queue, int* heavy_count, int heavy_threshold) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < count;
    const float4* base = bodies + (size_t)(live ? i : 0) * 8;
    const float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3], i0 = base[4], i1 = base[5];
    const Q ori = {q4.x, q4.y, q4.z, q4.w};
    const V3 pos = {p4.x, p4.y, p4.z};
    BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
    const bool is_kinematic = i0.x == 0 && i0.y == 0 && i0.z == 0 && i0.w == 0 && i1.x == 0 && i1.y == 0 && i1.z == 0;  // Bodies.cs:326-349
    const float sleep_energy = lengthSquared(vel.lin) + lengthSquared(vel.ang);                                           // :334, before the callback
    const unsigned long long integrates = __ballot(live && (integrate_velocity_for_kinematics || !is_kinematic));        // :323-331, one bit per lane of the wave
    const int first_lane_of_bundle = (threadIdx.x & 63) & ~(bundle_width - 1);
    const bool bundle_integrates = ((integrates >> first_lane_of_bundle) & ((1ull << bundle_width) - 1ull)) != 0;
    if (!live) return;
    if (bundle_integrates) velocity_callback(sp, vel, pos, i);  // :337-338 (never stored)
    const CollidableIn c = collidables[i];
    PredictedBounds r;
    if (heavy_queue && isHeavyShape(c, tables, heavy_threshold)) {  // a wave of predict_heavy_bounds_kernel does the box and the margin; the sleep counters are settled here
        r.activity = updateSleepCandidacy(sleep_energy, c.sleep_threshold, c.minimum_timesteps_under_threshold, c.activity);
        r.min[0] = r.min[1] = r.min[2] = r.max[0] = r.max[1] = r.max[2] = r.speculative_margin = 0.0f;
        heavy_queue[atomicAdd(heavy_count, 1)] = make_int2(i, bundle_integrates ? 1 : 0);
    } else {
        predictBoundsOfAnyShape(pos, ori, vel, sleep_energy, dt, c, tables, r);
    }
    out[i] = r;
    if (keep_activity) collidables[i].activity = r.activity;  // device-resident records: the sleep counters carry over to the next frame
}
#endif

// Second pass of PredictBoundingBoxes: one wave per queued body (compound, mesh, large hull), as many waves as the grid has looping over the queue.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(64) void predict_heavy_bounds_kernel(const float4* __restrict__ bodies, const CollidableIn* __restrict__ collidables, PredictedBounds* __restrict__ out, float dt,
                                                                   StepParams sp, ShapeTables tables, const int2* __restrict__ heavy_queue, const int* __restrict__ heavy_count) {
    const int queued = *heavy_count;
    for (int q = blockIdx.x; q < queued; q += gridDim.x) {
        const int2 item = heavy_queue[q];
        const float4* base = bodies + (size_t)item.x * 8;
        const float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3];
        BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
        if (item.y) velocity_callback(sp, vel, V3{p4.x, p4.y, p4.z}, item.x);
        const CollidableIn c = collidables[item.x];
        PredictedBounds r;
        heavyBounds((int)threadIdx.x, V3{p4.x, p4.y, p4.z}, Q{q4.x, q4.y, q4.z, q4.w}, vel, dt, c, tables, r);
        if (threadIdx.x == 0) {
            PredictedBounds* o = out + item.x;
            o->min[0] = r.min[0]; o->min[1] = r.min[1]; o->min[2] = r.min[2]; o->speculative_margin = r.speculative_margin;
            o->max[0] = r.max[0]; o->max[1] = r.max[1]; o->max[2] = r.max[2];
        }
    }
}
#endif

// ---- boundary exchange (one connected scene split across GPUs, BASELINE.json configs[4]) ----
// A boundary body exists on several ranks (owned on one, ghost elsewhere). Between passes every holder publishes what its own constraints did to the
// body's velocity since the last synchronisation point, the ranks sum those deltas (RCCL all-reduce, done by the caller), and every holder
// replaces its copy with snapshot + sum: block-Jacobi across the cut, Gauss-Seidel everywhere else.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ void boundary_snapshot_kernel(const float4* __restrict__ bodies, const int* __restrict__ indices, int count, float4* __restrict__ snapshot) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4* base = bodies + (size_t)indices[i] * 8;
    snapshot[2 * i] = base[2];
    snapshot[2 * i + 1] = base[3];
}
#endif
// `exact` (BEPUHIP_EXCHANGE_PER_BATCH_EXACT): the exchange runs after every batch, where at most ONE rank has touched a given body (a batch references a body
// once, and the shares keep the global batch indices), so instead of float differences the ranks exchange the XOR of the velocity's bit pattern with the
// snapshot's: all but one contribution are zero, an integer sum returns the toucher's pattern exactly, and every copy becomes bit-identical to what the
// unsplit solve holds at that point. `rows` (optional) scatters / gathers through the dense exchange buffer (row of body i = rows[i]; 6 words per row).
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ void boundary_deltas_kernel(const float4* __restrict__ bodies, const int* __restrict__ indices, int count, const float4* __restrict__ snapshot, float* __restrict__ out,
                                       const int* __restrict__ rows, int exact) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4* base = bodies + (size_t)indices[i] * 8;
    const float4 l = base[2], a = base[3], l0 = snapshot[2 * i], a0 = snapshot[2 * i + 1];
    float* o = out + (size_t)(rows ? rows[i] : i) * 6;
    if (exact) {
        unsigned* x = reinterpret_cast<unsigned*>(o);
        x[0] = __float_as_uint(l.x) ^ __float_as_uint(l0.x); x[1] = __float_as_uint(l.y) ^ __float_as_uint(l0.y); x[2] = __float_as_uint(l.z) ^ __float_as_uint(l0.z);
        x[3] = __float_as_uint(a.x) ^ __float_as_uint(a0.x); x[4] = __float_as_uint(a.y) ^ __float_as_uint(a0.y); x[5] = __float_as_uint(a.z) ^ __float_as_uint(a0.z);
        return;
    }
    o[0] = l.x - l0.x; o[1] = l.y - l0.y; o[2] = l.z - l0.z;
    o[3] = a.x - a0.x; o[4] = a.y - a0.y; o[5] = a.z - a0.z;
}
#endif
// `holders` (optional, indexed like the sums): the number of ranks holding the body; the summed deltas of mass-split copies are averaged (lattice.py).
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ void boundary_apply_kernel(float4* bodies, const int* __restrict__ indices, int count, float4* snapshot, const float* __restrict__ sums, const int* __restrict__ rows,
                                      const float* __restrict__ holders, int exact) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float4* base = bodies + (size_t)indices[i] * 8;
    const int row = rows ? rows[i] : i;
    const float* d = sums + (size_t)row * 6;
    float4 l0 = snapshot[2 * i], a0 = snapshot[2 * i + 1];
    float4 l, a;
    if (exact) {
        const unsigned* x = reinterpret_cast<const unsigned*>(d);
        l = make_float4(__uint_as_float(__float_as_uint(l0.x) ^ x[0]), __uint_as_float(__float_as_uint(l0.y) ^ x[1]), __uint_as_float(__float_as_uint(l0.z) ^ x[2]), l0.w);
        a = make_float4(__uint_as_float(__float_as_uint(a0.x) ^ x[3]), __uint_as_float(__float_as_uint(a0.y) ^ x[4]), __uint_as_float(__float_as_uint(a0.z) ^ x[5]), a0.w);
    } else if (holders) {
        const float k = holders[row];
        l = make_float4(l0.x + d[0] / k, l0.y + d[1] / k, l0.z + d[2] / k, l0.w);
        a = make_float4(a0.x + d[3] / k, a0.y + d[4] / k, a0.z + d[5] / k, a0.w);
    } else {
        l = make_float4(l0.x + d[0], l0.y + d[1], l0.z + d[2], l0.w);
        a = make_float4(a0.x + d[3], a0.y + d[4], a0.z + d[5], a0.w);
    }
    base[2] = l; base[3] = a;
    snapshot[2 * i] = l; snapshot[2 * i + 1] = a;  // the next exchange's deltas are relative to the synchronised value
}
#endif

// Ranged in-place update of one type batch's prestep / accumulated-impulse rows from the caller's AOSOA bundles (bepuhip_update_prestep /
// bepuhip_update_accumulated_impulses): one thread per constraint of the range, `fields` strided stores each. `device_index` maps the constraint's
// index inside the type batch to its slot in the SoA rows (identity unless the island schedule permuted the batch); null = identity.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void scatter_bundles_kernel(const float* __restrict__ bundles, float* __restrict__ rows, const int* __restrict__ device_index,
                                                              int first_constraint, int constraint_count, int fields, int stride, int W) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= constraint_count) return;
    const int h = first_constraint + j;
    const int d = device_index ? device_index[h] : h;
    const float* src = bundles + (size_t)(j / W) * fields * W + (j % W);
    for (int f = 0; f < fields; ++f) rows[(size_t)f * stride + d] = src[(size_t)f * W];
}
#endif
// ---- structural updates: the host's TypeProcessor mutations mirrored on the rows that live in HBM ----
// kind 0: TypeProcessor.Move (TypeProcessor.cs:578-592): lane `src` copied over lane `dst` (body references, prestep, accumulated impulses) — the swap-with-last of Remove.
// kind 1: AllocateInTypeBatch (:314-334): lane `dst` written from the payload (references, prestep), accumulated impulses cleared (GatherScatter.ClearLane :327).
// kind 2: UpdateForBodyMemoryMove (:807): one body reference of lane `dst` (body slot `src`) replaced by the payload word.
// kind 3: bepuhip_swap_constraints: lanes `src` and `dst` exchanged.
// kind 4 (round 6, the sequential fallback batch): the references of `src` lanes from lane `dst` on become -1 (RemoveBodyReferencesLane :301-311; a new bundle's lanes :287-296).
// kind 5 (round 6): `pad` lanes of EVERY row copied from lane `src` on to lane `dst` on — the last bundle moved into an emptied one (TypeProcessor.cs:660-667).
struct StructuralOp { unsigned refs_off, prestep_off, accum_off; int stride, nb, pf, imf, kind, src, dst; unsigned payload_off; int pad; };
static_assert(sizeof(StructuralOp) == 48, "uploaded as raw words");
// One workgroup per type batch: its operations run in the order the host issued them (a Move may read what an earlier append wrote), rows in parallel.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(64) void apply_structural_ops_kernel(unsigned* __restrict__ slab, const StructuralOp* __restrict__ ops, const int* __restrict__ group_begin,
                                                                  const unsigned* __restrict__ payload) {
    const int g = blockIdx.x, lane = threadIdx.x;
    for (int o = group_begin[g]; o < group_begin[g + 1]; ++o) {
        const StructuralOp op = ops[o];
        const int rows = op.nb + op.pf + op.imf;
        if (op.kind == 2) {
            if (lane == 0) slab[op.refs_off + (size_t)op.src * op.stride + op.dst] = payload[op.payload_off];
        } else if (op.kind == 4) {
            for (int e = lane; e < op.nb * op.src; e += 64) slab[op.refs_off + (size_t)(e / op.src) * op.stride + op.dst + (e % op.src)] = 0xFFFFFFFFu;
        } else if (op.kind == 5) {
            for (int e = lane; e < rows * op.pad; e += 64) {
                const int r = e / op.pad, l = e % op.pad;
                const size_t base = r < op.nb ? op.refs_off + (size_t)r * op.stride
                                  : (r < op.nb + op.pf ? op.prestep_off + (size_t)(r - op.nb) * op.stride : op.accum_off + (size_t)(r - op.nb - op.pf) * op.stride);
                slab[base + op.dst + l] = slab[base + op.src + l];
            }
        } else {
            for (int r = lane; r < rows; r += 64) {
                const size_t base = r < op.nb ? op.refs_off + (size_t)r * op.stride
                                  : (r < op.nb + op.pf ? op.prestep_off + (size_t)(r - op.nb) * op.stride : op.accum_off + (size_t)(r - op.nb - op.pf) * op.stride);
                unsigned v;
                if (op.kind == 0 || op.kind == 3) v = slab[base + op.src];
                else v = r < op.nb + op.pf ? payload[op.payload_off + r] : 0u;
                if (op.kind == 3) slab[base + op.src] = slab[base + op.dst];
                slab[base + op.dst] = v;
            }
        }
        __syncthreads();  // the next operation of this type batch sees this one's stores (one workgroup, global memory)
        __threadfence_block();
    }
}
#endif
// Structural updates that stay on the island layout (bepu_soft_updates.h): the final state of every device slot touched since the last flush. A live slot gets its
// encoded references, its packed local references, its prestep lane and zero impulses (TypeProcessor.cs:327); a freed one gets -1 references and local references that name a kinematic copy (kLrefDead).
struct SoftSlotOp { unsigned refs_off, lrefs_off, prestep_off, accum_off; int stride, bodies, prestep, impulse, slot, live; unsigned payload; int ranks; };  // ranks: rank words behind the local references (split plans: one per body slot)
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(64) void apply_soft_slots_kernel(unsigned* slab, const SoftSlotOp* __restrict__ ops, int count, const unsigned* __restrict__ payload) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const SoftSlotOp op = ops[i];
    const int lref_rows = (op.bodies + 1) / 2;
    if (!op.live) {
        for (int k = 0; k < op.bodies; ++k) slab[op.refs_off + (size_t)k * op.stride + op.slot] = 0xFFFFFFFFu;
        for (int k = 0; k < lref_rows; ++k) slab[op.lrefs_off + (size_t)k * op.stride + op.slot] = kLrefDead;
        return;
    }
    const unsigned* p = payload + op.payload;
    for (int k = 0; k < op.bodies; ++k) slab[op.refs_off + (size_t)k * op.stride + op.slot] = *p++;
    for (int k = 0; k < lref_rows; ++k) slab[op.lrefs_off + (size_t)k * op.stride + op.slot] = *p++;
    for (int k = 0; k < op.ranks; ++k) slab[op.lrefs_off + (size_t)(lref_rows + k) * op.stride + op.slot] = *p++;
    for (int f = 0; f < op.prestep; ++f) slab[op.prestep_off + (size_t)f * op.stride + op.slot] = *p++;
    for (int f = 0; f < op.impulse; ++f) slab[op.accum_off + (size_t)f * op.stride + op.slot] = 0u;
}
#endif
// Single bits of slab words (the conserving modes' marks on an island layout): set (1) or cleared (0); words that hold several marks are listed once per mark.
struct BitMark { size_t word; unsigned mask; unsigned pad; };
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void mark_bits_kernel(unsigned* slab, const BitMark* __restrict__ marks, int count, int set) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (set) atomicOr(&slab[marks[i].word], marks[i].mask); else atomicAnd(&slab[marks[i].word], ~marks[i].mask);
}
#endif
struct IndexPatch { int* table; int index, value, pad; };
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(64) void patch_index_kernel(const IndexPatch* __restrict__ patches, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) patches[i].table[patches[i].index] = patches[i].value;
}
#endif
// Caller's order -> island schedule, rows to rows (bepuhip_replan): permuted[r][device_index[h]] = rows[r][h]; a null index table is the identity.
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void permute_rows_kernel(const unsigned* __restrict__ src, int src_stride, unsigned* __restrict__ dst, int dst_stride, const int* __restrict__ device_index,
                                                           int count, int rows) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= count) return;
    const int d = device_index ? device_index[h] : h;
    for (int r = 0; r < rows; ++r) dst[(size_t)r * dst_stride + d] = src[(size_t)r * src_stride + h];
}
#endif
// Island schedule -> caller's order: rows[r][host index] = permuted[r][device index] (the first structural update leaves the island schedule).
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void unpermute_rows_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, const int* __restrict__ device_to_host, int count, int stride, int rows) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= count) return;
    const int h = device_to_host[d];
    if (h < 0) return;  // a free slot of the island layout
    for (int r = 0; r < rows; ++r) dst[(size_t)r * stride + h] = src[(size_t)r * stride + d];
}
#endif
// The inverse, for ranged read-back (bepuhip_get_*_range).
#ifndef BEPU_CLUSTER_UNIT  // launched by bepuhip.hip only: the cluster units do not carry a copy
__global__ __launch_bounds__(256) void gather_bundles_kernel(float* __restrict__ bundles, const float* __restrict__ rows, const int* __restrict__ device_index,
                                                             int first_constraint, int constraint_count, int fields, int stride, int W) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= constraint_count) return;
    const int h = first_constraint + j;
    const int d = device_index ? device_index[h] : h;
    float* dst = bundles + (size_t)(j / W) * fields * W + (j % W);
    for (int f = 0; f < fields; ++f) dst[(size_t)f * W] = rows[(size_t)f * stride + d];
}
#endif

}  // namespace
