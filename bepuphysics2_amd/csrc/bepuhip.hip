// libbepuhip — MI355X (gfx950) native constraint solver + pose integrator behind the C ABI of include/bepuhip.h.
//
// Data layout in HBM
//   bodies        : the reference's BodyDynamics AoS, 128 B per body = 8 x float4 (BodyProperties.cs:318-338):
//                   [0] orientation xyzw  [1] position xyz  [2] linear xyz  [3] angular xyz
//                   [4..5] local inverse inertia {xx,yx,yy,zx | zy,zz,invMass}  [6..7] world inverse inertia.
//                   One 128-byte line per gathered body; every access is a 16-byte vector load/store.
//   type batches  : per (batch, type) SoA slabs — refs[slot][stride], prestep[field][stride], accumulated[field][stride],
//                   stride = count rounded up to 64 lanes, so lane i of a wavefront reads consecutive dwords (coalesced 256 B/wave/field).
//                   Converted from the host's AOSOA (BundleIndexing.cs:50-60) on upload and back on download.
// Schedule (no host sync inside a frame; replayed from a hipGraph):
//   per substep: [incremental contact update, all batches, one launch] -> [integrate constrained bodies, one launch]
//                -> warm start: one launch per batch (all constraint types of the batch in one grid)
//                -> velocity iterations: one launch per batch per iteration
//   then one final pose-integration launch over all bodies.
// Integration is hoisted out of the first-touching constraint's warm start into the per-substep body kernel; SURVEY.md A.2
// gives the argument that this is value-identical (nothing reads a body between its integration and its first constraint).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/bepuhip.h"
#include "bepu_device_constraints.h"

#pragma clang fp contract(off)

using namespace bd;

// ------------------------------------------------------------------------------------------------
// Device side
// ------------------------------------------------------------------------------------------------
namespace {

constexpr unsigned kDynamicLimit = 1u << 30;  // Bodies_GatherScatter.cs:107-118
constexpr int kRefMask = 0x3FFFFFFF;
constexpr int kBlock = 64;  // one wavefront per workgroup: a batch rarely fills the chip, so spread waves over as many CUs as possible

struct StepParams {
    float dt, inv_dt;
    float gx, gy, gz;  // gravity * dt
    float lin_damp, ang_damp;
    int angular_mode;  // AngularIntegrationMode (PoseIntegrator.cs:20-38): 0 Nonconserving, 1 ConserveMomentum, 2 ConserveMomentumWithGyroscopicTorque
};

struct DevTypeBatch {
    int type_id, count, stride, block_begin;
    int* refs;
    float* prestep;
    float* accum;
};

// ---- cluster path descriptors (see cluster_kernel) ----
constexpr int kMaxPreds = 6;
constexpr int kFallbackBatchLimit = 64;
struct ClusterItem {  // <= 64 consecutive constraints of one type batch, all owned by one cluster; 64 bytes, staged in LDS
    int type_id, count, stride, start;                // start: index of the first constraint inside the (reordered) type batch
    unsigned lrefs_off, prestep_off, accum_off;       // word offsets into the constraint slab: lrefs[bodies][stride], prestep[pf][stride], accum[imf][stride]
    int batch_npred;                                  // bits 0-15 batch, 16-19 predecessor count, 20-23 cross-pass predecessor count, 24 / 25 overflow flags
    unsigned short pred[kMaxPreds];                   // cluster-relative indices of the items that last touched this item's dynamic bodies (same pass)
    unsigned short xpred[kMaxPreds];                  // for bodies this item touches FIRST in a pass: their last toucher (previous pass); may be the item itself
    int tb, shape;                                    // host bookkeeping: type batch, bodies | prestep floats << 8 | impulse floats << 16
};
static_assert(sizeof(ClusterItem) == 64, "ClusterItem is staged in LDS as four 16-byte vectors");
struct ClusterDesc { int body_begin, slot_count, item_begin, item_count, batch_item_offset; };
constexpr int kMaxClusterSubsteps = 16;
struct ClusterParams {
    int substeps, batch_count, integrate_velocity_for_kinematics;
    int iters[kMaxClusterSubsteps];
    StepParams sp;
};


struct DBody {
    V3 pos; Q ori; BodyVel vel; Inertia inertia;
    float linw, angw;  // padding lanes of the velocity float4s, preserved on store
};

template <int ACCESS>
__device__ __forceinline__ void load_body(const float4* __restrict__ bodies, int ref, DBody& b) {
    const float4* base = bodies + (size_t)(ref & kRefMask) * 8;
    if (ACCESS & kOri) { float4 q = base[0]; b.ori = {q.x, q.y, q.z, q.w}; } else b.ori = {0, 0, 0, 0};
    if (ACCESS & kPos) { float4 p = base[1]; b.pos = {p.x, p.y, p.z}; } else b.pos = {0, 0, 0};
    if (ACCESS & kLin) { float4 l = base[2]; b.vel.lin = {l.x, l.y, l.z}; b.linw = l.w; } else { b.vel.lin = {0, 0, 0}; b.linw = 0; }
    if (ACCESS & kAng) { float4 a = base[3]; b.vel.ang = {a.x, a.y, a.z}; b.angw = a.w; } else { b.vel.ang = {0, 0, 0}; b.angw = 0; }
    if (ACCESS & kInertia) {
        float4 i0 = base[6], i1 = base[7];
        b.inertia.t = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
        b.inertia.invMass = i1.z;
    } else { b.inertia.t = {0, 0, 0, 0, 0, 0}; b.inertia.invMass = 0; }
}
// ScatterVelocities: kinematic / empty references are never written (Bodies_GatherScatter.cs:675-682,717-724).
template <int ACCESS>
__device__ __forceinline__ void store_velocity(float4* bodies, int ref, const DBody& b) {
    if ((unsigned)ref >= kDynamicLimit) return;
    float4* base = bodies + (size_t)ref * 8;
    if (ACCESS & kLin) base[2] = make_float4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, b.linw);
    if (ACCESS & kAng) base[3] = make_float4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, b.angw);
}

enum { kStageWarmStart = 0, kStageSolve = 1, kStageIncremental = 2 };
// The constraint functions call their gate once, right before the first use of the bodies' velocities (everything before it depends on
// poses, inertias and prestep data only). The launch-per-batch kernels have the velocities in registers already.
struct NoGate {
    static constexpr bool kPin = false;
    __device__ __forceinline__ void operator()(BodyVel&, BodyVel&) const {}
};

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
template <int STAGE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void batch_kernel(const DevTypeBatch* __restrict__ tbs, int tb_begin, int tb_count, float4* bodies, float dt, float inv_dt) {
    const int b = blockIdx.x;
    int t = tb_begin;
    for (int k = 1; k < tb_count; ++k)
        if (b >= tbs[tb_begin + k].block_begin) t = tb_begin + k;
    const DevTypeBatch tb = tbs[t];
    const int i = (b - tb.block_begin) * kBlock + threadIdx.x;
    if (i >= tb.count) return;
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
        default: break;
    }
}

// Body flag bits (per body index).
enum { kFlagConstrained = 1, kFlagDynamicConstrained = 2, kFlagConstrainedKinematic = 4, kFlagClustered = 8 /* dynamic body owned by a cluster_kernel workgroup */ };

// Device-side equivalent of the merged constrained-body set of PrepareConstraintIntegrationResponsibilities
// (Solver_Solve.cs:1198-1207,1378-1381): every body referenced as dynamic gets integration inside the solver.
__global__ void mark_constrained_kernel(const int* __restrict__ refs, int count, int stride, int bodies_per_constraint, unsigned* flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    for (int k = 0; k < bodies_per_constraint; ++k) {
        int ref = refs[(size_t)k * stride + i];
        if (ref < 0) continue;
        // dynamic reference -> integrated inside the solver; kinematic reference -> member of Solver.ConstrainedKinematicHandles (Solver.cs:68)
        unsigned bits = kFlagConstrained | (((unsigned)ref < kDynamicLimit) ? kFlagDynamicConstrained : kFlagConstrainedKinematic);
        atomicOr(&flags[ref & kRefMask], bits);
    }
}
__global__ void mark_indices_kernel(const int* __restrict__ indices, int count, unsigned* flags, unsigned bits) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) atomicOr(&flags[indices[i] & kRefMask], bits);
}

__device__ __forceinline__ void velocity_callback(const StepParams& sp, BodyVel& v) {  // Demos/DemoCallbacks.cs:100-109
    V3 g = {sp.gx, sp.gy, sp.gz};
    v.lin = scale(add(v.lin, g), sp.lin_damp);
    v.ang = scale(v.ang, sp.ang_damp);
}

// Cluster path only: advance the constrained kinematic bodies in global memory through the in-solver substeps
// (PoseIntegrator.cs:451-535 applied substep_count times: substep 0 velocity only, later substeps pose then velocity).
__global__ void kinematic_substeps_kernel(float4* bodies, const int* __restrict__ indices, int count, int substeps, int integrate_velocity_for_kinematics, StepParams sp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float4* base = bodies + (size_t)(indices[i] & kRefMask) * 8;
    float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3];
    Q ori = {q4.x, q4.y, q4.z, q4.w};
    V3 pos = {p4.x, p4.y, p4.z};
    BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
    for (int s = 0; s < substeps; ++s) {
        if (s > 0) {
            pos = add(pos, scale(vel.lin, sp.dt));
            ori = integrateOrientation(ori, vel.ang, sp.dt * 0.5f);
        }
        if (integrate_velocity_for_kinematics) velocity_callback(sp, vel);
    }
    base[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
    base[1] = make_float4(pos.x, pos.y, pos.z, p4.w);
    if (integrate_velocity_for_kinematics) {
        base[2] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
        base[3] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
    }
}

// Per-substep integration of every constrained body — the work the reference fuses into the first-touching constraint's
// warm start (TypeProcessor.cs:1204-1283) plus the kinematic prepass (PoseIntegrator.cs:451-535).
// substep 0: velocity only; substep > 0: pose, then velocity. World inverse inertia is refreshed either way.
__global__ __launch_bounds__(256) void substep_integrate_kernel(float4* bodies, const unsigned* __restrict__ flags, int count, int integrate_pose,
                                                                 int integrate_velocity_for_kinematics, int skip_clustered, StepParams sp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned f = flags[i];
    float4* base = bodies + (size_t)i * 8;
    if (skip_clustered && (f & kFlagClustered)) return;  // integrated in LDS by the owning cluster_kernel workgroup
    if (f & kFlagDynamicConstrained) {
        float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3], i0 = base[4], i1 = base[5];
        Q ori = {q4.x, q4.y, q4.z, q4.w};
        V3 pos = {p4.x, p4.y, p4.z};
        BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
        Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
        Sym3 world;
        if (integrate_pose) {                                           // IntegratePoseAndVelocity, TypeProcessor.cs:1204-1248
            pos = add(pos, scale(vel.lin, sp.dt));                      // :1217
            const Q previousOrientation = ori;
            ori = integrateOrientation(ori, vel.ang, sp.dt * 0.5f);     // :1240
            world = rotateInverseInertia(local, ori);                   // :1242
            if (sp.angular_mode == 1) vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, local, world, vel.ang);                // :1224-1231
            else if (sp.angular_mode == 2) vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(ori, local, vel.ang, sp.dt);   // :1232-1238
            base[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
            base[1] = make_float4(pos.x, pos.y, pos.z, p4.w);
        } else {                                                        // IntegrateVelocity, TypeProcessor.cs:1251-1283
            world = rotateInverseInertia(local, ori);                   // :1262
            if (sp.angular_mode == 1) {
                const Q previousOrientation = integrateOrientation(ori, vel.ang, sp.dt * -0.5f);  // :1266 "integrating backwards"
                vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, local, world, vel.ang);
            } else if (sp.angular_mode == 2) {
                vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(ori, local, vel.ang, sp.dt);
            }
        }
        velocity_callback(sp, vel);                                     // :1244 / :1273-1281
        base[2] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
        base[3] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
        base[6] = make_float4(world.xx, world.yx, world.yy, world.zx);
        base[7] = make_float4(world.zy, world.zz, i1.z, base[7].w);
    } else if (f & kFlagConstrainedKinematic) {
        float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3];
        Q ori = {q4.x, q4.y, q4.z, q4.w};
        V3 pos = {p4.x, p4.y, p4.z};
        BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
        if (integrate_pose) {                                           // PoseIntegrator.cs:519-523
            pos = add(pos, scale(vel.lin, sp.dt));
            ori = integrateOrientation(ori, vel.ang, sp.dt * 0.5f);
            base[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
            base[1] = make_float4(pos.x, pos.y, pos.z, p4.w);
        }
        if (integrate_velocity_for_kinematics) {                        // :524-529, :481-485
            velocity_callback(sp, vel);
            base[2] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
            base[3] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
        }
    }
}

// Substep 0, conserving modes only: the reference transforms the angular velocity of EVERY lane of a conditionally integrating bundle before it saves
// the "previous velocity" it later restores non-integrating lanes to (TypeProcessor.cs:1264-1281), so a body that was integrated by an earlier batch
// is transformed once more when it shares a bundle (slot-wise) with a body that is integrated there. The host lists those bodies per batch
// (bundle membership depends on the host's bundle width); this kernel runs before the batch's warm start. Bodies within a batch are distinct.
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

// PoseIntegrator.IntegrateBundlesAfterSubstepping (PoseIntegrator.cs:537-693), one lane per body.
__global__ __launch_bounds__(256) void final_integrate_kernel(float4* bodies, const unsigned* __restrict__ flags, int count, float dt, float substep_dt, int substep_count,
                                                               int allow_substeps_for_unconstrained, int integrate_velocity_for_kinematics, int skip_clustered, StepParams sp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float4* base = bodies + (size_t)i * 8;
    if (skip_clustered && (flags[i] & kFlagClustered)) return;  // final pose already written by the owning cluster_kernel workgroup
    const bool unconstrained = !(flags[i] & kFlagConstrained);
    const float effective_dt = allow_substeps_for_unconstrained ? substep_dt : (unconstrained ? dt : substep_dt);  // :591-599
    const float half_dt = effective_dt * 0.5f;
    float4 q4 = base[0], p4 = base[1], l4 = base[2], a4 = base[3];
    Q ori = {q4.x, q4.y, q4.z, q4.w};
    V3 pos = {p4.x, p4.y, p4.z};
    BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
    if (unconstrained) {
        float4 i0 = base[4], i1 = base[5];
        const bool is_kinematic = i0.x == 0 && i0.y == 0 && i0.z == 0 && i0.w == 0 && i1.x == 0 && i1.y == 0 && i1.z == 0;  // Bodies.cs:326-349
        const bool velocity_mask = integrate_velocity_for_kinematics ? true : !is_kinematic;                                // :604-616
        const int steps = allow_substeps_for_unconstrained ? substep_count : 1;
        for (int s = 0; s < steps; ++s) {
            if (velocity_mask) velocity_callback(sp, vel);   // velocity -> pose for unconstrained bodies (:634-667)
            pos = add(pos, scale(vel.lin, effective_dt));
            if (sp.angular_mode == 1) {                      // :649-655
                const Q previousOrientation = ori;
                ori = integrateOrientation(ori, vel.ang, half_dt);
                const Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
                vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, local, rotateInverseInertia(local, ori), vel.ang);
            } else if (sp.angular_mode == 2) {               // :656-660
                ori = integrateOrientation(ori, vel.ang, half_dt);
                const Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
                vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(ori, local, vel.ang, effective_dt);
            } else {
                ori = integrateOrientation(ori, vel.ang, half_dt);
            }
        }
        if (velocity_mask) {
            base[2] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
            base[3] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
        }
    } else {
        ori = integrateOrientation(ori, vel.ang, half_dt);   // :684-691
        pos = add(pos, scale(vel.lin, effective_dt));
    }
    base[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
    base[1] = make_float4(pos.x, pos.y, pos.z, p4.w);
}



// ---- boundary exchange (one connected scene split across GPUs, BASELINE.json configs[4]) ----
// A boundary body exists on several ranks (owned on one, ghost elsewhere). Between passes every holder publishes what its own constraints did to the
// body's velocity since the last synchronisation point, the ranks sum those deltas (RCCL all-reduce, done by the caller), and every holder
// replaces its copy with snapshot + sum: block-Jacobi across the cut, Gauss-Seidel everywhere else.
__global__ void boundary_snapshot_kernel(const float4* __restrict__ bodies, const int* __restrict__ indices, int count, float4* __restrict__ snapshot) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4* base = bodies + (size_t)indices[i] * 8;
    snapshot[2 * i] = base[2];
    snapshot[2 * i + 1] = base[3];
}
__global__ void boundary_deltas_kernel(const float4* __restrict__ bodies, const int* __restrict__ indices, int count, const float4* __restrict__ snapshot, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4* base = bodies + (size_t)indices[i] * 8;
    const float4 l = base[2], a = base[3], l0 = snapshot[2 * i], a0 = snapshot[2 * i + 1];
    float* o = out + (size_t)i * 6;
    o[0] = l.x - l0.x; o[1] = l.y - l0.y; o[2] = l.z - l0.z;
    o[3] = a.x - a0.x; o[4] = a.y - a0.y; o[5] = a.z - a0.z;
}
__global__ void boundary_apply_kernel(float4* bodies, const int* __restrict__ indices, int count, float4* snapshot, const float* __restrict__ sums) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float4* base = bodies + (size_t)indices[i] * 8;
    const float* d = sums + (size_t)i * 6;
    float4 l0 = snapshot[2 * i], a0 = snapshot[2 * i + 1];
    const float4 l = make_float4(l0.x + d[0], l0.y + d[1], l0.z + d[2], l0.w);
    const float4 a = make_float4(a0.x + d[3], a0.y + d[4], a0.z + d[5], a0.w);
    base[2] = l; base[3] = a;
    snapshot[2 * i] = l; snapshot[2 * i + 1] = a;  // the next pass's deltas are relative to the synchronised value
}

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
// LDS: [8 planes of float4 x ncap body slots: the BodyDynamics record, one plane per 16-byte field][work items][flags, counters].
// Slot numbering is rotated by the host inside every group of 16 (slot = (i & ~15) | ((i + (i >> 4)) & 15)) so that the regular
// "same joint of consecutive ragdolls" access pattern (lane stride = island size) spreads over all 16 bank slots of ds_read_b128.
// ------------------------------------------------------------------------------------------------
constexpr int kPlanes = 8;
constexpr int kClusterThreads = 1024;

typedef __attribute__((address_space(1))) float gfloat;  // global
typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(3))) unsigned lds_u32;  // LDS: ds_read/ds_write, lgkmcnt only (a generic pointer would poll with flat loads and drag vmcnt in)

struct ClusterShared {
    float4* planes;        // [kPlanes][ncap]
    int ncap;
    ClusterItem* items;
    volatile lds_u32* flags;  // per item: epoch of the last completed pass
    lds_u32* batch_done;      // per batch: items completed, monotonic over passes (fallback for items with too many predecessors)
    int* lbib;                // batch -> first item of the cluster (batch_count + 1 entries)
    lds_u32* counter;         // item claim counter, monotonic
    int batch_count;
    unsigned* status;      // global: [0] != 0 when a wait ran out of patience (a scheduling bug, never expected); [1..7] first offender
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
        float4 i0 = base[6 * n], i1 = base[7 * n];
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
// wait makes that independent of the pipeline's internals.
__device__ __forceinline__ void publish_item(volatile lds_u32* flag, lds_u32* batch_counter, unsigned epoch) {
    unsigned long long saved;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "ds_write_b32 %[fa], %[e]\n\t"
        "ds_add_u32 %[ba], %[one]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sv] "=&s"(saved)
        : [fa] "v"(lds_address(flag)), [e] "v"(epoch), [ba] "v"(lds_address(batch_counter)), [one] "v"(1u)
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
// One bounded poll loop (all lanes read the same LDS word: a broadcast ds_read).
__device__ __forceinline__ void wait_word(const ClusterShared& sh, const volatile lds_u32* word, unsigned want, int kind, int k, int what) {
    unsigned spins = 0, seen;
    while ((seen = (unsigned)__builtin_amdgcn_readfirstlane((int)*word)) < want) {
        __builtin_amdgcn_s_sleep(1);  // 64 clocks; polling back to back or sleeping twice as long measures the same
        if (++spins > kSpinLimit) { report_stall(sh.status, *sh.counter, kind, k, what, want, seen); break; }
        if ((spins & 4095u) == 0 && __hip_atomic_load(sh.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // somebody already gave up
    }
}
// Block until every predecessor of the item has published: same-pass predecessors must have finished `epoch`; with CROSS (a Solve item: the warm start
// pass before it is not separated by a barrier) the last touchers of the bodies this item touches first must have finished `epoch - 1`.
template <bool CROSS>
__device__ __forceinline__ void wait_predecessors(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h, int k, unsigned epoch) {
    for (int q = 0; q < h.npred; ++q) {
        const int pred = __builtin_amdgcn_readfirstlane((int)it->pred[q]);
        wait_word(sh, sh.flags + pred, epoch, 1, k, pred);
    }
    if (CROSS) {
        for (int q = 0; q < h.nxpred; ++q) {
            const int pred = __builtin_amdgcn_readfirstlane((int)it->xpred[q]);
            wait_word(sh, sh.flags + pred, epoch - 1, 3, k, pred);
        }
    }
    if (h.overflow) {  // more predecessors than the item records: wait for every item of every earlier batch
        for (int b = 0; b < h.batch; ++b)
            wait_word(sh, (const volatile lds_u32*)sh.batch_done + b, epoch * (unsigned)__builtin_amdgcn_readfirstlane(sh.lbib[b + 1] - sh.lbib[b]), 2, k, b);
    }
    if (CROSS && h.xoverflow) {  // ... and for the whole previous pass
        for (int b = 0; b < sh.batch_count; ++b)
            wait_word(sh, (const volatile lds_u32*)sh.batch_done + b, (epoch - 1) * (unsigned)__builtin_amdgcn_readfirstlane(sh.lbib[b + 1] - sh.lbib[b]), 4, k, b);
    }
    asm volatile("" ::: "memory");  // nothing below may be hoisted above the polls
}

struct ItemStamps { unsigned long long loaded, pre_gate, post_gate; };  // trace builds only

// The gate the cluster path hands to the constraint functions: wait for the item's predecessors, then gather the velocities.
template <int ACC_A, int ACC_B, int BODIES, bool CROSS, bool TRACE>
struct ClusterGate {
    static constexpr bool kPin = true;  // the constraint pins its velocity-independent values before calling: they are computed while the predecessors still run
    const ClusterShared& sh; const ClusterItem* it; const ItemHeader& h; int k; unsigned epoch; int ra, rb; DBody& A; DBody& B; ItemStamps& stamps;
    __device__ __forceinline__ void operator()(BodyVel&, BodyVel&) const {
        if (TRACE) stamps.pre_gate = __builtin_readcyclecounter();
        wait_predecessors<CROSS>(sh, it, h, k, epoch);
        __builtin_amdgcn_s_setprio(3);  // from here to the publish the item is on its bodies' critical path: issue ahead of waves still preparing theirs
        if (TRACE) stamps.post_gate = __builtin_readcyclecounter();
        load_velocity_lds<ACC_A>(sh, ra, A);
        if (BODIES == 2) load_velocity_lds<ACC_B>(sh, rb, B);
    }
};


template <class F, int STAGE, bool TRACE>
__device__ __forceinline__ void run_cluster_constraint(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h, int k, int lane, unsigned epoch,
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
    const int ra = lrefs[i];
    const int rb = (F::bodies == 2) ? lrefs[stride + i] : -1;
    _Pragma("unroll") for (int f = 0; f < F::prestepFloats; ++f) p[f] = prestep[(size_t)f * stride + i];
    if (STAGE != kStageIncremental) { _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) a[f] = accum[(size_t)f * stride + i]; }
    DBody A, B;
    if (STAGE == kStageIncremental) {  // reads velocities, writes only this constraint's depths: no ordering inside the stage
        load_body_lds<kAccessOnlyVelocity>(sh, ra, A);
        if (F::bodies == 2) load_body_lds<kAccessOnlyVelocity>(sh, rb, B); else load_body_lds<0>(sh, 0, B);
        F::incrementalUpdate(dt, A.vel, B.vel, p);
        if constexpr (F::incremental) {
            if (active) { _Pragma("unroll") for (int cidx = 0; cidx < F::contacts; ++cidx) prestep[(size_t)F::depthRow(cidx) * stride + i] = p[F::depthRow(cidx)]; }
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
    ClusterGate<accA, accB, F::bodies, STAGE == kStageSolve, TRACE> gate{sh, it, h, k, epoch, ra, rb, A, B, stamps};
    if (STAGE == kStageWarmStart) F::warmStart(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, p, a, A.vel, B.vel, gate);
    else F::solve(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, dt, inv_dt, p, a, A.vel, B.vel, gate);
    store_velocity_lds<accA>(sh, active ? ra : -1, A);   // -1: never stored (same rule as kinematic / empty references)
    if (F::bodies == 2) store_velocity_lds<accB>(sh, active ? rb : -1, B);
    publish_item(sh.flags + k, sh.batch_done + h.batch, epoch);
    __builtin_amdgcn_s_setprio(0);
    if (STAGE == kStageSolve && active) {  // off the critical path: nothing reads the impulses before the next pass (a barrier away)
        _Pragma("unroll") for (int f = 0; f < F::impulseFloats; ++f) accum[(size_t)f * stride + i] = a[f];
    }
}

using DC1O = Contact<1, false>; using DC2O = Contact<2, false>; using DC3O = Contact<3, false>; using DC4O = Contact<4, false>;
using DC1T = Contact<1, true>; using DC2T = Contact<2, true>; using DC3T = Contact<3, true>; using DC4T = Contact<4, true>;

template <int STAGE, bool TRACE, bool WIDE>
__device__ __forceinline__ void run_cluster_item(const ClusterShared& sh, const ClusterItem* it, const ItemHeader& h, int k, int lane, unsigned epoch,
                                                 unsigned* __restrict__ slab, float dt, float inv_dt, ItemStamps& stamps) {
#define BEPU_CASE(ID, F) case ID: run_cluster_constraint<F, STAGE, TRACE>(sh, it, h, k, lane, epoch, slab, dt, inv_dt, stamps); break;
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
            if constexpr (STAGE != kStageIncremental) {  // only contacts need incremental updates (RequiresIncrementalSubstepUpdates)
                switch (h.type_id) {
                    BD_HOT_JOINT_TYPES(BEPU_CASE)
                    default:
                        if constexpr (WIDE) {  // SURVEY 8(f) types live in a second kernel variant: scenes made of the sixteen hot-path types keep the leaner one
                            switch (h.type_id) {
                                BD_WIDENED_JOINT_TYPES(BEPU_CASE)
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
template <int STAGE0, bool TRACE, bool WIDE>
__device__ __forceinline__ void run_cluster_sweep(const ClusterShared& sh, int item_count, int solve_items, int lane, int wave, unsigned epoch, unsigned claim_base,
                                                  unsigned* __restrict__ slab, float dt, float inv_dt, unsigned long long* trace) {
    for (;;) {
        const int v = (int)(claim_next(sh.counter) - claim_base);
        if (v >= item_count + solve_items) break;
        const bool second = v >= item_count;
        const int k = second ? v - item_count : v;
        const unsigned item_epoch = second ? epoch + 1 : epoch;
        const ClusterItem* it = sh.items + k;
        const ItemHeader h = read_item(it);
        unsigned long long t0 = 0;
        if (TRACE) t0 = __builtin_readcyclecounter();
        ItemStamps stamps = {0, 0, 0};
        if (STAGE0 == kStageWarmStart && !second) run_cluster_item<kStageWarmStart, TRACE, WIDE>(sh, it, h, k, lane, item_epoch, slab, dt, inv_dt, stamps);
        else run_cluster_item<kStageSolve, TRACE, WIDE>(sh, it, h, k, lane, item_epoch, slab, dt, inv_dt, stamps);
        if (TRACE && trace && blockIdx.x == 0 && lane == 0) {
            unsigned long long* rec = trace + ((size_t)(item_epoch - 1) * item_count + k) * 8;
            rec[4] = stamps.loaded; rec[5] = stamps.pre_gate; rec[6] = stamps.post_gate; rec[7] = 0;
            rec[0] = t0; rec[1] = __builtin_readcyclecounter();
            rec[2] = (unsigned long long)wave | ((unsigned long long)h.type_id << 8) | ((unsigned long long)h.batch << 16) | ((unsigned long long)((STAGE0 == kStageWarmStart && !second) ? kStageWarmStart : kStageSolve) << 32);
            rec[3] = (unsigned long long)h.count;
        }
    }
}

template <int THREADS, bool TRACE, bool WIDE>
__global__ __launch_bounds__(THREADS) void cluster_kernel(const ClusterDesc* __restrict__ clusters, const ClusterItem* __restrict__ items,
                                                                   const int* __restrict__ batch_item_begin, const int* __restrict__ cluster_bodies,
                                                                   float4* bodies, unsigned* __restrict__ slab, ClusterParams cp, int ncap, int max_items,
                                                                   unsigned long long* trace, unsigned* status, unsigned long long* cycles) {
    const unsigned long long kernel_t0 = __builtin_readcyclecounter();
    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    ClusterShared sh;
    sh.planes = lds;
    sh.ncap = ncap;
    sh.items = reinterpret_cast<ClusterItem*>(lds + kPlanes * ncap);
    unsigned* words = reinterpret_cast<unsigned*>(lds + kPlanes * ncap + max_items * (int)(sizeof(ClusterItem) / 16));
    sh.flags = (volatile lds_u32*)words;
    sh.batch_done = (lds_u32*)(words + max_items);
    sh.lbib = reinterpret_cast<int*>(words + max_items + kFallbackBatchLimit + 1);
    sh.counter = (lds_u32*)(words + max_items + 2 * (kFallbackBatchLimit + 1) + 1);
    sh.status = status;
    sh.batch_count = cp.batch_count;
    const ClusterDesc cd = clusters[blockIdx.x];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    const float dt = cp.sp.dt, inv_dt = cp.sp.inv_dt;
    const int* slots = cluster_bodies + cd.body_begin;  // slot -> body index (bit 30: kinematic, private read-only copy; -1: unused slot)
    // ---- stage the cluster in LDS: bodies (one plane per 16-byte field), work items, batch -> item ranges; clear the sync words ----
    for (int j = tid; j < cd.slot_count * kPlanes; j += blockDim.x) {
        const int slot = j >> 3, v = j & 7;
        const int g = slots[slot];
        lds[v * ncap + slot] = g >= 0 ? bodies[(size_t)(g & kRefMask) * 8 + v] : make_float4(0, 0, 0, 0);
    }
    {
        const int4* src = reinterpret_cast<const int4*>(items + cd.item_begin);
        int4* dst = reinterpret_cast<int4*>(sh.items);
        for (int j = tid; j < cd.item_count * (int)(sizeof(ClusterItem) / 16); j += blockDim.x) dst[j] = src[j];
    }
    for (int j = tid; j < max_items + kFallbackBatchLimit + 1; j += blockDim.x) words[j] = 0;  // flags + batch_done
    for (int j = tid; j <= cp.batch_count; j += blockDim.x) sh.lbib[j] = batch_item_begin[cd.batch_item_offset + j] - cd.item_begin;
    if (tid == 0) *sh.counter = 0;
    __syncthreads();

    unsigned epoch = 0, claim_base = 0;
    for (int s = 0; s < cp.substeps; ++s) {
        if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(&sh.status[10], (unsigned)s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (s > 0) {  // Solver_Solve.cs:1427-1439: contact depths advance with the pre-integration velocities
            for (int k = wave; k < cd.item_count; k += nwaves) {
                const ClusterItem* it = sh.items + k;
                const ItemHeader h = read_item(it);
                if (!isContactType(h.type_id)) continue;
                ItemStamps stamps = {0, 0, 0};
                run_cluster_item<kStageIncremental, false, WIDE>(sh, it, h, k, lane, 0u, slab, dt, inv_dt, stamps);
            }
            __syncthreads();
        }
        // Integration of every constrained body of the cluster (TypeProcessor.cs:1204-1283, PoseIntegrator.cs:451-535):
        // substep 0 velocity only, later substeps pose then velocity; world inverse inertia refreshed either way.
        for (int j = tid; j < cd.slot_count; j += blockDim.x) {
            const int g = slots[j];
            if (g < 0) continue;
            float4* r = lds + j;
            float4 q4 = r[0], p4 = r[ncap], l4 = r[2 * ncap], a4 = r[3 * ncap];
            Q ori = {q4.x, q4.y, q4.z, q4.w};
            V3 pos = {p4.x, p4.y, p4.z};
            BodyVel vel = {{l4.x, l4.y, l4.z}, {a4.x, a4.y, a4.z}};
            if (s > 0) {
                pos = add(pos, scale(vel.lin, dt));
                ori = integrateOrientation(ori, vel.ang, dt * 0.5f);
                r[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
                r[ncap] = make_float4(pos.x, pos.y, pos.z, p4.w);
            }
            if ((unsigned)g < kDynamicLimit) {
                const float4 i0 = r[4 * ncap], i1 = r[5 * ncap];
                Sym3 local = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
                Sym3 world = rotateInverseInertia(local, ori);
                velocity_callback(cp.sp, vel);
                r[2 * ncap] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
                r[3 * ncap] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
                r[6 * ncap] = make_float4(world.xx, world.yx, world.yy, world.zx);
                r[7 * ncap] = make_float4(world.zy, world.zz, i1.z, r[7 * ncap].w);
            } else if (cp.integrate_velocity_for_kinematics) {  // kinematic: private copy, same arithmetic as the global kinematic pass
                velocity_callback(cp.sp, vel);
                r[2 * ncap] = make_float4(vel.lin.x, vel.lin.y, vel.lin.z, l4.w);
                r[3 * ncap] = make_float4(vel.ang.x, vel.ang.y, vel.ang.z, a4.w);
            }
        }
        __syncthreads();
        ++epoch;
        const int fused = cp.iters[s] > 0 ? cd.item_count : 0;  // the first velocity iteration rides in the warm start's claim sequence
        run_cluster_sweep<kStageWarmStart, TRACE, WIDE>(sh, cd.item_count, fused, lane, wave, epoch, claim_base, slab, dt, inv_dt, trace);
        claim_base += cd.item_count + fused + nwaves;  // every wave makes exactly one failing claim per sweep
        if (fused) ++epoch;
        __syncthreads();
        for (int iter = 1; iter < cp.iters[s]; ++iter) {
            ++epoch;
            run_cluster_sweep<kStageSolve, TRACE, WIDE>(sh, cd.item_count, 0, lane, wave, epoch, claim_base, slab, dt, inv_dt, trace);
            claim_base += cd.item_count + nwaves;
            __syncthreads();
        }
    }
    // Trailing pose integration of constrained bodies (PoseIntegrator.cs:684-691) and write-back.
    for (int j = tid; j < cd.slot_count; j += blockDim.x) {
        const int g = slots[j];
        if ((unsigned)g >= kDynamicLimit) continue;  // unused slot, or kinematic (advanced in global memory by kinematic_substeps_kernel + the final pass)
        const float4* r = lds + j;
        float4 q4 = r[0], p4 = r[ncap], l4 = r[2 * ncap], a4 = r[3 * ncap];
        Q ori = {q4.x, q4.y, q4.z, q4.w};
        V3 pos = {p4.x, p4.y, p4.z};
        V3 lin = {l4.x, l4.y, l4.z}, ang = {a4.x, a4.y, a4.z};
        ori = integrateOrientation(ori, ang, dt * 0.5f);
        pos = add(pos, scale(lin, dt));
        float4* gb = bodies + (size_t)g * 8;
        gb[0] = make_float4(ori.x, ori.y, ori.z, ori.w);
        gb[1] = make_float4(pos.x, pos.y, pos.z, p4.w);
        gb[2] = l4;
        gb[3] = a4;
        gb[6] = r[6 * ncap];
        gb[7] = r[7 * ncap];
    }
    if (tid == 0) cycles[blockIdx.x] = __builtin_readcyclecounter() - kernel_t0;  // shader clocks this cluster took: a clock-frequency-independent measure
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Host side
// Ranged in-place update of one type batch's prestep / accumulated-impulse rows from the caller's AOSOA bundles (bepuhip_update_prestep /
// bepuhip_update_accumulated_impulses): one thread per constraint of the range, `fields` strided stores each. `device_index` maps the constraint's
// index inside the type batch to its slot in the SoA rows (identity unless the island schedule permuted the batch); null = identity.
__global__ __launch_bounds__(256) void scatter_bundles_kernel(const float* __restrict__ bundles, float* __restrict__ rows, const int* __restrict__ device_index,
                                                              int first_constraint, int constraint_count, int fields, int stride, int W) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= constraint_count) return;
    const int h = first_constraint + j;
    const int d = device_index ? device_index[h] : h;
    const float* src = bundles + (size_t)(j / W) * fields * W + (j % W);
    for (int f = 0; f < fields; ++f) rows[(size_t)f * stride + d] = src[(size_t)f * W];
}
// The inverse, for ranged read-back (bepuhip_get_*_range).
__global__ __launch_bounds__(256) void gather_bundles_kernel(float* __restrict__ bundles, const float* __restrict__ rows, const int* __restrict__ device_index,
                                                             int first_constraint, int constraint_count, int fields, int stride, int W) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= constraint_count) return;
    const int h = first_constraint + j;
    const int d = device_index ? device_index[h] : h;
    float* dst = bundles + (size_t)(j / W) * fields * W + (j % W);
    for (int f = 0; f < fields; ++f) dst[(size_t)f * W] = rows[(size_t)f * stride + d];
}

// ------------------------------------------------------------------------------------------------
// cluster_kernel instantiations: register budget (launch bounds) x trace x type set. The traced build exists for the 1024-thread budget only (a kernel
// compiled for 1024 threads runs any smaller workgroup).
#ifdef BEPUHIP_FAST_BUILD  // kernel-tuning builds (tools/): one register budget, hot-path type set only
#define BEPU_CLUSTER_VARIANTS(X) X(1024, false, false) X(1024, true, false)
#else
#define BEPU_CLUSTER_VARIANTS(X) X(512, false, false) X(768, false, false) X(1024, false, false) X(1024, true, false) \
                                 X(512, false, true) X(768, false, true) X(1024, false, true) X(1024, true, true)
#endif
static const void* cluster_kernel_variant(int threads, bool trace, bool wide) {
#ifdef BEPUHIP_FAST_BUILD
    return trace ? (const void*)cluster_kernel<1024, true, false> : (const void*)cluster_kernel<1024, false, false>;
#endif
    const int budget = trace ? 1024 : (threads > 768 ? 1024 : threads > 512 ? 768 : 512);
#define X(T, TR, W) if (budget == T && trace == TR && wide == W) return (const void*)cluster_kernel<T, TR, W>;
    BEPU_CLUSTER_VARIANTS(X)
#undef X
    return (const void*)cluster_kernel<1024, false, true>;
}

static thread_local std::string g_last_error;
static int32_t fail(int32_t code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(BEPUHIP_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

using C1O = Contact<1, false>; using C2O = Contact<2, false>; using C3O = Contact<3, false>; using C4O = Contact<4, false>;
using C1T = Contact<1, true>; using C2T = Contact<2, true>; using C3T = Contact<3, true>; using C4T = Contact<4, true>;
struct TypeInfoH { int bodies, prestep, impulse; bool incremental; };
static bool type_info(int id, TypeInfoH& t) {
#define TI(T) { t = {T::bodies, T::prestepFloats, T::impulseFloats, T::incremental}; return true; }
    switch (id) {
        case kContact1OneBody: TI(C1O) case kContact2OneBody: TI(C2O)
        case kContact3OneBody: TI(C3O) case kContact4OneBody: TI(C4O)
        case kContact1: TI(C1T) case kContact2: TI(C2T)
        case kContact3: TI(C3T) case kContact4: TI(C4T)
#define X(ID, T) case ID: TI(T)
        BD_JOINT_TYPES(X)
        BD_NONCONVEX_CONTACT_TYPES(X)
#undef X
    }
#undef TI
    return false;
}

static bool is_widened_type(int id) {
    switch (id) {
#define X(ID, T) case ID: return true;
        BD_WIDENED_JOINT_TYPES(X)
        BD_NONCONVEX_CONTACT_TYPES(X)
#undef X
    }
    return false;
}

struct HostTypeBatch {
    int batch, type_id, count, stride;
    TypeInfoH info;
    size_t refs_off, prestep_off, accum_off, lrefs_off;  // offsets (in 4-byte words) into the constraint slab
    std::vector<int32_t> perm;      // cluster path: device index -> host index inside the type batch (empty = identity)
    std::vector<int32_t> inv;       // host index -> device index (lazily built)
    int perm_inverse(int host_index) {
        if (inv.empty()) { inv.resize(perm.size()); for (size_t d = 0; d < perm.size(); ++d) inv[perm[d]] = (int32_t)d; }
        return inv[host_index];
    }
    int32_t* d_device_index = nullptr;  // device copy of `inv` for the ranged update / read-back kernels (allocated on first use)
    std::vector<int32_t> lrefs_soa; // cluster path: local (LDS) body indices
    std::vector<int32_t> refs_soa;
    std::vector<float> prestep_soa, accum_soa;  // host staging until end_constraints
};

struct GraphKey {
    std::vector<int> iterations;
    float dt;
    bepuhip_integrator integ;
    bool operator<(const GraphKey& o) const {
        if (iterations != o.iterations) return iterations < o.iterations;
        if (dt != o.dt) return dt < o.dt;
        return memcmp(&integ, &o.integ, sizeof(integ)) < 0;
    }
};

struct bepuhip_ctx {
    int device = 0, W = 8, flags = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    float4* d_bodies = nullptr;
    float4* d_bodies0 = nullptr;  // pristine snapshot for reset_state
    unsigned* d_flags = nullptr;
    int body_count = 0, body_capacity = 0;
    int* d_kin = nullptr;
    int kin_count = 0;
    std::vector<int32_t> kin_indices;
    // constraints
    bool building = false, built = false;
    int batch_count = 0;
    std::vector<HostTypeBatch> tbs;
    std::vector<int> batch_begin;        // tbs index of each batch's first type batch (size batch_count+1)
    std::vector<int> batch_blocks;       // grid size per batch
    uint32_t* d_slab = nullptr;          // all refs/prestep/accum
    uint32_t* d_slab0 = nullptr;         // pristine snapshot
    float* d_stage = nullptr;            // staging for ranged updates / read-backs (caller's AOSOA bundles)
    size_t stage_floats = 0;
    size_t slab_words = 0;
    DevTypeBatch* d_tbs = nullptr;       // per (batch) descriptors, solve/warm-start grids
    DevTypeBatch* d_inc_tbs = nullptr;   // incremental-update grid (contacts of all batches)
    int inc_tb_count = 0, inc_blocks = 0;
    int64_t total_constraints = 0;
    // cluster path
    bool clusters_enabled = false;
    bool has_widened_types = false;  // any type outside SURVEY 8(a)'s sixteen: selects the wider cluster_kernel variant
    int cluster_count = 0, cluster_max_slots = 0, cluster_max_items = 0, cluster_total_items = 0;
    ClusterDesc first_cluster = {0, 0, 0, 0, 0};
    int* d_requirk = nullptr;            // conserving angular modes: per batch, the bodies momentum_requirk_kernel transforms in substep 0
    std::vector<int> requirk_begin;     // batch -> offset into d_requirk (batch_count + 1 entries)
    int* d_boundary = nullptr;          // boundary body indices (see bepuhip_set_boundary_bodies)
    float4* d_boundary_snapshot = nullptr;
    float* d_boundary_buf = nullptr;     // count * 6 floats staging for host-pointer exchanges
    int boundary_count = 0;
    unsigned long long* d_cycles = nullptr;  // per cluster: shader clocks of the last cluster_kernel launch
    unsigned* d_status = nullptr;  // cluster schedule watchdog words (see report_stall)
    unsigned long long* d_trace = nullptr;  // optional per-item timeline of cluster 0 (diagnostics)
    size_t trace_words = 0;
    ClusterDesc* d_clusters = nullptr;
    ClusterItem* d_items = nullptr;
    int* d_batch_item_begin = nullptr;
    int* d_cluster_bodies = nullptr;
    int* d_clustered_dynamic = nullptr;
    int clustered_dynamic_count = 0;
    int* d_kinlist = nullptr;       // constrained kinematic body indices derived from the body references
    int kinlist_count = 0;
    // measurement
    float last_ms = 0;
    int64_t last_constraint_iterations = 0;
    bool profiling = false;
    float prof_ms[6] = {0, 0, 0, 0, 0, 0};
    int prof_launches[6] = {0, 0, 0, 0, 0, 0};
    std::map<GraphKey, hipGraphExec_t> graphs;
};

static void free_constraints(bepuhip_ctx* c) {
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    if (c->d_slab) hipFree(c->d_slab);
    if (c->d_slab0) hipFree(c->d_slab0);
    if (c->d_tbs) hipFree(c->d_tbs);
    if (c->d_inc_tbs) hipFree(c->d_inc_tbs);
    if (c->d_clusters) hipFree(c->d_clusters);
    if (c->d_items) hipFree(c->d_items);
    if (c->d_batch_item_begin) hipFree(c->d_batch_item_begin);
    if (c->d_cluster_bodies) hipFree(c->d_cluster_bodies);
    if (c->d_clustered_dynamic) hipFree(c->d_clustered_dynamic);
    if (c->d_kinlist) hipFree(c->d_kinlist);
    if (c->d_requirk) hipFree(c->d_requirk);
    c->d_requirk = nullptr; c->requirk_begin.clear();
    if (c->d_trace) hipFree(c->d_trace);
    c->d_trace = nullptr; c->trace_words = 0;
    if (c->d_cycles) hipFree(c->d_cycles);
    c->d_cycles = nullptr;
    c->d_clusters = nullptr; c->d_items = nullptr; c->d_batch_item_begin = nullptr; c->d_cluster_bodies = nullptr;
    c->d_clustered_dynamic = nullptr; c->d_kinlist = nullptr;
    c->clusters_enabled = false; c->cluster_count = 0; c->clustered_dynamic_count = 0; c->kinlist_count = 0;
    c->d_slab = c->d_slab0 = nullptr;
    c->d_tbs = c->d_inc_tbs = nullptr;
    for (auto& tb : c->tbs) if (tb.d_device_index) hipFree(tb.d_device_index);
    c->tbs.clear();
    c->built = false;
}


// ---- cluster planning (host, once per topology upload) ----
// Islands = connected components through dynamic bodies (kinematic references never connect: they are read-only to the solver).
// Whole islands are packed, in body-index order, into clusters of at most `cap` LDS-resident bodies; each type batch is
// reordered so that every cluster's constraints are contiguous (coalesced loads per <=64-lane work item), and every work item
// records which earlier items last touched its dynamic bodies (the only ordering the solve has to respect).
struct ClusterPlan {
    bool enabled = false;
    std::vector<ClusterDesc> clusters;
    std::vector<ClusterItem> items;
    std::vector<int> batch_item_begin, cluster_bodies, clustered_dynamic, kinlist;
    int max_slots = 0, max_items = 0;
};

static int env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : fallback;
}

constexpr size_t kLdsBudgetBytes = 160 * 1024 - 256;
static size_t cluster_sync_words(int max_items) { return (size_t)max_items + 2 * (kFallbackBatchLimit + 1) + 2; }
static size_t cluster_lds_bytes(int ncap, int max_items) {
    return (size_t)kPlanes * ncap * 16 + (size_t)max_items * sizeof(ClusterItem) + (cluster_sync_words(max_items) + 3) / 4 * 16;
}
// Slot rotation inside every group of 16 (see the LDS layout note above cluster_kernel).
static inline int rotated_slot(int i) { return (i & ~15) | ((i + (i >> 4)) & 15); }

static void plan_clusters(bepuhip_ctx* c, ClusterPlan& plan) {
    int universe = 0;
    for (auto& tb : c->tbs)
        for (int32_t r : tb.refs_soa)
            if (r >= 0) universe = std::max(universe, (r & kRefMask) + 1);
    // kinematic list (Solver.ConstrainedKinematicHandles equivalent), always built
    {
        std::vector<uint8_t> seen(universe, 0);
        for (auto& tb : c->tbs)
            for (int k = 0; k < tb.info.bodies; ++k)
                for (int i = 0; i < tb.count; ++i) {
                    int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r >= kDynamicLimit && r >= 0 && !seen[r & kRefMask]) { seen[r & kRefMask] = 1; plan.kinlist.push_back(r & kRefMask); }
                }
    }
    if ((c->flags & BEPUHIP_FLAG_NO_CLUSTERS) || env_int("BEPUHIP_NO_CLUSTERS", 0) || universe == 0 || c->total_constraints == 0 || c->batch_count > kFallbackBatchLimit) return;
    std::vector<int32_t> parent(universe);
    for (int i = 0; i < universe; ++i) parent[i] = i;
    auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    std::vector<uint8_t> is_dyn(universe, 0);
    for (auto& tb : c->tbs) {
        for (int i = 0; i < tb.count; ++i) {
            int first = -1;
            for (int k = 0; k < tb.info.bodies; ++k) {
                int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                if ((uint32_t)r >= kDynamicLimit) continue;
                is_dyn[r] = 1;
                if (first < 0) first = find(r);
                else { int o = find(r); if (o != first) { if (o < first) std::swap(o, first); parent[o] = first; } }
            }
            if (first < 0) return;  // a constraint with no dynamic body: leave everything to the global path
        }
    }
    // component sizes (root = smallest body index of the component)
    std::vector<int32_t> comp_size(universe, 0);
    int64_t total_dyn = 0;
    int32_t largest = 0;
    for (int i = 0; i < universe; ++i) if (is_dyn[i]) { largest = std::max(largest, ++comp_size[find(i)]); ++total_dyn; }
    for (int i = 0; i < universe; ++i) if (is_dyn[i]) find(i);  // full path compression: parent[i] is the root from here on
    int cap = env_int("BEPUHIP_CLUSTER_BODIES", 0);
    if (cap <= 0) {
        // default: one resident workgroup per CU (the kernel's LDS footprint admits one workgroup per CU), a single round
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        int64_t target = (total_dyn + (cus * 31 / 32) - 1) / std::max(1, cus * 31 / 32);
        cap = (int)std::min<int64_t>(std::max<int64_t>(target, 64), 1200);
    }
    if (largest > cap) cap = largest;
    // ---- phase A: find a cap whose clusters fit the LDS budget (no mutation yet) ----
    std::vector<int32_t> cluster_of(universe, -1);  // by component root
    std::vector<std::vector<int32_t>> cl_of_constraint(c->tbs.size());
    int nclusters = 0;
    for (int attempt = 0;; ++attempt) {
        nclusters = 0;
        int cur = 0;
        for (int i = 0; i < universe; ++i) {
            if (!is_dyn[i] || parent[i] != i) continue;  // roots only, ascending
            if (nclusters == 0 || cur + comp_size[i] > cap) { ++nclusters; cur = 0; }
            cluster_of[i] = nclusters - 1;
            cur += comp_size[i];
        }
        std::vector<int32_t> dyn_count(nclusters, 0), item_count(nclusters, 0);
        std::vector<std::vector<int32_t>> kin_seen(nclusters);
        for (int i = 0; i < universe; ++i) if (is_dyn[i]) dyn_count[cluster_of[parent[i]]]++;
        std::vector<int32_t> per_cluster(nclusters);
        for (size_t t = 0; t < c->tbs.size(); ++t) {
            HostTypeBatch& tb = c->tbs[t];
            cl_of_constraint[t].resize(tb.count);
            std::fill(per_cluster.begin(), per_cluster.end(), 0);
            for (int i = 0; i < tb.count; ++i) {
                int cl = -1;
                for (int k = 0; k < tb.info.bodies && cl < 0; ++k) {
                    int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r < kDynamicLimit) cl = cluster_of[parent[r]];
                }
                cl_of_constraint[t][i] = cl;
                per_cluster[cl]++;
                for (int k = 0; k < tb.info.bodies; ++k) {
                    int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r >= kDynamicLimit) {
                        auto& ks = kin_seen[cl];
                        if (std::find(ks.begin(), ks.end(), r & kRefMask) == ks.end()) ks.push_back(r & kRefMask);
                    }
                }
            }
            for (int cl = 0; cl < nclusters; ++cl) item_count[cl] += (per_cluster[cl] + 63) / 64;
        }
        int max_slots = 0, max_items = 0;
        for (int cl = 0; cl < nclusters; ++cl) {
            max_slots = std::max(max_slots, (dyn_count[cl] + (int)kin_seen[cl].size() + 15) / 16 * 16);
            max_items = std::max(max_items, item_count[cl]);
        }
        if (max_items < 65536 && cluster_lds_bytes(max_slots, max_items) <= kLdsBudgetBytes) break;
        if (cap <= largest || attempt > 24) return;  // an island (plus its work items) does not fit one workgroup: global path
        cap = std::max<int>(largest, cap * 7 / 8);
    }
    // ---- phase B: local slots, reordered type batches, work items with predecessor lists ----
    std::vector<std::vector<int32_t>> cl_bodies(nclusters);  // natural local order: dynamics ascending, kinematics appended on first use
    std::vector<int32_t> local_of(universe, -1);
    for (int i = 0; i < universe; ++i)
        if (is_dyn[i]) { int cl = cluster_of[parent[i]]; local_of[i] = (int)cl_bodies[cl].size(); cl_bodies[cl].push_back(i); plan.clustered_dynamic.push_back(i); }
    std::vector<std::vector<std::pair<int32_t, int32_t>>> cl_kin(nclusters);  // (kinematic body, natural local index)
    auto kin_local = [&](int cl, int body) {
        for (auto& kv : cl_kin[cl]) if (kv.first == body) return kv.second;
        int l = (int)cl_bodies[cl].size();
        cl_bodies[cl].push_back(body | (int)kDynamicLimit);
        cl_kin[cl].push_back({body, l});
        return l;
    };
    std::vector<std::vector<int32_t>> last_toucher(nclusters);  // by slot: cluster-relative index of the item that last touched the (dynamic) body
    for (int cl = 0; cl < nclusters; ++cl) last_toucher[cl].assign((cl_bodies[cl].size() + 15) / 16 * 16, -1);
    std::vector<std::vector<ClusterItem>> cl_items(nclusters);
    std::vector<std::vector<std::pair<int32_t, int32_t>>> first_touch(nclusters);  // (item, slot): the item is the slot's first toucher in a pass
    // Items are claimed in list order. Inside a batch any order is legal (a batch never references a body twice); the types that move the most data per
    // constraint go first so that their loads and velocity-independent work start as early as the claim sequence allows.
    std::vector<size_t> visit(c->tbs.size());
    for (size_t t = 0; t < visit.size(); ++t) visit[t] = t;
    if (env_int("BEPUHIP_CLUSTER_ORDER", 1) != 0)
        std::stable_sort(visit.begin(), visit.end(), [&](size_t a, size_t b) {
            const HostTypeBatch &x = c->tbs[a], &y = c->tbs[b];
            if (x.batch != y.batch) return x.batch < y.batch;
            return x.info.prestep + 2 * x.info.impulse > y.info.prestep + 2 * y.info.impulse;
        });
    for (size_t t : visit) {
        HostTypeBatch& tb = c->tbs[t];
        const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse;
        const std::vector<int32_t>& clc = cl_of_constraint[t];
        tb.perm.resize(tb.count);
        for (int i = 0; i < tb.count; ++i) tb.perm[i] = i;
        std::stable_sort(tb.perm.begin(), tb.perm.end(), [&](int a, int b) { return clc[a] < clc[b]; });
        std::vector<int32_t> refs((size_t)nb * tb.stride, -1), lrefs((size_t)nb * tb.stride, -1);
        std::vector<float> pre((size_t)pf * tb.stride, 0.0f), acc((size_t)imf * tb.stride, 0.0f);
        for (int d = 0; d < tb.count; ++d) {
            const int h = tb.perm[d], cl = clc[h];
            for (int k = 0; k < nb; ++k) {
                int32_t r = tb.refs_soa[(size_t)k * tb.stride + h];
                refs[(size_t)k * tb.stride + d] = r;
                lrefs[(size_t)k * tb.stride + d] = ((uint32_t)r < kDynamicLimit) ? rotated_slot(local_of[r]) : (rotated_slot(kin_local(cl, r & kRefMask)) | (int)kDynamicLimit);
            }
            for (int f = 0; f < pf; ++f) pre[(size_t)f * tb.stride + d] = tb.prestep_soa[(size_t)f * tb.stride + h];
            for (int f = 0; f < imf; ++f) acc[(size_t)f * tb.stride + d] = tb.accum_soa[(size_t)f * tb.stride + h];
        }
        tb.refs_soa.swap(refs); tb.prestep_soa.swap(pre); tb.accum_soa.swap(acc); tb.lrefs_soa.swap(lrefs);
        for (int d = 0; d < tb.count;) {
            const int cl = clc[tb.perm[d]];
            int e = d;
            while (e < tb.count && clc[tb.perm[e]] == cl) ++e;
            for (int s0 = d; s0 < e; s0 += 64) {
                ClusterItem it;
                memset(&it, 0, sizeof(it));
                it.type_id = tb.type_id; it.count = std::min(64, e - s0); it.stride = tb.stride; it.start = s0;
                it.tb = (int)t; it.shape = nb | (pf << 8) | (imf << 16);
                const int self = (int)cl_items[cl].size();
                int npred = 0, overflow = 0;
                std::vector<int32_t>& lt = last_toucher[cl];
                for (int j = s0; j < s0 + it.count; ++j)
                    for (int k = 0; k < nb; ++k) {
                        const int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + j];
                        if ((uint32_t)lr >= kDynamicLimit) continue;
                        if ((size_t)lr >= lt.size()) lt.resize((size_t)lr + 16, -1);
                        const int pred = lt[lr];
                        if (pred < 0) { first_touch[cl].push_back({self, lr}); continue; }  // this item is the body's first toucher in a pass
                        if (pred == self) continue;
                        bool known = false;
                        for (int q = 0; q < npred; ++q) known |= it.pred[q] == pred;
                        if (known) continue;
                        if (npred < kMaxPreds) it.pred[npred++] = (unsigned short)pred; else overflow = 1;
                    }
                for (int j = s0; j < s0 + it.count; ++j)
                    for (int k = 0; k < nb; ++k) {
                        const int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + j];
                        if ((uint32_t)lr < kDynamicLimit) lt[lr] = self;
                    }
                if (overflow) npred = 0;
                it.batch_npred = (tb.batch & 0xFFFF) | (npred << 16) | (overflow << 24);
                cl_items[cl].push_back(it);
            }
            d = e;
        }
    }
    // Cross-pass predecessors: the last toucher (end of a pass) of every body an item touches first.
    for (int cl = 0; cl < nclusters; ++cl) {
        for (auto& fs : first_touch[cl]) {
            ClusterItem& it = cl_items[cl][fs.first];
            const int last = last_toucher[cl][fs.second];
            int nx = (it.batch_npred >> 20) & 0xF;
            if ((it.batch_npred >> 25) & 1) continue;
            bool known = false;
            for (int q = 0; q < nx; ++q) known |= it.xpred[q] == last;
            if (known) continue;
            if (nx < kMaxPreds) { it.xpred[nx++] = (unsigned short)last; it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (nx << 20); }
            else it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (1 << 25);
        }
    }
    for (int cl = 0; cl < nclusters; ++cl) {
        ClusterDesc d;
        d.body_begin = (int)plan.cluster_bodies.size();
        d.slot_count = ((int)cl_bodies[cl].size() + 15) / 16 * 16;
        std::vector<int32_t> slots(d.slot_count, -1);
        for (size_t i = 0; i < cl_bodies[cl].size(); ++i) slots[rotated_slot((int)i)] = cl_bodies[cl][i];
        plan.cluster_bodies.insert(plan.cluster_bodies.end(), slots.begin(), slots.end());
        d.item_begin = (int)plan.items.size();
        d.item_count = (int)cl_items[cl].size();
        d.batch_item_offset = (int)plan.batch_item_begin.size();
        // items were appended in type-batch order == batch order
        int k = 0;
        for (int b = 0; b <= c->batch_count; ++b) {
            while (k < d.item_count && (cl_items[cl][k].batch_npred & 0xFFFF) < b) ++k;
            plan.batch_item_begin.push_back(d.item_begin + k);
        }
        plan.items.insert(plan.items.end(), cl_items[cl].begin(), cl_items[cl].end());
        plan.clusters.push_back(d);
        plan.max_slots = std::max(plan.max_slots, d.slot_count);
        plan.max_items = std::max(plan.max_items, d.item_count);
    }
    plan.enabled = nclusters > 0 && cluster_lds_bytes(plan.max_slots, plan.max_items) <= kLdsBudgetBytes;
}

extern "C" {

const char* bepuhip_last_error(void) { return g_last_error.c_str(); }

int32_t bepuhip_type_info(int32_t type_id, int32_t* bodies, int32_t* prestep_floats, int32_t* impulse_floats) {
    TypeInfoH t;
    if (!type_info(type_id, t)) return fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(type_id));
    if (bodies) *bodies = t.bodies;
    if (prestep_floats) *prestep_floats = t.prestep;
    if (impulse_floats) *impulse_floats = t.impulse;
    return BEPUHIP_OK;
}

int32_t bepuhip_create(const bepuhip_config* config, bepuhip_ctx** out_ctx) {
    if (!config || !out_ctx) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (config->bundle_width != 4 && config->bundle_width != 8 && config->bundle_width != 16)
        return fail(BEPUHIP_E_INVALID_ARGUMENT, "bundle_width must be 4, 8 or 16");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(BEPUHIP_E_DEVICE, "no HIP device available (the bepuhip product path has no CPU fallback)");
    if (config->device_ordinal < 0 || config->device_ordinal >= n) return fail(BEPUHIP_E_INVALID_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(config->device_ordinal));
    bepuhip_ctx* c = new bepuhip_ctx();
    c->device = config->device_ordinal;
    c->W = config->bundle_width;
    c->flags = config->flags;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev_start));
    HIP_TRY(hipEventCreate(&c->ev_stop));
    HIP_TRY(hipHostMalloc((void**)&c->d_status, 256, hipHostMallocMapped | hipHostMallocCoherent));  // host-visible while a kernel runs
    memset(c->d_status, 0, 256);
    *out_ctx = c;
    return BEPUHIP_OK;
}

int32_t bepuhip_destroy(bepuhip_ctx* c) {
    if (!c) return BEPUHIP_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    free_constraints(c);
    if (c->d_bodies) hipFree(c->d_bodies);
    if (c->d_bodies0) hipFree(c->d_bodies0);
    if (c->d_flags) hipFree(c->d_flags);
    if (c->d_kin) hipFree(c->d_kin);
    if (c->d_status) hipHostFree(c->d_status);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->d_boundary) hipFree(c->d_boundary);
    if (c->d_boundary_snapshot) hipFree(c->d_boundary_snapshot);
    if (c->d_boundary_buf) hipFree(c->d_boundary_buf);
    hipEventDestroy(c->ev_start);
    hipEventDestroy(c->ev_stop);
    hipStreamDestroy(c->stream);
    delete c;
    return BEPUHIP_OK;
}

static int32_t rebuild_flags(bepuhip_ctx* c) {
    if (!c->d_flags || c->body_count == 0) return BEPUHIP_OK;
    HIP_TRY(hipMemsetAsync(c->d_flags, 0, (size_t)c->body_count * 4, c->stream));
    if (c->built) {
        for (auto& tb : c->tbs) {
            if (tb.count == 0) continue;
            int blocks = (tb.count + 255) / 256;
            hipLaunchKernelGGL(mark_constrained_kernel, dim3(blocks), dim3(256), 0, c->stream, (const int*)(c->d_slab + tb.refs_off), tb.count, tb.stride, tb.info.bodies, c->d_flags);
        }
    }
    if (c->built && c->clusters_enabled && c->clustered_dynamic_count > 0) {
        hipLaunchKernelGGL(mark_indices_kernel, dim3((c->clustered_dynamic_count + 255) / 256), dim3(256), 0, c->stream, (const int*)c->d_clustered_dynamic,
                           c->clustered_dynamic_count, c->d_flags, (unsigned)kFlagClustered);
    }
    if (c->boundary_count > 0) {  // held by several ranks: always integrated inside the solver, whatever this rank's share of its constraints
        hipLaunchKernelGGL(mark_indices_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, (const int*)c->d_boundary, c->boundary_count, c->d_flags,
                           (unsigned)(kFlagConstrained | kFlagDynamicConstrained));
    }
    if (c->kin_count > 0) {
        hipLaunchKernelGGL(mark_indices_kernel, dim3((c->kin_count + 255) / 256), dim3(256), 0, c->stream, (const int*)c->d_kin, c->kin_count, c->d_flags, (unsigned)(kFlagConstrainedKinematic | kFlagConstrained));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}

int32_t bepuhip_set_bodies(bepuhip_ctx* c, const void* aos, int32_t count) {
    if (!c || (!aos && count > 0) || count < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad bodies argument");
    HIP_TRY(hipSetDevice(c->device));
    if (count > c->body_capacity) {
        if (c->d_bodies) hipFree(c->d_bodies);
        if (c->d_bodies0) hipFree(c->d_bodies0);
        if (c->d_flags) hipFree(c->d_flags);
        c->d_bodies = c->d_bodies0 = nullptr; c->d_flags = nullptr;
        HIP_TRY(hipMalloc((void**)&c->d_bodies, (size_t)count * 128));
        HIP_TRY(hipMalloc((void**)&c->d_bodies0, (size_t)count * 128));
        HIP_TRY(hipMalloc((void**)&c->d_flags, (size_t)count * 4));
        c->body_capacity = count;
    }
    const bool count_changed = count != c->body_count;
    c->body_count = count;
    if (count > 0) {
        HIP_TRY(hipMemcpyAsync(c->d_bodies, aos, (size_t)count * 128, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_bodies0, c->d_bodies, (size_t)count * 128, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (count_changed) return rebuild_flags(c);
    return BEPUHIP_OK;
}

int32_t bepuhip_begin_constraints(bepuhip_ctx* c, int32_t batch_count, int32_t fallback_batch_threshold) {
    if (!c || batch_count < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad batch count");
    if (batch_count > fallback_batch_threshold)
        return fail(BEPUHIP_E_UNSUPPORTED, "a sequential fallback batch exists (batch_count > FallbackBatchThreshold); use simulation.Solve");
    HIP_TRY(hipSetDevice(c->device));
    hipStreamSynchronize(c->stream);
    free_constraints(c);
    c->batch_count = batch_count;
    c->has_widened_types = false;
    c->building = true;
    return BEPUHIP_OK;
}

int32_t bepuhip_set_type_batch(bepuhip_ctx* c, int32_t batch_index, int32_t type_id, int32_t count, const int32_t* refs, const float* prestep, const float* accum) {
    if (!c || !c->building) return fail(BEPUHIP_E_STATE, "set_type_batch outside begin/end");
    if (batch_index < 0 || batch_index >= c->batch_count || count < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad batch index or count");
    if (!c->tbs.empty() && batch_index < c->tbs.back().batch) return fail(BEPUHIP_E_INVALID_ARGUMENT, "type batches must be supplied in batch order");
    if (count > 0 && (!refs || !prestep || !accum)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null buffer");
    HostTypeBatch tb;
    if (!type_info(type_id, tb.info)) return fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(type_id));
    tb.batch = batch_index; tb.type_id = type_id; tb.count = count;
    tb.stride = ((count + 63) / 64) * 64;
    const int W = c->W;
    const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse;
    tb.refs_soa.assign((size_t)nb * tb.stride, -1);
    tb.prestep_soa.assign((size_t)pf * tb.stride, 0.0f);
    tb.accum_soa.assign((size_t)imf * tb.stride, 0.0f);
    for (int i = 0; i < count; ++i) {  // AOSOA -> SoA (BundleIndexing.cs:50-60, TypeProcessor.cs:269-279)
        const size_t bundle = (size_t)(i / W), lane = (size_t)(i % W);
        for (int k = 0; k < nb; ++k) {
            int32_t r = refs[bundle * nb * W + (size_t)k * W + lane];
            if (r < 0) return fail(BEPUHIP_E_UNSUPPORTED, "empty (-1) body reference inside a type batch: sequential fallback layout is not supported");
            tb.refs_soa[(size_t)k * tb.stride + i] = r;
        }
        for (int f = 0; f < pf; ++f) tb.prestep_soa[(size_t)f * tb.stride + i] = prestep[bundle * pf * W + (size_t)f * W + lane];
        for (int f = 0; f < imf; ++f) tb.accum_soa[(size_t)f * tb.stride + i] = accum[bundle * imf * W + (size_t)f * W + lane];
    }
    c->has_widened_types = c->has_widened_types || is_widened_type(type_id);
    c->tbs.push_back(std::move(tb));
    return BEPUHIP_OK;
}

int32_t bepuhip_end_constraints(bepuhip_ctx* c) {
    if (!c || !c->building) return fail(BEPUHIP_E_STATE, "end_constraints without begin");
    HIP_TRY(hipSetDevice(c->device));
    c->building = false;
    size_t words = 0;
    c->total_constraints = 0;
    for (auto& tb : c->tbs) c->total_constraints += tb.count;
    // Bodies the reference re-transforms in substep 0 of the conserving angular modes (see momentum_requirk_kernel). Bundles are W consecutive
    // constraints in the HOST's order, so this runs before the island schedule permutes the type batches.
    {
        int universe = 0;
        for (auto& tb : c->tbs)
            for (int32_t r : tb.refs_soa)
                if (r >= 0) universe = std::max(universe, (r & kRefMask) + 1);
        std::vector<int32_t> first_batch(universe, INT32_MAX);
        for (auto& tb : c->tbs)
            for (int k = 0; k < tb.info.bodies; ++k)
                for (int i = 0; i < tb.count; ++i) {
                    const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r < kDynamicLimit) first_batch[r] = std::min(first_batch[r], tb.batch);
                }
        std::vector<std::vector<int32_t>> lists(c->batch_count);
        const int W = c->W;
        for (auto& tb : c->tbs) {
            if (tb.batch == 0) continue;  // batch 0 always integrates (Solver_Solve.cs:188-194): no conditional bundles
            for (int k = 0; k < tb.info.bodies; ++k)
                for (int b0 = 0; b0 < tb.count; b0 += W) {
                    const int b1 = std::min(tb.count, b0 + W);
                    bool any = false;
                    for (int i = b0; i < b1; ++i) {
                        const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                        any |= (uint32_t)r < kDynamicLimit && first_batch[r] == tb.batch;
                    }
                    if (!any) continue;
                    for (int i = b0; i < b1; ++i) {
                        const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                        if ((uint32_t)r < kDynamicLimit && first_batch[r] < tb.batch) lists[tb.batch].push_back(r);
                    }
                }
        }
        std::vector<int32_t> flat;
        c->requirk_begin.assign(c->batch_count + 1, 0);
        for (int b = 0; b < c->batch_count; ++b) { c->requirk_begin[b] = (int)flat.size(); flat.insert(flat.end(), lists[b].begin(), lists[b].end()); }
        c->requirk_begin[c->batch_count] = (int)flat.size();
        if (!flat.empty()) {
            HIP_TRY(hipMalloc((void**)&c->d_requirk, flat.size() * 4));
            HIP_TRY(hipMemcpy(c->d_requirk, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
        }
    }
    ClusterPlan plan;
    plan_clusters(c, plan);
    for (auto& tb : c->tbs) {
        tb.refs_off = words; words += tb.refs_soa.size();
        tb.prestep_off = words; words += tb.prestep_soa.size();
        tb.accum_off = words; words += tb.accum_soa.size();
        tb.lrefs_off = words; words += tb.lrefs_soa.size();
    }
    c->slab_words = words;
    if (words > 0) {
        HIP_TRY(hipMalloc((void**)&c->d_slab, words * 4));
        HIP_TRY(hipMalloc((void**)&c->d_slab0, words * 4));
        std::vector<uint32_t> host(words);
        for (auto& tb : c->tbs) {
            if (!tb.refs_soa.empty()) memcpy(&host[tb.refs_off], tb.refs_soa.data(), tb.refs_soa.size() * 4);
            if (!tb.prestep_soa.empty()) memcpy(&host[tb.prestep_off], tb.prestep_soa.data(), tb.prestep_soa.size() * 4);
            if (!tb.accum_soa.empty()) memcpy(&host[tb.accum_off], tb.accum_soa.data(), tb.accum_soa.size() * 4);
            if (!tb.lrefs_soa.empty()) memcpy(&host[tb.lrefs_off], tb.lrefs_soa.data(), tb.lrefs_soa.size() * 4);
            std::vector<int32_t>().swap(tb.lrefs_soa);
            std::vector<int32_t>().swap(tb.refs_soa);
            std::vector<float>().swap(tb.prestep_soa);
            std::vector<float>().swap(tb.accum_soa);
        }
        HIP_TRY(hipMemcpy(c->d_slab, host.data(), words * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_slab0, c->d_slab, words * 4, hipMemcpyDeviceToDevice));
    }
    // Descriptors: per batch grid layout.
    c->batch_begin.assign(c->batch_count + 1, 0);
    c->batch_blocks.assign(c->batch_count, 0);
    std::vector<DevTypeBatch> descs(c->tbs.size()), inc;
    {
        size_t t = 0;
        for (int b = 0; b < c->batch_count; ++b) {
            c->batch_begin[b] = (int)t;
            int blocks = 0;
            while (t < c->tbs.size() && c->tbs[t].batch == b) {
                auto& tb = c->tbs[t];
                DevTypeBatch d;
                d.type_id = tb.type_id; d.count = tb.count; d.stride = tb.stride; d.block_begin = blocks;
                d.refs = (int*)(c->d_slab + tb.refs_off);
                d.prestep = (float*)(c->d_slab + tb.prestep_off);
                d.accum = (float*)(c->d_slab + tb.accum_off);
                descs[t] = d;
                blocks += (tb.count + kBlock - 1) / kBlock;
                ++t;
            }
            c->batch_blocks[b] = blocks;
        }
        c->batch_begin[c->batch_count] = (int)t;
    }
    c->inc_blocks = 0;
    for (size_t t = 0; t < c->tbs.size(); ++t) {
        if (!c->tbs[t].info.incremental || c->tbs[t].count == 0) continue;
        DevTypeBatch d = descs[t];
        d.block_begin = c->inc_blocks;
        c->inc_blocks += (c->tbs[t].count + kBlock - 1) / kBlock;
        inc.push_back(d);
    }
    c->inc_tb_count = (int)inc.size();
    if (!descs.empty()) {
        HIP_TRY(hipMalloc((void**)&c->d_tbs, descs.size() * sizeof(DevTypeBatch)));
        HIP_TRY(hipMemcpy(c->d_tbs, descs.data(), descs.size() * sizeof(DevTypeBatch), hipMemcpyHostToDevice));
    }
    if (!inc.empty()) {
        HIP_TRY(hipMalloc((void**)&c->d_inc_tbs, inc.size() * sizeof(DevTypeBatch)));
        HIP_TRY(hipMemcpy(c->d_inc_tbs, inc.data(), inc.size() * sizeof(DevTypeBatch), hipMemcpyHostToDevice));
    }
    // cluster path tables
    auto upload_ints = [&](const void* src, size_t bytes, void** dst) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        hipError_t e = hipMalloc(dst, bytes);
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    c->kinlist_count = (int)plan.kinlist.size();
    HIP_TRY(upload_ints(plan.kinlist.data(), plan.kinlist.size() * 4, (void**)&c->d_kinlist));
    c->clusters_enabled = plan.enabled;
    if (plan.enabled) {
        c->cluster_count = (int)plan.clusters.size();
        c->cluster_max_slots = plan.max_slots;
        HIP_TRY(hipMalloc((void**)&c->d_cycles, plan.clusters.size() * 8));
        HIP_TRY(hipMemset(c->d_cycles, 0, plan.clusters.size() * 8));
        c->first_cluster = plan.clusters[0];
        c->cluster_max_items = plan.max_items;
        for (auto& it : plan.items) {  // resolve the items' slab offsets now that the slab layout exists
            const HostTypeBatch& tb = c->tbs[it.tb];
            it.lrefs_off = (unsigned)tb.lrefs_off;
            it.prestep_off = (unsigned)tb.prestep_off;
            it.accum_off = (unsigned)tb.accum_off;
        }
        c->clustered_dynamic_count = (int)plan.clustered_dynamic.size();
        HIP_TRY(upload_ints(plan.clusters.data(), plan.clusters.size() * sizeof(ClusterDesc), (void**)&c->d_clusters));
        HIP_TRY(upload_ints(plan.items.data(), plan.items.size() * sizeof(ClusterItem), (void**)&c->d_items));
        HIP_TRY(upload_ints(plan.batch_item_begin.data(), plan.batch_item_begin.size() * 4, (void**)&c->d_batch_item_begin));
        HIP_TRY(upload_ints(plan.cluster_bodies.data(), plan.cluster_bodies.size() * 4, (void**)&c->d_cluster_bodies));
        HIP_TRY(upload_ints(plan.clustered_dynamic.data(), plan.clustered_dynamic.size() * 4, (void**)&c->d_clustered_dynamic));
#define X(T, TR, W) (const void*)cluster_kernel<T, TR, W>,
        for (const void* fn : {BEPU_CLUSTER_VARIANTS(X)})
#undef X
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
    }
    c->built = true;
    return rebuild_flags(c);
}

int32_t bepuhip_set_constrained_kinematics(bepuhip_ctx* c, const int32_t* indices, int32_t count) {
    if (!c || count < 0 || (count > 0 && !indices)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad kinematic list");
    HIP_TRY(hipSetDevice(c->device));
    if (c->d_kin) { hipFree(c->d_kin); c->d_kin = nullptr; }
    c->kin_count = count;
    c->kin_indices.assign(indices, indices + count);
    for (int i = 0; i < count; ++i)
        if (indices[i] < 0 || indices[i] >= c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "kinematic body index out of range (call set_bodies first)");
    if (count > 0) {
        HIP_TRY(hipMalloc((void**)&c->d_kin, (size_t)count * 4));
        HIP_TRY(hipMemcpy(c->d_kin, indices, (size_t)count * 4, hipMemcpyHostToDevice));
    }
    return rebuild_flags(c);
}

int32_t bepuhip_sync(bepuhip_ctx* c);

struct Timed {
    bepuhip_ctx* c; int family; hipEvent_t a = nullptr, b = nullptr;
    Timed(bepuhip_ctx* c_, int f) : c(c_), family(f) {
        if (c->profiling) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, c->stream); }
    }
    ~Timed() {
        if (c->profiling) {
            hipEventRecord(b, c->stream); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            c->prof_ms[family] += ms; c->prof_launches[family] += 1;
            hipEventDestroy(a); hipEventDestroy(b);
        }
    }
};

static StepParams make_params(const bepuhip_integrator* in, float dt_for_callbacks, float dt, float inv_dt) {
    // DemoPoseIntegratorCallbacks.PrepareForIntegration (Demos/DemoCallbacks.cs:79-86)
    StepParams sp;
    float l = 1 - in->linear_damping, a = 1 - in->angular_damping;
    l = l < 0 ? 0 : (l > 1 ? 1 : l);
    a = a < 0 ? 0 : (a > 1 ? 1 : a);
    sp.lin_damp = powf(l, dt_for_callbacks);
    sp.ang_damp = powf(a, dt_for_callbacks);
    sp.gx = in->gravity[0] * dt_for_callbacks; sp.gy = in->gravity[1] * dt_for_callbacks; sp.gz = in->gravity[2] * dt_for_callbacks;
    sp.dt = dt; sp.inv_dt = inv_dt;
    sp.angular_mode = in->angular_integration_mode;
    return sp;
}

static void enqueue_requirk(bepuhip_ctx* c, int substep, int batch, const StepParams& sp) {
    if (substep != 0 || sp.angular_mode == 0 || c->requirk_begin.empty()) return;
    const int n = c->requirk_begin[batch + 1] - c->requirk_begin[batch];
    if (n > 0)
        hipLaunchKernelGGL(momentum_requirk_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_bodies, (const int*)(c->d_requirk + c->requirk_begin[batch]), n, sp);
}

// Enqueue every kernel of one Simulation.Solve on the context's stream (Solver_Solve.cs:1415-1479 + PoseIntegrator.cs:707-726).
static void enqueue_solve(bepuhip_ctx* c, float dt, int substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    const float substep_dt = dt / substeps;          // Solver_Solve.cs:1417
    const float inv_dt = 1.0f / substep_dt;          // :1421
    const StepParams sp = make_params(in, substep_dt, substep_dt, inv_dt);
    const int body_blocks = (c->body_count + 255) / 256;
    const size_t lds_bytes = cluster_lds_bytes(c->cluster_max_slots, c->cluster_max_items);
    // The island schedule implements AngularIntegrationMode.Nonconserving; the conserving modes run the launch-per-batch schedule.
    const bool use_clusters = c->clusters_enabled && substeps <= kMaxClusterSubsteps && lds_bytes <= kLdsBudgetBytes && in->angular_integration_mode == 0;
    const int skip_clustered = use_clusters ? 1 : 0;
    if (use_clusters) {
        // Every constraint belongs to an island small enough for one workgroup: the whole substep loop runs in ONE launch.
        ClusterParams cp;
        cp.substeps = substeps; cp.batch_count = c->batch_count; cp.integrate_velocity_for_kinematics = in->integrate_velocity_for_kinematics;
        for (int s = 0; s < kMaxClusterSubsteps; ++s) cp.iters[s] = s < substeps ? iterations[s] : 0;
        cp.sp = sp;
        {
            Timed t(c, 5);
            // Waves per cluster: 16 by default (four per SIMD; 12 is as fast when memory latency is low, 8 is slower everywhere); BEPUHIP_CLUSTER_THREADS selects 512 / 768 / 1024.
            const int req = env_int("BEPUHIP_CLUSTER_THREADS", kClusterThreads);
            const int threads = req >= 1024 ? 1024 : req >= 768 ? 768 : std::max(64, std::min(512, req / 64 * 64));
            void* args[] = {(void*)&c->d_clusters, (void*)&c->d_items, (void*)&c->d_batch_item_begin, (void*)&c->d_cluster_bodies, (void*)&c->d_bodies, (void*)&c->d_slab,
                            (void*)&cp, (void*)&c->cluster_max_slots, (void*)&c->cluster_max_items, (void*)&c->d_trace, (void*)&c->d_status, (void*)&c->d_cycles};
            const bool tr = c->d_trace != nullptr;
            const void* fn = cluster_kernel_variant(threads, tr, c->has_widened_types);
            hipLaunchKernel(fn, dim3(c->cluster_count), dim3(threads), args, lds_bytes, c->stream);
        }
        if (c->kinlist_count > 0) {
            Timed t(c, 1);
            hipLaunchKernelGGL(kinematic_substeps_kernel, dim3((c->kinlist_count + 63) / 64), dim3(64), 0, c->stream, c->d_bodies, (const int*)c->d_kinlist, c->kinlist_count, substeps,
                               in->integrate_velocity_for_kinematics, sp);
        }
    }
    for (int s = 0; s < substeps && !use_clusters; ++s) {
        if (s > 0 && c->inc_blocks > 0) {             // :1427-1439 (all batches in one grid: it reads velocities and writes only prestep depths)
            Timed t(c, 0);
            hipLaunchKernelGGL(batch_kernel<kStageIncremental>, dim3(c->inc_blocks), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_inc_tbs, 0, c->inc_tb_count, c->d_bodies, substep_dt, inv_dt);
        }
        if (body_blocks > 0) {                        // :1440-1445 + the integration half of GatherAndIntegrate
            Timed t(c, 1);
            hipLaunchKernelGGL(substep_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, s > 0 ? 1 : 0,
                               in->integrate_velocity_for_kinematics, skip_clustered, sp);
        }
        for (int b = 0; b < c->batch_count; ++b) {    // :1447-1463
            if (c->batch_blocks[b] == 0) continue;
            enqueue_requirk(c, s, b, sp);
            Timed t(c, 2);
            hipLaunchKernelGGL(batch_kernel<kStageWarmStart>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                               c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
        }
        for (int it = 0; it < iterations[s]; ++it) {  // :1464-1476
            for (int b = 0; b < c->batch_count; ++b) {
                if (c->batch_blocks[b] == 0) continue;
                Timed t(c, 3);
                hipLaunchKernelGGL(batch_kernel<kStageSolve>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                                   c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
            }
        }
    }
    if (body_blocks > 0) {                            // PoseIntegrator.cs:707-726
        const float vdt = in->allow_substeps_for_unconstrained ? substep_dt : dt;
        const StepParams fsp = make_params(in, vdt, vdt, 1.0f / vdt);
        Timed t(c, 4);
        hipLaunchKernelGGL(final_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, dt, substep_dt, substeps,
                           in->allow_substeps_for_unconstrained, in->integrate_velocity_for_kinematics, skip_clustered, fsp);
    }
}

static int32_t validate_solve(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    if (!c || !in || !iterations) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (!(dt > 0)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Timestep duration must be positive.");                  // Simulation.cs:318-319
    if (substeps < 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Substep count must be positive.");                    // SolveDescription.cs:42-47
    for (int s = 0; s < substeps; ++s)
        if (iterations[s] < 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Velocity iteration count must be positive.");
    if (in->angular_integration_mode < 0 || in->angular_integration_mode > 2) return fail(BEPUHIP_E_INVALID_ARGUMENT, "unknown AngularIntegrationMode");
    if (c->building) return fail(BEPUHIP_E_STATE, "solve between begin_constraints and end_constraints");
    return BEPUHIP_OK;
}

int32_t bepuhip_solve_async(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    int32_t st = validate_solve(c, dt, substeps, iterations, in);
    if (st != BEPUHIP_OK) return st;
    HIP_TRY(hipSetDevice(c->device));
    int64_t iters = 0;
    for (int s = 0; s < substeps; ++s) iters += c->total_constraints * (int64_t)(1 + iterations[s]);
    c->last_constraint_iterations = iters;
    if (c->profiling) { for (int i = 0; i < 6; ++i) { c->prof_ms[i] = 0; c->prof_launches[i] = 0; } }
    HIP_TRY(hipEventRecord(c->ev_start, c->stream));
    const bool use_graph = !(c->flags & BEPUHIP_FLAG_NO_GRAPH) && !c->profiling;
    if (use_graph) {
        GraphKey key;
        key.iterations.assign(iterations, iterations + substeps);
        key.dt = dt; key.integ = *in;
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            enqueue_solve(c, dt, substeps, iterations, in);
            HIP_TRY(hipStreamEndCapture(c->stream, &graph));
            hipGraphExec_t exec = nullptr;
            HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
            it = c->graphs.emplace(key, exec).first;
            // the event recorded before capture began is still first in stream order
        }
        HIP_TRY(hipGraphLaunch(it->second, c->stream));
    } else {
        enqueue_solve(c, dt, substeps, iterations, in);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
    return BEPUHIP_OK;
}

int32_t bepuhip_set_boundary_bodies(bepuhip_ctx* c, const int32_t* indices, int32_t count) {
    if (!c || count < 0 || (count > 0 && !indices)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad boundary list");
    HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < count; ++i)
        if (indices[i] < 0 || indices[i] >= c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "boundary body index out of range (call set_bodies first)");
    if (c->d_boundary) { hipFree(c->d_boundary); hipFree(c->d_boundary_snapshot); hipFree(c->d_boundary_buf); c->d_boundary = nullptr; c->d_boundary_snapshot = nullptr; c->d_boundary_buf = nullptr; }
    c->boundary_count = count;
    if (count > 0) {
        HIP_TRY(hipMalloc((void**)&c->d_boundary, (size_t)count * 4));
        HIP_TRY(hipMalloc((void**)&c->d_boundary_snapshot, (size_t)count * 32));
        HIP_TRY(hipMalloc((void**)&c->d_boundary_buf, (size_t)count * 24));
        HIP_TRY(hipMemcpy(c->d_boundary, indices, (size_t)count * 4, hipMemcpyHostToDevice));
    }
    return rebuild_flags(c);
}

int32_t bepuhip_boundary_deltas(bepuhip_ctx* c, float* out, int32_t out_is_device) {
    if (!c || (c->boundary_count > 0 && !out)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (c->boundary_count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    float* dst = out_is_device ? out : c->d_boundary_buf;
    hipLaunchKernelGGL(boundary_deltas_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, (const float4*)c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                       (const float4*)c->d_boundary_snapshot, dst);
    HIP_TRY(hipGetLastError());
    if (!out_is_device) HIP_TRY(hipMemcpyAsync(out, dst, (size_t)c->boundary_count * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the caller hands the buffer to a collective on another stream / the host next
    return BEPUHIP_OK;
}

int32_t bepuhip_boundary_apply(bepuhip_ctx* c, const float* sums, int32_t in_is_device) {
    if (!c || (c->boundary_count > 0 && !sums)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (c->boundary_count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const float* src = sums;
    if (!in_is_device) {
        HIP_TRY(hipMemcpyAsync(c->d_boundary_buf, sums, (size_t)c->boundary_count * 24, hipMemcpyHostToDevice, c->stream));
        src = c->d_boundary_buf;
    }
    hipLaunchKernelGGL(boundary_apply_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                       c->d_boundary_snapshot, src);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}

// Simulation.Solve with an exchange point after every pass: the launch-per-batch schedule, eager, host-synchronised at each call-back.
int32_t bepuhip_solve_exchanged(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in, bepuhip_exchange_fn fn, void* user) {
    int32_t st = validate_solve(c, dt, substeps, iterations, in);
    if (st != BEPUHIP_OK) return st;
    if (!fn) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null exchange call-back");
    if (c->clusters_enabled) return fail(BEPUHIP_E_STATE, "solve_exchanged needs a context created with BEPUHIP_FLAG_NO_CLUSTERS (a split scene is one island per rank anyway)");
    HIP_TRY(hipSetDevice(c->device));
    int64_t iters = 0;
    for (int s = 0; s < substeps; ++s) iters += c->total_constraints * (int64_t)(1 + iterations[s]);
    c->last_constraint_iterations = iters;
    HIP_TRY(hipEventRecord(c->ev_start, c->stream));
    const float substep_dt = dt / substeps, inv_dt = 1.0f / substep_dt;
    const StepParams sp = make_params(in, substep_dt, substep_dt, inv_dt);
    const int body_blocks = (c->body_count + 255) / 256;
    auto exchange = [&](int s, int pass) -> int32_t {
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int32_t r = fn(user, s, pass);
        if (r != 0) return fail(BEPUHIP_E_STATE, "exchange call-back failed with status " + std::to_string(r));
        return BEPUHIP_OK;
    };
    for (int s = 0; s < substeps; ++s) {
        if (s > 0 && c->inc_blocks > 0)
            hipLaunchKernelGGL(batch_kernel<kStageIncremental>, dim3(c->inc_blocks), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_inc_tbs, 0, c->inc_tb_count, c->d_bodies, substep_dt, inv_dt);
        if (body_blocks > 0)
            hipLaunchKernelGGL(substep_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, s > 0 ? 1 : 0,
                               in->integrate_velocity_for_kinematics, 0, sp);
        if (c->boundary_count > 0)  // deltas of this substep are relative to the integrated velocities (identical on every holder)
            hipLaunchKernelGGL(boundary_snapshot_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, (const float4*)c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                               c->d_boundary_snapshot);
        for (int b = 0; b < c->batch_count; ++b)
            if (c->batch_blocks[b] > 0) {
                enqueue_requirk(c, s, b, sp);
                hipLaunchKernelGGL(batch_kernel<kStageWarmStart>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                                   c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
            }
        if ((st = exchange(s, 0)) != BEPUHIP_OK) return st;
        for (int it = 0; it < iterations[s]; ++it) {
            for (int b = 0; b < c->batch_count; ++b)
                if (c->batch_blocks[b] > 0)
                    hipLaunchKernelGGL(batch_kernel<kStageSolve>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                                       c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
            if ((st = exchange(s, 1 + it)) != BEPUHIP_OK) return st;
        }
    }
    if (body_blocks > 0) {
        const float vdt = in->allow_substeps_for_unconstrained ? substep_dt : dt;
        const StepParams fsp = make_params(in, vdt, vdt, 1.0f / vdt);
        hipLaunchKernelGGL(final_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, dt, substep_dt, substeps,
                           in->allow_substeps_for_unconstrained, in->integrate_velocity_for_kinematics, 0, fsp);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
    return bepuhip_sync(c);
}

int32_t bepuhip_sync(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) == hipSuccess) c->last_ms = ms;
    if (c->clusters_enabled) {
        unsigned st[8];
        memcpy(st, c->d_status, sizeof(st));
        if (st[0] != 0) {
            memset(c->d_status, 0, 256);
            char msg[256];
            snprintf(msg, sizeof(msg), "cluster schedule stalled: cluster %u kind %u item %u waiting on %u, wanted %u saw %u, claim counter %u", st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
            return fail(BEPUHIP_E_DEVICE, msg);
        }
    }
    return BEPUHIP_OK;
}

int32_t bepuhip_solve(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    int32_t st = bepuhip_solve_async(c, dt, substeps, iterations, in);
    if (st != BEPUHIP_OK) return st;
    return bepuhip_sync(c);
}

int32_t bepuhip_get_bodies(bepuhip_ctx* c, void* out, int32_t count) {
    if (!c || !out || count < 0 || count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad get_bodies argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_bodies, (size_t)count * 128, hipMemcpyDeviceToHost));
    return BEPUHIP_OK;
}

static HostTypeBatch* find_tb(bepuhip_ctx* c, int batch, int type_id) {
    for (auto& tb : c->tbs) if (tb.batch == batch && tb.type_id == type_id) return &tb;
    return nullptr;
}
static int32_t download_aosoa(bepuhip_ctx* c, HostTypeBatch* tb, size_t off, int fields, float* out) {
    if (tb->count == 0) return BEPUHIP_OK;
    std::vector<float> soa((size_t)fields * tb->stride);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(soa.data(), c->d_slab + off, soa.size() * 4, hipMemcpyDeviceToHost));
    const int W = c->W;
    for (int i = 0; i < tb->count; ++i) {
        const size_t bundle = (size_t)(i / W), lane = (size_t)(i % W);
        const int d = tb->perm.empty() ? i : tb->perm_inverse(i);
        for (int f = 0; f < fields; ++f) out[bundle * fields * W + (size_t)f * W + lane] = soa[(size_t)f * tb->stride + d];
    }
    return BEPUHIP_OK;
}
int32_t bepuhip_get_accumulated_impulses(bepuhip_ctx* c, int32_t batch, int32_t type_id, float* out) {
    if (!c || !out || !c->built) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad argument or no constraints");
    HIP_TRY(hipSetDevice(c->device));
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb) return fail(BEPUHIP_E_INVALID_ARGUMENT, "no such type batch");
    return download_aosoa(c, tb, tb->accum_off, tb->info.impulse, out);
}
int32_t bepuhip_get_prestep(bepuhip_ctx* c, int32_t batch, int32_t type_id, float* out) {
    if (!c || !out || !c->built) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad argument or no constraints");
    HIP_TRY(hipSetDevice(c->device));
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb) return fail(BEPUHIP_E_INVALID_ARGUMENT, "no such type batch");
    return download_aosoa(c, tb, tb->prestep_off, tb->info.prestep, out);
}

// ---- Device-resident incremental updates: ranged rewrites of what already lives in HBM, no re-plan, no full re-upload ----
static int32_t stage_reserve(bepuhip_ctx* c, size_t floats) {
    if (floats <= c->stage_floats) return BEPUHIP_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_stage) hipFree(c->d_stage);
    c->d_stage = nullptr; c->stage_floats = 0;
    HIP_TRY(hipMalloc((void**)&c->d_stage, floats * 4));
    c->stage_floats = floats;
    return BEPUHIP_OK;
}
static int32_t device_index_of(bepuhip_ctx* c, HostTypeBatch* tb, const int** out) {
    *out = nullptr;
    if (tb->perm.empty()) return BEPUHIP_OK;  // launch-per-batch layout: host order
    if (!tb->d_device_index) {
        tb->perm_inverse(0);
        HIP_TRY(hipMalloc((void**)&tb->d_device_index, tb->inv.size() * 4));
        HIP_TRY(hipMemcpy(tb->d_device_index, tb->inv.data(), tb->inv.size() * 4, hipMemcpyHostToDevice));
    }
    *out = tb->d_device_index;
    return BEPUHIP_OK;
}
static int32_t bundle_range(bepuhip_ctx* c, int batch, int type_id, int first_bundle, int bundle_count, const void* buffer, HostTypeBatch** tb_out, int* first, int* n) {
    if (!c || !c->built) return fail(BEPUHIP_E_STATE, "no constraints uploaded");
    if (first_bundle < 0 || bundle_count < 0 || (!buffer && bundle_count > 0)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad bundle range");
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb) return fail(BEPUHIP_E_INVALID_ARGUMENT, "no such type batch");
    const int bundles = (tb->count + c->W - 1) / c->W;
    if (first_bundle + bundle_count > bundles) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bundle range exceeds the type batch");
    *tb_out = tb;
    *first = first_bundle * c->W;
    *n = std::min(bundle_count * c->W, tb->count - *first);  // trailing lanes of the last bundle are empty (TypeProcessor.cs:287-298)
    return BEPUHIP_OK;
}
static int32_t update_rows(bepuhip_ctx* c, int batch, int type_id, int first_bundle, int bundle_count, const float* bundles, bool prestep) {
    HostTypeBatch* tb; int first, n;
    int32_t st = bundle_range(c, batch, type_id, first_bundle, bundle_count, bundles, &tb, &first, &n);
    if (st != BEPUHIP_OK || n <= 0) return st;
    HIP_TRY(hipSetDevice(c->device));
    const int fields = prestep ? tb->info.prestep : tb->info.impulse;
    const size_t floats = (size_t)bundle_count * fields * c->W;
    if ((st = stage_reserve(c, floats)) != BEPUHIP_OK) return st;
    const int* index;
    if ((st = device_index_of(c, tb, &index)) != BEPUHIP_OK) return st;
    HIP_TRY(hipMemcpyAsync(c->d_stage, bundles, floats * 4, hipMemcpyHostToDevice, c->stream));
    const size_t off = prestep ? tb->prestep_off : tb->accum_off;
    for (uint32_t* slab : {c->d_slab, c->d_slab0})  // the pristine snapshot follows, so that reset_state restores "what the set_*/update_* calls uploaded"
        if (slab) scatter_bundles_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(c->d_stage, (float*)(slab + off), index, first, n, fields, tb->stride, c->W);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));  // the caller's buffer and the staging buffer are free again on return
    return BEPUHIP_OK;
}
static int32_t read_rows(bepuhip_ctx* c, int batch, int type_id, int first_bundle, int bundle_count, float* bundles_out, bool prestep) {
    HostTypeBatch* tb; int first, n;
    int32_t st = bundle_range(c, batch, type_id, first_bundle, bundle_count, bundles_out, &tb, &first, &n);
    if (st != BEPUHIP_OK || n <= 0) return st;
    HIP_TRY(hipSetDevice(c->device));
    const int fields = prestep ? tb->info.prestep : tb->info.impulse;
    const size_t floats = (size_t)bundle_count * fields * c->W;
    if ((st = stage_reserve(c, floats)) != BEPUHIP_OK) return st;
    const int* index;
    if ((st = device_index_of(c, tb, &index)) != BEPUHIP_OK) return st;
    HIP_TRY(hipMemsetAsync(c->d_stage, 0, floats * 4, c->stream));  // empty trailing lanes read back as zero
    const size_t off = prestep ? tb->prestep_off : tb->accum_off;
    gather_bundles_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(c->d_stage, (const float*)(c->d_slab + off), index, first, n, fields, tb->stride, c->W);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(bundles_out, c->d_stage, floats * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}
int32_t bepuhip_update_prestep(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, const float* prestep_bundles) {
    return update_rows(c, batch, type_id, first_bundle, bundle_count, prestep_bundles, true);
}
int32_t bepuhip_update_accumulated_impulses(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, const float* impulse_bundles) {
    return update_rows(c, batch, type_id, first_bundle, bundle_count, impulse_bundles, false);
}
int32_t bepuhip_get_prestep_range(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, float* prestep_bundles_out) {
    return read_rows(c, batch, type_id, first_bundle, bundle_count, prestep_bundles_out, true);
}
int32_t bepuhip_get_accumulated_impulses_range(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, float* impulse_bundles_out) {
    return read_rows(c, batch, type_id, first_bundle, bundle_count, impulse_bundles_out, false);
}
int32_t bepuhip_update_bodies(bepuhip_ctx* c, const void* aos, int32_t first, int32_t count) {
    if (!c || first < 0 || count < 0 || (!aos && count > 0) || first + count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad body range");
    if (count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->d_bodies + (size_t)first * 8, aos, (size_t)count * 128, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_bodies0 + (size_t)first * 8, c->d_bodies + (size_t)first * 8, (size_t)count * 128, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}
int32_t bepuhip_get_bodies_range(bepuhip_ctx* c, void* aos_out, int32_t first, int32_t count) {
    if (!c || first < 0 || count < 0 || (!aos_out && count > 0) || first + count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad body range");
    if (count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(aos_out, c->d_bodies + (size_t)first * 8, (size_t)count * 128, hipMemcpyDeviceToHost));
    return BEPUHIP_OK;
}

int32_t bepuhip_get_constrained_flags(bepuhip_ctx* c, uint8_t* out, int32_t count) {
    if (!c || !out || count < 0 || count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<unsigned> f((size_t)count);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (count > 0) HIP_TRY(hipMemcpy(f.data(), c->d_flags, (size_t)count * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < count; ++i) out[i] = (uint8_t)(f[i] & 3u);
    return BEPUHIP_OK;
}

int32_t bepuhip_last_solve_ms(bepuhip_ctx* c, float* ms) {
    if (!c || !ms) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *ms = c->last_ms;
    return BEPUHIP_OK;
}
int32_t bepuhip_set_profiling(bepuhip_ctx* c, int32_t enabled) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    c->profiling = enabled != 0;
    return BEPUHIP_OK;
}
int32_t bepuhip_get_profile(bepuhip_ctx* c, int32_t family, float* ms, int32_t* launches) {
    if (!c || family < 0 || family > 5) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad family");
    if (ms) *ms = c->prof_ms[family];
    if (launches) *launches = c->prof_launches[family];
    return BEPUHIP_OK;
}
int32_t bepuhip_set_cluster_trace(bepuhip_ctx* c, int32_t enabled) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);  // captured launches bake the trace pointer in
    c->graphs.clear();
    if (c->d_trace) { hipFree(c->d_trace); c->d_trace = nullptr; c->trace_words = 0; }
    if (enabled && c->clusters_enabled) {
        const ClusterDesc first = c->first_cluster;
        c->trace_words = (size_t)first.item_count * 8 * (size_t)kMaxClusterSubsteps * 8;  // up to 128 passes of cluster 0
        HIP_TRY(hipMalloc((void**)&c->d_trace, c->trace_words * 8));
        HIP_TRY(hipMemset(c->d_trace, 0, c->trace_words * 8));
    }
    return BEPUHIP_OK;
}
int32_t bepuhip_get_cluster_trace(bepuhip_ctx* c, uint64_t* out, int64_t capacity_words, int32_t* items_out) {
    if (!c || !out || !items_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (!c->d_trace) return fail(BEPUHIP_E_STATE, "cluster trace is not enabled (or the scene does not use the cluster schedule)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t n = std::min<size_t>(c->trace_words, (size_t)std::max<int64_t>(capacity_words, 0));
    HIP_TRY(hipMemcpy(out, c->d_trace, n * 8, hipMemcpyDeviceToHost));
    *items_out = c->first_cluster.item_count;
    return BEPUHIP_OK;
}
int32_t bepuhip_get_cluster_cycles(bepuhip_ctx* c, uint64_t* out, int32_t capacity, int32_t* count_out) {
    if (!c || !count_out || (capacity > 0 && !out)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *count_out = c->clusters_enabled ? c->cluster_count : 0;
    if (!c->clusters_enabled || capacity <= 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_cycles, (size_t)std::min(capacity, c->cluster_count) * 8, hipMemcpyDeviceToHost));
    return BEPUHIP_OK;
}
int32_t bepuhip_debug_status(bepuhip_ctx* c, uint32_t* out16) {
    if (!c || !out16) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    memcpy(out16, c->d_status, 64);
    return BEPUHIP_OK;
}
int32_t bepuhip_last_constraint_iterations(bepuhip_ctx* c, int64_t* out) {
    if (!c || !out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *out = c->last_constraint_iterations;
    return BEPUHIP_OK;
}
int32_t bepuhip_get_stream(bepuhip_ctx* c, void** out) {
    if (!c || !out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *out = (void*)c->stream;
    return BEPUHIP_OK;
}
int32_t bepuhip_reset_state(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (c->body_count > 0) HIP_TRY(hipMemcpyAsync(c->d_bodies, c->d_bodies0, (size_t)c->body_count * 128, hipMemcpyDeviceToDevice, c->stream));
    if (c->slab_words > 0) HIP_TRY(hipMemcpyAsync(c->d_slab, c->d_slab0, c->slab_words * 4, hipMemcpyDeviceToDevice, c->stream));
    return BEPUHIP_OK;
}

}  // extern "C"
