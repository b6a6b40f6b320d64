// TEST INFRASTRUCTURE — CPU oracle for the bepuphysics2 solver + pose-integrator hot path.
// PARITY UNPINNED: the reference (C#/.NET 8) cannot be built or run in this environment and its own tests hold
// no golden vectors for this path; this is our restatement of the C#, following the reference's
// control flow (integration fused into the first-touching constraint's warm start, per-bundle modes) rather
// than the product's reorganised schedule, so that it also checks the product's equivalence argument.
// oracle/pin/ holds the kit that pins it on a machine with .NET (C# ReferenceDumper + exporter + bit-for-bit comparison).
//
// Restates, single-threaded (threads=1) exactly as the reference's dispatcher==null path, and optionally with the
// reference's work-block/barrier scheme (threads>1) for use as bench.py's cpu_baseline ("port"):
//   Simulation.Solve                              BepuPhysics/Simulation.cs:278-290
//   Solver.PrepareConstraintIntegrationResponsibilities  BepuPhysics/Solver_Solve.cs:951-1044,1072-1388
//   Solver<T>.Solve (dispatcher == null)          BepuPhysics/Solver_Solve.cs:1415-1479
//   TypeProcessor.GatherAndIntegrate & friends    BepuPhysics/Constraints/TypeProcessor.cs:1155-1397
//   Two/OneBodyTypeProcessor loops                BepuPhysics/Constraints/TwoBodyTypeProcessor.cs:168-241, OneBodyTypeProcessor.cs:82-146
//   Bodies.GatherState / Scatter*                 BepuPhysics/Bodies_GatherScatter.cs:43-139,267-753 (scalar fallbacks as spec)
//   PoseIntegrator kinematic + final passes       BepuPhysics/PoseIntegrator.cs:451-726
//   DemoPoseIntegratorCallbacks                   Demos/DemoCallbacks.cs:79-109
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "bepu_constraints.h"
#include "bepu_bounds.h"

using namespace bo;
using C1O = Contact<1, false>; using C2O = Contact<2, false>; using C3O = Contact<3, false>; using C4O = Contact<4, false>;
using C1T = Contact<1, true>; using C2T = Contact<2, true>; using C3T = Contact<3, true>; using C4T = Contact<4, true>;

extern "C" {
struct OracleTypeBatch {
    int32_t type_id;
    int32_t constraint_count;
    int32_t* body_refs;  // AOSOA: (i/W)*bodies*W + slot*W + (i%W)        BepuPhysics/Constraints/TypeProcessor.cs:269-279
    float* prestep;      // AOSOA: (i/W)*F*W + f*W + (i%W)                BepuUtilities/BundleIndexing.cs:50-60
    float* accumulated;  // AOSOA likewise
};
struct OracleParams {
    float dt;
    int32_t substep_count;
    const int32_t* velocity_iterations;  // per substep (SolveDescription.VelocityIterationScheduler resolved by caller)
    float gravity[3];
    float linear_damping;
    float angular_damping;
    int32_t allow_substeps_for_unconstrained;
    int32_t integrate_velocity_for_kinematics;
    int32_t threads;
    // Optional: called after every pass (pass 0 = warm start of the substep, k = its k-th velocity iteration), for the split-scene tests
    // (the CPU stand-in of bepuhip_solve_exchanged, include/bepuhip.h). Null = the reference's plain loop.
    int32_t (*exchange)(void* user, int32_t substep, int32_t pass);
    void* exchange_user;
    int32_t angular_integration_mode;  // AngularIntegrationMode (PoseIntegrator.cs:20-38): 0 Nonconserving, 1 ConserveMomentum, 2 ConserveMomentumWithGyroscopicTorque
    int32_t fallback_batch_threshold;  // SolveDescription.FallbackBatchThreshold (SolveDescription.cs:38); 0 = the default 64. Batch index == threshold is the sequential fallback batch.
    // IPoseIntegratorCallbacks.IntegrateVelocity: 0 = DemoPoseIntegratorCallbacks (Demos/DemoCallbacks.cs:100-109, the fields above), 1 = PerBodyGravityDemoCallbacks
    // (Demos/Demos/PerBodyGravityDemo.cs:57-88; body_gravity[i] = the value of the body at index i), 2 = PlanetaryGravityCallbacks (Demos/Demos/PlanetDemo.cs:36-47)
    int32_t velocity_model;
    float planet_center[3];
    float planet_gravity;
    const float* body_gravity;
};
struct OracleScene {
    float* bodies;  // AoS BodyDynamics, 32 floats/body (BepuPhysics/BodyProperties.cs:11-46,258-338)
    int32_t body_count;
    const int32_t* index_to_handle;  // body index -> handle
    const int32_t* handle_to_index;  // handle -> body index
    int32_t handle_capacity;         // max handle + 1
    int32_t batch_count;
    const int32_t* type_batch_counts;  // per batch
    OracleTypeBatch* type_batches;     // flattened in batch order
    const int32_t* constrained_kinematic_handles;
    int32_t constrained_kinematic_count;
    int32_t bundle_width;  // Vector<float>.Count of the host that laid out the AOSOA buffers (4/8/16)
};
}

namespace {

constexpr int kMaxW = 16;
constexpr uint32_t kDynamicLimit = 1u << 30;      // Bodies_GatherScatter.cs:107-118
constexpr int32_t kBodyReferenceMask = 0x3FFFFFFF;

struct Callbacks {  // Demos/DemoCallbacks.cs:79-109; PerBodyGravityDemo.cs:57-88; PlanetDemo.cs:36-47
    V3 gravityDt;
    float linearDampingDt, angularDampingDt;
    int mode = 0;  // AngularIntegrationMode
    int model = 0;
    V3 planetCenter;
    float planetGravityDt = 0;
    const float* bodyGravity = nullptr;
    void prepare(const OracleParams& p, float dt) {
        mode = p.angular_integration_mode;
        model = p.velocity_model;
        planetCenter = {p.planet_center[0], p.planet_center[1], p.planet_center[2]};
        planetGravityDt = dt * p.planet_gravity;  // PlanetDemo.cs:39
        bodyGravity = p.body_gravity;
        float l = 1 - p.linear_damping, a = 1 - p.angular_damping;
        l = l < 0 ? 0 : (l > 1 ? 1 : l);
        a = a < 0 ? 0 : (a > 1 ? 1 : a);
        linearDampingDt = powf(l, dt);
        angularDampingDt = powf(a, dt);
        gravityDt = {p.gravity[0] * dt, p.gravity[1] * dt, p.gravity[2] * dt};
    }
    // IntegrateVelocity(bodyIndices, position, ..., dt, ref velocity): one lane. bodyIndex < 0 = a lane the caller masked out (its result is discarded).
    void integrateVelocity(BodyVel& v, const V3& position, int bodyIndex, float dt) const {
        if (model == 0) {
            v.lin = scale(add(v.lin, gravityDt), linearDampingDt);
            v.ang = scale(v.ang, angularDampingDt);
        } else if (model == 1) {  // PerBodyGravityDemo.cs:87: velocity.Linear.Y += new Vector<float>(gravityValues) * dt
            const float g = bodyIndex >= 0 ? bodyGravity[bodyIndex] : 0.0f;
            v.lin.y = v.lin.y + g * dt;
        } else {  // PlanetDemo.cs:44-46
            const V3 offset = sub(position, planetCenter);
            const float distance = sqrtf(offset.x * offset.x + offset.y * offset.y + offset.z * offset.z);
            const V3 scaled = {offset.x * planetGravityDt, offset.y * planetGravityDt, offset.z * planetGravityDt};  // Vector<float> * Vector3Wide (Vector3Wide.cs:390-397)
            const float cube = distance * distance * distance;
            const float inverse = 1.0f / (1.0f > cube ? 1.0f : cube);                                                  // Vector.Max(One, d^3); "/" = multiply by One / scalar (:357-365)
            v.lin = sub(v.lin, V3{scaled.x * inverse, scaled.y * inverse, scaled.z * inverse});
        }
    }
};

struct BodyState {
    V3 pos; Q ori; BodyVel vel; Inertia inertia;
};

inline void gatherState(const float* bodies, int32_t ref, bool worldInertia, BodyState& s) {  // Bodies_GatherScatter.cs:267-478 (scalar spec :43-105)
    if (ref < 0) { memset(&s, 0, sizeof(s)); return; }
    const float* b = bodies + (size_t)(ref & kBodyReferenceMask) * 32;
    s.ori = {b[0], b[1], b[2], b[3]};
    s.pos = {b[4], b[5], b[6]};
    s.vel.lin = {b[8], b[9], b[10]};
    s.vel.ang = {b[12], b[13], b[14]};
    const float* in = b + (worldInertia ? 24 : 16);
    s.inertia.t = {in[0], in[1], in[2], in[3], in[4], in[5]};
    s.inertia.invMass = in[6];
}
inline void scatterVelocities(float* bodies, int32_t ref, const BodyVel& v) {  // :626-753
    if ((uint32_t)ref >= kDynamicLimit) return;
    float* b = bodies + (size_t)ref * 32;
    b[8] = v.lin.x; b[9] = v.lin.y; b[10] = v.lin.z;
    b[12] = v.ang.x; b[13] = v.ang.y; b[14] = v.ang.z;
}
inline void scatterPose(float* bodies, int32_t index, V3 pos, Q ori) {  // :484-549
    float* b = bodies + (size_t)index * 32;
    b[0] = ori.x; b[1] = ori.y; b[2] = ori.z; b[3] = ori.w;
    b[4] = pos.x; b[5] = pos.y; b[6] = pos.z;
}
inline void scatterInertia(float* bodies, int32_t index, const Inertia& in) {  // :553-622 (world slot @ float 24)
    float* b = bodies + (size_t)index * 32 + 24;
    b[0] = in.t.xx; b[1] = in.t.yx; b[2] = in.t.yy; b[3] = in.t.zx; b[4] = in.t.zy; b[5] = in.t.zz; b[6] = in.invMass;
}

enum BatchMode { kAlways = 0, kNever = 1, kConditional = 2 };

struct Ctx {
    OracleScene* scene;
    const OracleParams* params;
    Callbacks cb;
    int W;
    // integrationFlags[batch][typeBatch][slot] as 64-bit word arrays; coarse[batch][typeBatch]   (Solver_Solve.cs:1072-1388)
    std::vector<std::vector<std::vector<std::vector<uint64_t>>>> flags;
    std::vector<std::vector<char>> coarse;
    std::vector<uint64_t> mergedConstrained;  // by handle
    std::vector<int> batchStart;              // index of first type batch of each batch
    int fallbackThreshold = 64;               // SolveDescription.FallbackBatchThreshold: batch index of the sequential fallback batch, if one exists
};

// TypeProcessor.cs:1204-1248, one lane.
inline void integratePoseAndVelocity(const Callbacks& cb, const Inertia& localInertia, float dt, bool mask, BodyState& s, Inertia& outInertia, int bodyIndex) {
    V3 newPosition = add(s.pos, scale(s.vel.lin, dt));
    s.pos = sel3(mask, newPosition, s.pos);
    outInertia.invMass = localInertia.invMass;
    BodyVel previousVelocity = s.vel;
    if (cb.mode == 1) {  // ConserveMomentum :1224-1231
        Q previousOrientation = s.ori;
        Q newOrientation = integrateOrientation(s.ori, s.vel.ang, dt * 0.5f);
        s.ori = mask ? newOrientation : s.ori;
        outInertia.t = rotateInverseInertia(localInertia.t, s.ori);
        s.vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, localInertia.t, outInertia.t, s.vel.ang);
    } else if (cb.mode == 2) {  // ConserveMomentumWithGyroscopicTorque :1232-1238
        Q newOrientation = integrateOrientation(s.ori, s.vel.ang, dt * 0.5f);
        s.ori = mask ? newOrientation : s.ori;
        outInertia.t = rotateInverseInertia(localInertia.t, s.ori);
        s.vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(s.ori, localInertia.t, s.vel.ang, dt);
    } else {
        Q newOrientation = integrateOrientation(s.ori, s.vel.ang, dt * 0.5f);
        s.ori = mask ? newOrientation : s.ori;
        outInertia.t = rotateInverseInertia(localInertia.t, s.ori);
    }
    cb.integrateVelocity(s.vel, s.pos, bodyIndex, dt);
    s.vel.lin = sel3(mask, s.vel.lin, previousVelocity.lin);
    s.vel.ang = sel3(mask, s.vel.ang, previousVelocity.ang);
}
// TypeProcessor.cs:1251-1283, one lane. NOTE (:1264-1281): in the conserving modes the angular velocity of EVERY lane of the bundle is transformed
// before the conditional branch saves `previousVelocity`, so a lane that does not integrate here (its body was integrated by an earlier batch) still
// leaves with a transformed angular velocity when another lane of its bundle integrates. Reproduced as is.
template <int Mode>
inline void integrateVelocity(const Callbacks& cb, const Inertia& localInertia, float dt, bool mask, BodyState& s, Inertia& outInertia, int bodyIndex) {
    outInertia.invMass = localInertia.invMass;
    outInertia.t = rotateInverseInertia(localInertia.t, s.ori);
    if (cb.mode == 1) {
        Q previousOrientation = integrateOrientation(s.ori, s.vel.ang, dt * -0.5f);  // "integrating backwards to get a previous orientation"
        s.vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, localInertia.t, outInertia.t, s.vel.ang);
    } else if (cb.mode == 2) {
        s.vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(s.ori, localInertia.t, s.vel.ang, dt);
    }
    if (Mode == kConditional) {
        BodyVel previousVelocity = s.vel;
        cb.integrateVelocity(s.vel, s.pos, bodyIndex, dt);
        s.vel.lin = sel3(mask, s.vel.lin, previousVelocity.lin);
        s.vel.ang = sel3(mask, s.vel.ang, previousVelocity.ang);
    } else {
        cb.integrateVelocity(s.vel, s.pos, bodyIndex, dt);
    }
}

// TypeProcessor.GatherAndIntegrate for one bundle and one body slot (TypeProcessor.cs:1298-1397).
template <int Mode, bool AllowPose>
inline void gatherAndIntegrateBundle(Ctx& c, const std::vector<uint64_t>* flagsForSlot, float dt, int bundleIndex, const int32_t* refs, int lanes, BodyState* out) {
    float* bodies = c.scene->bodies;
    const int W = c.W;
    if (Mode == kNever) {
        for (int l = 0; l < lanes; ++l) gatherState(bodies, refs[l], true, out[l]);
        return;
    }
    bool mask[kMaxW];
    bool anyIntegrate;
    if (Mode == kAlways) {
        for (int l = 0; l < lanes; ++l) mask[l] = (uint32_t)refs[l] < kDynamicLimit;  // :1312
        anyIntegrate = true;
    } else {
        // BundleShouldIntegrate, TypeProcessor.cs:1155-1202
        int constraintStartIndex = bundleIndex * W;
        int flagBundleIndex = constraintStartIndex >> 6;
        int flagInnerIndex = constraintStartIndex - (flagBundleIndex << 6);
        uint32_t flagMask = (W >= 32) ? 0xFFFFFFFFu : ((1u << W) - 1);
        uint32_t scalarIntegrationMask = ((uint32_t)((*flagsForSlot)[flagBundleIndex] >> flagInnerIndex)) & flagMask;
        anyIntegrate = scalarIntegrationMask != 0;
        for (int l = 0; l < lanes; ++l) mask[l] = (scalarIntegrationMask >> l) & 1u;
    }
    // Note the reference gathers *local* inertia for the whole bundle if any lane integrates, world otherwise (:1333).
    for (int l = 0; l < lanes; ++l) gatherState(bodies, refs[l], !anyIntegrate, out[l]);
    if (!anyIntegrate) return;
    for (int l = 0; l < lanes; ++l) {
        Inertia local = out[l].inertia, world;
        const int bodyIndex = mask[l] ? (refs[l] & kBodyReferenceMask) : -1;  // DecodeBodyIndices (TypeProcessor.cs:1285-1296): masked lanes carry -1
        if (AllowPose) integratePoseAndVelocity(c.cb, local, dt, mask[l], out[l], world, bodyIndex);
        else integrateVelocity<Mode>(c.cb, local, dt, mask[l], out[l], world, bodyIndex);
        out[l].inertia = world;
        if (mask[l]) {
            int32_t idx = refs[l] & kBodyReferenceMask;
            if (AllowPose) scatterPose(bodies, idx, out[l].pos, out[l].ori);
            scatterInertia(bodies, idx, world);
        }
    }
}

template <class F>
inline void loadLane(const OracleTypeBatch& tb, int W, int bundle, int lane, float* p, float* a) {
    const float* pb = tb.prestep + (size_t)bundle * F::prestepFloats * W + lane;
    for (int f = 0; f < F::prestepFloats; ++f) p[f] = pb[(size_t)f * W];
    const float* ab = tb.accumulated + (size_t)bundle * F::impulseFloats * W + lane;
    for (int f = 0; f < F::impulseFloats; ++f) a[f] = ab[(size_t)f * W];
}
template <class F>
inline void storeAccumulated(const OracleTypeBatch& tb, int W, int bundle, int lane, const float* a) {
    float* ab = tb.accumulated + (size_t)bundle * F::impulseFloats * W + lane;
    for (int f = 0; f < F::impulseFloats; ++f) ab[(size_t)f * W] = a[f];
}
template <class F>
inline void storePrestep(const OracleTypeBatch& tb, int W, int bundle, int lane, const float* p) {
    float* pb = tb.prestep + (size_t)bundle * F::prestepFloats * W + lane;
    for (int f = 0; f < F::prestepFloats; ++f) pb[(size_t)f * W] = p[f];
}

// Two/OneBodyTypeProcessor.WarmStart (TwoBodyTypeProcessor.cs:168-203, OneBodyTypeProcessor.cs:82-112)
template <class F, int Mode, bool AllowPose>
void warmStartRange(Ctx& c, const OracleTypeBatch& tb, const std::vector<std::vector<uint64_t>>* flagsForTypeBatch, float dt, int startBundle, int endBundle) {
    const int W = c.W;
    float* bodies = c.scene->bodies;
    BodyState sA[kMaxW], sB[kMaxW];
    for (int b = startBundle; b < endBundle; ++b) {
        // Trailing lanes of the last bundle hold -1 body refs (TypeProcessor.cs:287-298); the reference still runs them on zeros. We run all W lanes the same way.
        const int32_t* refsA = tb.body_refs + (size_t)b * F::bodies * W;
        const int32_t* refsB = refsA + W;
        gatherAndIntegrateBundle<Mode, AllowPose>(c, flagsForTypeBatch ? &(*flagsForTypeBatch)[0] : nullptr, dt, b, refsA, W, sA);
        if (F::bodies == 2) gatherAndIntegrateBundle<Mode, AllowPose>(c, flagsForTypeBatch ? &(*flagsForTypeBatch)[1] : nullptr, dt, b, refsB, W, sB);
        for (int l = 0; l < W; ++l) {
            if (b * W + l >= tb.constraint_count) break;  // padding lanes (refs == -1) produce discarded garbage in the reference; skipped here
            float p[40], a[16];
            loadLane<F>(tb, W, b, l, p, a);
            BodyState zero;
            memset(&zero, 0, sizeof(zero));
            BodyState& B = (F::bodies == 2) ? sB[l] : zero;
            F::warmStart(sA[l].pos, sA[l].ori, sA[l].inertia, B.pos, B.ori, B.inertia, p, a, sA[l].vel, B.vel);
            scatterVelocities(bodies, refsA[l], sA[l].vel);
            if (F::bodies == 2) scatterVelocities(bodies, refsB[l], B.vel);
        }
    }
}
// Two/OneBodyTypeProcessor.Solve (TwoBodyTypeProcessor.cs:205-225, OneBodyTypeProcessor.cs:114-130)
template <class F>
void solveRange(Ctx& c, const OracleTypeBatch& tb, float dt, float inverseDt, int startBundle, int endBundle) {
    const int W = c.W;
    float* bodies = c.scene->bodies;
    for (int b = startBundle; b < endBundle; ++b) {
        const int32_t* refsA = tb.body_refs + (size_t)b * F::bodies * W;
        const int32_t* refsB = refsA + W;
        for (int l = 0; l < W; ++l) {
            if (b * W + l >= tb.constraint_count) break;
            BodyState A, B;
            gatherState(bodies, refsA[l], true, A);
            if (F::bodies == 2) gatherState(bodies, refsB[l], true, B); else memset(&B, 0, sizeof(B));
            float p[40], a[16];
            loadLane<F>(tb, W, b, l, p, a);
            F::solve(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, dt, inverseDt, p, a, A.vel, B.vel);
            storeAccumulated<F>(tb, W, b, l, a);
            scatterVelocities(bodies, refsA[l], A.vel);
            if (F::bodies == 2) scatterVelocities(bodies, refsB[l], B.vel);
        }
    }
}
// Three/FourBodyTypeProcessor.WarmStart / Solve (ThreeBodyTypeProcessor.cs, FourBodyTypeProcessor.cs): the same bundle loop over F::bodies body slots;
// every slot integrates through GatherAndIntegrate with its own flags, the constraint function takes the bodies as arrays.
template <class F, int Mode, bool AllowPose>
void warmStartRangeMany(Ctx& c, const OracleTypeBatch& tb, const std::vector<std::vector<uint64_t>>* flagsForTypeBatch, float dt, int startBundle, int endBundle) {
    const int W = c.W;
    float* bodies = c.scene->bodies;
    BodyState st[F::bodies][kMaxW];
    for (int b = startBundle; b < endBundle; ++b) {
        const int32_t* refs = tb.body_refs + (size_t)b * F::bodies * W;
        for (int k = 0; k < F::bodies; ++k)
            gatherAndIntegrateBundle<Mode, AllowPose>(c, flagsForTypeBatch ? &(*flagsForTypeBatch)[k] : nullptr, dt, b, refs + k * W, W, st[k]);
        for (int l = 0; l < W; ++l) {
            if (b * W + l >= tb.constraint_count) break;
            float p[40], a[16];
            loadLane<F>(tb, W, b, l, p, a);
            V3 pos[F::bodies]; float inverseMass[F::bodies]; BodyVel vel[F::bodies];
            for (int k = 0; k < F::bodies; ++k) { pos[k] = st[k][l].pos; inverseMass[k] = st[k][l].inertia.invMass; vel[k] = st[k][l].vel; }
            F::warmStartN(pos, inverseMass, p, a, vel);
            for (int k = 0; k < F::bodies; ++k) scatterVelocities(bodies, refs[k * W + l], vel[k]);
        }
    }
}
template <class F>
void solveRangeMany(Ctx& c, const OracleTypeBatch& tb, float dt, float inverseDt, int startBundle, int endBundle) {
    const int W = c.W;
    float* bodies = c.scene->bodies;
    for (int b = startBundle; b < endBundle; ++b) {
        const int32_t* refs = tb.body_refs + (size_t)b * F::bodies * W;
        for (int l = 0; l < W; ++l) {
            if (b * W + l >= tb.constraint_count) break;
            V3 pos[F::bodies]; float inverseMass[F::bodies]; BodyVel vel[F::bodies];
            for (int k = 0; k < F::bodies; ++k) {
                BodyState s;
                gatherState(bodies, refs[k * W + l], true, s);
                pos[k] = s.pos; inverseMass[k] = s.inertia.invMass; vel[k] = s.vel;
            }
            float p[40], a[16];
            loadLane<F>(tb, W, b, l, p, a);
            F::solveN(pos, inverseMass, dt, inverseDt, p, a, vel);
            storeAccumulated<F>(tb, W, b, l, a);
            for (int k = 0; k < F::bodies; ++k) scatterVelocities(bodies, refs[k * W + l], vel[k]);
        }
    }
}
// IncrementallyUpdateForSubstep (TwoBodyTypeProcessor.cs:227-241, OneBodyTypeProcessor.cs:132-146)
template <class F>
void incrementalRange(Ctx& c, const OracleTypeBatch& tb, float dt, int startBundle, int endBundle) {
    const int W = c.W;
    float* bodies = c.scene->bodies;
    for (int b = startBundle; b < endBundle; ++b) {
        const int32_t* refsA = tb.body_refs + (size_t)b * F::bodies * W;
        const int32_t* refsB = refsA + W;
        for (int l = 0; l < W; ++l) {
            if (b * W + l >= tb.constraint_count) break;
            BodyState A, B;
            gatherState(bodies, refsA[l], true, A);
            if (F::bodies == 2) gatherState(bodies, refsB[l], true, B); else memset(&B, 0, sizeof(B));
            float p[40], a[16];
            loadLane<F>(tb, W, b, l, p, a);
            F::incrementalUpdate(dt, A.vel, B.vel, p);
            storePrestep<F>(tb, W, b, l, p);
        }
    }
}

enum Stage { kStageIncremental, kStageWarmStart, kStageSolve };

template <class F, int Mode, bool AllowPose>
void warmStartDispatch(Ctx& c, const OracleTypeBatch& tb, const std::vector<std::vector<uint64_t>>* flags, float dt, int startBundle, int endBundle) {
    if constexpr (F::bodies > 2) warmStartRangeMany<F, Mode, AllowPose>(c, tb, flags, dt, startBundle, endBundle);
    else warmStartRange<F, Mode, AllowPose>(c, tb, flags, dt, startBundle, endBundle);
}
template <class F>
void runTyped(Ctx& c, Stage stage, int batchIndex, int typeBatchIndex, const OracleTypeBatch& tb, int substepIndex, float dt, float inverseDt, int startBundle, int endBundle) {
    if (stage == kStageIncremental) {
        if constexpr (F::incremental) incrementalRange<F>(c, tb, dt, startBundle, endBundle);
        return;
    }
    if (stage == kStageSolve) {
        if constexpr (F::bodies > 2) solveRangeMany<F>(c, tb, dt, inverseDt, startBundle, endBundle);
        else solveRange<F>(c, tb, dt, inverseDt, startBundle, endBundle);
        return;
    }
    // WarmStartBlock, Solver_Solve.cs:185-210
    if (batchIndex == 0) {
        if (substepIndex == 0) warmStartDispatch<F, kAlways, false>(c, tb, nullptr, dt, startBundle, endBundle);
        else warmStartDispatch<F, kAlways, true>(c, tb, nullptr, dt, startBundle, endBundle);
    } else if (c.coarse[batchIndex][typeBatchIndex]) {
        auto* fl = &c.flags[batchIndex][typeBatchIndex];
        if (substepIndex == 0) warmStartDispatch<F, kConditional, false>(c, tb, fl, dt, startBundle, endBundle);
        else warmStartDispatch<F, kConditional, true>(c, tb, fl, dt, startBundle, endBundle);
    } else {
        if (substepIndex == 0) warmStartDispatch<F, kNever, false>(c, tb, nullptr, dt, startBundle, endBundle);
        else warmStartDispatch<F, kNever, true>(c, tb, nullptr, dt, startBundle, endBundle);
    }
}

bool typeInfo(int typeId, int& bodies, int& prestepFloats, int& impulseFloats, bool& incremental) {
#define TI(T) { bodies = T::bodies; prestepFloats = T::prestepFloats; impulseFloats = T::impulseFloats; incremental = T::incremental; return true; }
    switch (typeId) {
        case kContact1OneBody: TI(C1O) case kContact2OneBody: TI(C2O)
        case kContact3OneBody: TI(C3O) case kContact4OneBody: TI(C4O)
        case kContact1: TI(C1T) case kContact2: TI(C2T)
        case kContact3: TI(C3T) case kContact4: TI(C4T)
#define X(ID, T) case ID: TI(T)
        BO_JOINT_TYPES(X)
        BO_NONCONVEX_CONTACT_TYPES(X)
        BO_MANY_BODY_TYPES(X)
#undef X
    }
#undef TI
    return false;
}

void runBlock(Ctx& c, Stage stage, int batchIndex, int typeBatchIndex, int substepIndex, float dt, float inverseDt, int startBundle, int endBundle) {
    const OracleTypeBatch& tb = c.scene->type_batches[c.batchStart[batchIndex] + typeBatchIndex];
#define RT(T) runTyped<T>(c, stage, batchIndex, typeBatchIndex, tb, substepIndex, dt, inverseDt, startBundle, endBundle); break;
    switch (tb.type_id) {
        case kContact1OneBody: RT(C1O) case kContact2OneBody: RT(C2O)
        case kContact3OneBody: RT(C3O) case kContact4OneBody: RT(C4O)
        case kContact1: RT(C1T) case kContact2: RT(C2T)
        case kContact3: RT(C3T) case kContact4: RT(C4T)
#define X(ID, T) case ID: RT(T)
        BO_JOINT_TYPES(X)
        BO_NONCONVEX_CONTACT_TYPES(X)
        BO_MANY_BODY_TYPES(X)
#undef X
    }
#undef RT
}

inline int bundleCount(int count, int W) { return (count + W - 1) / W; }

// ---- Solver.PrepareConstraintIntegrationResponsibilities (Solver_Solve.cs:1072-1388; region pass :951-1044) ----
void prepareIntegrationResponsibilities(Ctx& c) {
    OracleScene& s = *c.scene;
    const int W = c.W;
    int words = (s.handle_capacity + 63) / 64;
    if (words < 1) words = 1;
    c.mergedConstrained.assign(words, 0);
    c.flags.assign(s.batch_count, {});
    c.coarse.assign(s.batch_count, {});
    // batchReferencedHandles[b]: dynamic body handles referenced by batch b (BepuPhysics/Solver.cs:33,1046-1051); kinematics excluded (:1058-1078).
    std::vector<uint64_t> batchHandles(words), firstObserved(words);
    for (int b = 0; b < s.batch_count; ++b) {
        std::fill(batchHandles.begin(), batchHandles.end(), 0);
        int tbCount = s.type_batch_counts[b];
        for (int t = 0; t < tbCount; ++t) {
            const OracleTypeBatch& tb = s.type_batches[c.batchStart[b] + t];
            int bodies, pf, imf; bool inc;
            typeInfo(tb.type_id, bodies, pf, imf, inc);
            for (int i = 0; i < tb.constraint_count; ++i) {
                for (int k = 0; k < bodies; ++k) {
                    int32_t ref = tb.body_refs[(size_t)(i / W) * bodies * W + k * W + (i % W)];
                    if ((uint32_t)ref < kDynamicLimit) {
                        int h = s.index_to_handle[ref];
                        batchHandles[h >> 6] |= 1ull << (h & 63);
                    }
                }
            }
        }
        if (b == 0) {
            c.mergedConstrained = batchHandles;  // :1141-1146
            continue;
        }
        for (int w = 0; w < words; ++w) {  // :1198-1207
            uint64_t mergeBundle = c.mergedConstrained[w], batchBundle = batchHandles[w];
            c.mergedConstrained[w] = mergeBundle | batchBundle;
            firstObserved[w] = ~mergeBundle & batchBundle;
        }
        c.flags[b].resize(tbCount);
        c.coarse[b].assign(tbCount, 0);
        std::unordered_map<int, uint64_t> fallbackEarliest;  // body index -> smallest (type batch << 32 | index) among its constraints in the fallback batch
        if (b == c.fallbackThreshold) {
            for (int t = 0; t < tbCount; ++t) {
                const OracleTypeBatch& tb = s.type_batches[c.batchStart[b] + t];
                int bodies, pf, imf; bool inc;
                typeInfo(tb.type_id, bodies, pf, imf, inc);
                for (int i = 0; i < tb.constraint_count; ++i)
                    for (int k = 0; k < bodies; ++k) {
                        int32_t ref = tb.body_refs[(size_t)(i / W) * bodies * W + k * W + (i % W)];
                        if (ref == -1) continue;
                        uint64_t slot = ((uint64_t)t << 32) | (uint32_t)i;
                        auto it = fallbackEarliest.find(ref & kBodyReferenceMask);
                        if (it == fallbackEarliest.end() || slot < it->second) fallbackEarliest[ref & kBodyReferenceMask] = slot;
                    }
            }
        }
        for (int t = 0; t < tbCount; ++t) {  // ComputeIntegrationResponsibilitiesForConstraintRegion
            const OracleTypeBatch& tb = s.type_batches[c.batchStart[b] + t];
            int bodies, pf, imf; bool inc;
            typeInfo(tb.type_id, bodies, pf, imf, inc);
            int flagWords = (tb.constraint_count + 63) / 64;
            if (flagWords < 1) flagWords = 1;
            c.flags[b][t].assign(bodies, std::vector<uint64_t>(flagWords + 1, 0));
            uint64_t mergedFlagBundles = 0;
            const bool fallback = b == c.fallbackThreshold;  // ComputeIntegrationResponsibilitiesForConstraintRegion<IsFallbackBatch>, Solver_Solve.cs:978-1020
            for (int i = 0; i < tb.constraint_count; ++i) {
                for (int k = 0; k < bodies; ++k) {
                    int32_t ref = tb.body_refs[(size_t)(i / W) * bodies * W + k * W + (i % W)];
                    if (fallback && ref == -1) continue;  // an empty lane of a fallback bundle (:983-986)
                    int bodyIndex = ref & kBodyReferenceMask;
                    int h = s.index_to_handle[bodyIndex];
                    if ((firstObserved[h >> 6] >> (h & 63)) & 1ull) {
                        if (fallback) {
                            // A body may appear in many constraints of the fallback batch: the EARLIEST slot, ordered by (type batch, index in type batch), integrates it (:1003-1019).
                            uint64_t currentSlot = ((uint64_t)t << 32) | (uint32_t)i;
                            if (fallbackEarliest.count(bodyIndex) == 0 || fallbackEarliest[bodyIndex] != currentSlot) continue;
                        }
                        c.flags[b][t][k][i >> 6] |= 1ull << (i & 63);
                        mergedFlagBundles |= 1;
                    }
                }
            }
            c.coarse[b][t] = mergedFlagBundles != 0;
        }
    }
    for (int i = 0; i < s.constrained_kinematic_count; ++i) {  // :1378-1381
        int h = s.constrained_kinematic_handles[i];
        c.mergedConstrained[h >> 6] |= 1ull << (h & 63);
    }
}

// ---- PoseIntegrator.IntegrateKinematicVelocities / IntegrateKinematicPosesAndVelocities (PoseIntegrator.cs:451-535) ----
void integrateKinematics(Ctx& c, float substepDt, bool poses) {
    OracleScene& s = *c.scene;
    float halfDt = substepDt * 0.5f;
    for (int i = 0; i < s.constrained_kinematic_count; ++i) {
        int idx = s.handle_to_index[s.constrained_kinematic_handles[i]];
        BodyState st;
        gatherState(s.bodies, idx, false, st);
        if (poses) {
            st.pos = add(st.pos, scale(st.vel.lin, substepDt));
            st.ori = integrateOrientation(st.ori, st.vel.ang, halfDt);
            scatterPose(s.bodies, idx, st.pos, st.ori);
        }
        if (c.params->integrate_velocity_for_kinematics) {
            c.cb.integrateVelocity(st.vel, st.pos, idx, substepDt);
            scatterVelocities(s.bodies, idx, st.vel);  // plain index: written (ScatterVelocities<AccessAll>, :485,:530)
        }
    }
}

// ---- PoseIntegrator.IntegrateBundlesAfterSubstepping (PoseIntegrator.cs:537-693), lane-wise ----
void integrateAfterSubstepping(Ctx& c, int start, int end) {
    OracleScene& s = *c.scene;
    const OracleParams& p = *c.params;
    float dt = p.dt, substepDt = p.dt / p.substep_count;
    for (int i = start; i < end; ++i) {
        int h = s.index_to_handle[i];
        bool unconstrained = !((c.mergedConstrained[h >> 6] >> (h & 63)) & 1ull);
        float effectiveDt = p.allow_substeps_for_unconstrained ? substepDt : (unconstrained ? dt : substepDt);
        float halfDt = effectiveDt * 0.5f;
        BodyState st;
        gatherState(s.bodies, i, false, st);
        bool isKinematic = st.inertia.t.xx == 0 && st.inertia.t.yx == 0 && st.inertia.t.yy == 0 && st.inertia.t.zx == 0 &&
                           st.inertia.t.zy == 0 && st.inertia.t.zz == 0 && st.inertia.invMass == 0;  // Bodies.IsKinematic, Bodies.cs:326-349
        bool velocityMask = p.integrate_velocity_for_kinematics ? unconstrained : (unconstrained && !isKinematic);
        if (unconstrained) {
            int steps = p.allow_substeps_for_unconstrained ? p.substep_count : 1;
            for (int stepIndex = 0; stepIndex < steps; ++stepIndex) {
                if (velocityMask) c.cb.integrateVelocity(st.vel, st.pos, i, effectiveDt);
                st.pos = add(st.pos, scale(st.vel.lin, effectiveDt));
                if (c.cb.mode == 1) {  // PoseIntegrator.cs:649-655
                    Q previousOrientation = st.ori;
                    st.ori = integrateOrientation(st.ori, st.vel.ang, halfDt);
                    Sym3 inverseInertiaTensor = rotateInverseInertia(st.inertia.t, st.ori);
                    st.vel.ang = integrateAngularVelocityConserveMomentum(previousOrientation, st.inertia.t, inverseInertiaTensor, st.vel.ang);
                } else if (c.cb.mode == 2) {  // :656-660
                    st.ori = integrateOrientation(st.ori, st.vel.ang, halfDt);
                    st.vel.ang = integrateAngularVelocityConserveMomentumWithGyroscopicTorque(st.ori, st.inertia.t, st.vel.ang, effectiveDt);
                } else {
                    st.ori = integrateOrientation(st.ori, st.vel.ang, halfDt);
                }
                scatterPose(s.bodies, i, st.pos, st.ori);
                if (velocityMask) scatterVelocities(s.bodies, i, st.vel);
            }
        } else {
            st.ori = integrateOrientation(st.ori, st.vel.ang, halfDt);
            st.pos = add(st.pos, scale(st.vel.lin, effectiveDt));
            scatterPose(s.bodies, i, st.pos, st.ori);
        }
    }
}

// ---- Threading: the reference's work-block + barrier scheme (Solver_Solve.cs:683-741,780-787,514), simplified:
// blocks of <= 1024 bundles, ~4 blocks per type batch per worker, claimed by atomic counter, one barrier per (stage,batch). ----
struct Block { int typeBatch, start, end; };
struct Pool {
    int n;
    std::vector<std::thread> threads;
    std::atomic<int> generation{0}, arrived{0}, next{0};
    std::atomic<bool> quit{false};
    const std::vector<Block>* blocks = nullptr;
    std::function<void(const Block&)>* fn = nullptr;
    void workerLoop() {
        int seen = 0;
        for (;;) {
            for (unsigned spins = 0; generation.load(std::memory_order_acquire) == seen; ++spins) {
                if (quit.load()) return;
                if (spins > 2000) std::this_thread::yield();
#if defined(__x86_64__)
                else __builtin_ia32_pause();
#endif
            }
            seen = generation.load(std::memory_order_acquire);
            drain();
            arrived.fetch_add(1, std::memory_order_acq_rel);
        }
    }
    void drain() {
        int count = (int)blocks->size();
        for (;;) {
            int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= count) break;
            (*fn)((*blocks)[i]);
        }
    }
    explicit Pool(int n_) : n(n_) {
        for (int i = 1; i < n; ++i) threads.emplace_back([this] { workerLoop(); });
    }
    ~Pool() {
        quit.store(true);
        for (auto& t : threads) t.join();
    }
    void run(const std::vector<Block>& b, std::function<void(const Block&)>& f) {
        if (b.empty()) return;
        blocks = &b; fn = &f;
        next.store(0); arrived.store(0);
        generation.fetch_add(1, std::memory_order_release);
        drain();
        for (unsigned spins = 0; arrived.load(std::memory_order_acquire) < n - 1; ++spins) {
            if (spins > 2000) std::this_thread::yield();
#if defined(__x86_64__)
            else __builtin_ia32_pause();
#endif
        }
    }
};

}  // namespace

extern "C" {

// Returns 0 on success, negative on unsupported input.
int oracle_solve(OracleScene* scene, const OracleParams* params) {
    Ctx c;
    c.scene = scene;
    c.params = params;
    c.W = scene->bundle_width;
    if (c.W < 1 || c.W > kMaxW || (64 % c.W) != 0) return -1;
    if (!(params->dt > 0) || params->substep_count < 1) return -2;
    c.batchStart.resize(scene->batch_count + 1);
    int acc = 0;
    for (int b = 0; b < scene->batch_count; ++b) { c.batchStart[b] = acc; acc += scene->type_batch_counts[b]; }
    c.batchStart[scene->batch_count] = acc;
    for (int t = 0; t < acc; ++t) {
        int bodies, pf, imf; bool inc;
        if (!typeInfo(scene->type_batches[t].type_id, bodies, pf, imf, inc)) return -3;
    }
    const int W = c.W;
    const int threads = params->threads < 1 ? 1 : params->threads;

    c.fallbackThreshold = params->fallback_batch_threshold > 0 ? params->fallback_batch_threshold : 64;
    if (scene->batch_count > c.fallbackThreshold + 1) return -5;  // at most FallbackBatchThreshold synchronized batches + the fallback batch (Solver.cs:1878-1884)
    // Simulation.Solve: prepass -> Solve -> IntegrateAfterSubstepping (Simulation.cs:278-290)
    prepareIntegrationResponsibilities(c);

    const float substepDt = params->dt / params->substep_count;  // Solver_Solve.cs:1417
    c.cb.prepare(*params, substepDt);                             // :1418
    const float inverseDt = 1.0f / substepDt;

    // The worker pool persists across calls (bench.py's cpu_baseline solves many frames): creating hundreds of threads per frame would be timed too.
    static std::unique_ptr<Pool> persistent;
    static std::mutex persistentMutex;
    std::unique_lock<std::mutex> poolLock(persistentMutex, std::defer_lock);
    Pool* pool = nullptr;
    if (threads > 1) {
        poolLock.lock();
        if (!persistent || persistent->n != threads) persistent.reset(new Pool(threads));
        pool = persistent.get();
    }
    std::vector<std::vector<Block>> batchBlocks(scene->batch_count);
    if (threads > 1) {
        for (int b = 0; b < scene->batch_count; ++b) {
            for (int t = 0; t < scene->type_batch_counts[b]; ++t) {
                int bundles = bundleCount(scene->type_batches[c.batchStart[b] + t].constraint_count, W);
                int target = 4 * threads;  // target blocks per batch per worker (:780-787)
                int per = (bundles + target - 1) / target;
                if (per < 1) per = 1;
                if (per > 1024) per = 1024;
                for (int s0 = 0; s0 < bundles; s0 += per) batchBlocks[b].push_back({t, s0, s0 + per < bundles ? s0 + per : bundles});
            }
        }
    }
    auto runStage = [&](Stage stage, int b, int substepIndex) {
        // The fallback batch is solved by one thread, bundle after bundle (Solver_Solve.cs:546-583): its bundles may share bodies with each other.
        // (Its incremental contact update only writes the constraint's own depths and is dispatched like any other batch, :154-164.)
        if (threads > 1 && !(b == c.fallbackThreshold && stage != kStageIncremental)) {
            std::function<void(const Block&)> f = [&](const Block& blk) { runBlock(c, stage, b, blk.typeBatch, substepIndex, substepDt, inverseDt, blk.start, blk.end); };
            pool->run(batchBlocks[b], f);
        } else {
            for (int t = 0; t < scene->type_batch_counts[b]; ++t) {
                int bundles = bundleCount(scene->type_batches[c.batchStart[b] + t].constraint_count, W);
                runBlock(c, stage, b, t, substepIndex, substepDt, inverseDt, 0, bundles);
            }
        }
    };

    for (int substepIndex = 0; substepIndex < params->substep_count; ++substepIndex) {  // :1425
        if (substepIndex > 0) {
            for (int b = 0; b < scene->batch_count; ++b) runStage(kStageIncremental, b, substepIndex);  // :1429-1439
            integrateKinematics(c, substepDt, true);                                                     // :1440
        } else if (params->integrate_velocity_for_kinematics) {
            integrateKinematics(c, substepDt, false);                                                    // :1444-1445
        }
        for (int b = 0; b < scene->batch_count; ++b) runStage(kStageWarmStart, b, substepIndex);         // :1447-1463
        if (params->exchange && params->exchange(params->exchange_user, substepIndex, 0) != 0) return -4;
        int iterations = params->velocity_iterations[substepIndex];
        for (int it = 0; it < iterations; ++it) {                                                        // :1464-1476
            for (int b = 0; b < scene->batch_count; ++b) runStage(kStageSolve, b, substepIndex);
            if (params->exchange && params->exchange(params->exchange_user, substepIndex, 1 + it) != 0) return -4;
        }
    }

    // PoseIntegrator.IntegrateAfterSubstepping (PoseIntegrator.cs:707-726)
    float velocityIntegrationTimestep = params->allow_substeps_for_unconstrained ? substepDt : params->dt;
    c.cb.prepare(*params, velocityIntegrationTimestep);
    if (threads > 1) {
        std::vector<Block> blocks;
        int per = (scene->body_count + 8 * threads - 1) / (8 * threads);
        if (per < 64) per = 64;
        for (int s0 = 0; s0 < scene->body_count; s0 += per) blocks.push_back({0, s0, s0 + per < scene->body_count ? s0 + per : scene->body_count});
        std::function<void(const Block&)> f = [&](const Block& blk) { integrateAfterSubstepping(c, blk.start, blk.end); };
        pool->run(blocks, f);
    } else {
        integrateAfterSubstepping(c, 0, scene->body_count);
    }
    return 0;
}

// Integration-responsibility prepass only, for parity-testing the host mirror's a2 implementation.
// out_merged: handle bitset (words = (handle_capacity+63)/64). out_flags: for every (batch>=1, typeBatch, slot) in order,
// (constraint_count+63)/64 words each, concatenated. out_coarse: one byte per type batch (flattened, batch 0 entries = 0).
int oracle_prepare_flags(OracleScene* scene, uint64_t* out_merged, uint64_t* out_flags, int64_t out_flags_capacity, uint8_t* out_coarse) {
    Ctx c;
    c.scene = scene;
    c.params = nullptr;
    c.W = scene->bundle_width;
    c.batchStart.resize(scene->batch_count + 1);
    int acc = 0;
    for (int b = 0; b < scene->batch_count; ++b) { c.batchStart[b] = acc; acc += scene->type_batch_counts[b]; }
    c.batchStart[scene->batch_count] = acc;
    prepareIntegrationResponsibilities(c);
    memcpy(out_merged, c.mergedConstrained.data(), c.mergedConstrained.size() * 8);
    int64_t o = 0;
    for (int t = 0; t < acc; ++t) out_coarse[t] = 0;
    for (int b = 1; b < scene->batch_count; ++b) {
        for (int t = 0; t < scene->type_batch_counts[b]; ++t) {
            out_coarse[c.batchStart[b] + t] = c.coarse[b][t];
            int words = (scene->type_batches[c.batchStart[b] + t].constraint_count + 63) / 64;
            for (auto& slot : c.flags[b][t]) {
                if (o + words > out_flags_capacity) return -1;
                memcpy(out_flags + o, slot.data(), (size_t)words * 8);
                o += words;
            }
        }
    }
    return 0;
}

// Per-function access for known-answer tests: apply WarmStart then Solve `iterations` times to a single lane
// (TwoBodyConstraintBenchmarks-style, DemoBenchmarks/TwoBodyConstraintBenchmarks.cs:19-37).
// bodyA/bodyB: 32-float BodyDynamics (world inertia slot used as-is); prestep/accumulated: flat lane arrays.
int oracle_constraint_iterate(int type_id, float* bodyA, float* bodyB, float* prestep, float* accumulated, float dt, int iterations) {
    int bodies, pf, imf; bool inc;
    if (!typeInfo(type_id, bodies, pf, imf, inc) || bodies > 2) return -3;  // (three- and four-body types are exercised through whole scenes)
    BodyState A, B;
    gatherState(bodyA, 0, true, A);
    if (bodies == 2) gatherState(bodyB, 0, true, B); else memset(&B, 0, sizeof(B));
    float inverseDt = 1.0f / dt;
    for (int i = 0; i < iterations; ++i) {
#define IT(T) T::warmStart(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, prestep, accumulated, A.vel, B.vel); \
              T::solve(A.pos, A.ori, A.inertia, B.pos, B.ori, B.inertia, dt, inverseDt, prestep, accumulated, A.vel, B.vel); break;
        switch (type_id) {
            case kContact1OneBody: IT(C1O) case kContact2OneBody: IT(C2O)
            case kContact3OneBody: IT(C3O) case kContact4OneBody: IT(C4O)
            case kContact1: IT(C1T) case kContact2: IT(C2T)
            case kContact3: IT(C3T) case kContact4: IT(C4T)
#define X(ID, T) case ID: IT(T)
            BO_JOINT_TYPES(X)
            BO_NONCONVEX_CONTACT_TYPES(X)
#undef X
        }
#undef IT
    }
    scatterVelocities(bodyA, 0, A.vel);
    if (bodies == 2) scatterVelocities(bodyB, 0, B.vel);
    return 0;
}

// Scalar math probes for unit tests (MathHelper.Sin/Cos/Acos restatements).
// PoseIntegrator.PredictBoundingBoxes (PoseIntegrator.cs:307-370) over `count` bodies: sleep candidacy from the stored velocity, the velocity callback for
// the full dt on a copy (integration mask: non-kinematic bodies unless IntegrateVelocityForKinematics), bounds of the primitive convex shapes.
int oracle_predict_bounding_boxes_hulls(const float* bodies, int count, const OracleParams* params, const CollidableIn* collidables, PredictedBounds* out, const float* hull_points,
                                        const int* hull_begin, int hull_count);
int oracle_predict_bounding_boxes(const float* bodies, int count, const OracleParams* params, const CollidableIn* collidables, PredictedBounds* out) {
    return oracle_predict_bounding_boxes_hulls(bodies, count, params, collidables, out, nullptr, nullptr, 0);
}
// ... with convex hulls (ConvexHull.Id = 5, ConvexHull.cs:319-364): collidable.shape[0] = hull index, hull h = points [hull_begin[h], hull_begin[h + 1]).
int oracle_predict_bounding_boxes_shapes(const float* bodies, int count, const OracleParams* params, const CollidableIn* collidables, PredictedBounds* out, const float* hull_points,
                                         const int* hull_begin, int hull_count, const CompoundChildIn* children, const int* child_begin, int compound_count, const float* triangles,
                                         const int* triangle_begin, const float* mesh_scales, int mesh_count, int bundle_width);
int oracle_predict_bounding_boxes_hulls(const float* bodies, int count, const OracleParams* params, const CollidableIn* collidables, PredictedBounds* out, const float* hull_points,
                                        const int* hull_begin, int hull_count) {
    return oracle_predict_bounding_boxes_shapes(bodies, count, params, collidables, out, hull_points, hull_begin, hull_count, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 8);
}
// ... and with compounds (Compound.Id = 6, BigCompound.Id = 7: collidable.shape[0] = compound index) and meshes (Mesh.Id = 8: shape[0] = mesh index).
int oracle_predict_bounding_boxes_shapes(const float* bodies, int count, const OracleParams* params, const CollidableIn* collidables, PredictedBounds* out, const float* hull_points,
                                         const int* hull_begin, int hull_count, const CompoundChildIn* children, const int* child_begin, int compound_count, const float* triangles,
                                         const int* triangle_begin, const float* mesh_scales, int mesh_count, int bundle_width) {
    if (!bodies || !params || !collidables || !out || count < 0 || !(params->dt > 0) || (bundle_width != 4 && bundle_width != 8 && bundle_width != 16)) return -1;
    const ShapeTables tables = {{hull_points, hull_begin, hull_count}, {children, child_begin, compound_count}, {triangles, triangle_begin, mesh_scales, mesh_count}};
    for (int i = 0; i < count; ++i) {  // indices must name table entries
        const int t = collidables[i].shape_type, k = (int)collidables[i].shape[0];
        if ((t == kShapeCompound || t == kShapeBigCompound) && (k < 0 || k >= compound_count)) return -2;
        if (t == kShapeMesh && (k < 0 || k >= mesh_count)) return -2;
        if (t > kShapeMesh) return -2;
    }
    Callbacks cb;
    cb.prepare(*params, params->dt);   // PredictBoundingBoxes(dt, ...) -> Callbacks.PrepareForIntegration(dt)
    // PoseIntegrator.cs:315-368, bundle by bundle (bundle_width = Vector<float>.Count of the host): the callback runs on the WHOLE bundle as soon as one of its lanes
    // is to be integrated (:337-338), and nothing masks its result afterwards — the demo callbacks ignore the mask (DemoCallbacks.cs:99-109, relying on the caller
    // to discard inactive lanes as the interface promises, PoseIntegrator.cs:91) — so a kinematic body that shares a bundle with a dynamic one is predicted with
    // gravity and damping applied, and one in an all-kinematic bundle is not. Restated as written.
    for (int bundleStart = 0; bundleStart < count; bundleStart += bundle_width) {
        const int countInBundle = count - bundleStart < bundle_width ? count - bundleStart : bundle_width;
        BodyState lanes[16];
        float sleepEnergy[16];
        bool anyLaneIntegrates = false;
        for (int lane = 0; lane < countInBundle; ++lane) {
            BodyState& st = lanes[lane];
            gatherState(bodies, bundleStart + lane, false, st);  // GatherState<AccessAll>(laneIndices, false, ...): local inertia
            const bool isKinematic = st.inertia.t.xx == 0 && st.inertia.t.yx == 0 && st.inertia.t.yy == 0 && st.inertia.t.zx == 0 && st.inertia.t.zy == 0 &&
                                     st.inertia.t.zz == 0 && st.inertia.invMass == 0;                       // Bodies.IsKinematic, Bodies.cs:326-349
            sleepEnergy[lane] = lengthSquared(st.vel.lin) + lengthSquared(st.vel.ang);                     // :334, before the callback
            if (params->integrate_velocity_for_kinematics || !isKinematic) anyLaneIntegrates = true;       // :323-331
        }
        for (int lane = 0; lane < countInBundle; ++lane) {
            if (anyLaneIntegrates) cb.integrateVelocity(lanes[lane].vel, lanes[lane].pos, bundleStart + lane, params->dt);  // :337-338 (never stored)
            predictBoundsOfAnyShape(lanes[lane].pos, lanes[lane].ori, lanes[lane].vel, sleepEnergy[lane], params->dt, collidables[bundleStart + lane], tables, out[bundleStart + lane]);
        }
    }
    return 0;
}

void oracle_math_probe(const float* x, int n, float* out_sin, float* out_cos, float* out_acos) {
    for (int i = 0; i < n; ++i) { out_sin[i] = bsin(x[i]); out_cos[i] = bcos(x[i]); out_acos[i] = bacos(x[i]); }
}

int oracle_type_info(int type_id, int* bodies, int* prestep_floats, int* impulse_floats, int* incremental) {
    int b, p, i; bool inc;
    if (!typeInfo(type_id, b, p, i, inc)) return -3;
    *bodies = b; *prestep_floats = p; *impulse_floats = i; *incremental = inc;
    return 0;
}

}  // extern "C"
