// oracle/wide — TEST INFRASTRUCTURE (checker + bench.py's cpu_baseline leg); nothing under bepuphysics2_amd/ may use it. Parity unpinned (see wide_vec.h).
//
// The reference's CPU solver path restated in its own shape: AOSOA bundles of Vector<float>.Count = 8 lanes, AVX 8x8 transposes for the body
// gather/scatter (BepuPhysics/Bodies_GatherScatter.cs:267-753), integration fused into the first warm start that touches a body through the
// integration-responsibility flags (BepuPhysics/Solver_Solve.cs:951-1388, Constraints/TypeProcessor.cs:1155-1397), and the multithreaded
// work-block / sync-stage scheduler of Solver_Solve.cs:297-946. Transcribed from the C# only; see wide_vec.h for why that matters.
//
// C ABI: wide_solve(scene, params) — same marshalling structs as tests/oracle_ffi.py builds (a data format, not shared code).
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include <sched.h>

#include "wide_contacts.h"
#include "wide_joints_more.h"

namespace wide {

// ------------------------------------------------------------------------------------------------------------------ marshalling (tests/oracle_ffi.py)
struct SceneTypeBatch { int32_t type_id, constraint_count; void* body_refs; void* prestep; void* accumulated; };
struct SceneParams {
    float dt; int32_t substep_count; const int32_t* velocity_iterations; float gravity[3]; float linear_damping, angular_damping;
    int32_t allow_substeps_for_unconstrained, integrate_velocity_for_kinematics, threads; void* exchange; void* exchange_user; int32_t angular_integration_mode;
    int32_t fallback_batch_threshold;  // SolveDescription.FallbackBatchThreshold; 0 = 64
    int32_t velocity_model;            // which of the reference's callback structs PoseIntegratorCallbacks below stands for (0 demo, 1 per-body gravity, 2 planet)
    float planet_center[3]; float planet_gravity; const float* body_gravity;
};
struct SceneDesc {
    float* bodies; int32_t body_count; const int32_t* index_to_handle; const int32_t* handle_to_index; int32_t handle_capacity; int32_t batch_count;
    const int32_t* type_batch_counts; SceneTypeBatch* type_batches; const int32_t* constrained_kinematic_handles; int32_t constrained_kinematic_count; int32_t bundle_width;
};

// ------------------------------------------------------------------------------------------------------------------ BepuUtilities/Collections/IndexSet.cs
struct IndexSet {
    std::vector<uint64_t> Flags;
    bool Contains(int index) const {
        int packedIndex = index >> 6;
        return packedIndex < (int)Flags.size() && (Flags[packedIndex] & (1ull << (index & 63))) > 0;
    }
    void AddUnsafely(int index) { Flags[index >> 6] |= 1ull << (index & 63); }
    void Set(int index) {
        if ((index >> 6) >= (int)Flags.size()) Flags.resize((index >> 6) + 1, 0);
        AddUnsafely(index);
    }
    static int GetBundleCapacity(int count) { return (count + 63) >> 6; }
};

// ------------------------------------------------------------------------------------------------------------------ Bodies_GatherScatter.cs
constexpr int KinematicMask = 1 << 30;                      // :109
constexpr uint32_t DynamicLimit = (uint32_t)KinematicMask;  // :113
constexpr int BodyReferenceMask = 0x3FFFFFFF;               // :118

enum AngularIntegrationMode { Nonconserving = 0, ConserveMomentum = 1, ConserveMomentumWithGyroscopicTorque = 2 };  // PoseIntegrator.cs:20-38

static inline void Transpose8x8(const __m256 m[8], __m256 out[8]) {  // the unpack / shuffle / permute network of Bodies_GatherScatter.cs:323-356
    __m256 n0 = _mm256_unpacklo_ps(m[0], m[1]), n1 = _mm256_unpacklo_ps(m[2], m[3]), n2 = _mm256_unpacklo_ps(m[4], m[5]), n3 = _mm256_unpacklo_ps(m[6], m[7]);
    __m256 n4 = _mm256_unpackhi_ps(m[0], m[1]), n5 = _mm256_unpackhi_ps(m[2], m[3]), n6 = _mm256_unpackhi_ps(m[4], m[5]), n7 = _mm256_unpackhi_ps(m[6], m[7]);
    constexpr int lo = 0 | (1 << 2) | (0 << 4) | (1 << 6), hi = 2 | (3 << 2) | (2 << 4) | (3 << 6);
    __m256 o0 = _mm256_shuffle_ps(n0, n1, lo), o1 = _mm256_shuffle_ps(n2, n3, lo), o2 = _mm256_shuffle_ps(n4, n5, lo), o3 = _mm256_shuffle_ps(n6, n7, lo);
    __m256 o4 = _mm256_shuffle_ps(n0, n1, hi), o5 = _mm256_shuffle_ps(n2, n3, hi), o6 = _mm256_shuffle_ps(n4, n5, hi), o7 = _mm256_shuffle_ps(n6, n7, hi);
    out[0] = _mm256_permute2f128_ps(o0, o1, 0 | (2 << 4));
    out[1] = _mm256_permute2f128_ps(o4, o5, 0 | (2 << 4));
    out[2] = _mm256_permute2f128_ps(o2, o3, 0 | (2 << 4));
    out[3] = _mm256_permute2f128_ps(o6, o7, 0 | (2 << 4));
    out[4] = _mm256_permute2f128_ps(o0, o1, 1 | (3 << 4));
    out[5] = _mm256_permute2f128_ps(o4, o5, 1 | (3 << 4));
    out[6] = _mm256_permute2f128_ps(o2, o3, 1 | (3 << 4));
    out[7] = _mm256_permute2f128_ps(o6, o7, 1 | (3 << 4));
}

struct Bodies {
    float* states;  // ActiveSet.DynamicsState: 32 floats per body (BodyProperties.cs:318-338)
    int count;

    // GatherState<TAccessFilter> (:267-478). The access filters only decide which outputs are left uninitialised; the constraint functions never read
    // those, so everything is gathered here. Empty lanes (negative reference) read as zero.
    void GatherState(VI encodedBodyIndices, bool worldInertia, Vector3Wide& position, QuaternionWide& orientation, BodyVelocityWide& velocity, BodyInertiaWide& inertia) const {
        const float* s[8];
        bool empty[8];
        for (int i = 0; i < 8; ++i) {
            int bodyIndex = encodedBodyIndices[i];
            empty[i] = bodyIndex < 0;
            s[i] = states + (size_t)(bodyIndex & BodyReferenceMask) * 32;
        }
        __m256 m[8], t[8];
        const __m256 zero = _mm256_setzero_ps();
        for (int i = 0; i < 8; ++i) m[i] = empty[i] ? zero : _mm256_loadu_ps(s[i]);
        Transpose8x8(m, t);
        orientation.X = (VF)t[0]; orientation.Y = (VF)t[1]; orientation.Z = (VF)t[2]; orientation.W = (VF)t[3];
        position.X = (VF)t[4]; position.Y = (VF)t[5]; position.Z = (VF)t[6];
        for (int i = 0; i < 8; ++i) m[i] = empty[i] ? zero : _mm256_loadu_ps(s[i] + 8);
        Transpose8x8(m, t);
        velocity.Linear.X = (VF)t[0]; velocity.Linear.Y = (VF)t[1]; velocity.Linear.Z = (VF)t[2];
        velocity.Angular.X = (VF)t[4]; velocity.Angular.Y = (VF)t[5]; velocity.Angular.Z = (VF)t[6];
        const int offsetInFloats = worldInertia ? 24 : 16;  // :414
        for (int i = 0; i < 8; ++i) m[i] = empty[i] ? zero : _mm256_loadu_ps(s[i] + offsetInFloats);
        Transpose8x8(m, t);
        inertia.InverseInertiaTensor.XX = (VF)t[0]; inertia.InverseInertiaTensor.YX = (VF)t[1]; inertia.InverseInertiaTensor.YY = (VF)t[2];
        inertia.InverseInertiaTensor.ZX = (VF)t[3]; inertia.InverseInertiaTensor.ZY = (VF)t[4]; inertia.InverseInertiaTensor.ZZ = (VF)t[5];
        inertia.InverseMass = (VF)t[6];
    }
    void GatherVelocity(VI encodedBodyIndices, BodyVelocityWide& velocity) const {  // GatherState<AccessOnlyVelocity>
        __m256 m[8], t[8];
        const __m256 zero = _mm256_setzero_ps();
        for (int i = 0; i < 8; ++i) {
            int bodyIndex = encodedBodyIndices[i];
            m[i] = bodyIndex < 0 ? zero : _mm256_loadu_ps(states + (size_t)(bodyIndex & BodyReferenceMask) * 32 + 8);
        }
        Transpose8x8(m, t);
        velocity.Linear.X = (VF)t[0]; velocity.Linear.Y = (VF)t[1]; velocity.Linear.Z = (VF)t[2];
        velocity.Angular.X = (VF)t[4]; velocity.Angular.Y = (VF)t[5]; velocity.Angular.Z = (VF)t[6];
    }
    // ScatterPose (:484-549): 32 bytes at float offset 0 for lanes whose mask is set; the eighth float repeats position.Z (:502, "Laze alert").
    void ScatterPose(const Vector3Wide& position, const QuaternionWide& orientation, VI encodedBodyIndices, VI mask) {
        __m256 m[8] = {(__m256)orientation.X, (__m256)orientation.Y, (__m256)orientation.Z, (__m256)orientation.W,
                       (__m256)position.X,    (__m256)position.Y,    (__m256)position.Z,    (__m256)position.Z};
        __m256 t[8];
        Transpose8x8(m, t);
        for (int i = 0; i < 8; ++i)
            if (mask[i] != 0) _mm256_storeu_ps(states + (size_t)encodedBodyIndices[i] * 32, t[i]);
    }
    // ScatterInertia (:553-622): world inertia slot, float offset 24.
    void ScatterInertia(const BodyInertiaWide& inertia, VI encodedBodyIndices, VI mask) {
        __m256 m[8] = {(__m256)inertia.InverseInertiaTensor.XX, (__m256)inertia.InverseInertiaTensor.YX, (__m256)inertia.InverseInertiaTensor.YY, (__m256)inertia.InverseInertiaTensor.ZX,
                       (__m256)inertia.InverseInertiaTensor.ZY, (__m256)inertia.InverseInertiaTensor.ZZ, (__m256)inertia.InverseMass,             (__m256)inertia.InverseMass};
        __m256 t[8];
        Transpose8x8(m, t);
        for (int i = 0; i < 8; ++i)
            if (mask[i] != 0) _mm256_storeu_ps(states + (size_t)encodedBodyIndices[i] * 32 + 24, t[i]);
    }
    // ScatterVelocities<TAccessFilter> (:626-753): skipped for kinematic / empty references; 16-byte store when only one of the two is accessed.
    template <bool Linear, bool Angular> void ScatterVelocities(const BodyVelocityWide& sourceVelocities, const VI& encodedBodyIndices) {
        static_assert(Linear || Angular, "filter accesses no velocity");
        if constexpr (Linear != Angular) {
            const Vector3Wide& v = Linear ? sourceVelocities.Linear : sourceVelocities.Angular;
            const int targetOffset = Linear ? 8 : 12;
            __m256 m[8] = {(__m256)v.X, (__m256)v.Y, (__m256)v.Z, (__m256)v.Z, (__m256)v.X, (__m256)v.Y, (__m256)v.Z, (__m256)v.Z};
            __m256 t[8];
            Transpose8x8(m, t);  // t[i] = [x y z z | x y z z] of lane i
            for (int i = 0; i < 8; ++i) {
                uint32_t index = (uint32_t)encodedBodyIndices[i];
                if (index < DynamicLimit) _mm_storeu_ps(states + (size_t)index * 32 + targetOffset, _mm256_castps256_ps128(t[i]));
            }
        } else {
            __m256 m[8] = {(__m256)sourceVelocities.Linear.X,  (__m256)sourceVelocities.Linear.Y,  (__m256)sourceVelocities.Linear.Z,  (__m256)sourceVelocities.Linear.Z,
                           (__m256)sourceVelocities.Angular.X, (__m256)sourceVelocities.Angular.Y, (__m256)sourceVelocities.Angular.Z, (__m256)sourceVelocities.Angular.Z};
            __m256 t[8];
            Transpose8x8(m, t);
            for (int i = 0; i < 8; ++i) {
                uint32_t index = (uint32_t)encodedBodyIndices[i];
                if (index < DynamicLimit) _mm256_storeu_ps(states + (size_t)index * 32 + 8, t[i]);
            }
        }
    }
    static VI IsKinematic(const BodyInertiaWide& inertia) {  // Bodies.cs:326
        return Equals(BitwiseOrF(BitwiseOrF(BitwiseOrF(inertia.InverseMass, inertia.InverseInertiaTensor.XX), BitwiseOrF(inertia.InverseInertiaTensor.YX, inertia.InverseInertiaTensor.YY)),
                                 BitwiseOrF(BitwiseOrF(inertia.InverseInertiaTensor.ZX, inertia.InverseInertiaTensor.ZY), inertia.InverseInertiaTensor.ZZ)),
                      kZero);
    }
};

// ------------------------------------------------------------------------------------------------------------------ Demos/DemoCallbacks.cs:20-109
struct PoseIntegratorCallbacks {
    float Gravity[3], LinearDamping, AngularDamping;
    int AngularIntegrationMode;
    bool AllowSubstepsForUnconstrainedBodies, IntegrateVelocityForKinematics;
    Vector3Wide gravityWideDt;
    VF linearDampingDt, angularDampingDt;
    static float Clamp(float value, float min, float max) { return value < min ? min : (value > max ? max : value); }  // MathHelper.Clamp
    // Three of the reference's callback structs behind one name, picked by VelocityModel: 0 DemoPoseIntegratorCallbacks (Demos/DemoCallbacks.cs:20-109),
    // 1 PerBodyGravityDemoCallbacks (Demos/Demos/PerBodyGravityDemo.cs:20-89; BodyGravities holds the value of the body at each INDEX here — the demo's handle lookup
    // is done by the caller), 2 PlanetaryGravityCallbacks (Demos/Demos/PlanetDemo.cs:19-48).
    int VelocityModel = 0;
    float PlanetCenter[3] = {0, 0, 0}, PlanetGravity = 0;
    const float* BodyGravities = nullptr;
    float gravityDt = 0;
    void PrepareForIntegration(float dt) {                                                                          // :79
        linearDampingDt = vf(powf(Clamp(1 - LinearDamping, 0, 1), dt));
        angularDampingDt = vf(powf(Clamp(1 - AngularDamping, 0, 1), dt));
        gravityWideDt = Vector3Wide::Broadcast(Gravity[0] * dt, Gravity[1] * dt, Gravity[2] * dt);
        gravityDt = dt * PlanetGravity;                                                                             // PlanetDemo.cs:39
    }
    void IntegrateVelocity(const VI& bodyIndices, const Vector3Wide& position, const QuaternionWide& orientation, const BodyInertiaWide& localInertia, const VI& integrationMask,
                           int workerIndex, const VF& dt, BodyVelocityWide& velocity) const {  // :99
        if (VelocityModel == 1) {                                                                                   // PerBodyGravityDemo.cs:57-88
            VF gravityValues = kZero;                                                                               // stackalloc float[Vector<float>.Count]
            for (int bundleSlotIndex = 0; bundleSlotIndex < W; ++bundleSlotIndex) {
                const int bodyIndex = bodyIndices[bundleSlotIndex];
                if (bodyIndex >= 0) gravityValues[bundleSlotIndex] = BodyGravities[bodyIndex];                      // BodyGravities[bodies.ActiveSet.IndexToHandle[bodyIndex]]
            }
            velocity.Linear.Y = velocity.Linear.Y + gravityValues * dt;                                             // velocity.Linear.Y += new Vector<float>(gravityValues) * dt
            return;
        }
        if (VelocityModel == 2) {                                                                                   // PlanetDemo.cs:42-47
            Vector3Wide offset = position - Vector3Wide::Broadcast(PlanetCenter[0], PlanetCenter[1], PlanetCenter[2]);
            VF distance = Vector3Wide::Length(offset);
            Vector3Wide scaled = vf(gravityDt) * offset;
            VF inverse = kOne / Max(kOne, distance * distance * distance);                                          // Vector3Wide operator / (Vector3Wide.cs:357-365)
            velocity.Linear = velocity.Linear - scaled * inverse;
            return;
        }
        velocity.Linear = (velocity.Linear + gravityWideDt) * linearDampingDt;
        velocity.Angular = velocity.Angular * angularDampingDt;
    }
};

// ------------------------------------------------------------------------------------------------------------------ PoseIntegrator.cs:146-253
namespace PoseIntegration {
static inline void Integrate(const QuaternionWide& start, const Vector3Wide& angularVelocity, const VF& halfDt, QuaternionWide& integrated) {  // :146
    VF speed;
    Vector3Wide::Length(angularVelocity, speed);
    VF halfAngle = speed * halfDt;
    QuaternionWide q;
    VF s = MathHelper::Sin(halfAngle);
    VF scale = s / speed;
    q.X = angularVelocity.X * scale;
    q.Y = angularVelocity.Y * scale;
    q.Z = angularVelocity.Z * scale;
    q.W = MathHelper::Cos(halfAngle);
    QuaternionWide end;
    QuaternionWide::ConcatenateWithoutOverlap(start, q, end);
    end = QuaternionWide::Normalize(end);
    VI speedValid = GreaterThan(speed, vf(1e-15f));
    QuaternionWide result;  // `integrated` may alias `start` (PoseIntegrator.cs:525)
    result.X = ConditionalSelect(speedValid, end.X, start.X);
    result.Y = ConditionalSelect(speedValid, end.Y, start.Y);
    result.Z = ConditionalSelect(speedValid, end.Z, start.Z);
    result.W = ConditionalSelect(speedValid, end.W, start.W);
    integrated = result;
}
static inline void RotateInverseInertia(const Symmetric3x3Wide& localInverseInertiaTensor, const QuaternionWide& orientation, Symmetric3x3Wide& rotatedInverseInertiaTensor) {  // :167
    Matrix3x3Wide orientationMatrix;
    Matrix3x3Wide::CreateFromQuaternion(orientation, orientationMatrix);
    Symmetric3x3Wide::RotationSandwich(orientationMatrix, localInverseInertiaTensor, rotatedInverseInertiaTensor);
}
static inline void FallbackIfInertiaIncompatible(const Vector3Wide& previousAngularVelocity, Vector3Wide& angularVelocity) {  // :181
    VF infinity = vf(INFINITY);
    VI useNewVelocity = BitwiseAnd(LessThan(Abs(angularVelocity.X), infinity), BitwiseAnd(LessThan(Abs(angularVelocity.Y), infinity), LessThan(Abs(angularVelocity.Z), infinity)));
    angularVelocity.X = ConditionalSelect(useNewVelocity, angularVelocity.X, previousAngularVelocity.X);
    angularVelocity.Y = ConditionalSelect(useNewVelocity, angularVelocity.Y, previousAngularVelocity.Y);
    angularVelocity.Z = ConditionalSelect(useNewVelocity, angularVelocity.Z, previousAngularVelocity.Z);
}
static inline void IntegrateAngularVelocityConserveMomentum(const QuaternionWide& previousOrientation, const Symmetric3x3Wide& localInverseInertia,
                                                            const Symmetric3x3Wide& worldInverseInertia, Vector3Wide& angularVelocity) {  // :193
    Matrix3x3Wide previousOrientationMatrix;
    Matrix3x3Wide::CreateFromQuaternion(previousOrientation, previousOrientationMatrix);
    Vector3Wide localPreviousAngularVelocity, localAngularMomentum, angularMomentum;
    Matrix3x3Wide::TransformByTransposedWithoutOverlap(angularVelocity, previousOrientationMatrix, localPreviousAngularVelocity);
    Symmetric3x3Wide localInertiaTensor;
    Symmetric3x3Wide::Invert(localInverseInertia, localInertiaTensor);
    Symmetric3x3Wide::TransformWithoutOverlap(localPreviousAngularVelocity, localInertiaTensor, localAngularMomentum);
    Matrix3x3Wide::Transform(localAngularMomentum, previousOrientationMatrix, angularMomentum);
    Vector3Wide previousVelocity = angularVelocity;
    Symmetric3x3Wide::TransformWithoutOverlap(angularMomentum, worldInverseInertia, angularVelocity);
    FallbackIfInertiaIncompatible(previousVelocity, angularVelocity);
}
static inline void IntegrateAngularVelocityConserveMomentumWithGyroscopicTorque(const QuaternionWide& orientation, const Symmetric3x3Wide& localInverseInertia,
                                                                                Vector3Wide& angularVelocity, const VF& dt) {  // :209
    Matrix3x3Wide orientationMatrix;
    Matrix3x3Wide::CreateFromQuaternion(orientation, orientationMatrix);
    Vector3Wide localAngularVelocity, localAngularMomentum;
    Matrix3x3Wide::TransformByTransposedWithoutOverlap(angularVelocity, orientationMatrix, localAngularVelocity);
    Symmetric3x3Wide localInertiaTensor;
    Symmetric3x3Wide::Invert(localInverseInertia, localInertiaTensor);
    Symmetric3x3Wide::TransformWithoutOverlap(localAngularVelocity, localInertiaTensor, localAngularMomentum);
    Vector3Wide residual = dt * Vector3Wide::Cross(localAngularMomentum, localAngularVelocity);
    Matrix3x3Wide skewMomentum, skewVelocity;
    Matrix3x3Wide::CreateCrossProduct(localAngularMomentum, skewMomentum);
    Matrix3x3Wide::CreateCrossProduct(localAngularVelocity, skewVelocity);
    Matrix3x3Wide transformedSkewVelocity = skewVelocity * localInertiaTensor;
    Matrix3x3Wide changeOverDt, change;
    Matrix3x3Wide::Subtract(transformedSkewVelocity, skewMomentum, changeOverDt);
    Matrix3x3Wide::Scale(changeOverDt, dt, change);
    Matrix3x3Wide jacobian = localInertiaTensor + change;
    Matrix3x3Wide inverseJacobian;
    Matrix3x3Wide::Invert(jacobian, inverseJacobian);
    Vector3Wide newtonStep;
    Matrix3x3Wide::Transform(residual, inverseJacobian, newtonStep);
    localAngularVelocity = localAngularVelocity - newtonStep;
    Vector3Wide previousVelocity = angularVelocity;
    Matrix3x3Wide::Transform(localAngularVelocity, orientationMatrix, angularVelocity);
    FallbackIfInertiaIncompatible(previousVelocity, angularVelocity);
}
}  // namespace PoseIntegration

// ------------------------------------------------------------------------------------------------------------------ Constraints/TypeProcessor.cs:1147-1397
enum BundleIntegrationMode { None = 0, Partial = 1, All = 2 };
enum BatchIntegrationMode { BatchShouldAlwaysIntegrate, BatchShouldNeverIntegrate, BatchShouldConditionallyIntegrate };

static inline BundleIntegrationMode BundleShouldIntegrate(int bundleIndex, const IndexSet& integrationFlags, VI& integrationMask) {  // :1155
    int constraintStartIndex = bundleIndex * W;
    int flagBundleIndex = constraintStartIndex >> 6;
    int flagInnerIndex = constraintStartIndex - (flagBundleIndex << 6);
    int flagMask = (1 << W) - 1;
    int scalarIntegrationMask = ((int)(integrationFlags.Flags[flagBundleIndex] >> flagInnerIndex)) & flagMask;
    if (scalarIntegrationMask == flagMask) {
        integrationMask = vi(-1);
        return All;
    } else if (scalarIntegrationMask > 0) {
        VI selectors = {1, 2, 4, 8, 16, 32, 64, 128};
        VI scalarBroadcast = vi(scalarIntegrationMask);
        VI selected = BitwiseAnd(selectors, scalarBroadcast);
        integrationMask = EqualsI(selected, selectors);
        return Partial;
    }
    integrationMask = vi(0);
    return None;
}

static inline void IntegratePoseAndVelocity(PoseIntegratorCallbacks& integratorCallbacks, VI& bodyIndices, const BodyInertiaWide& localInertia, float dt, const VI& integrationMask,
                                            Vector3Wide& position, QuaternionWide& orientation, BodyVelocityWide& velocity, int workerIndex, BodyInertiaWide& inertia) {  // :1204
    VF dtWide = vf(dt);
    Vector3Wide newPosition = position + velocity.Linear * dtWide;
    Vector3Wide::ConditionalSelect(integrationMask, newPosition, position, position);
    QuaternionWide newOrientation;
    inertia.InverseMass = localInertia.InverseMass;
    BodyVelocityWide previousVelocity = velocity;
    if (integratorCallbacks.AngularIntegrationMode == ConserveMomentum) {
        QuaternionWide previousOrientation = orientation;
        PoseIntegration::Integrate(orientation, velocity.Angular, dtWide * vf(0.5f), newOrientation);
        QuaternionWide::ConditionalSelect(integrationMask, newOrientation, orientation, orientation);
        PoseIntegration::RotateInverseInertia(localInertia.InverseInertiaTensor, orientation, inertia.InverseInertiaTensor);
        PoseIntegration::IntegrateAngularVelocityConserveMomentum(previousOrientation, localInertia.InverseInertiaTensor, inertia.InverseInertiaTensor, velocity.Angular);
    } else if (integratorCallbacks.AngularIntegrationMode == ConserveMomentumWithGyroscopicTorque) {
        PoseIntegration::Integrate(orientation, velocity.Angular, dtWide * vf(0.5f), newOrientation);
        QuaternionWide::ConditionalSelect(integrationMask, newOrientation, orientation, orientation);
        PoseIntegration::RotateInverseInertia(localInertia.InverseInertiaTensor, orientation, inertia.InverseInertiaTensor);
        PoseIntegration::IntegrateAngularVelocityConserveMomentumWithGyroscopicTorque(orientation, localInertia.InverseInertiaTensor, velocity.Angular, dtWide);
    } else {
        PoseIntegration::Integrate(orientation, velocity.Angular, dtWide * vf(0.5f), newOrientation);
        QuaternionWide::ConditionalSelect(integrationMask, newOrientation, orientation, orientation);
        PoseIntegration::RotateInverseInertia(localInertia.InverseInertiaTensor, orientation, inertia.InverseInertiaTensor);
    }
    integratorCallbacks.IntegrateVelocity(bodyIndices, position, orientation, localInertia, integrationMask, workerIndex, vf(dt), velocity);
    Vector3Wide::ConditionalSelect(integrationMask, velocity.Linear, previousVelocity.Linear, velocity.Linear);
    Vector3Wide::ConditionalSelect(integrationMask, velocity.Angular, previousVelocity.Angular, velocity.Angular);
}

template <BatchIntegrationMode TBatchIntegrationMode>
static inline void IntegrateVelocity(PoseIntegratorCallbacks& integratorCallbacks, VI& bodyIndices, const BodyInertiaWide& localInertia, float dt, const VI& integrationMask,
                                     const Vector3Wide& position, const QuaternionWide& orientation, BodyVelocityWide& velocity, int workerIndex, BodyInertiaWide& inertia) {  // :1251
    inertia.InverseMass = localInertia.InverseMass;
    PoseIntegration::RotateInverseInertia(localInertia.InverseInertiaTensor, orientation, inertia.InverseInertiaTensor);
    if (integratorCallbacks.AngularIntegrationMode == ConserveMomentum) {
        QuaternionWide previousOrientation;
        PoseIntegration::Integrate(orientation, velocity.Angular, vf(dt * -0.5f), previousOrientation);
        PoseIntegration::IntegrateAngularVelocityConserveMomentum(previousOrientation, localInertia.InverseInertiaTensor, inertia.InverseInertiaTensor, velocity.Angular);
    } else if (integratorCallbacks.AngularIntegrationMode == ConserveMomentumWithGyroscopicTorque) {
        PoseIntegration::IntegrateAngularVelocityConserveMomentumWithGyroscopicTorque(orientation, localInertia.InverseInertiaTensor, velocity.Angular, vf(dt));
    }
    if constexpr (TBatchIntegrationMode == BatchShouldConditionallyIntegrate) {
        BodyVelocityWide previousVelocity = velocity;
        integratorCallbacks.IntegrateVelocity(bodyIndices, position, orientation, localInertia, integrationMask, workerIndex, vf(dt), velocity);
        Vector3Wide::ConditionalSelect(integrationMask, velocity.Linear, previousVelocity.Linear, velocity.Linear);
        Vector3Wide::ConditionalSelect(integrationMask, velocity.Angular, previousVelocity.Angular, velocity.Angular);
    } else {
        integratorCallbacks.IntegrateVelocity(bodyIndices, position, orientation, localInertia, integrationMask, workerIndex, vf(dt), velocity);
    }
}

static inline VI DecodeBodyIndices(VI encodedBodyIndices, VI integrationMask) {  // :1292
    return (encodedBodyIndices & vi(BodyReferenceMask)) | OnesComplement(integrationMask);
}

template <BatchIntegrationMode TBatchIntegrationMode, bool TShouldIntegratePoses>
static inline void GatherAndIntegrate(Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks, const IndexSet* integrationFlags, int bodyIndexInConstraint, float dt, int workerIndex,
                                      int bundleIndex, VI& encodedBodyIndices, Vector3Wide& position, QuaternionWide& orientation, BodyVelocityWide& velocity, BodyInertiaWide& inertia) {  // :1298
    if constexpr (TShouldIntegratePoses) {
        if constexpr (TBatchIntegrationMode == BatchShouldAlwaysIntegrate) {
            VI integrationMask = (VI)(((__v8su)encodedBodyIndices) < ((__v8su)vi((int)DynamicLimit)));
            BodyInertiaWide localInertia;
            bodies.GatherState(encodedBodyIndices, false, position, orientation, velocity, localInertia);
            VI decodedBodyIndices = DecodeBodyIndices(encodedBodyIndices, integrationMask);
            IntegratePoseAndVelocity(integratorCallbacks, decodedBodyIndices, localInertia, dt, integrationMask, position, orientation, velocity, workerIndex, inertia);
            bodies.ScatterPose(position, orientation, encodedBodyIndices, integrationMask);
            bodies.ScatterInertia(inertia, encodedBodyIndices, integrationMask);
        } else if constexpr (TBatchIntegrationMode == BatchShouldNeverIntegrate) {
            bodies.GatherState(encodedBodyIndices, true, position, orientation, velocity, inertia);
        } else {
            VI integrationMask;
            BundleIntegrationMode bundleIntegrationMode = BundleShouldIntegrate(bundleIndex, integrationFlags[bodyIndexInConstraint], integrationMask);
            BodyInertiaWide gatheredInertia;
            bodies.GatherState(encodedBodyIndices, bundleIntegrationMode == None, position, orientation, velocity, gatheredInertia);
            if (bundleIntegrationMode != None) {
                VI decodedBodyIndices = DecodeBodyIndices(encodedBodyIndices, integrationMask);
                IntegratePoseAndVelocity(integratorCallbacks, decodedBodyIndices, gatheredInertia, dt, integrationMask, position, orientation, velocity, workerIndex, inertia);
                bodies.ScatterPose(position, orientation, encodedBodyIndices, integrationMask);
                bodies.ScatterInertia(inertia, encodedBodyIndices, integrationMask);
            } else {
                inertia = gatheredInertia;
            }
        }
    } else {
        if constexpr (TBatchIntegrationMode == BatchShouldAlwaysIntegrate) {
            VI integrationMask = (VI)(((__v8su)encodedBodyIndices) < ((__v8su)vi((int)DynamicLimit)));
            BodyInertiaWide localInertia;
            bodies.GatherState(encodedBodyIndices, false, position, orientation, velocity, localInertia);
            VI decodedBodyIndices = DecodeBodyIndices(encodedBodyIndices, integrationMask);
            IntegrateVelocity<TBatchIntegrationMode>(integratorCallbacks, decodedBodyIndices, localInertia, dt, integrationMask, position, orientation, velocity, workerIndex, inertia);
            bodies.ScatterInertia(inertia, encodedBodyIndices, integrationMask);
        } else if constexpr (TBatchIntegrationMode == BatchShouldNeverIntegrate) {
            bodies.GatherState(encodedBodyIndices, true, position, orientation, velocity, inertia);
        } else {
            VI integrationMask;
            BundleIntegrationMode bundleIntegrationMode = BundleShouldIntegrate(bundleIndex, integrationFlags[bodyIndexInConstraint], integrationMask);
            BodyInertiaWide gatheredInertia;
            bodies.GatherState(encodedBodyIndices, bundleIntegrationMode == None, position, orientation, velocity, gatheredInertia);
            if (bundleIntegrationMode != None) {
                VI decodedBodyIndices = DecodeBodyIndices(encodedBodyIndices, integrationMask);
                IntegrateVelocity<TBatchIntegrationMode>(integratorCallbacks, decodedBodyIndices, gatheredInertia, dt, integrationMask, position, orientation, velocity, workerIndex, inertia);
                bodies.ScatterInertia(inertia, encodedBodyIndices, integrationMask);
            } else {
                inertia = gatheredInertia;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ type batches and processors
struct TypeBatch {  // Constraints/TypeBatch.cs:10-36
    int TypeId, ConstraintCount, BundleCount;
    VI* BodyReferences;
    VF* PrestepData;
    VF* AccumulatedImpulses;
};

struct TypeProcessor {  // Constraints/TypeProcessor.cs (the part the solve uses)
    int BodiesPerConstraint = 0, PrestepFloats = 0, ImpulseFloats = 0;
    bool RequiresIncrementalSubstepUpdates = false;
    virtual ~TypeProcessor() {}
    virtual void WarmStart(BatchIntegrationMode mode, bool allowPoseIntegration, TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies,
                           PoseIntegratorCallbacks& integratorCallbacks, float dt, float inverseDt, int startBundle, int exclusiveEndBundle, int workerIndex) = 0;
    virtual void Solve(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) = 0;
    virtual void IncrementallyUpdateForSubstep(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) {}
    // DemoBenchmarks/TwoBodyConstraintBenchmarks.cs:19-37 (and OneBodyConstraintBenchmarks.cs): `iterations` x (WarmStart; Solve) on broadcast inputs.
    virtual void Microbenchmark(Bodies& bodies, float* prestepLane, float* accumulatedLane, float dt, int iterations) = 0;
};

template <typename T> static inline void BroadcastLanes(T& wideStruct, const float* lane) {
    VF* fields = reinterpret_cast<VF*>(&wideStruct);
    for (size_t f = 0; f < sizeof(T) / sizeof(VF); ++f) fields[f] = vf(lane[f]);
}
template <typename T> static inline void ReadFirstLanes(const T& wideStruct, float* lane) {
    const VF* fields = reinterpret_cast<const VF*>(&wideStruct);
    for (size_t f = 0; f < sizeof(T) / sizeof(VF); ++f) lane[f] = fields[f][0];
}

struct Filter { bool Linear, Angular; };  // the part of IBodyAccessFilter a scatter depends on (Constraints/IBodyAccessFilter.cs)
constexpr Filter AccessAll{true, true}, AccessNoPose{true, true}, AccessNoPosition{true, true}, AccessOnlyAngular{false, true}, AccessOnlyAngularWithoutPose{false, true};

// Constraints/TwoBodyTypeProcessor.cs:168-241
template <typename TConstraintFunctions, bool WSALinear, bool WSBLinear, bool SALinear, bool SBLinear, bool Incremental>
struct TwoBodyTypeProcessor : TypeProcessor {
    typedef typename TConstraintFunctions::Prestep TPrestepData;
    typedef typename TConstraintFunctions::Impulses TAccumulatedImpulse;
    struct TwoBodyReferences { VI IndexA, IndexB; };  // :13
    TwoBodyTypeProcessor() {
        BodiesPerConstraint = 2;
        PrestepFloats = sizeof(TPrestepData) / sizeof(VF);
        ImpulseFloats = sizeof(TAccumulatedImpulse) / sizeof(VF);
        RequiresIncrementalSubstepUpdates = Incremental;
    }
    template <BatchIntegrationMode TBatchIntegrationMode, bool TAllowPoseIntegration>
    void WarmStartImpl(TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks, float dt, float inverseDt, int startBundle,
                       int exclusiveEndBundle, int workerIndex) {  // :168
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        TwoBodyReferences* bodyReferencesBundles = (TwoBodyReferences*)typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            TPrestepData& prestep = prestepBundles[i];
            TAccumulatedImpulse& accumulatedImpulses = accumulatedImpulsesBundles[i];
            TwoBodyReferences& references = bodyReferencesBundles[i];
            Vector3Wide positionA, positionB;
            QuaternionWide orientationA, orientationB;
            BodyVelocityWide wsvA, wsvB;
            BodyInertiaWide inertiaA, inertiaB;
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 0, dt, workerIndex, i, references.IndexA, positionA, orientationA,
                                                                             wsvA, inertiaA);
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 1, dt, workerIndex, i, references.IndexB, positionB, orientationB,
                                                                             wsvB, inertiaB);
            TConstraintFunctions::WarmStart(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, prestep, accumulatedImpulses, wsvA, wsvB);
            if constexpr (TBatchIntegrationMode == BatchShouldNeverIntegrate) {
                bodies.ScatterVelocities<WSALinear, true>(wsvA, references.IndexA);
                bodies.ScatterVelocities<WSBLinear, true>(wsvB, references.IndexB);
            } else {
                bodies.ScatterVelocities<true, true>(wsvA, references.IndexA);
                bodies.ScatterVelocities<true, true>(wsvB, references.IndexB);
            }
        }
    }
    void WarmStart(BatchIntegrationMode mode, bool allowPoseIntegration, TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks,
                   float dt, float inverseDt, int startBundle, int exclusiveEndBundle, int workerIndex) override {
#define WIDE_DISPATCH(MODE)                                                                                                                             \
    if (allowPoseIntegration) WarmStartImpl<MODE, true>(typeBatch, integrationFlags, bodies, integratorCallbacks, dt, inverseDt, startBundle, exclusiveEndBundle, workerIndex); \
    else WarmStartImpl<MODE, false>(typeBatch, integrationFlags, bodies, integratorCallbacks, dt, inverseDt, startBundle, exclusiveEndBundle, workerIndex);
        if (mode == BatchShouldAlwaysIntegrate) { WIDE_DISPATCH(BatchShouldAlwaysIntegrate) }
        else if (mode == BatchShouldNeverIntegrate) { WIDE_DISPATCH(BatchShouldNeverIntegrate) }
        else { WIDE_DISPATCH(BatchShouldConditionallyIntegrate) }
    }
    void Solve(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) override {  // :205
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        TwoBodyReferences* bodyReferencesBundles = (TwoBodyReferences*)typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            TPrestepData& prestep = prestepBundles[i];
            TAccumulatedImpulse& accumulatedImpulses = accumulatedImpulsesBundles[i];
            TwoBodyReferences& references = bodyReferencesBundles[i];
            Vector3Wide positionA, positionB;
            QuaternionWide orientationA, orientationB;
            BodyVelocityWide wsvA, wsvB;
            BodyInertiaWide inertiaA, inertiaB;
            bodies.GatherState(references.IndexA, true, positionA, orientationA, wsvA, inertiaA);
            bodies.GatherState(references.IndexB, true, positionB, orientationB, wsvB, inertiaB);
            TConstraintFunctions::Solve(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, dt, inverseDt, prestep, accumulatedImpulses, wsvA, wsvB);
            bodies.ScatterVelocities<SALinear, true>(wsvA, references.IndexA);
            bodies.ScatterVelocities<SBLinear, true>(wsvB, references.IndexB);
        }
    }
    void Microbenchmark(Bodies& bodies, float* prestepLane, float* accumulatedLane, float dt, int iterations) override {
        TPrestepData prestep;
        TAccumulatedImpulse accumulatedImpulse;
        BroadcastLanes(prestep, prestepLane);
        BroadcastLanes(accumulatedImpulse, accumulatedLane);
        Vector3Wide positionA, positionB;
        QuaternionWide orientationA, orientationB;
        BodyVelocityWide velocityA, velocityB;
        BodyInertiaWide inertiaA, inertiaB;
        VI indexA = vi(0), indexB = vi(1);
        bodies.GatherState(indexA, true, positionA, orientationA, velocityA, inertiaA);
        bodies.GatherState(indexB, true, positionB, orientationB, velocityB, inertiaB);
        const float inverseDt = 1.0f / dt;
        for (int i = 0; i < iterations; ++i) {
            TConstraintFunctions::WarmStart(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, prestep, accumulatedImpulse, velocityA, velocityB);
            TConstraintFunctions::Solve(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, dt, inverseDt, prestep, accumulatedImpulse, velocityA, velocityB);
        }
        indexA = VI{0, -1, -1, -1, -1, -1, -1, -1};
        indexB = VI{1, -1, -1, -1, -1, -1, -1, -1};
        bodies.ScatterVelocities<true, true>(velocityA, indexA);
        bodies.ScatterVelocities<true, true>(velocityB, indexB);
        ReadFirstLanes(prestep, prestepLane);
        ReadFirstLanes(accumulatedImpulse, accumulatedLane);
    }
    void IncrementallyUpdateForSubstep(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) override {  // :227
        if constexpr (Incremental) {
            TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
            TwoBodyReferences* bodyReferencesBundles = (TwoBodyReferences*)typeBatch.BodyReferences;
            VF dtWide = vf(dt);
            for (int i = startBundle; i < exclusiveEndBundle; ++i) {
                BodyVelocityWide wsvA, wsvB;
                bodies.GatherVelocity(bodyReferencesBundles[i].IndexA, wsvA);
                bodies.GatherVelocity(bodyReferencesBundles[i].IndexB, wsvB);
                TConstraintFunctions::IncrementallyUpdateForSubstep(dtWide, wsvA, wsvB, prestepBundles[i]);
            }
        }
    }
};

// Constraints/OneBodyTypeProcessor.cs:82-146 (contacts: AccessNoPose everywhere, :149-150)
template <typename TConstraintFunctions, bool Incremental = true> struct OneBodyTypeProcessor : TypeProcessor {
    typedef typename TConstraintFunctions::Prestep TPrestepData;
    typedef typename TConstraintFunctions::Impulses TAccumulatedImpulse;
    OneBodyTypeProcessor() {
        BodiesPerConstraint = 1;
        PrestepFloats = sizeof(TPrestepData) / sizeof(VF);
        ImpulseFloats = sizeof(TAccumulatedImpulse) / sizeof(VF);
        RequiresIncrementalSubstepUpdates = Incremental;
    }
    template <BatchIntegrationMode TBatchIntegrationMode, bool TAllowPoseIntegration>
    void WarmStartImpl(TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks, float dt, float inverseDt, int startBundle,
                       int exclusiveEndBundle, int workerIndex) {  // :82
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        VI* bodyReferencesBundles = typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            VI& references = bodyReferencesBundles[i];
            Vector3Wide positionA;
            QuaternionWide orientationA;
            BodyVelocityWide wsvA;
            BodyInertiaWide inertiaA;
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 0, dt, workerIndex, i, references, positionA, orientationA, wsvA,
                                                                             inertiaA);
            TConstraintFunctions::WarmStart(positionA, orientationA, inertiaA, prestepBundles[i], accumulatedImpulsesBundles[i], wsvA);
            bodies.ScatterVelocities<true, true>(wsvA, references);  // AccessNoPose and AccessAll both write linear + angular
        }
    }
    void WarmStart(BatchIntegrationMode mode, bool allowPoseIntegration, TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks,
                   float dt, float inverseDt, int startBundle, int exclusiveEndBundle, int workerIndex) override {
        if (mode == BatchShouldAlwaysIntegrate) { WIDE_DISPATCH(BatchShouldAlwaysIntegrate) }
        else if (mode == BatchShouldNeverIntegrate) { WIDE_DISPATCH(BatchShouldNeverIntegrate) }
        else { WIDE_DISPATCH(BatchShouldConditionallyIntegrate) }
    }
    void Solve(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) override {  // :115
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        VI* bodyReferencesBundles = typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            VI& references = bodyReferencesBundles[i];
            Vector3Wide positionA;
            QuaternionWide orientationA;
            BodyVelocityWide wsvA;
            BodyInertiaWide inertiaA;
            bodies.GatherState(references, true, positionA, orientationA, wsvA, inertiaA);
            TConstraintFunctions::Solve(positionA, orientationA, inertiaA, dt, inverseDt, prestepBundles[i], accumulatedImpulsesBundles[i], wsvA);
            bodies.ScatterVelocities<true, true>(wsvA, references);
        }
    }
    void Microbenchmark(Bodies& bodies, float* prestepLane, float* accumulatedLane, float dt, int iterations) override {
        TPrestepData prestep;
        TAccumulatedImpulse accumulatedImpulse;
        BroadcastLanes(prestep, prestepLane);
        BroadcastLanes(accumulatedImpulse, accumulatedLane);
        Vector3Wide positionA;
        QuaternionWide orientationA;
        BodyVelocityWide velocityA;
        BodyInertiaWide inertiaA;
        VI indexA = vi(0);
        bodies.GatherState(indexA, true, positionA, orientationA, velocityA, inertiaA);
        const float inverseDt = 1.0f / dt;
        for (int i = 0; i < iterations; ++i) {
            TConstraintFunctions::WarmStart(positionA, orientationA, inertiaA, prestep, accumulatedImpulse, velocityA);
            TConstraintFunctions::Solve(positionA, orientationA, inertiaA, dt, inverseDt, prestep, accumulatedImpulse, velocityA);
        }
        indexA = VI{0, -1, -1, -1, -1, -1, -1, -1};
        bodies.ScatterVelocities<true, true>(velocityA, indexA);
        ReadFirstLanes(prestep, prestepLane);
        ReadFirstLanes(accumulatedImpulse, accumulatedLane);
    }
    void IncrementallyUpdateForSubstep(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) override {  // :134
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        VI* bodyReferencesBundles = typeBatch.BodyReferences;
        VF dtWide = vf(dt);
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            BodyVelocityWide wsvA;
            bodies.GatherVelocity(bodyReferencesBundles[i], wsvA);
            TConstraintFunctions::IncrementallyUpdateForSubstep(dtWide, wsvA, prestepBundles[i]);
        }
    }
};

// Constraints/ThreeBodyTypeProcessor.cs:83-179. Both users (AreaConstraint, and VolumeConstraint below) declare AccessOnlyLinear everywhere: the scatter writes the linear half
// only, and since the angular half that was gathered is not touched by either constraint, writing both halves back (what this template does) leaves the same bytes.
template <typename TConstraintFunctions> struct ThreeBodyTypeProcessor : TypeProcessor {
    typedef typename TConstraintFunctions::Prestep TPrestepData;
    typedef typename TConstraintFunctions::Impulses TAccumulatedImpulse;
    struct ThreeBodyReferences { VI IndexA, IndexB, IndexC; };  // :9
    ThreeBodyTypeProcessor() {
        BodiesPerConstraint = 3;
        PrestepFloats = sizeof(TPrestepData) / sizeof(VF);
        ImpulseFloats = sizeof(TAccumulatedImpulse) / sizeof(VF);
        RequiresIncrementalSubstepUpdates = false;  // AreaConstraint.cs:190
    }
    template <BatchIntegrationMode TBatchIntegrationMode, bool TAllowPoseIntegration>
    void WarmStartImpl(TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks, float dt, float inverseDt, int startBundle,
                       int exclusiveEndBundle, int workerIndex) {  // :83
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        ThreeBodyReferences* bodyReferencesBundles = (ThreeBodyReferences*)typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            ThreeBodyReferences& references = bodyReferencesBundles[i];
            Vector3Wide positionA, positionB, positionC;
            QuaternionWide orientationA, orientationB, orientationC;
            BodyVelocityWide wsvA, wsvB, wsvC;
            BodyInertiaWide inertiaA, inertiaB, inertiaC;
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 0, dt, workerIndex, i, references.IndexA, positionA, orientationA,
                                                                             wsvA, inertiaA);
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 1, dt, workerIndex, i, references.IndexB, positionB, orientationB,
                                                                             wsvB, inertiaB);
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 2, dt, workerIndex, i, references.IndexC, positionC, orientationC,
                                                                             wsvC, inertiaC);
            TConstraintFunctions::WarmStart(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, positionC, orientationC, inertiaC, prestepBundles[i],
                                            accumulatedImpulsesBundles[i], wsvA, wsvB, wsvC);
            bodies.ScatterVelocities<true, true>(wsvA, references.IndexA);
            bodies.ScatterVelocities<true, true>(wsvB, references.IndexB);
            bodies.ScatterVelocities<true, true>(wsvC, references.IndexC);
        }
    }
    void WarmStart(BatchIntegrationMode mode, bool allowPoseIntegration, TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks,
                   float dt, float inverseDt, int startBundle, int exclusiveEndBundle, int workerIndex) override {
        if (mode == BatchShouldAlwaysIntegrate) { WIDE_DISPATCH(BatchShouldAlwaysIntegrate) }
        else if (mode == BatchShouldNeverIntegrate) { WIDE_DISPATCH(BatchShouldNeverIntegrate) }
        else { WIDE_DISPATCH(BatchShouldConditionallyIntegrate) }
    }
    void Solve(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) override {  // :128
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        ThreeBodyReferences* bodyReferencesBundles = (ThreeBodyReferences*)typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            ThreeBodyReferences& references = bodyReferencesBundles[i];
            Vector3Wide positionA, positionB, positionC;
            QuaternionWide orientationA, orientationB, orientationC;
            BodyVelocityWide wsvA, wsvB, wsvC;
            BodyInertiaWide inertiaA, inertiaB, inertiaC;
            bodies.GatherState(references.IndexA, true, positionA, orientationA, wsvA, inertiaA);
            bodies.GatherState(references.IndexB, true, positionB, orientationB, wsvB, inertiaB);
            bodies.GatherState(references.IndexC, true, positionC, orientationC, wsvC, inertiaC);
            TConstraintFunctions::Solve(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, positionC, orientationC, inertiaC, dt, inverseDt, prestepBundles[i],
                                        accumulatedImpulsesBundles[i], wsvA, wsvB, wsvC);
            bodies.ScatterVelocities<true, true>(wsvA, references.IndexA);
            bodies.ScatterVelocities<true, true>(wsvB, references.IndexB);
            bodies.ScatterVelocities<true, true>(wsvC, references.IndexC);
        }
    }
    void Microbenchmark(Bodies& bodies, float* prestepLane, float* accumulatedLane, float dt, int iterations) override {
        TPrestepData prestep;
        TAccumulatedImpulse accumulatedImpulse;
        BroadcastLanes(prestep, prestepLane);
        BroadcastLanes(accumulatedImpulse, accumulatedLane);
        Vector3Wide position[3];
        QuaternionWide orientation[3];
        BodyVelocityWide velocity[3];
        BodyInertiaWide inertia[3];
        for (int k = 0; k < 3; ++k) bodies.GatherState(vi(k), true, position[k], orientation[k], velocity[k], inertia[k]);
        const float inverseDt = 1.0f / dt;
        for (int i = 0; i < iterations; ++i) {
            TConstraintFunctions::WarmStart(position[0], orientation[0], inertia[0], position[1], orientation[1], inertia[1], position[2], orientation[2], inertia[2], prestep,
                                            accumulatedImpulse, velocity[0], velocity[1], velocity[2]);
            TConstraintFunctions::Solve(position[0], orientation[0], inertia[0], position[1], orientation[1], inertia[1], position[2], orientation[2], inertia[2], dt, inverseDt, prestep,
                                        accumulatedImpulse, velocity[0], velocity[1], velocity[2]);
        }
        for (int k = 0; k < 3; ++k) {
            VI index = VI{k, -1, -1, -1, -1, -1, -1, -1};
            bodies.ScatterVelocities<true, true>(velocity[k], index);
        }
        ReadFirstLanes(prestep, prestepLane);
        ReadFirstLanes(accumulatedImpulse, accumulatedLane);
    }
};

// Constraints/FourBodyTypeProcessor.cs:104-192
template <typename TConstraintFunctions> struct FourBodyTypeProcessor : TypeProcessor {
    typedef typename TConstraintFunctions::Prestep TPrestepData;
    typedef typename TConstraintFunctions::Impulses TAccumulatedImpulse;
    struct FourBodyReferences { VI IndexA, IndexB, IndexC, IndexD; };  // :12
    FourBodyTypeProcessor() {
        BodiesPerConstraint = 4;
        PrestepFloats = sizeof(TPrestepData) / sizeof(VF);
        ImpulseFloats = sizeof(TAccumulatedImpulse) / sizeof(VF);
        RequiresIncrementalSubstepUpdates = false;  // VolumeConstraint.cs:179
    }
    template <BatchIntegrationMode TBatchIntegrationMode, bool TAllowPoseIntegration>
    void WarmStartImpl(TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks, float dt, float inverseDt, int startBundle,
                       int exclusiveEndBundle, int workerIndex) {  // :104
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        FourBodyReferences* bodyReferencesBundles = (FourBodyReferences*)typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            FourBodyReferences& references = bodyReferencesBundles[i];
            Vector3Wide positionA, positionB, positionC, positionD;
            QuaternionWide orientationA, orientationB, orientationC, orientationD;
            BodyVelocityWide wsvA, wsvB, wsvC, wsvD;
            BodyInertiaWide inertiaA, inertiaB, inertiaC, inertiaD;
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 0, dt, workerIndex, i, references.IndexA, positionA, orientationA,
                                                                             wsvA, inertiaA);
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 1, dt, workerIndex, i, references.IndexB, positionB, orientationB,
                                                                             wsvB, inertiaB);
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 2, dt, workerIndex, i, references.IndexC, positionC, orientationC,
                                                                             wsvC, inertiaC);
            GatherAndIntegrate<TBatchIntegrationMode, TAllowPoseIntegration>(bodies, integratorCallbacks, integrationFlags, 3, dt, workerIndex, i, references.IndexD, positionD, orientationD,
                                                                             wsvD, inertiaD);
            TConstraintFunctions::WarmStart(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, positionC, orientationC, inertiaC, positionD, orientationD, inertiaD,
                                            prestepBundles[i], accumulatedImpulsesBundles[i], wsvA, wsvB, wsvC, wsvD);
            bodies.ScatterVelocities<true, true>(wsvA, references.IndexA);
            bodies.ScatterVelocities<true, true>(wsvB, references.IndexB);
            bodies.ScatterVelocities<true, true>(wsvC, references.IndexC);
            bodies.ScatterVelocities<true, true>(wsvD, references.IndexD);
        }
    }
    void WarmStart(BatchIntegrationMode mode, bool allowPoseIntegration, TypeBatch& typeBatch, const IndexSet* integrationFlags, Bodies& bodies, PoseIntegratorCallbacks& integratorCallbacks,
                   float dt, float inverseDt, int startBundle, int exclusiveEndBundle, int workerIndex) override {
        if (mode == BatchShouldAlwaysIntegrate) { WIDE_DISPATCH(BatchShouldAlwaysIntegrate) }
        else if (mode == BatchShouldNeverIntegrate) { WIDE_DISPATCH(BatchShouldNeverIntegrate) }
        else { WIDE_DISPATCH(BatchShouldConditionallyIntegrate) }
    }
    void Solve(TypeBatch& typeBatch, Bodies& bodies, float dt, float inverseDt, int startBundle, int exclusiveEndBundle) override {  // :150
        TPrestepData* prestepBundles = (TPrestepData*)typeBatch.PrestepData;
        FourBodyReferences* bodyReferencesBundles = (FourBodyReferences*)typeBatch.BodyReferences;
        TAccumulatedImpulse* accumulatedImpulsesBundles = (TAccumulatedImpulse*)typeBatch.AccumulatedImpulses;
        for (int i = startBundle; i < exclusiveEndBundle; ++i) {
            FourBodyReferences& references = bodyReferencesBundles[i];
            Vector3Wide positionA, positionB, positionC, positionD;
            QuaternionWide orientationA, orientationB, orientationC, orientationD;
            BodyVelocityWide wsvA, wsvB, wsvC, wsvD;
            BodyInertiaWide inertiaA, inertiaB, inertiaC, inertiaD;
            bodies.GatherState(references.IndexA, true, positionA, orientationA, wsvA, inertiaA);
            bodies.GatherState(references.IndexB, true, positionB, orientationB, wsvB, inertiaB);
            bodies.GatherState(references.IndexC, true, positionC, orientationC, wsvC, inertiaC);
            bodies.GatherState(references.IndexD, true, positionD, orientationD, wsvD, inertiaD);
            TConstraintFunctions::Solve(positionA, orientationA, inertiaA, positionB, orientationB, inertiaB, positionC, orientationC, inertiaC, positionD, orientationD, inertiaD, dt, inverseDt,
                                        prestepBundles[i], accumulatedImpulsesBundles[i], wsvA, wsvB, wsvC, wsvD);
            bodies.ScatterVelocities<true, true>(wsvA, references.IndexA);
            bodies.ScatterVelocities<true, true>(wsvB, references.IndexB);
            bodies.ScatterVelocities<true, true>(wsvC, references.IndexC);
            bodies.ScatterVelocities<true, true>(wsvD, references.IndexD);
        }
    }
    void Microbenchmark(Bodies& bodies, float* prestepLane, float* accumulatedLane, float dt, int iterations) override {
        TPrestepData prestep;
        TAccumulatedImpulse accumulatedImpulse;
        BroadcastLanes(prestep, prestepLane);
        BroadcastLanes(accumulatedImpulse, accumulatedLane);
        Vector3Wide position[4];
        QuaternionWide orientation[4];
        BodyVelocityWide velocity[4];
        BodyInertiaWide inertia[4];
        for (int k = 0; k < 4; ++k) bodies.GatherState(vi(k), true, position[k], orientation[k], velocity[k], inertia[k]);
        const float inverseDt = 1.0f / dt;
        for (int i = 0; i < iterations; ++i) {
            TConstraintFunctions::WarmStart(position[0], orientation[0], inertia[0], position[1], orientation[1], inertia[1], position[2], orientation[2], inertia[2], position[3],
                                            orientation[3], inertia[3], prestep, accumulatedImpulse, velocity[0], velocity[1], velocity[2], velocity[3]);
            TConstraintFunctions::Solve(position[0], orientation[0], inertia[0], position[1], orientation[1], inertia[1], position[2], orientation[2], inertia[2], position[3], orientation[3],
                                        inertia[3], dt, inverseDt, prestep, accumulatedImpulse, velocity[0], velocity[1], velocity[2], velocity[3]);
        }
        for (int k = 0; k < 4; ++k) {
            VI index = VI{k, -1, -1, -1, -1, -1, -1, -1};
            bodies.ScatterVelocities<true, true>(velocity[k], index);
        }
        ReadFirstLanes(prestep, prestepLane);
        ReadFirstLanes(accumulatedImpulse, accumulatedLane);
    }
};
#undef WIDE_DISPATCH

// Joint function structs have no incremental update; give the two-body template a uniform call.
template <typename F> struct NoIncremental : F {
    static void IncrementallyUpdateForSubstep(const VF&, const BodyVelocityWide&, const BodyVelocityWide&, typename F::Prestep&) {}
};

template <typename F> struct NoIncrementalOneBody : F {
    static void IncrementallyUpdateForSubstep(const VF&, const BodyVelocityWide&, typename F::Prestep&) {}
};

static TypeProcessor* CreateProcessor(int typeId) {  // BepuPhysics/DefaultTypes.cs:20-63 (ids) + the filter lists of each *TypeProcessor class
    switch (typeId) {
        case 0: return new OneBodyTypeProcessor<ContactOneBodyFunctions<1>>();
        case 1: return new OneBodyTypeProcessor<ContactOneBodyFunctions<2>>();
        case 2: return new OneBodyTypeProcessor<ContactOneBodyFunctions<3>>();
        case 3: return new OneBodyTypeProcessor<ContactOneBodyFunctions<4>>();
        case 4: return new TwoBodyTypeProcessor<ContactFunctions<1>, true, true, true, true, true>();  // TwoBodyContactTypeProcessor: AccessNoPose x4 (TwoBodyTypeProcessor.cs:244)
        case 5: return new TwoBodyTypeProcessor<ContactFunctions<2>, true, true, true, true, true>();
        case 6: return new TwoBodyTypeProcessor<ContactFunctions<3>, true, true, true, true, true>();
        case 7: return new TwoBodyTypeProcessor<ContactFunctions<4>, true, true, true, true, true>();
        case 22: return new TwoBodyTypeProcessor<NoIncremental<BallSocketFunctions>, true, true, true, true, false>();        // BallSocket.cs:102 NoPosition, NoPosition, All, All
        case 23: return new TwoBodyTypeProcessor<NoIncremental<AngularHingeFunctions>, false, false, false, false, false>();  // AngularHinge.cs:225 OnlyAngular...
        case 25: return new TwoBodyTypeProcessor<NoIncremental<SwingLimitFunctions>, false, false, false, false, false>();    // SwingLimit.cs:171
        case 26: return new TwoBodyTypeProcessor<NoIncremental<TwistServoFunctions>, false, false, false, false, false>();    // TwistServo.cs:224
        case 27: return new TwoBodyTypeProcessor<NoIncremental<TwistLimitFunctions>, false, false, false, false, false>();    // TwistLimit.cs:139
        case 30: return new TwoBodyTypeProcessor<NoIncremental<AngularMotorFunctions>, false, false, false, false, false>();  // AngularMotor.cs:96
        case 46: return new TwoBodyTypeProcessor<NoIncremental<SwivelHingeFunctions>, true, true, true, true, false>();       // SwivelHinge.cs:216
        case 47: return new TwoBodyTypeProcessor<NoIncremental<HingeFunctions>, true, true, true, true, false>();             // Hinge.cs:224
        // widened set (wide_joints_more.h)
        case 28: return new TwoBodyTypeProcessor<NoIncremental<TwistMotorFunctions>, false, false, false, false, false>();        // TwistMotor.cs:130 OnlyAngular x4
        case 29: return new TwoBodyTypeProcessor<NoIncremental<AngularServoConstraint>, false, false, false, false, false>();      // AngularServo.cs:141 OnlyAngularWithoutPose x2, OnlyAngular x2
        case 41: return new TwoBodyTypeProcessor<NoIncremental<AngularAxisMotorFunctions>, false, false, false, false, false>();  // AngularAxisMotor.cs:109
        case 52: return new TwoBodyTypeProcessor<NoIncremental<BallSocketMotorFunctions>, true, true, true, true, false>();       // BallSocketMotor.cs:99 NoOrientation, All, All, All
        case 53: return new TwoBodyTypeProcessor<NoIncremental<BallSocketServoFunctions>, true, true, true, true, false>();       // BallSocketServo.cs:109 NoPosition x2, All x2
        case 24: return new TwoBodyTypeProcessor<NoIncremental<AngularSwivelHingeFunctions>, false, false, false, false, false>();    // AngularSwivelHinge.cs:151 OnlyAngular x4
        case 54: return new TwoBodyTypeProcessor<NoIncremental<AngularAxisGearMotorFunctions>, false, false, false, false, false>();  // AngularAxisGearMotor.cs:117
        case 35: return new TwoBodyTypeProcessor<NoIncremental<CenterDistanceConstraintFunctions>, true, true, true, true, false>();  // CenterDistanceConstraint.cs:135 OnlyLinear x4 (the angular halves go back unchanged)
        case 33: return new TwoBodyTypeProcessor<NoIncremental<DistanceServoFunctions>, true, true, true, true, false>();    // DistanceServo.cs:232 All x4
        case 34: return new TwoBodyTypeProcessor<NoIncremental<DistanceLimitFunctions>, true, true, true, true, false>();    // DistanceLimit.cs:184
        case 38: return new TwoBodyTypeProcessor<NoIncremental<LinearAxisServoFunctions>, true, true, true, true, false>();  // LinearAxisServo.cs:250
        case 39: return new TwoBodyTypeProcessor<NoIncremental<LinearAxisMotorFunctions>, true, true, true, true, false>();  // LinearAxisMotor.cs:113
        case 40: return new TwoBodyTypeProcessor<NoIncremental<LinearAxisLimitFunctions>, true, true, true, true, false>();  // LinearAxisLimit.cs:156
        case 42: return new OneBodyTypeProcessor<NoIncrementalOneBody<OneBodyAngularServoFunctions>, false>();  // OneBodyAngularServo.cs:111 OnlyAngular x2 (the linear halves go back unchanged)
        case 43: return new OneBodyTypeProcessor<NoIncrementalOneBody<OneBodyAngularMotorFunctions>, false>();  // OneBodyAngularMotor.cs:95
        case 44: return new OneBodyTypeProcessor<NoIncrementalOneBody<OneBodyLinearServoFunctions>, false>();   // OneBodyLinearServo.cs:148 All x2
        case 45: return new OneBodyTypeProcessor<NoIncrementalOneBody<OneBodyLinearMotorFunctions>, false>();   // OneBodyLinearMotor.cs:102 NoPosition x2
        case 31: return new TwoBodyTypeProcessor<NoIncremental<WeldFunctions>, true, true, true, true, false>();  // Weld.cs:222 NoPosition, NoPose, All, All
        case 37: return new TwoBodyTypeProcessor<NoIncremental<PointOnLineServoFunctions>, true, true, true, true, false>();  // PointOnLineServo.cs:195 All x4
        case 8: return new OneBodyTypeProcessor<ContactNonconvexOneBodyFunctions<2>>();  // ContactNonconvexTypes.cs:187 (OneBodyContactTypeProcessor: AccessNoPose x2)
        case 9: return new OneBodyTypeProcessor<ContactNonconvexOneBodyFunctions<3>>();
        case 10: return new OneBodyTypeProcessor<ContactNonconvexOneBodyFunctions<4>>();
        case 15: return new TwoBodyTypeProcessor<ContactNonconvexTwoBodyFunctions<2>, true, true, true, true, true>();  // ContactNonconvexTypes.cs:104 (TwoBodyContactTypeProcessor: AccessNoPose x4)
        case 16: return new TwoBodyTypeProcessor<ContactNonconvexTwoBodyFunctions<3>, true, true, true, true, true>();
        case 17: return new TwoBodyTypeProcessor<ContactNonconvexTwoBodyFunctions<4>, true, true, true, true, true>();
        case 36: return new ThreeBodyTypeProcessor<AreaConstraintFunctions>();   // AreaConstraint.cs:199 OnlyLinear x6
        case 32: return new FourBodyTypeProcessor<VolumeConstraintFunctions>();  // VolumeConstraint.cs:188 OnlyLinear x8
        case 55: return new TwoBodyTypeProcessor<NoIncremental<CenterDistanceLimitFunctions>, true, true, true, true, false>();       // CenterDistanceLimit.cs:134
        default: return nullptr;
    }
}

// ------------------------------------------------------------------------------------------------------------------ a minimal IThreadDispatcher
// BepuUtilities/ThreadDispatcher.cs: DispatchWorkers(body, maximumWorkerCount) runs body(workerIndex) on every worker (the caller is worker 0) and returns when all are done.
// Where the workers run (the cpu_baseline leg of bench.py; VERDICT r3 weak #11: unpinned workers on a shared two-socket host scaled NEGATIVELY beyond 16 threads).
// PinPlan lists the CPUs the process may use, ordered so that the first N entries are the best place for N workers: all on ONE socket (the one with the most
// allowed CPUs), one hardware thread per physical core first, sibling threads after, other sockets last. Worker i is pinned to entry i (WIDE_PIN=0: no pinning).
struct PinPlan {
    std::vector<int> cpus;
    int firstSocketCores = 0, firstSocketCpus = 0;
    static int ReadInt(const std::string& path, int fallback) {
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return fallback;
        int v = fallback;
        if (fscanf(f, "%d", &v) != 1) v = fallback;
        fclose(f);
        return v;
    }
    static const PinPlan& Get() {
        static const PinPlan plan = [] {
            PinPlan p;
            cpu_set_t allowed;
            CPU_ZERO(&allowed);
            if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return p;
            struct Cpu { int id, package, core; };
            std::vector<Cpu> all;
            for (int c = 0; c < CPU_SETSIZE; ++c)
                if (CPU_ISSET(c, &allowed)) {
                    const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(c) + "/topology/";
                    all.push_back({c, ReadInt(base + "physical_package_id", 0), ReadInt(base + "core_id", c)});
                }
            std::map<int, int> perPackage;
            for (auto& c : all) perPackage[c.package]++;
            std::vector<int> packages;
            for (auto& kv : perPackage) packages.push_back(kv.first);
            std::stable_sort(packages.begin(), packages.end(), [&](int a, int b) { return perPackage[a] > perPackage[b]; });
            for (size_t pi = 0; pi < packages.size(); ++pi) {
                std::set<int> coresSeen;
                std::vector<int> first, siblings;
                for (auto& c : all)
                    if (c.package == packages[pi]) (coresSeen.insert(c.core).second ? first : siblings).push_back(c.id);
                p.cpus.insert(p.cpus.end(), first.begin(), first.end());
                p.cpus.insert(p.cpus.end(), siblings.begin(), siblings.end());
                if (pi == 0) { p.firstSocketCores = (int)first.size(); p.firstSocketCpus = (int)(first.size() + siblings.size()); }
            }
            return p;
        }();
        return plan;
    }
    static bool Enabled() { const char* v = getenv("WIDE_PIN"); return !(v && *v == '0'); }
    static void PinCurrentThread(int worker) {
        const PinPlan& plan = Get();
        if (!Enabled() || plan.cpus.empty()) return;
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(plan.cpus[(size_t)worker % plan.cpus.size()], &one);
        sched_setaffinity(0, sizeof(one), &one);
    }
};
// The calling thread is worker 0: pinned for the duration of a call, its own affinity restored afterwards.
struct ScopedMainPin {
    cpu_set_t saved;
    bool active = false;
    explicit ScopedMainPin(bool wanted) {
        if (!wanted || !PinPlan::Enabled() || PinPlan::Get().cpus.empty()) return;
        active = sched_getaffinity(0, sizeof(saved), &saved) == 0;
        if (active) PinPlan::PinCurrentThread(0);
    }
    ~ScopedMainPin() { if (active) sched_setaffinity(0, sizeof(saved), &saved); }
};

class ThreadDispatcher {
  public:
    explicit ThreadDispatcher(int threadCount) : threadCount_(threadCount) {
        for (int i = 1; i < threadCount; ++i) threads_.emplace_back([this, i] { PinPlan::PinCurrentThread(i); Loop(i); });
    }
    ~ThreadDispatcher() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int ThreadCount() const { return threadCount_; }
    void DispatchWorkers(const std::function<void(int)>& body) {
        if (threadCount_ == 1) {
            body(0);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(m_);
            body_ = &body;
            remaining_.store(threadCount_ - 1, std::memory_order_relaxed);
            ++generation_;
        }
        cv_.notify_all();
        body(0);
        while (remaining_.load(std::memory_order_acquire) != 0) _mm_pause();
    }

  private:
    void Loop(int workerIndex) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* body;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                body = body_;
            }
            (*body)(workerIndex);
            remaining_.fetch_sub(1, std::memory_order_release);
        }
    }
    int threadCount_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(int)>* body_ = nullptr;
    std::atomic<int> remaining_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// BepuPhysics/LocalSpinWait.cs:20-48: workers spin 1, 2, 4 pauses, then yield their time slice on every further poll ("being pretty aggressive about yielding
// produces the best results"); the main thread never yields (Solver_Solve.cs:386-398, Thread.SpinWait(3)).
struct LocalSpinWait {
    int WaitCount = 0;
    static constexpr int YieldThreshold = 3;
    void SpinOnce() {
        if (WaitCount >= YieldThreshold) {
            std::this_thread::yield();
        } else {
            for (int i = 0; i < (1 << WaitCount); ++i) _mm_pause();
            ++WaitCount;
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------ the solver (Solver_Solve.cs)
struct ConstraintBatch { std::vector<TypeBatch> TypeBatches; };

struct WorkBlock { int BatchIndex, TypeBatchIndex, StartBundle, End; };  // Solver_Solve.cs:22-29
struct IntegrationWorkBlock { int StartBundleIndex, EndBundleIndex; };
enum SolverStageType { IncrementalUpdate, IntegrateConstrainedKinematics, WarmStartStage, SolveStage };
struct SolverSyncStage { std::atomic<int>* Claims; int ClaimCount; int WorkBlockStartIndex; SolverStageType StageType; int BatchIndex; };

struct Solver {
    Bodies bodies;
    PoseIntegratorCallbacks Callbacks;
    const int32_t* IndexToHandle;
    const int32_t* HandleToIndex;
    int HandleCapacity;
    std::vector<ConstraintBatch> Batches;
    std::vector<std::unique_ptr<TypeProcessor>> TypeProcessors;  // indexed by type id
    std::vector<int> ConstrainedKinematicHandles;
    std::vector<IndexSet> batchReferencedHandles;
    int substepCount, VelocityIterationCount;
    std::vector<int> velocityIterations;  // the VelocityIterationScheduler's answers (Solver_Solve.cs:743-751)
    int FallbackBatchThreshold = 64;  // SolveDescription.cs:38; Batches[FallbackBatchThreshold], when it exists, is the sequential fallback batch (Solver.cs:1878-1884)

    // ---- integration responsibilities (:951-1388)
    std::vector<std::vector<std::vector<IndexSet>>> integrationFlags;  // [batch][typeBatch][bodyIndexInConstraint]
    std::vector<std::vector<uint8_t>> coarseBatchIntegrationResponsibilities;
    std::vector<IndexSet> bodiesFirstObservedInBatches;
    IndexSet mergedConstrainedBodyHandles;

    // activeSet.Constraints[bodyIndex] scanned for the earliest slot in the fallback batch (:1003-1013), precomputed once per solve: body index -> (type batch << 32 | index)
    std::unordered_map<int, uint64_t> fallbackEarliestSlot;
    template <bool IsFallbackBatch>
    bool ComputeIntegrationResponsibilitiesForConstraintRegion(int batchIndex, int typeBatchIndex, int constraintStart, int exclusiveConstraintEnd) {  // :951
        IndexSet& firstObservedForBatch = bodiesFirstObservedInBatches[batchIndex];
        std::vector<IndexSet>& integrationFlagsForTypeBatch = integrationFlags[batchIndex][typeBatchIndex];
        TypeBatch& typeBatch = Batches[batchIndex].TypeBatches[typeBatchIndex];
        const int32_t* typeBatchBodyReferences = (const int32_t*)typeBatch.BodyReferences;
        int bodiesPerConstraintInTypeBatch = TypeProcessors[typeBatch.TypeId]->BodiesPerConstraint;
        int intsPerBundle = W * bodiesPerConstraintInTypeBatch;
        int bundleStartIndex = constraintStart / W;
        int bundleEndIndex = (exclusiveConstraintEnd + W - 1) / W;
        for (int bundleIndex = bundleStartIndex; bundleIndex < bundleEndIndex; ++bundleIndex) {
            int bundleStartIndexInConstraints = bundleIndex * W;
            int countInBundle = std::min(W, typeBatch.ConstraintCount - bundleStartIndexInConstraints);
            const int32_t* bundleBodyReferencesStart = typeBatchBodyReferences + bundleIndex * intsPerBundle;
            for (int bodyIndexInConstraint = 0; bodyIndexInConstraint < bodiesPerConstraintInTypeBatch; ++bodyIndexInConstraint) {
                IndexSet& integrationFlagsForBodyInConstraint = integrationFlagsForTypeBatch[bodyIndexInConstraint];
                const int32_t* bundleStart = bundleBodyReferencesStart + bodyIndexInConstraint * W;
                for (int bundleInnerIndex = 0; bundleInnerIndex < countInBundle; ++bundleInnerIndex) {
                    int bodyIndex;
                    if constexpr (IsFallbackBatch) {
                        int rawBodyIndex = bundleStart[bundleInnerIndex];
                        if (rawBodyIndex == -1) continue;  // fallback bundles can hold empty lanes anywhere (:983-986)
                        bodyIndex = rawBodyIndex & BodyReferenceMask;
                    } else {
                        bodyIndex = bundleStart[bundleInnerIndex] & BodyReferenceMask;
                    }
                    int bodyHandle = IndexToHandle[bodyIndex];
                    if (firstObservedForBatch.Contains(bodyHandle)) {
                        if constexpr (IsFallbackBatch) {
                            // the body may appear in several constraints of this batch: the earliest slot integrates it (:1003-1019)
                            int indexInTypeBatch = bundleStartIndexInConstraints + bundleInnerIndex;
                            uint64_t currentSlot = ((uint64_t)typeBatchIndex << 32) | (uint32_t)indexInTypeBatch;
                            if (currentSlot == fallbackEarliestSlot[bodyIndex]) integrationFlagsForBodyInConstraint.AddUnsafely(indexInTypeBatch);
                        } else {
                            integrationFlagsForBodyInConstraint.AddUnsafely(bundleStartIndexInConstraints + bundleInnerIndex);
                        }
                    }
                }
            }
        }
        int flagBundleCount = IndexSet::GetBundleCapacity(typeBatch.ConstraintCount);
        uint64_t mergedFlagBundles = 0;
        for (int bodyIndexInConstraint = 0; bodyIndexInConstraint < bodiesPerConstraintInTypeBatch; ++bodyIndexInConstraint)
            for (int i = 0; i < flagBundleCount; ++i) mergedFlagBundles |= integrationFlagsForTypeBatch[bodyIndexInConstraint].Flags[i];
        return mergedFlagBundles != 0;
    }

    void PrepareConstraintIntegrationResponsibilities(ThreadDispatcher* threadDispatcher) {  // :1072
        int batchCount = (int)Batches.size();
        int mergedWords = (HandleCapacity - 1 + 64) / 64;  // (HighestPossiblyClaimedId + 64) / 64
        if (mergedWords < 1) mergedWords = 1;
        mergedConstrainedBodyHandles.Flags.assign(mergedWords, 0);
        if (batchCount == 0) return;
        integrationFlags.assign(batchCount, {});
        coarseBatchIntegrationResponsibilities.assign(batchCount, {});
        for (int batchIndex = 1; batchIndex < batchCount; ++batchIndex) {
            ConstraintBatch& batch = Batches[batchIndex];
            integrationFlags[batchIndex].resize(batch.TypeBatches.size());
            coarseBatchIntegrationResponsibilities[batchIndex].assign(batch.TypeBatches.size(), 0);
            for (size_t typeBatchIndex = 0; typeBatchIndex < batch.TypeBatches.size(); ++typeBatchIndex) {
                TypeBatch& typeBatch = batch.TypeBatches[typeBatchIndex];
                int bodiesPerConstraint = TypeProcessors[typeBatch.TypeId]->BodiesPerConstraint;
                auto& flagsForTypeBatch = integrationFlags[batchIndex][typeBatchIndex];
                flagsForTypeBatch.resize(bodiesPerConstraint);
                for (int b = 0; b < bodiesPerConstraint; ++b) flagsForTypeBatch[b].Flags.assign(IndexSet::GetBundleCapacity(typeBatch.ConstraintCount) + 1, 0);
            }
        }
        bodiesFirstObservedInBatches.assign(batchCount, {});
        {
            size_t copyLength = std::min(mergedConstrainedBodyHandles.Flags.size(), batchReferencedHandles[0].Flags.size());
            for (size_t i = 0; i < copyLength; ++i) mergedConstrainedBodyHandles.Flags[i] = batchReferencedHandles[0].Flags[i];
        }
        std::vector<uint8_t> batchHasAnyIntegrationResponsibilities(batchCount, 0);
        for (int batchIndex = 1; batchIndex < batchCount; ++batchIndex) {  // :1150-1209 (scalar form of the merge)
            IndexSet& batchHandles = batchReferencedHandles[batchIndex];
            IndexSet& firstObservedInBatch = bodiesFirstObservedInBatches[batchIndex];
            int flagBundleCount = (int)std::min(mergedConstrainedBodyHandles.Flags.size(), batchHandles.Flags.size());
            firstObservedInBatch.Flags.assign(flagBundleCount, 0);
            uint64_t horizontalMerge = 0;
            for (int flagBundleIndex = 0; flagBundleIndex < flagBundleCount; ++flagBundleIndex) {
                uint64_t mergeBundle = mergedConstrainedBodyHandles.Flags[flagBundleIndex];
                uint64_t batchBundle = batchHandles.Flags[flagBundleIndex];
                mergedConstrainedBodyHandles.Flags[flagBundleIndex] = mergeBundle | batchBundle;
                uint64_t firstObservedBundle = ~mergeBundle & batchBundle;
                horizontalMerge |= firstObservedBundle;
                firstObservedInBatch.Flags[flagBundleIndex] = firstObservedBundle;
            }
            batchHasAnyIntegrationResponsibilities[batchIndex] = horizontalMerge != 0;
        }
        const int synchronizedBatchCount = std::min(batchCount, FallbackBatchThreshold);  // GetSynchronizedBatchCount, Solver.cs:1878
        const bool fallbackExists = batchCount > FallbackBatchThreshold;
        fallbackEarliestSlot.clear();
        if (fallbackExists) {
            ConstraintBatch& batch = Batches[FallbackBatchThreshold];
            for (int j = 0; j < (int)batch.TypeBatches.size(); ++j) {
                TypeBatch& typeBatch = batch.TypeBatches[j];
                int bodiesPerConstraint = TypeProcessors[typeBatch.TypeId]->BodiesPerConstraint;
                const int32_t* refs = (const int32_t*)typeBatch.BodyReferences;
                for (int i = 0; i < typeBatch.ConstraintCount; ++i)
                    for (int k = 0; k < bodiesPerConstraint; ++k) {
                        int32_t raw = refs[(i >> 3) * bodiesPerConstraint * W + k * W + (i & 7)];
                        if (raw == -1) continue;
                        uint64_t slot = ((uint64_t)j << 32) | (uint32_t)i;
                        auto it = fallbackEarliestSlot.find(raw & BodyReferenceMask);
                        if (it == fallbackEarliestSlot.end() || slot < it->second) fallbackEarliestSlot[raw & BodyReferenceMask] = slot;
                    }
            }
        }
        bool useSingleThreadedPath = true;
        if (threadDispatcher != nullptr && threadDispatcher->ThreadCount() > 1) {  // :1218-1271
            struct Job { int batch, typeBatch, start, end; };
            std::vector<Job> jobs;
            int constraintCount = 0;
            const int targetJobSize = 2048;
            for (int batchIndex = 1; batchIndex < batchCount; ++batchIndex) {
                if (!batchHasAnyIntegrationResponsibilities[batchIndex]) continue;
                ConstraintBatch& batch = Batches[batchIndex];
                for (int typeBatchIndex = 0; typeBatchIndex < (int)batch.TypeBatches.size(); ++typeBatchIndex) {
                    TypeBatch& typeBatch = batch.TypeBatches[typeBatchIndex];
                    constraintCount += typeBatch.ConstraintCount;
                    int jobCountForTypeBatch = (typeBatch.ConstraintCount + targetJobSize - 1) / targetJobSize;
                    for (int i = 0; i < jobCountForTypeBatch; ++i) {
                        int jobStart = i * targetJobSize;
                        int jobEnd = std::min(jobStart + targetJobSize, typeBatch.ConstraintCount);
                        jobs.push_back({batchIndex, typeBatchIndex, jobStart, jobEnd});
                    }
                }
            }
            if (constraintCount > 4096 + threadDispatcher->ThreadCount() * 1024) {
                useSingleThreadedPath = false;
                std::vector<uint8_t> jobAlignedIntegrationResponsibilities(jobs.size(), 0);
                std::atomic<int> nextJob{0};
                threadDispatcher->DispatchWorkers([&](int) {
                    int jobIndex;
                    while ((jobIndex = nextJob.fetch_add(1)) < (int)jobs.size()) {
                        const Job& job = jobs[jobIndex];
                        jobAlignedIntegrationResponsibilities[jobIndex] = job.batch == FallbackBatchThreshold
                                                                              ? ComputeIntegrationResponsibilitiesForConstraintRegion<true>(job.batch, job.typeBatch, job.start, job.end)
                                                                              : ComputeIntegrationResponsibilitiesForConstraintRegion<false>(job.batch, job.typeBatch, job.start, job.end);
                    }
                });
                for (size_t i = 0; i < jobs.size(); ++i) coarseBatchIntegrationResponsibilities[jobs[i].batch][jobs[i].typeBatch] |= jobAlignedIntegrationResponsibilities[i];
            }
        }
        if (useSingleThreadedPath) {
            for (int i = 1; i < synchronizedBatchCount; ++i) {
                if (!batchHasAnyIntegrationResponsibilities[i]) continue;
                ConstraintBatch& batch = Batches[i];
                for (int j = 0; j < (int)batch.TypeBatches.size(); ++j)
                    coarseBatchIntegrationResponsibilities[i][j] = ComputeIntegrationResponsibilitiesForConstraintRegion<false>(i, j, 0, batch.TypeBatches[j].ConstraintCount);
            }
            if (fallbackExists && batchHasAnyIntegrationResponsibilities[FallbackBatchThreshold]) {  // :1285-1293
                ConstraintBatch& batch = Batches[FallbackBatchThreshold];
                for (int j = 0; j < (int)batch.TypeBatches.size(); ++j)
                    coarseBatchIntegrationResponsibilities[FallbackBatchThreshold][j] =
                        ComputeIntegrationResponsibilitiesForConstraintRegion<true>(FallbackBatchThreshold, j, 0, batch.TypeBatches[j].ConstraintCount);
            }
        }
        for (int handle : ConstrainedKinematicHandles) mergedConstrainedBodyHandles.AddUnsafely(handle);  // :1378
    }

    // ---- stage bodies (:185-295)
    void WarmStartBlock(bool allowPoseIntegration, int workerIndex, int batchIndex, int typeBatchIndex, int startBundle, int endBundle, TypeBatch& typeBatch, TypeProcessor* typeProcessor,
                        float dt, float inverseDt) {  // :185
        if (batchIndex == 0) {
            typeProcessor->WarmStart(BatchShouldAlwaysIntegrate, allowPoseIntegration, typeBatch, nullptr, bodies, Callbacks, dt, inverseDt, startBundle, endBundle, workerIndex);
        } else if (coarseBatchIntegrationResponsibilities[batchIndex][typeBatchIndex]) {
            typeProcessor->WarmStart(BatchShouldConditionallyIntegrate, allowPoseIntegration, typeBatch, integrationFlags[batchIndex][typeBatchIndex].data(), bodies, Callbacks, dt, inverseDt,
                                     startBundle, endBundle, workerIndex);
        } else {
            typeProcessor->WarmStart(BatchShouldNeverIntegrate, allowPoseIntegration, typeBatch, integrationFlags[batchIndex][typeBatchIndex].data(), bodies, Callbacks, dt, inverseDt,
                                     startBundle, endBundle, workerIndex);
        }
    }

    // PoseIntegrator.cs:451-535
    void IntegrateKinematicVelocities(int bundleStartIndex, int bundleEndIndex, float substepDt, int workerIndex) {  // :451
        int bodyCount = (int)ConstrainedKinematicHandles.size();
        VF bundleDt = vf(substepDt);
        BodyInertiaWide zeroInertia;
        std::memset(&zeroInertia, 0, sizeof(zeroInertia));
        for (int bundleIndex = bundleStartIndex; bundleIndex < bundleEndIndex; ++bundleIndex) {
            int bundleBaseIndex = bundleIndex * W;
            int countInBundle = std::min(bodyCount - bundleBaseIndex, W);
            VI bodyIndices = vi(0);
            for (int i = 0; i < countInBundle; ++i) bodyIndices[i] = HandleToIndex[ConstrainedKinematicHandles[bundleBaseIndex + i]];
            VI existingMask = CreateMaskForCountInBundle(countInBundle);
            VI trailingMask = OnesComplement(existingMask);
            VI bodyIndicesVector = BitwiseOr(trailingMask, bodyIndices);
            Vector3Wide position;
            QuaternionWide orientation;
            BodyVelocityWide velocity;
            BodyInertiaWide unused;
            bodies.GatherState(bodyIndicesVector, false, position, orientation, velocity, unused);
            Callbacks.IntegrateVelocity(bodyIndicesVector, position, orientation, zeroInertia, existingMask, workerIndex, bundleDt, velocity);
            bodies.ScatterVelocities<true, true>(velocity, bodyIndicesVector);
        }
    }
    void IntegrateKinematicPosesAndVelocities(int bundleStartIndex, int bundleEndIndex, float substepDt, int workerIndex) {  // :493
        int bodyCount = (int)ConstrainedKinematicHandles.size();
        VF bundleDt = vf(substepDt);
        VF halfDt = bundleDt * vf(0.5f);
        BodyInertiaWide zeroInertia;
        std::memset(&zeroInertia, 0, sizeof(zeroInertia));
        for (int bundleIndex = bundleStartIndex; bundleIndex < bundleEndIndex; ++bundleIndex) {
            int bundleBaseIndex = bundleIndex * W;
            int countInBundle = std::min(bodyCount - bundleBaseIndex, W);
            VI bodyIndices = vi(0);
            for (int i = 0; i < countInBundle; ++i) bodyIndices[i] = HandleToIndex[ConstrainedKinematicHandles[bundleBaseIndex + i]];
            VI existingMask = CreateMaskForCountInBundle(countInBundle);
            VI trailingMask = OnesComplement(existingMask);
            VI bodyIndicesVector = BitwiseOr(trailingMask, bodyIndices);
            Vector3Wide position;
            QuaternionWide orientation;
            BodyVelocityWide velocity;
            BodyInertiaWide unused;
            bodies.GatherState(bodyIndicesVector, false, position, orientation, velocity, unused);
            position = position + velocity.Linear * bundleDt;
            PoseIntegration::Integrate(orientation, velocity.Angular, halfDt, orientation);
            bodies.ScatterPose(position, orientation, bodyIndicesVector, existingMask);
            if (Callbacks.IntegrateVelocityForKinematics) {
                Callbacks.IntegrateVelocity(bodyIndicesVector, position, orientation, zeroInertia, existingMask, workerIndex, bundleDt, velocity);
                bodies.ScatterVelocities<true, true>(velocity, bodyIndicesVector);
            }
        }
    }
    static VI CreateMaskForCountInBundle(int countInBundle) {  // BepuUtilities/BundleIndexing.cs:88
        return (VI)(vf((float)countInBundle) > VF{0.f, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f});
    }
    static VI CreateTrailingMaskForCountInBundle(int countInBundle) {  // :63
        return (VI)(vf((float)countInBundle) <= VF{0.f, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f});
    }
    static int GetBundleCount(int elementCount) { return (elementCount + W - 1) / W; }

    int GetVelocityIterationCountForSubstepIndex(int substepIndex) const { return velocityIterations[substepIndex]; }  // :743

    // ---- single-threaded substep loop (:1415-1479)
    void SolveSingleThreaded(float totalDt) {
        float substepDt = totalDt / substepCount;
        Callbacks.PrepareForIntegration(substepDt);
        float inverseDt = 1.0f / substepDt;
        int batchCount = (int)Batches.size();
        int kinematicBundles = GetBundleCount((int)ConstrainedKinematicHandles.size());
        for (int substepIndex = 0; substepIndex < substepCount; ++substepIndex) {
            if (substepIndex > 0) {
                for (int i = 0; i < batchCount; ++i) {
                    ConstraintBatch& batch = Batches[i];
                    for (size_t j = 0; j < batch.TypeBatches.size(); ++j) {
                        TypeBatch& typeBatch = batch.TypeBatches[j];
                        TypeProcessor* processor = TypeProcessors[typeBatch.TypeId].get();
                        if (processor->RequiresIncrementalSubstepUpdates) processor->IncrementallyUpdateForSubstep(typeBatch, bodies, substepDt, inverseDt, 0, typeBatch.BundleCount);
                    }
                }
                IntegrateKinematicPosesAndVelocities(0, kinematicBundles, substepDt, 0);
            } else {
                if (Callbacks.IntegrateVelocityForKinematics) IntegrateKinematicVelocities(0, kinematicBundles, substepDt, 0);
            }
            for (int i = 0; i < batchCount; ++i) {
                ConstraintBatch& batch = Batches[i];
                for (size_t j = 0; j < batch.TypeBatches.size(); ++j) {
                    TypeBatch& typeBatch = batch.TypeBatches[j];
                    WarmStartBlock(substepIndex != 0, 0, i, (int)j, 0, typeBatch.BundleCount, typeBatch, TypeProcessors[typeBatch.TypeId].get(), substepDt, inverseDt);
                }
            }
            int velocityIterationCount = GetVelocityIterationCountForSubstepIndex(substepIndex);
            for (int iterationIndex = 0; iterationIndex < velocityIterationCount; ++iterationIndex) {
                for (int i = 0; i < batchCount; ++i) {
                    ConstraintBatch& batch = Batches[i];
                    for (size_t j = 0; j < batch.TypeBatches.size(); ++j) {
                        TypeBatch& typeBatch = batch.TypeBatches[j];
                        TypeProcessors[typeBatch.TypeId]->Solve(typeBatch, bodies, substepDt, inverseDt, 0, typeBatch.BundleCount);
                    }
                }
            }
        }
    }

    // ---- multithreaded substep loop (:297-946)
    struct SubstepMultithreadingContext {
        std::vector<SolverSyncStage> Stages;
        std::vector<WorkBlock> IncrementalUpdateBlocks, ConstraintBlocks;
        std::vector<IntegrationWorkBlock> KinematicIntegrationBlocks;
        std::vector<int> ConstraintBatchBoundaries;
        float Dt, InverseDt;
        int WorkerCount;
        alignas(64) std::atomic<int> SyncIndex;
        alignas(64) std::atomic<int> CompletedWorkBlockCount;
        std::vector<int> VelocityIterationCounts;
        int HighestVelocityIterationCount;
    } substepContext;

    // Not in the reference: time-stamp-counter ticks every worker spent INSIDE work blocks (one slot per cache line). bench.py's cpu_baseline reports the busy fraction
    // (sum / (threads x wall)) and the inflation of the work itself against one thread next to the thread curve: what the sync stages cost and what memory contention costs.
    std::vector<uint64_t> workerWorkTicks;
    template <typename TStageFunction>
    void ExecuteWorkerStage(TStageFunction& stageFunction, int workerIndex, int workerStart, int availableBlocksStartIndex, std::atomic<int>* claims, int claimCount, int previousSyncIndex,
                            int syncIndex, std::atomic<int>& completedWorkBlocks) {  // :297
        if (workerStart == -1) return;
        int workBlockIndex = workerStart;
        int locallyCompletedCount = 0;
        uint64_t ticks = 0;
        for (;;) {
            int expected = previousSyncIndex;
            if (!claims[workBlockIndex].compare_exchange_strong(expected, syncIndex)) break;
            uint64_t t0 = __rdtsc();
            stageFunction(availableBlocksStartIndex + workBlockIndex, workerIndex);
            ticks += __rdtsc() - t0;
            ++locallyCompletedCount;
            ++workBlockIndex;
            if (workBlockIndex >= claimCount) workBlockIndex = 0;
        }
        workBlockIndex = workerStart - 1;
        for (;;) {
            if (workBlockIndex < 0) workBlockIndex = claimCount - 1;
            int expected = previousSyncIndex;
            if (!claims[workBlockIndex].compare_exchange_strong(expected, syncIndex)) break;
            uint64_t t0 = __rdtsc();
            stageFunction(availableBlocksStartIndex + workBlockIndex, workerIndex);
            ticks += __rdtsc() - t0;
            ++locallyCompletedCount;
            workBlockIndex--;
        }
        workerWorkTicks[(size_t)workerIndex * 8] += ticks;
        completedWorkBlocks.fetch_add(locallyCompletedCount);
    }
    template <typename TStageFunction>
    void ExecuteMainStage(TStageFunction& stageFunction, int workerIndex, int workerStart, SolverSyncStage& stage, int previousSyncIndex, int syncIndex) {  // :361
        int availableBlocksCount = stage.ClaimCount;
        if (availableBlocksCount == 0) return;
        if (availableBlocksCount == 1) {
            uint64_t t0 = __rdtsc();
            stageFunction(stage.WorkBlockStartIndex, workerIndex);
            workerWorkTicks[(size_t)workerIndex * 8] += __rdtsc() - t0;
        } else {
            substepContext.SyncIndex.store(syncIndex, std::memory_order_release);
            ExecuteWorkerStage(stageFunction, workerIndex, workerStart, stage.WorkBlockStartIndex, stage.Claims, stage.ClaimCount, previousSyncIndex, syncIndex,
                               substepContext.CompletedWorkBlockCount);
            while (substepContext.CompletedWorkBlockCount.load(std::memory_order_acquire) != availableBlocksCount) { _mm_pause(); _mm_pause(); _mm_pause(); }  // Thread.SpinWait(3), :395-398
            substepContext.CompletedWorkBlockCount.store(0, std::memory_order_relaxed);
        }
    }
    int GetPreviousSyncIndexForIncrementalUpdate(int substepIndex, int syncIndex, int syncStagesPerSubstep) { return substepIndex == 1 ? 0 : std::max(0, syncIndex - syncStagesPerSubstep); }  // :409
    int GetPreviousSyncIndexForIntegrateConstrainedKinematics(int substepIndex, int syncIndex, int syncStagesPerSubstep) {                                                                    // :414
        return substepIndex == 1 ? (Callbacks.IntegrateVelocityForKinematics ? 2 : 0) : std::max(0, syncIndex - syncStagesPerSubstep);
    }
    int GetWarmStartLookback(int substepIndex, int synchronizedBatchCount) {  // :421
        int warmStartLookback = synchronizedBatchCount + 2;
        if (substepIndex > 0) warmStartLookback += synchronizedBatchCount * (substepContext.HighestVelocityIterationCount - substepContext.VelocityIterationCounts[substepIndex - 1]);
        return warmStartLookback;
    }
    static int GetUniformlyDistributedStart(int workerIndex, int blockCount, int workerCount, int offset) {  // :445
        if (blockCount <= workerCount) return workerIndex < blockCount ? offset + workerIndex : -1;
        int blocksPerWorker = blockCount / workerCount;
        int remainder = blockCount - blocksPerWorker * workerCount;
        return offset + blocksPerWorker * workerIndex + std::min(remainder, workerIndex);
    }

    void SolveWorker(int workerIndex) {  // :458
        int workerCount = substepContext.WorkerCount;
        int incrementalUpdateWorkerStart = GetUniformlyDistributedStart(workerIndex, (int)substepContext.IncrementalUpdateBlocks.size(), workerCount, 0);
        int kinematicIntegrationWorkerStart = GetUniformlyDistributedStart(workerIndex, (int)substepContext.KinematicIntegrationBlocks.size(), workerCount, 0);
        int synchronizedBatchCount = std::min((int)Batches.size(), FallbackBatchThreshold);
        const bool fallbackExists = (int)Batches.size() > FallbackBatchThreshold;
        std::vector<int> batchStarts(Batches.size());
        for (int batchIndex = 0; batchIndex < synchronizedBatchCount; ++batchIndex) {
            int batchOffset = batchIndex > 0 ? substepContext.ConstraintBatchBoundaries[batchIndex - 1] : 0;
            int batchCount = substepContext.ConstraintBatchBoundaries[batchIndex] - batchOffset;
            batchStarts[batchIndex] = GetUniformlyDistributedStart(workerIndex, batchCount, workerCount, 0);
        }
        const float Dt = substepContext.Dt, InverseDt = substepContext.InverseDt;
        int stageSubstepIndex = 0;
        auto incrementalUpdateStage = [&](int blockIndex, int worker) {  // :260
            WorkBlock& block = substepContext.IncrementalUpdateBlocks[blockIndex];
            TypeBatch& typeBatch = Batches[block.BatchIndex].TypeBatches[block.TypeBatchIndex];
            TypeProcessors[typeBatch.TypeId]->IncrementallyUpdateForSubstep(typeBatch, bodies, Dt, InverseDt, block.StartBundle, block.End);
        };
        auto integrateConstrainedKinematicsStage = [&](int blockIndex, int worker) {  // :275
            IntegrationWorkBlock& block = substepContext.KinematicIntegrationBlocks[blockIndex];
            if (stageSubstepIndex == 0) IntegrateKinematicVelocities(block.StartBundleIndex, block.EndBundleIndex, Dt, worker);
            else IntegrateKinematicPosesAndVelocities(block.StartBundleIndex, block.EndBundleIndex, Dt, worker);
        };
        auto warmstartStage = [&](int blockIndex, int worker) {  // :219
            WorkBlock& block = substepContext.ConstraintBlocks[blockIndex];
            TypeBatch& typeBatch = Batches[block.BatchIndex].TypeBatches[block.TypeBatchIndex];
            WarmStartBlock(stageSubstepIndex != 0, worker, block.BatchIndex, block.TypeBatchIndex, block.StartBundle, block.End, typeBatch, TypeProcessors[typeBatch.TypeId].get(), Dt, InverseDt);
        };
        auto solveStage = [&](int blockIndex, int worker) {  // :244
            WorkBlock& block = substepContext.ConstraintBlocks[blockIndex];
            TypeBatch& typeBatch = Batches[block.BatchIndex].TypeBatches[block.TypeBatchIndex];
            TypeProcessors[typeBatch.TypeId]->Solve(typeBatch, bodies, Dt, InverseDt, block.StartBundle, block.End);
        };
        int maximumSyncStagesPerSubstep = 2 + synchronizedBatchCount * (1 + substepContext.HighestVelocityIterationCount);
        if (workerIndex == 0) {
            for (int substepIndex = 0; substepIndex < substepCount; ++substepIndex) {
                int syncIndex = substepIndex * maximumSyncStagesPerSubstep + 1;
                if (substepIndex > 0)
                    ExecuteMainStage(incrementalUpdateStage, workerIndex, incrementalUpdateWorkerStart, substepContext.Stages[0],
                                     GetPreviousSyncIndexForIncrementalUpdate(substepIndex, syncIndex, maximumSyncStagesPerSubstep), syncIndex);
                ++syncIndex;
                if (substepIndex > 0 || Callbacks.IntegrateVelocityForKinematics) {
                    stageSubstepIndex = substepIndex;
                    ExecuteMainStage(integrateConstrainedKinematicsStage, workerIndex, kinematicIntegrationWorkerStart, substepContext.Stages[1],
                                     GetPreviousSyncIndexForIntegrateConstrainedKinematics(substepIndex, syncIndex, maximumSyncStagesPerSubstep), syncIndex);
                }
                stageSubstepIndex = substepIndex;
                int warmStartLookback = GetWarmStartLookback(substepIndex, synchronizedBatchCount);
                for (int batchIndex = 0; batchIndex < synchronizedBatchCount; ++batchIndex) {
                    ++syncIndex;
                    ExecuteMainStage(warmstartStage, workerIndex, batchStarts[batchIndex], substepContext.Stages[batchIndex + 2], std::max(0, syncIndex - warmStartLookback), syncIndex);
                }
                if (fallbackExists) {  // :546-563: the fallback batch runs on worker 0, bundle after bundle
                    ConstraintBatch& batch = Batches[FallbackBatchThreshold];
                    for (int j = 0; j < (int)batch.TypeBatches.size(); ++j) {
                        TypeBatch& typeBatch = batch.TypeBatches[j];
                        WarmStartBlock(substepIndex != 0, 0, FallbackBatchThreshold, j, 0, typeBatch.BundleCount, typeBatch, TypeProcessors[typeBatch.TypeId].get(), Dt, InverseDt);
                    }
                }
                int velocityIterationCountForSubstep = substepContext.VelocityIterationCounts[substepIndex];
                for (int iterationIndex = 0; iterationIndex < velocityIterationCountForSubstep; ++iterationIndex) {
                    for (int batchIndex = 0; batchIndex < synchronizedBatchCount; ++batchIndex) {
                        ++syncIndex;
                        ExecuteMainStage(solveStage, workerIndex, batchStarts[batchIndex], substepContext.Stages[batchIndex + 2], std::max(0, syncIndex - synchronizedBatchCount), syncIndex);
                    }
                    if (fallbackExists) {  // :574-583
                        ConstraintBatch& batch = Batches[FallbackBatchThreshold];
                        for (int j = 0; j < (int)batch.TypeBatches.size(); ++j) {
                            TypeBatch& typeBatch = batch.TypeBatches[j];
                            TypeProcessors[typeBatch.TypeId]->Solve(typeBatch, bodies, Dt, InverseDt, 0, typeBatch.BundleCount);
                        }
                    }
                }
            }
            substepContext.SyncIndex.store(INT32_MIN, std::memory_order_release);
        } else {
            int latestCompletedSyncIndex = 0;
            int syncIndexInSubstep = -1;
            int substepIndex = 0;
            for (;;) {
                int syncIndex;
                LocalSpinWait spinWait;
                while (latestCompletedSyncIndex == (syncIndex = substepContext.SyncIndex.load(std::memory_order_acquire))) spinWait.SpinOnce();  // :599-605
                if (syncIndex == INT32_MIN) break;
                int syncStepsSinceLast = syncIndex - latestCompletedSyncIndex;
                syncIndexInSubstep += syncStepsSinceLast;
                while (syncIndexInSubstep >= maximumSyncStagesPerSubstep) {
                    syncIndexInSubstep -= maximumSyncStagesPerSubstep;
                    ++substepIndex;
                }
                SolverSyncStage& stage = substepContext.Stages[syncIndexInSubstep];
                stageSubstepIndex = substepIndex;
                switch (stage.StageType) {
                    case IncrementalUpdate:
                        ExecuteWorkerStage(incrementalUpdateStage, workerIndex, incrementalUpdateWorkerStart, 0, stage.Claims, stage.ClaimCount,
                                           GetPreviousSyncIndexForIncrementalUpdate(substepIndex, syncIndex, maximumSyncStagesPerSubstep), syncIndex, substepContext.CompletedWorkBlockCount);
                        break;
                    case IntegrateConstrainedKinematics:
                        ExecuteWorkerStage(integrateConstrainedKinematicsStage, workerIndex, kinematicIntegrationWorkerStart, 0, stage.Claims, stage.ClaimCount,
                                           GetPreviousSyncIndexForIntegrateConstrainedKinematics(substepIndex, syncIndex, maximumSyncStagesPerSubstep), syncIndex,
                                           substepContext.CompletedWorkBlockCount);
                        break;
                    case WarmStartStage:
                        ExecuteWorkerStage(warmstartStage, workerIndex, batchStarts[stage.BatchIndex], stage.WorkBlockStartIndex, stage.Claims, stage.ClaimCount,
                                           std::max(0, syncIndex - GetWarmStartLookback(substepIndex, synchronizedBatchCount)), syncIndex, substepContext.CompletedWorkBlockCount);
                        break;
                    case SolveStage:
                        ExecuteWorkerStage(solveStage, workerIndex, batchStarts[stage.BatchIndex], stage.WorkBlockStartIndex, stage.Claims, stage.ClaimCount,
                                           std::max(0, syncIndex - synchronizedBatchCount), syncIndex, substepContext.CompletedWorkBlockCount);
                        break;
                }
                latestCompletedSyncIndex = syncIndex;
            }
        }
    }

    std::vector<IntegrationWorkBlock> BuildKinematicIntegrationWorkBlocks(int minimumBlockSizeInBundles, int maximumBlockSizeInBundles, int targetBlockCount) {  // :655
        std::vector<IntegrationWorkBlock> workBlocks;
        int bundleCount = GetBundleCount((int)ConstrainedKinematicHandles.size());
        if (bundleCount > 0) {
            int targetBundlesPerBlock = bundleCount / targetBlockCount;
            if (targetBundlesPerBlock < minimumBlockSizeInBundles) targetBundlesPerBlock = minimumBlockSizeInBundles;
            if (targetBundlesPerBlock > maximumBlockSizeInBundles) targetBundlesPerBlock = maximumBlockSizeInBundles;
            int blockCount = (bundleCount + targetBundlesPerBlock - 1) / targetBundlesPerBlock;
            int bundlesPerBlock = bundleCount / blockCount;
            int remainder = bundleCount - bundlesPerBlock * blockCount;
            int previousEnd = 0;
            for (int i = 0; i < blockCount; ++i) {
                int bundleCountForBlock = bundlesPerBlock;
                if (i < remainder) ++bundleCountForBlock;
                workBlocks.push_back({previousEnd, previousEnd + bundleCountForBlock});
                previousEnd += bundleCountForBlock;
            }
        }
        return workBlocks;
    }
    void BuildWorkBlocks(int minimumBlockSizeInBundles, int maximumBlockSizeInBundles, int targetBlocksPerBatch, bool incrementalFilter, std::vector<WorkBlock>& workBlocks,
                         std::vector<int>& batchBoundaries) {  // :683
        int batchCount = incrementalFilter ? (int)Batches.size() : std::min((int)Batches.size(), FallbackBatchThreshold);
        workBlocks.clear();
        batchBoundaries.assign(batchCount, 0);
        float inverseMinimumBlockSizeInBundles = 1.0f / minimumBlockSizeInBundles;
        float inverseMaximumBlockSizeInBundles = 1.0f / maximumBlockSizeInBundles;
        auto allow = [&](int typeId) { return !incrementalFilter || TypeProcessors[typeId]->RequiresIncrementalSubstepUpdates; };
        for (int batchIndex = 0; batchIndex < batchCount; ++batchIndex) {
            std::vector<TypeBatch>& typeBatches = Batches[batchIndex].TypeBatches;
            int bundleCount = 0;
            for (auto& tb : typeBatches)
                if (allow(tb.TypeId)) bundleCount += tb.BundleCount;
            for (int typeBatchIndex = 0; typeBatchIndex < (int)typeBatches.size(); ++typeBatchIndex) {
                TypeBatch& typeBatch = typeBatches[typeBatchIndex];
                if (!allow(typeBatch.TypeId)) continue;
                float typeBatchSizeFraction = typeBatch.BundleCount / (float)bundleCount;
                float typeBatchMaximumBlockCount = typeBatch.BundleCount * inverseMinimumBlockSizeInBundles;
                float typeBatchMinimumBlockCount = typeBatch.BundleCount * inverseMaximumBlockSizeInBundles;
                int typeBatchBlockCount = std::max(1, (int)std::min(typeBatchMaximumBlockCount, std::max(typeBatchMinimumBlockCount, targetBlocksPerBatch * typeBatchSizeFraction)));
                int previousEnd = 0;
                int baseBlockSizeInBundles = typeBatch.BundleCount / typeBatchBlockCount;
                int remainder = typeBatch.BundleCount - baseBlockSizeInBundles * typeBatchBlockCount;
                for (int newBlockIndex = 0; newBlockIndex < typeBatchBlockCount; ++newBlockIndex) {
                    int blockBundleCount = newBlockIndex < remainder ? baseBlockSizeInBundles + 1 : baseBlockSizeInBundles;
                    workBlocks.push_back({batchIndex, typeBatchIndex, previousEnd, previousEnd + blockBundleCount});
                    previousEnd += blockBundleCount;
                }
            }
            batchBoundaries[batchIndex] = (int)workBlocks.size();
        }
    }

    void ExecuteMultithreaded(float dt, ThreadDispatcher& threadDispatcher) {  // :753
        int workerCount = substepContext.WorkerCount = threadDispatcher.ThreadCount();
        workerWorkTicks.assign((size_t)workerCount * 8, 0);
        substepContext.Dt = dt;
        substepContext.InverseDt = 1.0f / dt;
        substepContext.VelocityIterationCounts.resize(substepCount);
        for (int i = 0; i < substepCount; ++i) substepContext.VelocityIterationCounts[i] = GetVelocityIterationCountForSubstepIndex(i);
        const int targetBlocksPerBatchPerWorker = 4;
        const int minimumBlockSizeInBundles = 1;
        const int maximumBlockSizeInBundles = 1024;
        int targetBlocksPerBatch = workerCount * targetBlocksPerBatchPerWorker;
        std::vector<int> incrementalUpdateBatchBoundaries;
        BuildWorkBlocks(minimumBlockSizeInBundles, maximumBlockSizeInBundles, targetBlocksPerBatch, false, substepContext.ConstraintBlocks, substepContext.ConstraintBatchBoundaries);
        BuildWorkBlocks(minimumBlockSizeInBundles, maximumBlockSizeInBundles, targetBlocksPerBatch, true, substepContext.IncrementalUpdateBlocks, incrementalUpdateBatchBoundaries);
        substepContext.KinematicIntegrationBlocks = BuildKinematicIntegrationWorkBlocks(minimumBlockSizeInBundles, maximumBlockSizeInBundles, targetBlocksPerBatch);
        substepContext.SyncIndex.store(0);
        substepContext.CompletedWorkBlockCount.store(0);
        int incrementalCount = (int)substepContext.IncrementalUpdateBlocks.size();
        int kinematicCount = (int)substepContext.KinematicIntegrationBlocks.size();
        int totalConstraintBatchWorkBlockCount = substepContext.ConstraintBatchBoundaries.empty() ? 0 : substepContext.ConstraintBatchBoundaries.back();
        int totalClaimCount = incrementalCount + kinematicCount + totalConstraintBatchWorkBlockCount;
        int stagesPerIteration = std::min((int)Batches.size(), FallbackBatchThreshold);
        substepContext.HighestVelocityIterationCount = 0;
        for (int c : substepContext.VelocityIterationCounts) substepContext.HighestVelocityIterationCount = std::max(c, substepContext.HighestVelocityIterationCount);
        substepContext.Stages.assign(2 + stagesPerIteration * (1 + substepContext.HighestVelocityIterationCount), SolverSyncStage{});
        std::unique_ptr<std::atomic<int>[]> claims(new std::atomic<int>[std::max(totalClaimCount, 1)]);
        for (int i = 0; i < totalClaimCount; ++i) claims[i].store(0, std::memory_order_relaxed);
        substepContext.Stages[0] = SolverSyncStage{claims.get(), incrementalCount, 0, IncrementalUpdate, 0};
        substepContext.Stages[1] = SolverSyncStage{claims.get() + incrementalCount, kinematicCount, 0, IntegrateConstrainedKinematics, 0};
        int targetStageIndex = 2;
        int preambleClaimCount = incrementalCount + kinematicCount;
        int claimStart = preambleClaimCount;
        for (int batchIndex = 0; batchIndex < stagesPerIteration; ++batchIndex) {
            int stageIndex = targetStageIndex++;
            int batchStart = batchIndex == 0 ? 0 : substepContext.ConstraintBatchBoundaries[batchIndex - 1];
            int workBlocksInBatch = substepContext.ConstraintBatchBoundaries[batchIndex] - batchStart;
            substepContext.Stages[stageIndex] = SolverSyncStage{claims.get() + claimStart, workBlocksInBatch, batchStart, WarmStartStage, batchIndex};
            claimStart += workBlocksInBatch;
        }
        for (int iterationIndex = 0; iterationIndex < substepContext.HighestVelocityIterationCount; ++iterationIndex) {
            claimStart = preambleClaimCount;
            for (int batchIndex = 0; batchIndex < stagesPerIteration; ++batchIndex) {
                int stageIndex = targetStageIndex++;
                int batchStart = batchIndex == 0 ? 0 : substepContext.ConstraintBatchBoundaries[batchIndex - 1];
                int workBlocksInBatch = substepContext.ConstraintBatchBoundaries[batchIndex] - batchStart;
                substepContext.Stages[stageIndex] = SolverSyncStage{claims.get() + claimStart, workBlocksInBatch, batchStart, SolveStage, batchIndex};
                claimStart += workBlocksInBatch;
            }
        }
        if (!Batches.empty()) threadDispatcher.DispatchWorkers([this](int workerIndex) { SolveWorker(workerIndex); });
    }

    void Solve(float totalDt, ThreadDispatcher* threadDispatcher) {  // :1415
        if (threadDispatcher == nullptr || threadDispatcher->ThreadCount() == 1) {
            SolveSingleThreaded(totalDt);
        } else {
            float substepDt = totalDt / substepCount;
            Callbacks.PrepareForIntegration(substepDt);
            ExecuteMultithreaded(substepDt, *threadDispatcher);
        }
    }

    // ---- PoseIntegrator.IntegrateAfterSubstepping (PoseIntegrator.cs:537-726)
    void IntegrateBundlesAfterSubstepping(int bundleStartIndex, int bundleEndIndex, float dt, float substepDt, int substepCountArg, int workerIndex) {  // :537
        int bodyCount = bodies.count;
        VF bundleDt = vf(dt);
        VF bundleSubstepDt = vf(substepDt);
        for (int i = bundleStartIndex; i < bundleEndIndex; ++i) {
            int bundleBaseIndex = i * W;
            int countInBundle = std::min(bodyCount - bundleBaseIndex, W);
            VI unconstrainedMask = vi(0), bodyIndices = vi(0);
            bool anyBodyInBundleIsUnconstrained = false;
            for (int innerIndex = 0; innerIndex < countInBundle; ++innerIndex) {
                int bodyIndex = bundleBaseIndex + innerIndex;
                bodyIndices[innerIndex] = bodyIndex;
                int bodyHandle = IndexToHandle[bodyIndex];
                if (mergedConstrainedBodyHandles.Contains(bodyHandle)) {
                    unconstrainedMask[innerIndex] = 0;
                } else {
                    unconstrainedMask[innerIndex] = -1;
                    anyBodyInBundleIsUnconstrained = true;
                }
            }
            if (countInBundle < W) {
                VI trailingMask = CreateTrailingMaskForCountInBundle(countInBundle);
                bodyIndices = BitwiseOr(bodyIndices, trailingMask);
                unconstrainedMask = AndNot(unconstrainedMask, trailingMask);
            }
            VF bundleEffectiveDt;
            if (Callbacks.AllowSubstepsForUnconstrainedBodies) bundleEffectiveDt = bundleSubstepDt;
            else bundleEffectiveDt = ConditionalSelect(unconstrainedMask, bundleDt, bundleSubstepDt);
            VF halfDt = bundleEffectiveDt * vf(0.5f);
            Vector3Wide position;
            QuaternionWide orientation;
            BodyVelocityWide velocity;
            BodyInertiaWide localInertia;
            bodies.GatherState(bodyIndices, false, position, orientation, velocity, localInertia);
            VI unconstrainedVelocityIntegrationMask;
            bool anyBodyInBundleNeedsVelocityIntegration;
            if (Callbacks.IntegrateVelocityForKinematics) {
                unconstrainedVelocityIntegrationMask = unconstrainedMask;
                anyBodyInBundleNeedsVelocityIntegration = anyBodyInBundleIsUnconstrained;
            } else {
                VI isKinematic = Bodies::IsKinematic(localInertia);
                unconstrainedVelocityIntegrationMask = AndNot(unconstrainedMask, isKinematic);
                anyBodyInBundleNeedsVelocityIntegration = LessThanAny(unconstrainedVelocityIntegrationMask, vi(0));
            }
            VI velocityMaskedBodyIndices = BitwiseOr(bodyIndices, OnesComplement(unconstrainedVelocityIntegrationMask));
            if (anyBodyInBundleIsUnconstrained) {
                int integrationStepCount = Callbacks.AllowSubstepsForUnconstrainedBodies ? substepCountArg : 1;
                for (int stepIndex = 0; stepIndex < integrationStepCount; ++stepIndex) {
                    BodyVelocityWide previousVelocity = velocity;
                    if (anyBodyInBundleNeedsVelocityIntegration) {
                        Callbacks.IntegrateVelocity(velocityMaskedBodyIndices, position, orientation, localInertia, unconstrainedVelocityIntegrationMask, workerIndex, bundleEffectiveDt, velocity);
                        Vector3Wide::ConditionalSelect(unconstrainedVelocityIntegrationMask, velocity.Linear, previousVelocity.Linear, velocity.Linear);
                        Vector3Wide::ConditionalSelect(unconstrainedVelocityIntegrationMask, velocity.Angular, previousVelocity.Angular, velocity.Angular);
                    }
                    position = position + velocity.Linear * bundleEffectiveDt;
                    if (Callbacks.AngularIntegrationMode == ConserveMomentum) {
                        QuaternionWide previousOrientation = orientation;
                        PoseIntegration::Integrate(orientation, velocity.Angular, halfDt, orientation);
                        Symmetric3x3Wide inverseInertiaTensor;
                        PoseIntegration::RotateInverseInertia(localInertia.InverseInertiaTensor, orientation, inverseInertiaTensor);
                        PoseIntegration::IntegrateAngularVelocityConserveMomentum(previousOrientation, localInertia.InverseInertiaTensor, inverseInertiaTensor, velocity.Angular);
                    } else if (Callbacks.AngularIntegrationMode == ConserveMomentumWithGyroscopicTorque) {
                        PoseIntegration::Integrate(orientation, velocity.Angular, halfDt, orientation);
                        PoseIntegration::IntegrateAngularVelocityConserveMomentumWithGyroscopicTorque(orientation, localInertia.InverseInertiaTensor, velocity.Angular, bundleEffectiveDt);
                    } else {
                        PoseIntegration::Integrate(orientation, velocity.Angular, halfDt, orientation);
                    }
                    VI integratePoseMask = CreateMaskForCountInBundle(countInBundle);
                    if (Callbacks.AllowSubstepsForUnconstrainedBodies) {
                        if (stepIndex > 0) integratePoseMask = BitwiseAnd(integratePoseMask, unconstrainedMask);
                    }
                    bodies.ScatterPose(position, orientation, bodyIndices, integratePoseMask);
                    if (anyBodyInBundleNeedsVelocityIntegration) bodies.ScatterVelocities<true, true>(velocity, velocityMaskedBodyIndices);
                }
            } else {
                PoseIntegration::Integrate(orientation, velocity.Angular, halfDt, orientation);
                position = position + velocity.Linear * bundleEffectiveDt;
                VI integratePoseMask = CreateMaskForCountInBundle(countInBundle);
                bodies.ScatterPose(position, orientation, bodyIndices, integratePoseMask);
            }
        }
    }
    void IntegrateAfterSubstepping(float dt, int substepCountArg, ThreadDispatcher* threadDispatcher) {  // :707
        float substepDt = dt / substepCountArg;
        float velocityIntegrationTimestep = Callbacks.AllowSubstepsForUnconstrainedBodies ? substepDt : dt;
        Callbacks.PrepareForIntegration(velocityIntegrationTimestep);
        int bundleCount = GetBundleCount(bodies.count);
        if (threadDispatcher != nullptr && threadDispatcher->ThreadCount() > 1) {
            const int jobsPerWorker = 2;  // PrepareForMultithreadedExecution, :410
            int targetJobCount = threadDispatcher->ThreadCount() * jobsPerWorker;
            int jobSize = bundleCount / targetJobCount;
            if (jobSize == 0) jobSize = 1;
            int jobCount = bundleCount / jobSize;
            if (jobSize * jobCount < bundleCount) ++jobCount;
            std::atomic<int> availableJobCount{jobCount};
            threadDispatcher->DispatchWorkers([&](int workerIndex) {
                for (;;) {  // TryGetJob, :383
                    int jobIndex = availableJobCount.fetch_sub(1) - 1;
                    if (jobIndex < 0) break;
                    int start = jobIndex * jobSize;
                    int exclusiveEnd = std::min(start + jobSize, bundleCount);
                    IntegrateBundlesAfterSubstepping(start, exclusiveEnd, dt, substepDt, substepCountArg, workerIndex);
                }
            });
        } else {
            IntegrateBundlesAfterSubstepping(0, bundleCount, dt, substepDt, substepCountArg, 0);
        }
    }
};

// First touch decides which NUMA node a page lives on: large buffers are copied by threads pinned to the cores the workers will run on (PinPlan: one socket), 2 MB
// chunks dealt round-robin, so the session's memory sits on that socket, spread over its memory controllers — not wherever the caller's thread happened to run.
static void* AlignedCopy(const void* src, size_t bytes) {  // BufferPool blocks are 128-byte aligned (BepuUtilities/Memory/BufferPool.cs:42)
    void* p = nullptr;
    if (posix_memalign(&p, 128, bytes ? bytes : 128) != 0) return nullptr;
    const size_t chunk = (size_t)2 << 20;
    const PinPlan& plan = PinPlan::Get();
    const int touchers = (int)std::min<size_t>({(size_t)16, (size_t)std::max(1, plan.firstSocketCores), (bytes + chunk - 1) / chunk});
    if (!PinPlan::Enabled() || touchers <= 1) {
        std::memcpy(p, src, bytes);
        return p;
    }
    std::vector<std::thread> threads;
    for (int t = 0; t < touchers; ++t)
        threads.emplace_back([=] {
            PinPlan::PinCurrentThread(t);
            for (size_t at = (size_t)t * chunk; at < bytes; at += (size_t)touchers * chunk) std::memcpy((char*)p + at, (const char*)src + at, std::min(chunk, bytes - at));
        });
    for (auto& th : threads) th.join();
    return p;
}

static std::mutex g_dispatcherMutex;
static std::unique_ptr<ThreadDispatcher> g_dispatcher;

// A scene held the way the reference holds it between frames: bodies and type batches in 128-byte aligned memory the session owns (BufferPool.cs:42,83 — the
// reference's buffers are always aligned and never marshalled), batchReferencedHandles maintained with the constraint set (Solver.cs:1046-1051: updated on Add /
// Remove, not rebuilt per frame). Built once by wide_session_create; wide_session_solve then runs exactly what Simulation.Solve runs (Simulation.cs:278-290):
// PrepareConstraintIntegrationResponsibilities, Solve, IntegrateAfterSubstepping — nothing else, so that it can be timed as the reference's frame would be.
struct Session {
    Solver solver;
    struct Owned { void* aligned; void* original; size_t bytes; };
    std::vector<Owned> owned;                       // [0] = bodies, then per type batch: references, prestep, accumulated impulses
    std::vector<int32_t> indexToHandle, handleToIndex;
    ~Session() {
        for (auto& o : owned) free(o.aligned);
    }
    void* Own(void* p, size_t bytes) {
        void* a = AlignedCopy(p, bytes);
        owned.push_back({a, p, bytes});
        return a;
    }
};

static int ApplyParams(Solver& solver, const SceneParams* params) {
    if (params->exchange != nullptr) return 3;    // the split-lattice exchange hook belongs to oracle/, not here
    if (params->dt <= 0 || params->substep_count < 1) return 1;
    solver.Callbacks.Gravity[0] = params->gravity[0];
    solver.Callbacks.Gravity[1] = params->gravity[1];
    solver.Callbacks.Gravity[2] = params->gravity[2];
    solver.Callbacks.LinearDamping = params->linear_damping;
    solver.Callbacks.VelocityModel = params->velocity_model;
    solver.Callbacks.PlanetCenter[0] = params->planet_center[0]; solver.Callbacks.PlanetCenter[1] = params->planet_center[1]; solver.Callbacks.PlanetCenter[2] = params->planet_center[2];
    solver.Callbacks.PlanetGravity = params->planet_gravity;
    solver.Callbacks.BodyGravities = params->body_gravity;
    solver.Callbacks.AngularDamping = params->angular_damping;
    solver.Callbacks.AngularIntegrationMode = params->angular_integration_mode;
    solver.Callbacks.AllowSubstepsForUnconstrainedBodies = params->allow_substeps_for_unconstrained != 0;
    solver.Callbacks.IntegrateVelocityForKinematics = params->integrate_velocity_for_kinematics != 0;
    solver.substepCount = params->substep_count;
    solver.velocityIterations.assign(params->velocity_iterations, params->velocity_iterations + params->substep_count);
    for (int v : solver.velocityIterations)
        if (v < 1) return 1;
    solver.FallbackBatchThreshold = params->fallback_batch_threshold > 0 ? params->fallback_batch_threshold : 64;
    return 0;
}

static int CreateSession(SceneDesc* scene, SceneParams* params, std::unique_ptr<Session>& out) {
    if (scene->bundle_width != W) return 2;       // this restatement is the AVX2 host shape only; results do not depend on the width (lanes are independent)
    std::unique_ptr<Session> session(new Session);
    Solver& solver = session->solver;
    if (int status = ApplyParams(solver, params)) return status;
    solver.bodies.states = (float*)session->Own(scene->bodies, (size_t)scene->body_count * 128);
    solver.bodies.count = scene->body_count;
    session->indexToHandle.assign(scene->index_to_handle, scene->index_to_handle + scene->body_count);
    session->handleToIndex.assign(scene->handle_to_index, scene->handle_to_index + scene->handle_capacity);
    solver.IndexToHandle = session->indexToHandle.data();
    solver.HandleToIndex = session->handleToIndex.data();
    solver.HandleCapacity = scene->handle_capacity;
    solver.TypeProcessors.resize(64);
    solver.Batches.resize(scene->batch_count);
    if (scene->batch_count > solver.FallbackBatchThreshold + 1) return 4;  // at most FallbackBatchThreshold synchronized batches + the fallback batch
    solver.batchReferencedHandles.assign(scene->batch_count, {});
    int flat = 0;
    for (int b = 0; b < scene->batch_count; ++b) {
        solver.batchReferencedHandles[b].Flags.assign((scene->handle_capacity + 63) / 64 + 1, 0);
        for (int j = 0; j < scene->type_batch_counts[b]; ++j, ++flat) {
            SceneTypeBatch& s = scene->type_batches[flat];
            if (s.type_id < 0 || s.type_id >= 64) return 5;
            if (!solver.TypeProcessors[s.type_id]) solver.TypeProcessors[s.type_id].reset(CreateProcessor(s.type_id));
            TypeProcessor* processor = solver.TypeProcessors[s.type_id].get();
            if (processor == nullptr) return 5;
            TypeBatch typeBatch;
            typeBatch.TypeId = s.type_id;
            typeBatch.ConstraintCount = s.constraint_count;
            typeBatch.BundleCount = Solver::GetBundleCount(s.constraint_count);
            typeBatch.BodyReferences = (VI*)session->Own(s.body_refs, (size_t)typeBatch.BundleCount * processor->BodiesPerConstraint * 32);
            typeBatch.PrestepData = (VF*)session->Own(s.prestep, (size_t)typeBatch.BundleCount * processor->PrestepFloats * 32);
            typeBatch.AccumulatedImpulses = (VF*)session->Own(s.accumulated, (size_t)typeBatch.BundleCount * processor->ImpulseFloats * 32);
            // batchReferencedHandles (Solver.cs:1046-1051): the handles of the DYNAMIC bodies each batch references.
            const int32_t* refs = (const int32_t*)typeBatch.BodyReferences;
            for (int c = 0; c < s.constraint_count; ++c)
                for (int k = 0; k < processor->BodiesPerConstraint; ++k) {
                    int32_t encoded = refs[(c >> 3) * processor->BodiesPerConstraint * W + k * W + (c & 7)];
                    if ((uint32_t)encoded < DynamicLimit) solver.batchReferencedHandles[b].Set(scene->index_to_handle[encoded]);
                }
            solver.Batches[b].TypeBatches.push_back(typeBatch);
        }
    }
    solver.ConstrainedKinematicHandles.assign(scene->constrained_kinematic_handles, scene->constrained_kinematic_handles + scene->constrained_kinematic_count);
    out = std::move(session);
    return 0;
}

// Simulation.Solve (Simulation.cs:278-290), `frames` times on the session's state. phase_seconds (optional, 4 doubles): the time spent in each of the three calls, and
// the seconds all workers together spent inside Solve's work blocks (the rest of threads x Solve time is waiting at sync stages).
static int SolveSession(Session& session, float dt, int threads, int frames, double* phaseSeconds) {
    Solver& solver = session.solver;
    if (!(dt > 0) || frames < 0) return 1;
    ThreadDispatcher* dispatcher = nullptr;
    std::unique_lock<std::mutex> dispatcherLock(g_dispatcherMutex, std::defer_lock);
    if (threads > 1) {
        dispatcherLock.lock();
        if (!g_dispatcher || g_dispatcher->ThreadCount() != threads) g_dispatcher.reset(new ThreadDispatcher(threads));
        dispatcher = g_dispatcher.get();
    }
    ScopedMainPin mainPin(true);  // worker 0 (and the single-threaded path) on the plan's first core
    using Clock = std::chrono::steady_clock;
    double phases[4] = {0, 0, 0, 0};
    for (int frame = 0; frame < frames; ++frame) {
        auto t0 = Clock::now();
        solver.PrepareConstraintIntegrationResponsibilities(dispatcher);
        auto t1 = Clock::now();
        uint64_t tick1 = __rdtsc();
        solver.Solve(dt, dispatcher);
        uint64_t tick2 = __rdtsc();
        auto t2 = Clock::now();
        if (dispatcher && tick2 > tick1) {  // seconds the workers spent inside work blocks, summed over workers (ticks converted with this Solve's own wall time)
            uint64_t sum = 0;
            for (size_t w = 0; w < solver.workerWorkTicks.size(); w += 8) sum += solver.workerWorkTicks[w];
            phases[3] += (double)sum * std::chrono::duration<double>(t2 - t1).count() / (double)(tick2 - tick1);
        } else {
            phases[3] += std::chrono::duration<double>(t2 - t1).count();
        }
        solver.IntegrateAfterSubstepping(dt, solver.substepCount, dispatcher);
        auto t3 = Clock::now();
        phases[0] += std::chrono::duration<double>(t1 - t0).count();
        phases[1] += std::chrono::duration<double>(t2 - t1).count();
        phases[2] += std::chrono::duration<double>(t3 - t2).count();
    }
    if (phaseSeconds) { phaseSeconds[0] = phases[0]; phaseSeconds[1] = phases[1]; phaseSeconds[2] = phases[2]; phaseSeconds[3] = phases[3]; }
    return 0;
}

static void ReadSession(Session& session) {  // the session's state back into the buffers it was created from
    for (auto& o : session.owned) std::memcpy(o.original, o.aligned, o.bytes);
}

static int SolveScene(SceneDesc* scene, SceneParams* params) {
    std::unique_ptr<Session> session;
    if (int status = CreateSession(scene, params, session)) return status;
    if (int status = SolveSession(*session, params->dt, params->threads, 1, nullptr)) return status;
    ReadSession(*session);
    return 0;
}

}  // namespace wide

#include "wide_bounds.h"

namespace wide {
// The records of include/bepuhip.h as plain data (the same bytes oracle_ffi hands to the scalar oracle).
struct CollidableRecord { int32_t shape_type; float shape[9]; float minimum_speculative_margin, maximum_speculative_margin; int32_t allow_expansion_beyond_speculative_margin;
                          float sleep_threshold; int32_t minimum_timesteps_under_threshold; int32_t activity; };
struct PredictedRecord { float min[3]; float speculative_margin; float max[3]; int32_t activity; };
struct CompoundChildRecord { int32_t shape_type; float shape[9]; float local_position[3]; float local_orientation[4]; };

static int AddConvexShape(bounds::Shapes& shapes, int type, const float* s) {  // Shapes.Add<TShape>: the index of the shape inside its type's batch
    using namespace bounds;
    switch (type) {
        case SphereId: shapes.spheres.push_back({s[0]}); return (int)shapes.spheres.size() - 1;
        case CapsuleId: shapes.capsules.push_back({s[0], s[1]}); return (int)shapes.capsules.size() - 1;
        case BoxId: shapes.boxes.push_back({s[0], s[1], s[2]}); return (int)shapes.boxes.size() - 1;
        case TriangleId: shapes.triangles.push_back({{s[0], s[1], s[2]}, {s[3], s[4], s[5]}, {s[6], s[7], s[8]}}); return (int)shapes.triangles.size() - 1;
        case CylinderId: shapes.cylinders.push_back({s[0], s[1]}); return (int)shapes.cylinders.size() - 1;
        case ConvexHullId: return (int)s[0];
        default: return -1;
    }
}

static int PredictBoundingBoxesOfScene(const float* bodyStates, int count, const SceneParams* params, const CollidableRecord* collidables, PredictedRecord* out, const float* hullPoints,
                                       const int* hullBegin, int hullCount, const CompoundChildRecord* children, const int* childBegin, int compoundCount, const float* triangles,
                                       const int* triangleBegin, const float* meshScales, int meshCount) {
    using namespace bounds;
    if (!bodyStates || !params || !collidables || !out || count < 0 || !(params->dt > 0)) return -1;
    World world;
    for (int h = 0; h < hullCount; ++h) {  // ConvexHullHelper.CreateShape (ConvexHullHelper.cs:1050-1063): bundles of points, the last vertex repeated into the unused lanes
        ConvexHull hull;
        int first = hullBegin[h], lastIndex = hullBegin[h + 1] - hullBegin[h] - 1;
        hull.Points.resize((size_t)(lastIndex + W) / W);
        for (size_t bundleIndex = 0; bundleIndex < hull.Points.size(); ++bundleIndex)
            for (int innerIndex = 0; innerIndex < W; ++innerIndex) {
                int index = (int)bundleIndex * W + innerIndex;
                if (index > lastIndex) index = lastIndex;
                const float* point = hullPoints + 3 * (size_t)(first + index);
                hull.Points[bundleIndex].X[innerIndex] = point[0]; hull.Points[bundleIndex].Y[innerIndex] = point[1]; hull.Points[bundleIndex].Z[innerIndex] = point[2];
            }
        world.shapes.hulls.push_back(std::move(hull));
    }
    for (int k = 0; k < compoundCount; ++k) {
        Compound compound;
        for (int j = childBegin[k]; j < childBegin[k + 1]; ++j) {
            const CompoundChildRecord& record = children[j];
            if (record.shape_type < 0 || record.shape_type > ConvexHullId) return -2;
            CompoundChild child;
            child.ShapeType = record.shape_type;
            child.ShapeIndex = AddConvexShape(world.shapes, record.shape_type, record.shape);
            if (child.ShapeType == ConvexHullId && (child.ShapeIndex < 0 || child.ShapeIndex >= hullCount)) return -2;
            child.LocalPosition = {record.local_position[0], record.local_position[1], record.local_position[2]};
            child.LocalOrientation = {record.local_orientation[0], record.local_orientation[1], record.local_orientation[2], record.local_orientation[3]};
            compound.Children.push_back(child);
        }
        world.shapes.compounds.push_back(std::move(compound));
    }
    for (int m = 0; m < meshCount; ++m) {
        Mesh mesh;
        mesh.scale = {meshScales[3 * m], meshScales[3 * m + 1], meshScales[3 * m + 2]};
        for (int t = triangleBegin[m]; t < triangleBegin[m + 1]; ++t) {
            const float* v = triangles + 9 * (size_t)t;
            mesh.Triangles.push_back({{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}});
        }
        world.shapes.meshes.push_back(std::move(mesh));
    }
    world.collidables.resize(count);
    world.activities.resize(count);
    world.boundsMin.assign(count, bounds::Vector3{0, 0, 0});
    world.boundsMax.assign(count, bounds::Vector3{0, 0, 0});
    for (int i = 0; i < count; ++i) {
        const CollidableRecord& record = collidables[i];
        Collidable& collidable = world.collidables[i];
        collidable.ShapeType = record.shape_type;
        if (record.shape_type >= 0 && record.shape_type <= ConvexHullId) collidable.ShapeIndex = AddConvexShape(world.shapes, record.shape_type, record.shape);
        else collidable.ShapeIndex = (int)record.shape[0];
        const int table = record.shape_type == ConvexHullId ? hullCount : (record.shape_type == CompoundId || record.shape_type == BigCompoundId) ? compoundCount : record.shape_type == MeshId ? meshCount : -1;
        if (record.shape_type > MeshId || (table >= 0 && (collidable.ShapeIndex < 0 || collidable.ShapeIndex >= table))) return -2;
        collidable.MinimumSpeculativeMargin = record.minimum_speculative_margin;
        collidable.MaximumSpeculativeMargin = record.maximum_speculative_margin;
        collidable.SpeculativeMargin = 0;
        collidable.AllowExpansionBeyondSpeculativeMargin = record.allow_expansion_beyond_speculative_margin != 0;
        world.activities[i] = {record.sleep_threshold, record.minimum_timesteps_under_threshold, record.activity & 0xFF, (record.activity & 0x100) != 0};
    }
    Bodies bodies;
    bodies.states = const_cast<float*>(bodyStates);  // read only here: the integrated velocities are never stored (:336)
    bodies.count = count;
    PoseIntegratorCallbacks callbacks;
    callbacks.Gravity[0] = params->gravity[0]; callbacks.Gravity[1] = params->gravity[1]; callbacks.Gravity[2] = params->gravity[2];
    callbacks.LinearDamping = params->linear_damping;
    callbacks.VelocityModel = params->velocity_model;
    callbacks.PlanetCenter[0] = params->planet_center[0]; callbacks.PlanetCenter[1] = params->planet_center[1]; callbacks.PlanetCenter[2] = params->planet_center[2];
    callbacks.PlanetGravity = params->planet_gravity;
    callbacks.BodyGravities = params->body_gravity;
    callbacks.AngularDamping = params->angular_damping;
    callbacks.AngularIntegrationMode = params->angular_integration_mode;
    callbacks.AllowSubstepsForUnconstrainedBodies = params->allow_substeps_for_unconstrained != 0;
    callbacks.IntegrateVelocityForKinematics = params->integrate_velocity_for_kinematics != 0;
    // PoseIntegrator.PredictBoundingBoxes(dt, pool, threadDispatcher) (:372-420), single threaded: prepare the callbacks for the full dt, one batcher, every bundle, flush.
    callbacks.PrepareForIntegration(params->dt);
    BoundingBoxBatcher batcher(world, params->dt);
    PredictBoundingBoxes(bodies, callbacks, world, 0, (count + W - 1) / W, params->dt, batcher, 0);
    batcher.Flush();
    for (int i = 0; i < count; ++i) {
        out[i].min[0] = world.boundsMin[i].X; out[i].min[1] = world.boundsMin[i].Y; out[i].min[2] = world.boundsMin[i].Z;
        out[i].max[0] = world.boundsMax[i].X; out[i].max[1] = world.boundsMax[i].Y; out[i].max[2] = world.boundsMax[i].Z;
        out[i].speculative_margin = world.collidables[i].SpeculativeMargin;
        out[i].activity = world.activities[i].TimestepsUnderThresholdCount | (world.activities[i].SleepCandidate ? 0x100 : 0);
    }
    return 0;
}
}  // namespace wide

extern "C" {
int wide_solve(wide::SceneDesc* scene, wide::SceneParams* params) { return wide::SolveScene(scene, params); }
// The persistent form (what bench.py's cpu_baseline times): create once, solve frames in place, read back, destroy.
void* wide_session_create(wide::SceneDesc* scene, wide::SceneParams* params, int* status) {
    std::unique_ptr<wide::Session> session;
    int s = wide::CreateSession(scene, params, session);
    if (status) *status = s;
    return s == 0 ? session.release() : nullptr;
}
int wide_session_solve(void* session, float dt, int threads, int frames, double* phase_seconds) {
    return session ? wide::SolveSession(*(wide::Session*)session, dt, threads, frames, phase_seconds) : -1;
}
int wide_session_read(void* session) {
    if (!session) return -1;
    wide::ReadSession(*(wide::Session*)session);
    return 0;
}
void wide_session_destroy(void* session) { delete (wide::Session*)session; }
// What the pinning plan looks like on this host: out[0] CPUs the process may use, out[1] physical cores of the socket the workers start on, out[2] hardware threads of
// that socket, out[3] 1 if pinning is on (WIDE_PIN != 0).
void wide_pin_plan(int* out) {
    const wide::PinPlan& plan = wide::PinPlan::Get();
    out[0] = (int)plan.cpus.size(); out[1] = plan.firstSocketCores; out[2] = plan.firstSocketCpus; out[3] = wide::PinPlan::Enabled() ? 1 : 0;
}
int wide_predict_bounding_boxes(const float* bodies, int count, const wide::SceneParams* params, const wide::CollidableRecord* collidables, wide::PredictedRecord* out,
                                const float* hull_points, const int* hull_begin, int hull_count, const wide::CompoundChildRecord* children, const int* child_begin, int compound_count,
                                const float* triangles, const int* triangle_begin, const float* mesh_scales, int mesh_count) {
    return wide::PredictBoundingBoxesOfScene(bodies, count, params, collidables, out, hull_points, hull_begin, hull_count, children, child_begin, compound_count, triangles, triangle_begin,
                                             mesh_scales, mesh_count);
}

// Per-type probe for unit tests: `iterations` x (WarmStart; Solve) on broadcast inputs, mirroring the reference's microbenchmarks
// (DemoBenchmarks/TwoBodyConstraintBenchmarks.cs:19-117). body_a / body_b: one 32-float BodyDynamics record each (world inertia slot used);
// prestep / accumulated: one lane each, updated in place; the bodies' velocities are written back.
int wide_constraint_iterate(int type_id, float* body_a, float* body_b, float* prestep, float* accumulated, float dt, int iterations) {
    using namespace wide;
    std::unique_ptr<TypeProcessor> processor(type_id >= 0 && type_id < 64 ? CreateProcessor(type_id) : nullptr);
    if (!processor) return -3;
    alignas(32) float pair[64];
    std::memcpy(pair, body_a, 128);
    if (body_b != nullptr) std::memcpy(pair + 32, body_b, 128); else std::memset(pair + 32, 0, 128);
    Bodies bodies{pair, 2};
    processor->Microbenchmark(bodies, prestep, accumulated, dt, iterations);
    std::memcpy(body_a, pair, 128);
    if (body_b != nullptr) std::memcpy(body_b, pair + 32, 128);
    return 0;
}

int wide_math_probe(const float* x, int n, float* sin_out, float* cos_out, float* acos_out) {
    using namespace wide;
    for (int i = 0; i < n; i += 8) {
        VF v = kZero;
        for (int k = 0; k < 8 && i + k < n; ++k) v[k] = x[i + k];
        VF s = MathHelper::Sin(v), c = MathHelper::Cos(v), a = MathHelper::Acos(v);
        for (int k = 0; k < 8 && i + k < n; ++k) {
            sin_out[i + k] = s[k];
            cos_out[i + k] = c[k];
            acos_out[i + k] = a[k];
        }
    }
    return 0;
}
}
