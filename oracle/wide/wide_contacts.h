// oracle/wide — TEST INFRASTRUCTURE (parity unpinned, see wide_vec.h). Convex contact constraints, transcribed bundle-for-bundle from
// BepuPhysics/Constraints/Contact/*.cs. The reference generates Contact2/3/4 from a T4 template (ContactConvexTypes.tt); here the same
// template is a C++ template over the contact count, with the reference's own special cases kept: Contact1 (no friction centre, no 1/N,
// twist lever arm = depth, ContactConvexTypes.cs:303-328,952-979) and the per-count ComputeFrictionCenter overloads (:124-196).
#pragma once
#include "wide_joints.h"

namespace wide {

struct ConvexContactWide { Vector3Wide OffsetA; VF Depth; };  // ContactConvexCommon.cs:6
struct MaterialPropertiesWide {                                // ContactConvexCommon.cs:12
    VF FrictionCoefficient;
    SpringSettingsWide SpringSettings;
    VF MaximumRecoveryVelocity;
};

namespace PenetrationLimit {  // Contact/PenetrationLimit.cs
static inline void ComputeCorrectiveImpulse(const BodyVelocityWide& wsvA, const BodyVelocityWide& wsvB, const Vector3Wide& normal, const Vector3Wide& angularA,
                                            const Vector3Wide& angularB, const VF& biasVelocity, const VF& softnessImpulseScale, const VF& effectiveMass, VF& accumulatedImpulse,
                                            VF& correctiveCSI) {  // :10
    VF csvaLinear, csvaAngular, negatedCSVBLinear, csvbAngular;
    Vector3Wide::Dot(wsvA.Linear, normal, csvaLinear);
    Vector3Wide::Dot(wsvA.Angular, angularA, csvaAngular);
    Vector3Wide::Dot(wsvB.Linear, normal, negatedCSVBLinear);
    Vector3Wide::Dot(wsvB.Angular, angularB, csvbAngular);
    VF negatedCSI = accumulatedImpulse * softnessImpulseScale + (csvaLinear - negatedCSVBLinear + csvaAngular + csvbAngular - biasVelocity) * effectiveMass;
    VF previousAccumulated = accumulatedImpulse;
    accumulatedImpulse = Max(kZero, accumulatedImpulse - negatedCSI);
    correctiveCSI = accumulatedImpulse - previousAccumulated;
}
static inline void UpdatePenetrationDepth(const VF& dt, const Vector3Wide& contactOffsetA, const Vector3Wide& offsetB, const Vector3Wide& normal, const BodyVelocityWide& velocityA,
                                          const BodyVelocityWide& velocityB, VF& penetrationDepth) {  // :29
    Vector3Wide wxra, contactVelocityA, contactOffsetB, wxrb, contactVelocityB, contactVelocityDifference;
    Vector3Wide::CrossWithoutOverlap(velocityA.Angular, contactOffsetA, wxra);
    Vector3Wide::Add(wxra, velocityA.Linear, contactVelocityA);
    Vector3Wide::Subtract(contactOffsetA, offsetB, contactOffsetB);
    Vector3Wide::CrossWithoutOverlap(velocityB.Angular, contactOffsetB, wxrb);
    Vector3Wide::Add(wxrb, velocityB.Linear, contactVelocityB);
    Vector3Wide::Subtract(contactVelocityA, contactVelocityB, contactVelocityDifference);
    VF estimatedDepthChangeVelocity;
    Vector3Wide::Dot(normal, contactVelocityDifference, estimatedDepthChangeVelocity);
    penetrationDepth -= estimatedDepthChangeVelocity * dt;
}
static inline void ApplyImpulse(const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector3Wide& normal, const Vector3Wide& angularA, const Vector3Wide& angularB,
                                const VF& correctiveImpulse, BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :46
    VF linearVelocityChangeA = correctiveImpulse * inertiaA.InverseMass;
    Vector3Wide correctiveVelocityALinearVelocity, correctiveAngularImpulseA, correctiveVelocityAAngularVelocity;
    Vector3Wide::Scale(normal, linearVelocityChangeA, correctiveVelocityALinearVelocity);
    Vector3Wide::Scale(angularA, correctiveImpulse, correctiveAngularImpulseA);
    Symmetric3x3Wide::TransformWithoutOverlap(correctiveAngularImpulseA, inertiaA.InverseInertiaTensor, correctiveVelocityAAngularVelocity);
    VF linearVelocityChangeB = correctiveImpulse * inertiaB.InverseMass;
    Vector3Wide correctiveVelocityBLinearVelocity, correctiveAngularImpulseB, correctiveVelocityBAngularVelocity;
    Vector3Wide::Scale(normal, linearVelocityChangeB, correctiveVelocityBLinearVelocity);
    Vector3Wide::Scale(angularB, correctiveImpulse, correctiveAngularImpulseB);
    Symmetric3x3Wide::TransformWithoutOverlap(correctiveAngularImpulseB, inertiaB.InverseInertiaTensor, correctiveVelocityBAngularVelocity);
    Vector3Wide::Add(wsvA.Linear, correctiveVelocityALinearVelocity, wsvA.Linear);
    Vector3Wide::Add(wsvA.Angular, correctiveVelocityAAngularVelocity, wsvA.Angular);
    Vector3Wide::Subtract(wsvB.Linear, correctiveVelocityBLinearVelocity, wsvB.Linear);
    Vector3Wide::Add(wsvB.Angular, correctiveVelocityBAngularVelocity, wsvB.Angular);
}
static inline void WarmStart(const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector3Wide& normal, const Vector3Wide& contactOffsetA,
                             const Vector3Wide& contactOffsetB, const VF& accumulatedImpulse, BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :67
    Vector3Wide angularA, angularB;
    Vector3Wide::CrossWithoutOverlap(contactOffsetA, normal, angularA);
    Vector3Wide::CrossWithoutOverlap(normal, contactOffsetB, angularB);
    ApplyImpulse(inertiaA, inertiaB, normal, angularA, angularB, accumulatedImpulse, wsvA, wsvB);
}
static inline void Solve(const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector3Wide& normal, const Vector3Wide& contactOffsetA, const Vector3Wide& contactOffsetB,
                         const VF& depth, const VF& positionErrorToVelocity, const VF& effectiveMassCFMScale, const VF& maximumRecoveryVelocity, const VF& inverseDt,
                         const VF& softnessImpulseScale, VF& accumulatedImpulse, BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :78
    Vector3Wide angularA, angularB;
    Vector3Wide::CrossWithoutOverlap(contactOffsetA, normal, angularA);
    Vector3Wide::CrossWithoutOverlap(normal, contactOffsetB, angularB);
    VF angularA0, angularB0;
    Symmetric3x3Wide::VectorSandwich(angularA, inertiaA.InverseInertiaTensor, angularA0);
    Symmetric3x3Wide::VectorSandwich(angularB, inertiaB.InverseInertiaTensor, angularB0);
    VF linear = inertiaA.InverseMass + inertiaB.InverseMass;
    VF effectiveMass = effectiveMassCFMScale / (linear + angularA0 + angularB0);
    VF biasVelocity = Min(depth * inverseDt, Min(depth * positionErrorToVelocity, maximumRecoveryVelocity));
    VF correctiveCSI;
    ComputeCorrectiveImpulse(wsvA, wsvB, normal, angularA, angularB, biasVelocity, softnessImpulseScale, effectiveMass, accumulatedImpulse, correctiveCSI);
    ApplyImpulse(inertiaA, inertiaB, normal, angularA, angularB, correctiveCSI, wsvA, wsvB);
}
}  // namespace PenetrationLimit

namespace PenetrationLimitOneBody {  // Contact/PenetrationLimitOneBody.cs
static inline void ComputeCorrectiveImpulse(const BodyVelocityWide& wsvA, const Vector3Wide& normal, const Vector3Wide& angularA, const VF& biasVelocity, const VF& softnessImpulseScale,
                                            const VF& effectiveMass, VF& accumulatedImpulse, VF& correctiveCSI) {  // :10
    VF csvaLinear, csvaAngular;
    Vector3Wide::Dot(wsvA.Linear, normal, csvaLinear);
    Vector3Wide::Dot(wsvA.Angular, angularA, csvaAngular);
    VF negatedCSI = accumulatedImpulse * softnessImpulseScale + (csvaLinear + csvaAngular - biasVelocity) * effectiveMass;
    VF previousAccumulated = accumulatedImpulse;
    accumulatedImpulse = Max(kZero, accumulatedImpulse - negatedCSI);
    correctiveCSI = accumulatedImpulse - previousAccumulated;
}
static inline void UpdatePenetrationDepth(const VF& dt, const Vector3Wide& contactOffset, const Vector3Wide& normal, const BodyVelocityWide& velocity, VF& penetrationDepth) {  // :27
    Vector3Wide wxr, contactVelocity;
    Vector3Wide::CrossWithoutOverlap(velocity.Angular, contactOffset, wxr);
    Vector3Wide::Add(wxr, velocity.Linear, contactVelocity);
    VF estimatedDepthChange;
    Vector3Wide::Dot(normal, contactVelocity, estimatedDepthChange);
    penetrationDepth -= estimatedDepthChange * dt;
}
static inline void ApplyImpulse(const BodyInertiaWide& inertiaA, const Vector3Wide& normal, const Vector3Wide& angularA, const VF& correctiveImpulse, BodyVelocityWide& wsvA) {  // :40
    VF linearVelocityChangeA = correctiveImpulse * inertiaA.InverseMass;
    Vector3Wide correctiveVelocityALinearVelocity, correctiveAngularImpulseA, correctiveVelocityAAngularVelocity;
    Vector3Wide::Scale(normal, linearVelocityChangeA, correctiveVelocityALinearVelocity);
    Vector3Wide::Scale(angularA, correctiveImpulse, correctiveAngularImpulseA);
    Symmetric3x3Wide::TransformWithoutOverlap(correctiveAngularImpulseA, inertiaA.InverseInertiaTensor, correctiveVelocityAAngularVelocity);
    Vector3Wide::Add(wsvA.Linear, correctiveVelocityALinearVelocity, wsvA.Linear);
    Vector3Wide::Add(wsvA.Angular, correctiveVelocityAAngularVelocity, wsvA.Angular);
}
static inline void WarmStart(const BodyInertiaWide& inertiaA, const Vector3Wide& normal, const Vector3Wide& contactOffsetA, const VF& accumulatedImpulse, BodyVelocityWide& wsvA) {  // :52
    Vector3Wide angularA;
    Vector3Wide::CrossWithoutOverlap(contactOffsetA, normal, angularA);
    ApplyImpulse(inertiaA, normal, angularA, accumulatedImpulse, wsvA);
}
static inline void Solve(const BodyInertiaWide& inertiaA, const Vector3Wide& normal, const Vector3Wide& contactOffsetA, const VF& depth, const VF& positionErrorToVelocity,
                         const VF& effectiveMassCFMScale, const VF& maximumRecoveryVelocity, const VF& inverseDt, const VF& softnessImpulseScale, VF& accumulatedImpulse,
                         BodyVelocityWide& wsvA) {  // :60
    Vector3Wide angularA;
    Vector3Wide::CrossWithoutOverlap(contactOffsetA, normal, angularA);
    VF angularA0;
    Symmetric3x3Wide::VectorSandwich(angularA, inertiaA.InverseInertiaTensor, angularA0);
    VF effectiveMass = effectiveMassCFMScale / (inertiaA.InverseMass + angularA0);
    VF biasVelocity = Min(depth * inverseDt, Min(depth * positionErrorToVelocity, maximumRecoveryVelocity));
    VF correctiveCSI;
    ComputeCorrectiveImpulse(wsvA, normal, angularA, biasVelocity, softnessImpulseScale, effectiveMass, accumulatedImpulse, correctiveCSI);
    ApplyImpulse(inertiaA, normal, angularA, correctiveCSI, wsvA);
}
}  // namespace PenetrationLimitOneBody

static inline void SandwichScale(const Matrix2x3Wide& m, const VF& scale, Symmetric2x2Wide& result) {  // BepuUtilities/Symmetric2x2Wide.cs:23
    result.XX = scale * (m.X.X * m.X.X + m.X.Y * m.X.Y + m.X.Z * m.X.Z);
    result.YX = scale * (m.Y.X * m.X.X + m.Y.Y * m.X.Y + m.Y.Z * m.X.Z);
    result.YY = scale * (m.Y.X * m.Y.X + m.Y.Y * m.Y.Y + m.Y.Z * m.Y.Z);
}

namespace TangentFriction {  // Contact/TangentFriction.cs
struct Jacobians { Matrix2x3Wide LinearA, AngularA, AngularB; };  // :12
static inline void ComputeJacobians(const Vector3Wide& tangentX, const Vector3Wide& tangentY, const Vector3Wide& offsetA, const Vector3Wide& offsetB, Jacobians& jacobians) {  // :20
    jacobians.LinearA.X = tangentX;
    jacobians.LinearA.Y = tangentY;
    Vector3Wide::CrossWithoutOverlap(offsetA, tangentX, jacobians.AngularA.X);
    Vector3Wide::CrossWithoutOverlap(offsetA, tangentY, jacobians.AngularA.Y);
    Vector3Wide::CrossWithoutOverlap(tangentX, offsetB, jacobians.AngularB.X);
    Vector3Wide::CrossWithoutOverlap(tangentY, offsetB, jacobians.AngularB.Y);
}
static inline void ApplyImpulse(const Jacobians& jacobians, const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector2Wide& correctiveImpulse,
                                BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :59
    Vector3Wide linearImpulseA, angularImpulseA, angularImpulseB;
    Matrix2x3Wide::Transform(correctiveImpulse, jacobians.LinearA, linearImpulseA);
    Matrix2x3Wide::Transform(correctiveImpulse, jacobians.AngularA, angularImpulseA);
    Matrix2x3Wide::Transform(correctiveImpulse, jacobians.AngularB, angularImpulseB);
    BodyVelocityWide correctiveVelocityA, correctiveVelocityB;
    Vector3Wide::Scale(linearImpulseA, inertiaA.InverseMass, correctiveVelocityA.Linear);
    Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseA, inertiaA.InverseInertiaTensor, correctiveVelocityA.Angular);
    Vector3Wide::Scale(linearImpulseA, inertiaB.InverseMass, correctiveVelocityB.Linear);
    Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseB, inertiaB.InverseInertiaTensor, correctiveVelocityB.Angular);
    Vector3Wide::Add(wsvA.Linear, correctiveVelocityA.Linear, wsvA.Linear);
    Vector3Wide::Add(wsvA.Angular, correctiveVelocityA.Angular, wsvA.Angular);
    Vector3Wide::Subtract(wsvB.Linear, correctiveVelocityB.Linear, wsvB.Linear);
    Vector3Wide::Add(wsvB.Angular, correctiveVelocityB.Angular, wsvB.Angular);
}
static inline void ComputeCorrectiveImpulse(const BodyVelocityWide& wsvA, const BodyVelocityWide& wsvB, const Symmetric2x2Wide& effectiveMass, const Jacobians& jacobians,
                                            const VF& maximumImpulse, Vector2Wide& accumulatedImpulse, Vector2Wide& correctiveCSI) {  // :77
    Vector2Wide csvaLinear, csvaAngular, csvbLinear, csvbAngular;
    Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvA.Linear, jacobians.LinearA, csvaLinear);
    Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvA.Angular, jacobians.AngularA, csvaAngular);
    Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvB.Linear, jacobians.LinearA, csvbLinear);
    Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvB.Angular, jacobians.AngularB, csvbAngular);
    Vector2Wide csvLinear, csvAngular, csv, csi;
    Vector2Wide::Subtract(csvbLinear, csvaLinear, csvLinear);
    Vector2Wide::Add(csvaAngular, csvbAngular, csvAngular);
    Vector2Wide::Subtract(csvLinear, csvAngular, csv);
    Symmetric2x2Wide::TransformWithoutOverlap(csv, effectiveMass, csi);
    Vector2Wide previousAccumulated = accumulatedImpulse;
    Vector2Wide::Add(accumulatedImpulse, csi, accumulatedImpulse);
    VF accumulatedMagnitude;
    Vector2Wide::Length(accumulatedImpulse, accumulatedMagnitude);
    VF scale = Min(kOne, maximumImpulse / Max(vf(1e-16f), accumulatedMagnitude));
    Vector2Wide::Scale(accumulatedImpulse, scale, accumulatedImpulse);
    Vector2Wide::Subtract(accumulatedImpulse, previousAccumulated, correctiveCSI);
}
static inline void WarmStart(const Vector3Wide& tangentX, const Vector3Wide& tangentY, const Vector3Wide& offsetToManifoldCenterA, const Vector3Wide& offsetToManifoldCenterB,
                             const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector2Wide& accumulatedImpulse, BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :107
    Jacobians jacobians;
    ComputeJacobians(tangentX, tangentY, offsetToManifoldCenterA, offsetToManifoldCenterB, jacobians);
    ApplyImpulse(jacobians, inertiaA, inertiaB, accumulatedImpulse, wsvA, wsvB);
}
static inline void Solve(const Vector3Wide& tangentX, const Vector3Wide& tangentY, const Vector3Wide& offsetToManifoldCenterA, const Vector3Wide& offsetToManifoldCenterB,
                         const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const VF& maximumImpulse, Vector2Wide& accumulatedImpulse, BodyVelocityWide& wsvA,
                         BodyVelocityWide& wsvB) {  // :118
    Jacobians jacobians;
    ComputeJacobians(tangentX, tangentY, offsetToManifoldCenterA, offsetToManifoldCenterB, jacobians);
    Symmetric2x2Wide linearContributionA, linearContributionB, angularContributionA, angularContributionB;
    SandwichScale(jacobians.LinearA, inertiaA.InverseMass, linearContributionA);
    SandwichScale(jacobians.LinearA, inertiaB.InverseMass, linearContributionB);
    Symmetric3x3Wide::MatrixSandwich(jacobians.AngularA, inertiaA.InverseInertiaTensor, angularContributionA);
    Symmetric3x3Wide::MatrixSandwich(jacobians.AngularB, inertiaB.InverseInertiaTensor, angularContributionB);
    Symmetric2x2Wide linear, angular, inverseEffectiveMass, effectiveMass;
    Symmetric2x2Wide::Add(linearContributionA, linearContributionB, linear);
    Symmetric2x2Wide::Add(angularContributionA, angularContributionB, angular);
    Symmetric2x2Wide::Add(linear, angular, inverseEffectiveMass);
    Symmetric2x2Wide::InvertWithoutOverlap(inverseEffectiveMass, effectiveMass);
    Vector2Wide correctiveCSI;
    ComputeCorrectiveImpulse(wsvA, wsvB, effectiveMass, jacobians, maximumImpulse, accumulatedImpulse, correctiveCSI);
    ApplyImpulse(jacobians, inertiaA, inertiaB, correctiveCSI, wsvA, wsvB);
}
}  // namespace TangentFriction

namespace TangentFrictionOneBody {  // Contact/TangentFrictionOneBody.cs
struct Jacobians { Matrix2x3Wide LinearA, AngularA; };  // :12
static inline void ComputeJacobians(const Vector3Wide& tangentX, const Vector3Wide& tangentY, const Vector3Wide& offsetA, Jacobians& jacobians) {  // :20
    jacobians.LinearA.X = tangentX;
    jacobians.LinearA.Y = tangentY;
    Vector3Wide::CrossWithoutOverlap(offsetA, tangentX, jacobians.AngularA.X);
    Vector3Wide::CrossWithoutOverlap(offsetA, tangentY, jacobians.AngularA.Y);
}
static inline void ApplyImpulse(const Jacobians& jacobians, const BodyInertiaWide& inertiaA, const Vector2Wide& correctiveImpulse, BodyVelocityWide& wsvA) {  // :33
    Vector3Wide linearImpulseA, angularImpulseA;
    Matrix2x3Wide::Transform(correctiveImpulse, jacobians.LinearA, linearImpulseA);
    Matrix2x3Wide::Transform(correctiveImpulse, jacobians.AngularA, angularImpulseA);
    BodyVelocityWide correctiveVelocityA;
    Vector3Wide::Scale(linearImpulseA, inertiaA.InverseMass, correctiveVelocityA.Linear);
    Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseA, inertiaA.InverseInertiaTensor, correctiveVelocityA.Angular);
    Vector3Wide::Add(wsvA.Linear, correctiveVelocityA.Linear, wsvA.Linear);
    Vector3Wide::Add(wsvA.Angular, correctiveVelocityA.Angular, wsvA.Angular);
}
static inline void ComputeCorrectiveImpulse(const BodyVelocityWide& wsvA, const Symmetric2x2Wide& effectiveMass, const Jacobians& jacobians, const VF& maximumImpulse,
                                            Vector2Wide& accumulatedImpulse, Vector2Wide& correctiveCSI) {  // :46
    Vector2Wide csvaLinear, csvaAngular, csv, negativeCSI;
    Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvA.Linear, jacobians.LinearA, csvaLinear);
    Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvA.Angular, jacobians.AngularA, csvaAngular);
    Vector2Wide::Add(csvaLinear, csvaAngular, csv);
    Symmetric2x2Wide::TransformWithoutOverlap(csv, effectiveMass, negativeCSI);
    Vector2Wide previousAccumulated = accumulatedImpulse;
    Vector2Wide::Subtract(accumulatedImpulse, negativeCSI, accumulatedImpulse);
    VF accumulatedMagnitude;
    Vector2Wide::Length(accumulatedImpulse, accumulatedMagnitude);
    VF scale = Min(kOne, maximumImpulse / Max(vf(1e-16f), accumulatedMagnitude));
    Vector2Wide::Scale(accumulatedImpulse, scale, accumulatedImpulse);
    Vector2Wide::Subtract(accumulatedImpulse, previousAccumulated, correctiveCSI);
}
static inline void WarmStart(const Vector3Wide& tangentX, const Vector3Wide& tangentY, const Vector3Wide& offsetToManifoldCenterA, const BodyInertiaWide& inertiaA,
                             const Vector2Wide& accumulatedImpulse, BodyVelocityWide& wsvA) {  // :68
    Jacobians jacobians;
    ComputeJacobians(tangentX, tangentY, offsetToManifoldCenterA, jacobians);
    ApplyImpulse(jacobians, inertiaA, accumulatedImpulse, wsvA);
}
static inline void Solve(const Vector3Wide& tangentX, const Vector3Wide& tangentY, const Vector3Wide& offsetToManifoldCenterA, const BodyInertiaWide& inertiaA, const VF& maximumImpulse,
                         Vector2Wide& accumulatedImpulse, BodyVelocityWide& wsvA) {  // :77
    Jacobians jacobians;
    ComputeJacobians(tangentX, tangentY, offsetToManifoldCenterA, jacobians);
    Symmetric2x2Wide linearContributionA, angularContributionA, inverseEffectiveMass, effectiveMass;
    SandwichScale(jacobians.LinearA, inertiaA.InverseMass, linearContributionA);
    Symmetric3x3Wide::MatrixSandwich(jacobians.AngularA, inertiaA.InverseInertiaTensor, angularContributionA);
    Symmetric2x2Wide::Add(linearContributionA, angularContributionA, inverseEffectiveMass);
    Symmetric2x2Wide::InvertWithoutOverlap(inverseEffectiveMass, effectiveMass);
    Vector2Wide correctiveCSI;
    ComputeCorrectiveImpulse(wsvA, effectiveMass, jacobians, maximumImpulse, accumulatedImpulse, correctiveCSI);
    ApplyImpulse(jacobians, inertiaA, correctiveCSI, wsvA);
}
}  // namespace TangentFrictionOneBody

namespace TwistFriction {  // Contact/TwistFriction.cs
static inline void ApplyImpulse(const Vector3Wide& angularJacobianA, const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const VF& correctiveImpulse, BodyVelocityWide& wsvA,
                                BodyVelocityWide& wsvB) {  // :16
    Vector3Wide worldCorrectiveImpulseA, worldCorrectiveVelocityA, worldCorrectiveVelocityB;
    Vector3Wide::Scale(angularJacobianA, correctiveImpulse, worldCorrectiveImpulseA);
    Symmetric3x3Wide::TransformWithoutOverlap(worldCorrectiveImpulseA, inertiaA.InverseInertiaTensor, worldCorrectiveVelocityA);
    Symmetric3x3Wide::TransformWithoutOverlap(worldCorrectiveImpulseA, inertiaB.InverseInertiaTensor, worldCorrectiveVelocityB);
    Vector3Wide::Add(wsvA.Angular, worldCorrectiveVelocityA, wsvA.Angular);
    Vector3Wide::Subtract(wsvB.Angular, worldCorrectiveVelocityB, wsvB.Angular);
}
static inline void ComputeCorrectiveImpulse(const Vector3Wide& angularJacobianA, const VF& effectiveMass, const BodyVelocityWide& wsvA, const BodyVelocityWide& wsvB,
                                            const VF& maximumImpulse, VF& accumulatedImpulse, VF& correctiveCSI) {  // :27
    VF csvA, negatedCSVB;
    Vector3Wide::Dot(wsvA.Angular, angularJacobianA, csvA);
    Vector3Wide::Dot(wsvB.Angular, angularJacobianA, negatedCSVB);
    VF negatedCSI = (csvA - negatedCSVB) * effectiveMass;
    VF previousAccumulated = accumulatedImpulse;
    accumulatedImpulse = Min(maximumImpulse, Max(neg(maximumImpulse), accumulatedImpulse - negatedCSI));
    correctiveCSI = accumulatedImpulse - previousAccumulated;
}
static inline void WarmStart(const Vector3Wide& angularJacobianA, const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const VF& accumulatedImpulse, BodyVelocityWide& wsvA,
                             BodyVelocityWide& wsvB) {  // :43
    ApplyImpulse(angularJacobianA, inertiaA, inertiaB, accumulatedImpulse, wsvA, wsvB);
}
static inline void Solve(const Vector3Wide& angularJacobianA, const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const VF& maximumImpulse, VF& accumulatedImpulse,
                         BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :50
    VF angularA, angularB;
    Symmetric3x3Wide::VectorSandwich(angularJacobianA, inertiaA.InverseInertiaTensor, angularA);
    Symmetric3x3Wide::VectorSandwich(angularJacobianA, inertiaB.InverseInertiaTensor, angularB);
    VF inverseEffectiveMass = angularA + angularB;
    VI inverseIsZero = Equals(kZero, inverseEffectiveMass);
    VF effectiveMass = ConditionalSelect(inverseIsZero, kZero, kOne / inverseEffectiveMass);
    VF correctiveCSI;
    ComputeCorrectiveImpulse(angularJacobianA, effectiveMass, wsvA, wsvB, maximumImpulse, accumulatedImpulse, correctiveCSI);
    ApplyImpulse(angularJacobianA, inertiaA, inertiaB, correctiveCSI, wsvA, wsvB);
}
}  // namespace TwistFriction

namespace TwistFrictionOneBody {  // Contact/TwistFrictionOneBody.cs
static inline void ApplyImpulse(const Vector3Wide& angularJacobianA, const BodyInertiaWide& inertiaA, const VF& correctiveImpulse, BodyVelocityWide& wsvA) {  // :19
    Vector3Wide worldCorrectiveImpulseA, worldCorrectiveVelocityA;
    Vector3Wide::Scale(angularJacobianA, correctiveImpulse, worldCorrectiveImpulseA);
    Symmetric3x3Wide::TransformWithoutOverlap(worldCorrectiveImpulseA, inertiaA.InverseInertiaTensor, worldCorrectiveVelocityA);
    Vector3Wide::Add(wsvA.Angular, worldCorrectiveVelocityA, wsvA.Angular);
}
static inline void ComputeCorrectiveImpulse(const Vector3Wide& angularJacobianA, const VF& effectiveMass, const BodyVelocityWide& wsvA, const VF& maximumImpulse, VF& accumulatedImpulse,
                                            VF& correctiveCSI) {  // :28
    VF csvA;
    Vector3Wide::Dot(wsvA.Angular, angularJacobianA, csvA);
    VF negativeCSI = csvA * effectiveMass;
    VF previousAccumulated = accumulatedImpulse;
    accumulatedImpulse = Min(maximumImpulse, Max(neg(maximumImpulse), accumulatedImpulse - negativeCSI));
    correctiveCSI = accumulatedImpulse - previousAccumulated;
}
static inline void WarmStart(const Vector3Wide& angularJacobianA, const BodyInertiaWide& inertiaA, const VF& accumulatedImpulse, BodyVelocityWide& wsvA) {  // :44
    ApplyImpulse(angularJacobianA, inertiaA, accumulatedImpulse, wsvA);
}
static inline void Solve(const Vector3Wide& angularJacobianA, const BodyInertiaWide& inertiaA, const VF& maximumImpulse, VF& accumulatedImpulse, BodyVelocityWide& wsvA) {  // :50
    VF angularA;
    Symmetric3x3Wide::VectorSandwich(angularJacobianA, inertiaA.InverseInertiaTensor, angularA);
    VI inverseIsZero = Equals(kZero, angularA);
    VF effectiveMass = ConditionalSelect(inverseIsZero, kZero, kOne / angularA);
    VF correctiveCSI;
    ComputeCorrectiveImpulse(angularJacobianA, effectiveMass, wsvA, maximumImpulse, accumulatedImpulse, correctiveCSI);
    ApplyImpulse(angularJacobianA, inertiaA, correctiveCSI, wsvA);
}
}  // namespace TwistFrictionOneBody

namespace FrictionHelpers {  // ContactConvexTypes.cs:121-198
static inline VF Weight(const VF& depth) { return ConditionalSelect(LessThan(depth, kZero), kZero, kOne); }
static inline void ComputeFrictionCenter(const Vector3Wide& offsetA0, const Vector3Wide& offsetA1, const VF& depth0, const VF& depth1, Vector3Wide& center) {  // :124
    VF weight0 = Weight(depth0);
    VF weight1 = Weight(depth1);
    VF weightSum = weight0 + weight1;
    VI useFallback = Equals(weightSum, kZero);
    weightSum = ConditionalSelect(useFallback, vf(2), weightSum);
    VF inverseWeightSum = kOne / weightSum;
    weight0 = ConditionalSelect(useFallback, inverseWeightSum, weight0 * inverseWeightSum);
    weight1 = ConditionalSelect(useFallback, inverseWeightSum, weight1 * inverseWeightSum);
    Vector3Wide a0Contribution, a1Contribution;
    Vector3Wide::Scale(offsetA0, weight0, a0Contribution);
    Vector3Wide::Scale(offsetA1, weight1, a1Contribution);
    Vector3Wide::Add(a0Contribution, a1Contribution, center);
}
static inline void ComputeFrictionCenter(const Vector3Wide& offsetA0, const Vector3Wide& offsetA1, const Vector3Wide& offsetA2, const VF& depth0, const VF& depth1, const VF& depth2,
                                         Vector3Wide& center) {  // :145
    VF weight0 = Weight(depth0);
    VF weight1 = Weight(depth1);
    VF weight2 = Weight(depth2);
    VF weightSum = weight0 + weight1 + weight2;
    VI useFallback = Equals(weightSum, kZero);
    weightSum = ConditionalSelect(useFallback, vf(3), weightSum);
    VF inverseWeightSum = kOne / weightSum;
    weight0 = ConditionalSelect(useFallback, inverseWeightSum, weight0 * inverseWeightSum);
    weight1 = ConditionalSelect(useFallback, inverseWeightSum, weight1 * inverseWeightSum);
    weight2 = ConditionalSelect(useFallback, inverseWeightSum, weight2 * inverseWeightSum);
    Vector3Wide a0Contribution, a1Contribution, a2Contribution, a0a1;
    Vector3Wide::Scale(offsetA0, weight0, a0Contribution);
    Vector3Wide::Scale(offsetA1, weight1, a1Contribution);
    Vector3Wide::Scale(offsetA2, weight2, a2Contribution);
    Vector3Wide::Add(a0Contribution, a1Contribution, a0a1);
    Vector3Wide::Add(a0a1, a2Contribution, center);
}
static inline void ComputeFrictionCenter(const Vector3Wide& offsetA0, const Vector3Wide& offsetA1, const Vector3Wide& offsetA2, const Vector3Wide& offsetA3, const VF& depth0,
                                         const VF& depth1, const VF& depth2, const VF& depth3, Vector3Wide& center) {  // :170
    VF weight0 = Weight(depth0);
    VF weight1 = Weight(depth1);
    VF weight2 = Weight(depth2);
    VF weight3 = Weight(depth3);
    VF weightSum = weight0 + weight1 + weight2 + weight3;
    VI useFallback = Equals(weightSum, kZero);
    weightSum = ConditionalSelect(useFallback, vf(4), weightSum);
    VF inverseWeightSum = kOne / weightSum;
    weight0 = ConditionalSelect(useFallback, inverseWeightSum, weight0 * inverseWeightSum);
    weight1 = ConditionalSelect(useFallback, inverseWeightSum, weight1 * inverseWeightSum);
    weight2 = ConditionalSelect(useFallback, inverseWeightSum, weight2 * inverseWeightSum);
    weight3 = ConditionalSelect(useFallback, inverseWeightSum, weight3 * inverseWeightSum);
    Vector3Wide a0Contribution, a1Contribution, a2Contribution, a3Contribution, a0a1, a2a3;
    Vector3Wide::Scale(offsetA0, weight0, a0Contribution);
    Vector3Wide::Scale(offsetA1, weight1, a1Contribution);
    Vector3Wide::Scale(offsetA2, weight2, a2Contribution);
    Vector3Wide::Scale(offsetA3, weight3, a3Contribution);
    Vector3Wide::Add(a0Contribution, a1Contribution, a0a1);
    Vector3Wide::Add(a2Contribution, a3Contribution, a2a3);
    Vector3Wide::Add(a0a1, a2a3, center);
}
}  // namespace FrictionHelpers

static inline VF Distance(const Vector3Wide& a, const Vector3Wide& b) {  // Vector3Wide.Distance, value-returning form, Vector3Wide.cs:657
    VF x = b.X - a.X;
    VF y = b.Y - a.Y;
    VF z = b.Z - a.Z;
    return SquareRoot(x * x + y * y + z * z);
}

// Prestep / accumulated impulse layouts (field order = AOSOA float order): ContactConvexTypes.cs:258-266,406-415,901-910,1418-1430 and :92-99.
template <int N> struct ContactOneBodyPrestepData { ConvexContactWide Contact[N]; Vector3Wide Normal; MaterialPropertiesWide MaterialProperties; };
template <int N> struct ContactPrestepData { ConvexContactWide Contact[N]; Vector3Wide OffsetB; Vector3Wide Normal; MaterialPropertiesWide MaterialProperties; };
template <int N> struct ContactAccumulatedImpulses { Vector2Wide Tangent; VF Penetration[N]; VF Twist; };

template <int N> static inline void ManifoldCenter(const ConvexContactWide* c, Vector3Wide& center) {
    if constexpr (N == 2) FrictionHelpers::ComputeFrictionCenter(c[0].OffsetA, c[1].OffsetA, c[0].Depth, c[1].Depth, center);
    else if constexpr (N == 3) FrictionHelpers::ComputeFrictionCenter(c[0].OffsetA, c[1].OffsetA, c[2].OffsetA, c[0].Depth, c[1].Depth, c[2].Depth, center);
    else FrictionHelpers::ComputeFrictionCenter(c[0].OffsetA, c[1].OffsetA, c[2].OffsetA, c[3].OffsetA, c[0].Depth, c[1].Depth, c[2].Depth, c[3].Depth, center);
}

// Contact{N}OneBodyFunctions: ContactConvexTypes.cs:292-332 (N=1), :441-487, :602-652, :773-827.
template <int N> struct ContactOneBodyFunctions {
    typedef ContactOneBodyPrestepData<N> Prestep;
    typedef ContactAccumulatedImpulses<N> Impulses;
    static void IncrementallyUpdateForSubstep(const VF& dt, const BodyVelocityWide& velocityA, Prestep& prestep) {
        for (int i = 0; i < N; ++i)
            PenetrationLimitOneBody::UpdatePenetrationDepth(dt, prestep.Contact[i].OffsetA, prestep.Normal, velocityA, prestep.Contact[i].Depth);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, Prestep& prestep, Impulses& accumulatedImpulses,
                          BodyVelocityWide& wsvA) {
        Vector3Wide x, z;
        Helpers::BuildOrthonormalBasis(prestep.Normal, x, z);
        if constexpr (N == 1) {
            TangentFrictionOneBody::WarmStart(x, z, prestep.Contact[0].OffsetA, inertiaA, accumulatedImpulses.Tangent, wsvA);
        } else {
            Vector3Wide offsetToManifoldCenterA;
            ManifoldCenter<N>(prestep.Contact, offsetToManifoldCenterA);
            TangentFrictionOneBody::WarmStart(x, z, offsetToManifoldCenterA, inertiaA, accumulatedImpulses.Tangent, wsvA);
        }
        for (int i = 0; i < N; ++i)
            PenetrationLimitOneBody::WarmStart(inertiaA, prestep.Normal, prestep.Contact[i].OffsetA, accumulatedImpulses.Penetration[i], wsvA);
        TwistFrictionOneBody::WarmStart(prestep.Normal, inertiaA, accumulatedImpulses.Twist, wsvA);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, float dt, float inverseDt, Prestep& prestep,
                      Impulses& accumulatedImpulses, BodyVelocityWide& wsvA) {
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.MaterialProperties.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF inverseDtWide = vf(inverseDt);
        for (int i = 0; i < N; ++i)
            PenetrationLimitOneBody::Solve(inertiaA, prestep.Normal, prestep.Contact[i].OffsetA, prestep.Contact[i].Depth, positionErrorToVelocity, effectiveMassCFMScale,
                                           prestep.MaterialProperties.MaximumRecoveryVelocity, inverseDtWide, softnessImpulseScale, accumulatedImpulses.Penetration[i], wsvA);
        Vector3Wide x, z;
        Helpers::BuildOrthonormalBasis(prestep.Normal, x, z);
        if constexpr (N == 1) {
            VF maximumTangentImpulse = prestep.MaterialProperties.FrictionCoefficient * (accumulatedImpulses.Penetration[0]);
            TangentFrictionOneBody::Solve(x, z, prestep.Contact[0].OffsetA, inertiaA, maximumTangentImpulse, accumulatedImpulses.Tangent, wsvA);
            VF maximumTwistImpulse = prestep.MaterialProperties.FrictionCoefficient * accumulatedImpulses.Penetration[0] * Max(kZero, prestep.Contact[0].Depth);
            TwistFrictionOneBody::Solve(prestep.Normal, inertiaA, maximumTwistImpulse, accumulatedImpulses.Twist, wsvA);
        } else {
            VF premultipliedFrictionCoefficient = vf(1.0f / (float)N) * prestep.MaterialProperties.FrictionCoefficient;
            VF penetrationSum = accumulatedImpulses.Penetration[0] + accumulatedImpulses.Penetration[1];
            for (int i = 2; i < N; ++i) penetrationSum = penetrationSum + accumulatedImpulses.Penetration[i];
            VF maximumTangentImpulse = premultipliedFrictionCoefficient * (penetrationSum);
            Vector3Wide offsetToManifoldCenterA;
            ManifoldCenter<N>(prestep.Contact, offsetToManifoldCenterA);
            TangentFrictionOneBody::Solve(x, z, offsetToManifoldCenterA, inertiaA, maximumTangentImpulse, accumulatedImpulses.Tangent, wsvA);
            VF leverSum = accumulatedImpulses.Penetration[0] * Distance(offsetToManifoldCenterA, prestep.Contact[0].OffsetA) +
                          accumulatedImpulses.Penetration[1] * Distance(offsetToManifoldCenterA, prestep.Contact[1].OffsetA);
            for (int i = 2; i < N; ++i) leverSum = leverSum + accumulatedImpulses.Penetration[i] * Distance(offsetToManifoldCenterA, prestep.Contact[i].OffsetA);
            VF maximumTwistImpulse = premultipliedFrictionCoefficient * (leverSum);
            TwistFrictionOneBody::Solve(prestep.Normal, inertiaA, maximumTwistImpulse, accumulatedImpulses.Twist, wsvA);
        }
    }
};

// Contact{N}Functions: ContactConvexTypes.cs:941-983 (N=1), :1103-1151, :1277-1329, :1461-1517.
template <int N> struct ContactFunctions {
    typedef ContactPrestepData<N> Prestep;
    typedef ContactAccumulatedImpulses<N> Impulses;
    static void IncrementallyUpdateForSubstep(const VF& dt, const BodyVelocityWide& velocityA, const BodyVelocityWide& velocityB, Prestep& prestep) {
        for (int i = 0; i < N; ++i)
            PenetrationLimit::UpdatePenetrationDepth(dt, prestep.Contact[i].OffsetA, prestep.OffsetB, prestep.Normal, velocityA, velocityB, prestep.Contact[i].Depth);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {
        Vector3Wide x, z;
        Helpers::BuildOrthonormalBasis(prestep.Normal, x, z);
        Vector3Wide offsetToManifoldCenterB;
        if constexpr (N == 1) {
            Vector3Wide::Subtract(prestep.Contact[0].OffsetA, prestep.OffsetB, offsetToManifoldCenterB);
            TangentFriction::WarmStart(x, z, prestep.Contact[0].OffsetA, offsetToManifoldCenterB, inertiaA, inertiaB, accumulatedImpulses.Tangent, wsvA, wsvB);
        } else {
            Vector3Wide offsetToManifoldCenterA;
            ManifoldCenter<N>(prestep.Contact, offsetToManifoldCenterA);
            Vector3Wide::Subtract(offsetToManifoldCenterA, prestep.OffsetB, offsetToManifoldCenterB);
            TangentFriction::WarmStart(x, z, offsetToManifoldCenterA, offsetToManifoldCenterB, inertiaA, inertiaB, accumulatedImpulses.Tangent, wsvA, wsvB);
        }
        for (int i = 0; i < N; ++i)
            PenetrationLimit::WarmStart(inertiaA, inertiaB, prestep.Normal, prestep.Contact[i].OffsetA, prestep.Contact[i].OffsetA - prestep.OffsetB, accumulatedImpulses.Penetration[i],
                                        wsvA, wsvB);
        TwistFriction::WarmStart(prestep.Normal, inertiaA, inertiaB, accumulatedImpulses.Twist, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.MaterialProperties.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF inverseDtWide = vf(inverseDt);
        for (int i = 0; i < N; ++i)
            PenetrationLimit::Solve(inertiaA, inertiaB, prestep.Normal, prestep.Contact[i].OffsetA, prestep.Contact[i].OffsetA - prestep.OffsetB, prestep.Contact[i].Depth,
                                    positionErrorToVelocity, effectiveMassCFMScale, prestep.MaterialProperties.MaximumRecoveryVelocity, inverseDtWide, softnessImpulseScale,
                                    accumulatedImpulses.Penetration[i], wsvA, wsvB);
        Vector3Wide x, z;
        Helpers::BuildOrthonormalBasis(prestep.Normal, x, z);
        Vector3Wide offsetToManifoldCenterB;
        if constexpr (N == 1) {
            VF maximumTangentImpulse = prestep.MaterialProperties.FrictionCoefficient * (accumulatedImpulses.Penetration[0]);
            Vector3Wide::Subtract(prestep.Contact[0].OffsetA, prestep.OffsetB, offsetToManifoldCenterB);
            TangentFriction::Solve(x, z, prestep.Contact[0].OffsetA, offsetToManifoldCenterB, inertiaA, inertiaB, maximumTangentImpulse, accumulatedImpulses.Tangent, wsvA, wsvB);
            VF maximumTwistImpulse = prestep.MaterialProperties.FrictionCoefficient * accumulatedImpulses.Penetration[0] * Max(kZero, prestep.Contact[0].Depth);
            TwistFriction::Solve(prestep.Normal, inertiaA, inertiaB, maximumTwistImpulse, accumulatedImpulses.Twist, wsvA, wsvB);
        } else {
            VF premultipliedFrictionCoefficient = vf(1.0f / (float)N) * prestep.MaterialProperties.FrictionCoefficient;
            VF penetrationSum = accumulatedImpulses.Penetration[0] + accumulatedImpulses.Penetration[1];
            for (int i = 2; i < N; ++i) penetrationSum = penetrationSum + accumulatedImpulses.Penetration[i];
            VF maximumTangentImpulse = premultipliedFrictionCoefficient * (penetrationSum);
            Vector3Wide offsetToManifoldCenterA;
            ManifoldCenter<N>(prestep.Contact, offsetToManifoldCenterA);
            Vector3Wide::Subtract(offsetToManifoldCenterA, prestep.OffsetB, offsetToManifoldCenterB);
            TangentFriction::Solve(x, z, offsetToManifoldCenterA, offsetToManifoldCenterB, inertiaA, inertiaB, maximumTangentImpulse, accumulatedImpulses.Tangent, wsvA, wsvB);
            VF leverSum = accumulatedImpulses.Penetration[0] * Distance(offsetToManifoldCenterA, prestep.Contact[0].OffsetA) +
                          accumulatedImpulses.Penetration[1] * Distance(offsetToManifoldCenterA, prestep.Contact[1].OffsetA);
            for (int i = 2; i < N; ++i) leverSum = leverSum + accumulatedImpulses.Penetration[i] * Distance(offsetToManifoldCenterA, prestep.Contact[i].OffsetA);
            VF maximumTwistImpulse = premultipliedFrictionCoefficient * (leverSum);
            TwistFriction::Solve(prestep.Normal, inertiaA, inertiaB, maximumTwistImpulse, accumulatedImpulses.Twist, wsvA, wsvB);
        }
    }
};

}  // namespace wide
