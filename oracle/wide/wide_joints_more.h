// oracle/wide — TEST INFRASTRUCTURE (parity unpinned: no run of the reference behind it). Constraint types beyond the sixteen of SURVEY.md 8(a), transcribed from the C# alone like the rest of this directory
// (the widened set gets its second, independent reading one type at a time).
#pragma once
#include "wide_joints.h"
#include "wide_contacts.h"

namespace wide {

// QuaternionWide.GetAxisAngleFromQuaternion (BepuUtilities/QuaternionWide.cs:227-243)
static inline void GetAxisAngleFromQuaternion(const QuaternionWide& q, Vector3Wide& axis, VF& angle) {
    VI shouldNegate = LessThan(q.W, kZero);
    axis.X = ConditionalSelect(shouldNegate, neg(q.X), q.X);
    axis.Y = ConditionalSelect(shouldNegate, neg(q.Y), q.Y);
    axis.Z = ConditionalSelect(shouldNegate, neg(q.Z), q.Z);
    VF qw = ConditionalSelect(shouldNegate, neg(q.W), q.W);
    VF axisLength;
    Vector3Wide::Length(axis, axisLength);
    Vector3Wide::Scale(axis, kOne / axisLength, axis);
    VI useFallback = LessThan(axisLength, vf(1e-14f));
    axis.X = ConditionalSelect(useFallback, kOne, axis.X);
    axis.Y = ConditionalSelect(useFallback, kZero, axis.Y);
    axis.Z = ConditionalSelect(useFallback, kZero, axis.Z);
    VF halfAngle = MathHelper::Acos(qw);
    angle = vf(2) * halfAngle;
}

namespace ServoSettingsMore {  // the Vector3Wide overloads of ServoSettingsWide.ComputeClampedBiasVelocity (BepuPhysics/Constraints/ServoSettings.cs:116-143)
static inline void ComputeClampedBiasVelocity(const Vector3Wide& errorAxis, const VF& errorLength, const VF& positionErrorToBiasVelocity, const ServoSettingsWide& servoSettings, float dt,
                                              float inverseDt, Vector3Wide& clampedBiasVelocity, VF& maximumImpulse) {  // :116
    VF baseSpeed = Min(servoSettings.BaseSpeed, errorLength * vf(inverseDt));
    VF unclampedBiasSpeed = errorLength * positionErrorToBiasVelocity;
    VF targetSpeed = Max(baseSpeed, unclampedBiasSpeed);
    VF scale = Min(kOne, servoSettings.MaximumSpeed / targetSpeed);
    VI useFallback = LessThan(targetSpeed, vf(1e-10f));
    scale = ConditionalSelect(useFallback, kOne, scale);
    Vector3Wide::Scale(errorAxis, scale * unclampedBiasSpeed, clampedBiasVelocity);
    maximumImpulse = servoSettings.MaximumForce * vf(dt);
}
static inline void ComputeClampedBiasVelocity(const Vector3Wide& error, const VF& positionErrorToBiasVelocity, const ServoSettingsWide& servoSettings, float dt, float inverseDt,
                                              Vector3Wide& clampedBiasVelocity, VF& maximumImpulse) {  // :132
    VF errorLength;
    Vector3Wide::Length(error, errorLength);
    Vector3Wide errorAxis;
    Vector3Wide::Scale(error, kOne / errorLength, errorAxis);
    VI useFallback = LessThan(errorLength, vf(1e-10f));
    errorAxis.X = ConditionalSelect(useFallback, kZero, errorAxis.X);
    errorAxis.Y = ConditionalSelect(useFallback, kZero, errorAxis.Y);
    errorAxis.Z = ConditionalSelect(useFallback, kZero, errorAxis.Z);
    ComputeClampedBiasVelocity(errorAxis, errorLength, positionErrorToBiasVelocity, servoSettings, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
}
}  // namespace ServoSettingsMore

// BallSocketShared.Solve with an impulse limit (BepuPhysics/Constraints/BallSocketShared.cs:126-135)
static inline void BallSocketSolveClamped(BodyVelocityWide& velocityA, BodyVelocityWide& velocityB, const Vector3Wide& offsetA, const Vector3Wide& offsetB, const Vector3Wide& biasVelocity,
                                          const Symmetric3x3Wide& effectiveMass, const VF& softnessImpulseScale, const VF& maximumImpulse, Vector3Wide& accumulatedImpulse,
                                          const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB) {
    Vector3Wide correctiveImpulse;
    BallSocketShared::ComputeCorrectiveImpulse(velocityA, velocityB, offsetA, offsetB, biasVelocity, effectiveMass, softnessImpulseScale, accumulatedImpulse, correctiveImpulse);
    ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulse, correctiveImpulse);
    BallSocketShared::ApplyImpulse(velocityA, velocityB, offsetA, offsetB, inertiaA, inertiaB, correctiveImpulse);
}

// ---------------------------------------------------------------------------------------------------------------- AngularServo (type id 29)
struct AngularServoPrestepData { QuaternionWide TargetRelativeRotationLocalA; SpringSettingsWide SpringSettings; ServoSettingsWide ServoSettings; };  // AngularServo.cs:62
struct AngularServoConstraint {  // AngularServoFunctions, AngularServo.cs:69 (its ApplyImpulse :73 is wide_joints.h's AngularServoFunctions::ApplyImpulse, which AngularMotor shares)
    typedef AngularServoPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void ApplyImpulse(Vector3Wide& angularVelocityA, Vector3Wide& angularVelocityB, const Symmetric3x3Wide& impulseToVelocityA, const Symmetric3x3Wide& negatedImpulseToVelocityB,
                             const Vector3Wide& csi) {
        AngularServoFunctions::ApplyImpulse(angularVelocityA, angularVelocityB, impulseToVelocityA, negatedImpulseToVelocityB, csi);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :101
        ApplyImpulse(wsvA.Angular, wsvB.Angular, inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :106
        QuaternionWide targetOrientationB, inverseTarget, errorRotation;
        QuaternionWide::ConcatenateWithoutOverlap(prestep.TargetRelativeRotationLocalA, orientationA, targetOrientationB);
        QuaternionWide::Conjugate(targetOrientationB, inverseTarget);
        QuaternionWide::ConcatenateWithoutOverlap(inverseTarget, orientationB, errorRotation);
        Vector3Wide errorAxis;
        VF errorLength;
        GetAxisAngleFromQuaternion(errorRotation, errorAxis, errorLength);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Symmetric3x3Wide unsoftenedInverseEffectiveMass, unsoftenedEffectiveMass;
        Symmetric3x3Wide::Add(inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, unsoftenedInverseEffectiveMass);
        Symmetric3x3Wide::Invert(unsoftenedInverseEffectiveMass, unsoftenedEffectiveMass);
        Vector3Wide clampedBiasVelocity;
        VF maximumImpulse;
        ServoSettingsMore::ComputeClampedBiasVelocity(errorAxis, errorLength, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
        Vector3Wide csv, csi, softnessComponent;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, csv);
        Vector3Wide::Subtract(clampedBiasVelocity, csv, csv);
        Symmetric3x3Wide::TransformWithoutOverlap(csv, unsoftenedEffectiveMass, csi);
        csi = csi * effectiveMassCFMScale;
        Vector3Wide::Scale(accumulatedImpulses, softnessImpulseScale, softnessComponent);
        Vector3Wide::Subtract(csi, softnessComponent, csi);
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(wsvA.Angular, wsvB.Angular, inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- TwistMotor (type id 28)
struct TwistMotorPrestepData { Vector3Wide LocalAxisA, LocalAxisB; VF TargetVelocity; MotorSettingsWide Settings; };  // TwistMotor.cs:69
struct TwistMotorFunctions {                                                                                        // TwistMotor.cs:77
    typedef TwistMotorPrestepData Prestep;
    typedef VF Impulses;
    static void ComputeJacobian(const QuaternionWide& orientationA, const QuaternionWide& orientationB, const Vector3Wide& localAxisA, const Vector3Wide& localAxisB, Vector3Wide& jacobianA) {  // :80
        Vector3Wide axisA, axisB;
        QuaternionWide::TransformWithoutOverlap(localAxisA, orientationA, axisA);
        QuaternionWide::TransformWithoutOverlap(localAxisB, orientationB, axisB);
        Vector3Wide::Add(axisA, axisB, jacobianA);
        VF length;
        Vector3Wide::Length(jacobianA, length);
        Vector3Wide::Scale(jacobianA, kOne / length, jacobianA);
        Vector3Wide::ConditionalSelect(LessThan(length, vf(1e-10f)), axisA, jacobianA, jacobianA);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :91
        Vector3Wide jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(orientationA, orientationB, prestep.LocalAxisA, prestep.LocalAxisB, jacobianA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        TwistServoFunctions::ApplyImpulse(wsvA.Angular, wsvB.Angular, impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :99
        Vector3Wide jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(orientationA, orientationB, prestep.LocalAxisA, prestep.LocalAxisB, jacobianA);
        VF unsoftenedInverseEffectiveMass;
        TwistServoFunctions::ComputeEffectiveMassContributions(inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB,
                                                               unsoftenedInverseEffectiveMass);
        VF effectiveMassCFMScale, softnessImpulseScale, maximumImpulse;
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        VF effectiveMass = effectiveMassCFMScale / unsoftenedInverseEffectiveMass;
        Vector3Wide velocityToImpulseA;
        Vector3Wide::Scale(jacobianA, effectiveMass, velocityToImpulseA);
        VF biasImpulse = prestep.TargetVelocity * effectiveMass;
        Vector3Wide netVelocity;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, netVelocity);
        VF csiVelocityComponent;
        Vector3Wide::Dot(netVelocity, velocityToImpulseA, csiVelocityComponent);
        VF csi = biasImpulse - accumulatedImpulses * softnessImpulseScale - csiVelocityComponent;
        VF previousAccumulatedImpulse = accumulatedImpulses;
        accumulatedImpulses = Max(Min(accumulatedImpulses + csi, maximumImpulse), neg(maximumImpulse));
        csi = accumulatedImpulses - previousAccumulatedImpulse;
        TwistServoFunctions::ApplyImpulse(wsvA.Angular, wsvB.Angular, impulseToVelocityA, negatedImpulseToVelocityB, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- AngularAxisMotor (type id 41)
struct AngularAxisMotorPrestepData { Vector3Wide LocalAxisA; VF TargetVelocity; MotorSettingsWide Settings; };  // AngularAxisMotor.cs:62
struct AngularAxisMotorFunctions {                                                                            // AngularAxisMotor.cs:69
    typedef AngularAxisMotorPrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const Vector3Wide& impulseToVelocityA, const Vector3Wide& negatedImpulseToVelocityB, const VF& csi, Vector3Wide& angularVelocityA, Vector3Wide& angularVelocityB) {  // :72
        angularVelocityA = angularVelocityA + impulseToVelocityA * csi;
        angularVelocityB = angularVelocityB - negatedImpulseToVelocityB * csi;
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :79
        Vector3Wide axis, jIA, jIB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalAxisA, orientationA, axis);
        Symmetric3x3Wide::TransformWithoutOverlap(axis, inertiaA.InverseInertiaTensor, jIA);
        Symmetric3x3Wide::TransformWithoutOverlap(axis, inertiaB.InverseInertiaTensor, jIB);
        ApplyImpulse(jIA, jIB, accumulatedImpulses, wsvA.Angular, wsvB.Angular);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :88
        Vector3Wide jA, jIA, jIB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalAxisA, orientationA, jA);
        Symmetric3x3Wide::TransformWithoutOverlap(jA, inertiaA.InverseInertiaTensor, jIA);
        VF contributionA, contributionB;
        Vector3Wide::Dot(jA, jIA, contributionA);
        Symmetric3x3Wide::TransformWithoutOverlap(jA, inertiaB.InverseInertiaTensor, jIB);
        Vector3Wide::Dot(jA, jIB, contributionB);
        VF effectiveMassCFMScale, softnessImpulseScale, maximumImpulse;
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        VF csi = (prestep.TargetVelocity + Vector3Wide::Dot(wsvB.Angular, jA) - Vector3Wide::Dot(wsvA.Angular, jA)) * effectiveMassCFMScale / (contributionA + contributionB) -
                 accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(jIA, jIB, csi, wsvA.Angular, wsvB.Angular);
    }
};

// ---------------------------------------------------------------------------------------------------------------- BallSocketMotor (type id 52)
struct BallSocketMotorPrestepData { Vector3Wide LocalOffsetB, TargetVelocityLocalA; MotorSettingsWide Settings; };  // BallSocketMotor.cs:60
struct BallSocketMotorFunctions {                                                                                 // BallSocketMotor.cs:67
    typedef BallSocketMotorPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :69
        Vector3Wide targetOffsetB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetB, orientationB, targetOffsetB);
        BallSocketShared::ApplyImpulse(wsvA, wsvB, (positionB - positionA) + targetOffsetB, targetOffsetB, inertiaA, inertiaB, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :75
        Vector3Wide targetOffsetB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetB, orientationB, targetOffsetB);
        Vector3Wide offsetA = (positionB - positionA) + targetOffsetB;
        VF effectiveMassCFMScale, softnessImpulseScale, maximumImpulse;
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        Symmetric3x3Wide effectiveMass;
        BallSocketShared::ComputeEffectiveMass(inertiaA, inertiaB, offsetA, targetOffsetB, effectiveMassCFMScale, effectiveMass);
        Vector3Wide biasVelocity, temp;
        QuaternionWide::TransformWithoutOverlap(prestep.TargetVelocityLocalA, orientationA, temp);  // QuaternionWide.Transform = TransformWithoutOverlap into a temporary (:283-287)
        biasVelocity = temp;
        Vector3Wide::Negate(biasVelocity, biasVelocity);
        BallSocketSolveClamped(wsvA, wsvB, offsetA, targetOffsetB, biasVelocity, effectiveMass, softnessImpulseScale, maximumImpulse, accumulatedImpulses, inertiaA, inertiaB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- BallSocketServo (type id 53)
struct BallSocketServoPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB; SpringSettingsWide SpringSettings; ServoSettingsWide ServoSettings; };  // BallSocketServo.cs:68
struct BallSocketServoFunctions {                                                                                                                  // BallSocketServo.cs:76
    typedef BallSocketServoPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :78
        Vector3Wide offsetA, offsetB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetA, orientationA, offsetA);
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetB, orientationB, offsetB);
        BallSocketShared::ApplyImpulse(wsvA, wsvB, offsetA, offsetB, inertiaA, inertiaB, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :85
        Vector3Wide offsetA, offsetB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetA, orientationA, offsetA);
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetB, orientationB, offsetB);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Symmetric3x3Wide effectiveMass;
        BallSocketShared::ComputeEffectiveMass(inertiaA, inertiaB, offsetA, offsetB, effectiveMassCFMScale, effectiveMass);
        Vector3Wide ab = positionB - positionA;
        Vector3Wide anchorB, error, biasVelocity;
        Vector3Wide::Add(ab, offsetB, anchorB);
        Vector3Wide::Subtract(anchorB, offsetA, error);
        VF maximumImpulse;
        ServoSettingsMore::ComputeClampedBiasVelocity(error, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, biasVelocity, maximumImpulse);
        BallSocketSolveClamped(wsvA, wsvB, offsetA, offsetB, biasVelocity, effectiveMass, softnessImpulseScale, maximumImpulse, accumulatedImpulses, inertiaA, inertiaB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- AngularSwivelHinge (type id 24)
struct AngularSwivelHingePrestepData { Vector3Wide LocalSwivelAxisA, LocalHingeAxisB; SpringSettingsWide SpringSettings; };  // AngularSwivelHinge.cs:64
struct AngularSwivelHingeFunctions {                                                                                          // AngularSwivelHinge.cs:71
    typedef AngularSwivelHingePrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const Vector3Wide& impulseToVelocityA, const Vector3Wide& negatedImpulseToVelocityB, const VF& csi, Vector3Wide& angularVelocityA, Vector3Wide& angularVelocityB) {  // :74
        Vector3Wide velocityChangeA, negatedVelocityChangeB;
        Vector3Wide::Scale(impulseToVelocityA, csi, velocityChangeA);
        Vector3Wide::Add(angularVelocityA, velocityChangeA, angularVelocityA);
        Vector3Wide::Scale(negatedImpulseToVelocityB, csi, negatedVelocityChangeB);
        Vector3Wide::Subtract(angularVelocityB, negatedVelocityChangeB, angularVelocityB);
    }
    static void ComputeJacobian(const Vector3Wide& localSwivelAxisA, const Vector3Wide& localHingeAxisB, const QuaternionWide& orientationA, const QuaternionWide& orientationB,
                                Vector3Wide& swivelAxis, Vector3Wide& hingeAxis, Vector3Wide& jacobianA) {  // :83
        QuaternionWide::TransformWithoutOverlap(localSwivelAxisA, orientationA, swivelAxis);
        QuaternionWide::TransformWithoutOverlap(localHingeAxisB, orientationB, hingeAxis);
        Vector3Wide::CrossWithoutOverlap(swivelAxis, hingeAxis, jacobianA);
        Vector3Wide fallbackJacobian;
        Helpers::FindPerpendicular(swivelAxis, fallbackJacobian);
        VF jacobianLengthSquared;
        Vector3Wide::Dot(jacobianA, jacobianA, jacobianLengthSquared);
        VI useFallback = LessThan(jacobianLengthSquared, vf(1e-3f));
        Vector3Wide::ConditionalSelect(useFallback, fallbackJacobian, jacobianA, jacobianA);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :97
        Vector3Wide swivelAxis, hingeAxis, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(prestep.LocalSwivelAxisA, prestep.LocalHingeAxisB, orientationA, orientationB, swivelAxis, hingeAxis, jacobianA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses, wsvA.Angular, wsvB.Angular);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :105
        Vector3Wide swivelAxis, hingeAxis, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(prestep.LocalSwivelAxisA, prestep.LocalHingeAxisB, orientationA, orientationB, swivelAxis, hingeAxis, jacobianA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        VF angularA, angularB;
        Vector3Wide::Dot(impulseToVelocityA, jacobianA, angularA);
        Vector3Wide::Dot(negatedImpulseToVelocityB, jacobianA, angularB);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF effectiveMass = effectiveMassCFMScale / (angularA + angularB);
        VF error;
        Vector3Wide::Dot(hingeAxis, swivelAxis, error);
        VF biasVelocity = neg(positionErrorToVelocity * error);
        Vector3Wide difference;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, difference);
        VF csv;
        Vector3Wide::Dot(difference, jacobianA, csv);
        VF csi = effectiveMass * (biasVelocity - csv) - accumulatedImpulses * softnessImpulseScale;
        accumulatedImpulses = accumulatedImpulses + csi;
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, csi, wsvA.Angular, wsvB.Angular);
    }
};

// ---------------------------------------------------------------------------------------------------------------- AngularAxisGearMotor (type id 54)
struct AngularAxisGearMotorPrestepData { Vector3Wide LocalAxisA; VF VelocityScale; MotorSettingsWide Settings; };  // AngularAxisGearMotor.cs:63
struct AngularAxisGearMotorFunctions {                                                                           // AngularAxisGearMotor.cs:70
    typedef AngularAxisGearMotorPrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const Vector3Wide& impulseToVelocityA, const Vector3Wide& negatedImpulseToVelocityB, const VF& csi, Vector3Wide& angularVelocityA, Vector3Wide& angularVelocityB) {  // :73
        angularVelocityA = angularVelocityA + impulseToVelocityA * csi;
        angularVelocityB = angularVelocityB - negatedImpulseToVelocityB * csi;
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :80
        Vector3Wide axis, jA, impulseToVelocityA, negatedImpulseToVelocityB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalAxisA, orientationA, axis);
        Vector3Wide::Scale(axis, prestep.VelocityScale, jA);
        Symmetric3x3Wide::TransformWithoutOverlap(jA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(axis, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses, wsvA.Angular, wsvB.Angular);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :90
        Vector3Wide axis, jA, impulseToVelocityA, negatedImpulseToVelocityB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalAxisA, orientationA, axis);
        Vector3Wide::Scale(axis, prestep.VelocityScale, jA);
        Symmetric3x3Wide::TransformWithoutOverlap(jA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        VF contributionA, contributionB;
        Vector3Wide::Dot(jA, impulseToVelocityA, contributionA);
        Symmetric3x3Wide::TransformWithoutOverlap(axis, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        Vector3Wide::Dot(axis, negatedImpulseToVelocityB, contributionB);
        VF effectiveMassCFMScale, softnessImpulseScale, maximumImpulse;
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        VF effectiveMass = effectiveMassCFMScale / (contributionA + contributionB);
        VF unscaledCSVA, negatedCSVB;
        Vector3Wide::Dot(wsvA.Angular, jA, unscaledCSVA);
        Vector3Wide::Dot(wsvB.Angular, axis, negatedCSVB);
        VF csi = (negatedCSVB - unscaledCSVA) * effectiveMass - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses, wsvA.Angular, wsvB.Angular);  // as written in the reference (:108): the accumulated impulse, not csi
    }
};

// MathHelper.FastReciprocal / FastReciprocalSquareRoot (BepuUtilities/MathHelper.cs:380-412). Default: the branch every target without the x86 approximation
// instructions takes (`Vector<float>.One / v`, :392 and :409) — what oracle/ and the device restate too.
// -DWIDE_FAST_RECIPROCAL_X86: the branch an AVX host takes (`Avx.Reciprocal` :384, `Avx.ReciprocalSqrt` :401 = vrcpps / vrsqrtps, the very instructions; relative
// error <= 1.5 * 2^-12, low bits vendor-specific). Built as wide/libbepu_wide_rcpx86.so and used ONLY to measure how far the portable branch (= the device) is from
// what the reference computes on the x86 host beside the GPU (tests/test_fast_reciprocal.py) — it is not a parity checker for the device.
#ifdef WIDE_FAST_RECIPROCAL_X86
static inline VF FastReciprocal(VF v) { return (VF)_mm256_rcp_ps((__m256)v); }
static inline VF FastReciprocalSquareRoot(VF v) { return (VF)_mm256_rsqrt_ps((__m256)v); }
#else
static inline VF FastReciprocal(VF v) { return kOne / v; }
static inline VF FastReciprocalSquareRoot(VF v) { return kOne / SquareRoot(v); }
#endif

// ---------------------------------------------------------------------------------------------------------------- CenterDistanceConstraint (type id 35)
struct CenterDistancePrestepData { VF TargetDistance; SpringSettingsWide SpringSettings; };  // CenterDistanceConstraint.cs:63
struct CenterDistanceConstraintFunctions {                                                  // CenterDistanceConstraint.cs:69
    typedef CenterDistancePrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const Vector3Wide& jacobianA, const VF& inverseMassA, const VF& inverseMassB, const VF& impulse, BodyVelocityWide& a, BodyVelocityWide& b) {  // :72
        Vector3Wide changeA, negatedChangeB;
        Vector3Wide::Scale(jacobianA, impulse * inverseMassA, changeA);
        Vector3Wide::Scale(jacobianA, impulse * inverseMassB, negatedChangeB);
        Vector3Wide::Add(a.Linear, changeA, a.Linear);
        Vector3Wide::Subtract(b.Linear, negatedChangeB, b.Linear);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :82
        Vector3Wide ab = positionB - positionA;
        VF lengthSquared = ab.X * ab.X + ab.Y * ab.Y + ab.Z * ab.Z;
        VF inverseDistance = FastReciprocalSquareRoot(lengthSquared);
        VI useFallback = LessThan(lengthSquared, vf(1e-10f));
        Vector3Wide jacobianA;
        Vector3Wide::Scale(ab, inverseDistance, jacobianA);
        jacobianA.X = ConditionalSelect(useFallback, kOne, jacobianA.X);
        jacobianA.Y = ConditionalSelect(useFallback, kZero, jacobianA.Y);
        jacobianA.Z = ConditionalSelect(useFallback, kZero, jacobianA.Z);
        ApplyImpulse(jacobianA, inertiaA.InverseMass, inertiaB.InverseMass, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulse,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :98
        Vector3Wide ab = positionB - positionA;
        VF distance = SquareRoot(ab.X * ab.X + ab.Y * ab.Y + ab.Z * ab.Z);
        VF inverseDistance = FastReciprocal(distance);
        VI useFallback = LessThan(distance, vf(1e-5f));
        Vector3Wide jacobianA;
        Vector3Wide::Scale(ab, inverseDistance, jacobianA);
        jacobianA.X = ConditionalSelect(useFallback, kOne, jacobianA.X);
        jacobianA.Y = ConditionalSelect(useFallback, kZero, jacobianA.Y);
        jacobianA.Z = ConditionalSelect(useFallback, kZero, jacobianA.Z);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF effectiveMass = effectiveMassCFMScale / (inertiaA.InverseMass + inertiaB.InverseMass);
        VF biasVelocity = (distance - prestep.TargetDistance) * positionErrorToVelocity;
        VF linearCSVA, negatedCSVB;
        Vector3Wide::Dot(wsvA.Linear, jacobianA, linearCSVA);
        Vector3Wide::Dot(wsvB.Linear, jacobianA, negatedCSVB);
        VF csi = (biasVelocity - (linearCSVA - negatedCSVB)) * effectiveMass - accumulatedImpulse * softnessImpulseScale;
        accumulatedImpulse = accumulatedImpulse + csi;
        ApplyImpulse(jacobianA, inertiaA.InverseMass, inertiaB.InverseMass, csi, wsvA, wsvB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- CenterDistanceLimit (type id 55)
struct CenterDistanceLimitPrestepData { VF MinimumDistance, MaximumDistance; SpringSettingsWide SpringSettings; };  // CenterDistanceLimit.cs:71
struct CenterDistanceLimitFunctions {                                                                              // CenterDistanceLimit.cs:78
    typedef CenterDistanceLimitPrestepData Prestep;
    typedef VF Impulses;
    static void ComputeJacobian(VF minimumDistance, VF maximumDistance, const Vector3Wide& positionA, const Vector3Wide& positionB, Vector3Wide& jacobianA, VF& distance, VI& useMinimum) {  // :81
        Vector3Wide ab = positionB - positionA;
        distance = SquareRoot(ab.X * ab.X + ab.Y * ab.Y + ab.Z * ab.Z);
        VF inverseDistance = FastReciprocal(distance);
        VI useFallback = LessThan(distance, vf(1e-5f));
        Vector3Wide::Scale(ab, inverseDistance, jacobianA);
        jacobianA.X = ConditionalSelect(useFallback, kOne, jacobianA.X);
        jacobianA.Y = ConditionalSelect(useFallback, kZero, jacobianA.Y);
        jacobianA.Z = ConditionalSelect(useFallback, kZero, jacobianA.Z);
        useMinimum = LessThan(Abs(distance - minimumDistance), Abs(distance - maximumDistance));
        Vector3Wide negated = -jacobianA;
        Vector3Wide::ConditionalSelect(useMinimum, negated, jacobianA, jacobianA);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :96
        Vector3Wide jacobianA;
        VF distance;
        VI useMinimum;
        ComputeJacobian(prestep.MinimumDistance, prestep.MaximumDistance, positionA, positionB, jacobianA, distance, useMinimum);
        CenterDistanceConstraintFunctions::ApplyImpulse(jacobianA, inertiaA.InverseMass, inertiaB.InverseMass, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulse,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :102
        Vector3Wide jacobianA;
        VF distance;
        VI useMinimum;
        ComputeJacobian(prestep.MinimumDistance, prestep.MaximumDistance, positionA, positionB, jacobianA, distance, useMinimum);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF effectiveMass = effectiveMassCFMScale / (inertiaA.InverseMass + inertiaB.InverseMass);
        VF error = ConditionalSelect(useMinimum, prestep.MinimumDistance - distance, distance - prestep.MaximumDistance);
        VF biasVelocity = Min(error * vf(inverseDt), error * positionErrorToVelocity);  // InequalityHelpers.ComputeBiasVelocity (InequalityHelpers.cs:9-12)
        VF csv = Vector3Wide::Dot(wsvA.Linear, jacobianA) - Vector3Wide::Dot(wsvB.Linear, jacobianA);
        VF csi = neg(accumulatedImpulse) * softnessImpulseScale - effectiveMass * (csv - biasVelocity);
        InequalityHelpers::ClampPositive(accumulatedImpulse, csi);
        CenterDistanceConstraintFunctions::ApplyImpulse(jacobianA, inertiaA.InverseMass, inertiaB.InverseMass, csi, wsvA, wsvB);
    }
};

namespace InequalityHelpers {  // BepuPhysics/Constraints/InequalityHelpers.cs:9-12
static inline void ComputeBiasVelocity(VF error, VF positionErrorToVelocity, float inverseDt, VF& biasVelocity) { biasVelocity = Min(error * vf(inverseDt), error * positionErrorToVelocity); }
}  // namespace InequalityHelpers

// ---------------------------------------------------------------------------------------------------------------- DistanceServo (type id 33)
struct DistanceServoPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB; VF TargetDistance; ServoSettingsWide ServoSettings; SpringSettingsWide SpringSettings; };  // DistanceServo.cs:97
struct DistanceServoFunctions {                                                                                                                                    // DistanceServo.cs:106
    typedef DistanceServoPrestepData Prestep;
    typedef VF Impulses;
    static void GetDistance(const QuaternionWide& orientationA, const Vector3Wide& ab, const QuaternionWide& orientationB, const Vector3Wide& localOffsetA, const Vector3Wide& localOffsetB,
                            Vector3Wide& anchorOffsetA, Vector3Wide& anchorOffsetB, Vector3Wide& anchorOffset, VF& distance) {  // :108
        QuaternionWide::TransformWithoutOverlap(localOffsetA, orientationA, anchorOffsetA);
        QuaternionWide::TransformWithoutOverlap(localOffsetB, orientationB, anchorOffsetB);
        Vector3Wide anchorB;
        Vector3Wide::Add(anchorOffsetB, ab, anchorB);
        Vector3Wide::Subtract(anchorB, anchorOffsetA, anchorOffset);
        Vector3Wide::Length(anchorOffset, distance);
    }
    static void ComputeJacobian(const VF& distance, const Vector3Wide& anchorOffsetA, const Vector3Wide& anchorOffsetB, Vector3Wide& direction, Vector3Wide& angularJA, Vector3Wide& angularJB) {  // :119
        VI needFallback = LessThan(distance, vf(1e-9f));
        direction.X = ConditionalSelect(needFallback, kOne, direction.X);
        direction.Y = ConditionalSelect(needFallback, kZero, direction.Y);
        direction.Z = ConditionalSelect(needFallback, kZero, direction.Z);
        Vector3Wide::CrossWithoutOverlap(anchorOffsetA, direction, angularJA);
        Vector3Wide::CrossWithoutOverlap(direction, anchorOffsetB, angularJB);
    }
    static void ComputeTransforms(const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector3Wide& anchorOffsetA, const Vector3Wide& anchorOffsetB, const VF& distance,
                                  Vector3Wide& direction, float dt, const SpringSettingsWide& springSettings, VF& positionErrorToVelocity, VF& softnessImpulseScale, VF& effectiveMass,
                                  Vector3Wide& angularJA, Vector3Wide& angularJB, Vector3Wide& angularImpulseToVelocityA, Vector3Wide& angularImpulseToVelocityB) {  // :132
        ComputeJacobian(distance, anchorOffsetA, anchorOffsetB, direction, angularJA, angularJB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJA, inertiaA.InverseInertiaTensor, angularImpulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJB, inertiaB.InverseInertiaTensor, angularImpulseToVelocityB);
        VF angularContributionA, angularContributionB;
        Vector3Wide::Dot(angularJA, angularImpulseToVelocityA, angularContributionA);
        Vector3Wide::Dot(angularJB, angularImpulseToVelocityB, angularContributionB);
        VF inverseEffectiveMass = inertiaA.InverseMass + inertiaB.InverseMass + angularContributionA + angularContributionB;
        VF effectiveMassCFMScale;
        SpringSettingsWide::ComputeSpringiness(springSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        effectiveMass = effectiveMassCFMScale / inverseEffectiveMass;
    }
    static void ApplyImpulse(const VF& inverseMassA, const VF& inverseMassB, const Vector3Wide& direction, const Vector3Wide& angularImpulseToVelocityA,
                             const Vector3Wide& angularImpulseToVelocityB, const VF& csi, BodyVelocityWide& velocityA, BodyVelocityWide& velocityB) {  // :173
        Vector3Wide linearVelocityChangeA, angularVelocityChangeA, negatedLinearVelocityChangeB, angularVelocityChangeB;
        Vector3Wide::Scale(direction, csi * inverseMassA, linearVelocityChangeA);
        Vector3Wide::Scale(angularImpulseToVelocityA, csi, angularVelocityChangeA);
        Vector3Wide::Add(linearVelocityChangeA, velocityA.Linear, velocityA.Linear);
        Vector3Wide::Add(angularVelocityChangeA, velocityA.Angular, velocityA.Angular);
        Vector3Wide::Scale(direction, csi * inverseMassB, negatedLinearVelocityChangeB);
        Vector3Wide::Scale(angularImpulseToVelocityB, csi, angularVelocityChangeB);
        Vector3Wide::Subtract(velocityB.Linear, negatedLinearVelocityChangeB, velocityB.Linear);
        Vector3Wide::Add(angularVelocityChangeB, velocityB.Angular, velocityB.Angular);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :189
        Vector3Wide anchorOffsetA, anchorOffsetB, anchorOffset, direction, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        VF distance;
        GetDistance(orientationA, positionB - positionA, orientationB, prestep.LocalOffsetA, prestep.LocalOffsetB, anchorOffsetA, anchorOffsetB, anchorOffset, distance);
        Vector3Wide::Scale(anchorOffset, kOne / distance, direction);
        ComputeJacobian(distance, anchorOffsetA, anchorOffsetB, direction, angularJA, angularJB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJA, inertiaA.InverseInertiaTensor, angularImpulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJB, inertiaB.InverseInertiaTensor, angularImpulseToVelocityB);
        ApplyImpulse(inertiaA.InverseMass, inertiaB.InverseMass, direction, angularImpulseToVelocityA, angularImpulseToVelocityB, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :199
        Vector3Wide anchorOffsetA, anchorOffsetB, anchorOffset, direction, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        VF distance, positionErrorToVelocity, softnessImpulseScale, effectiveMass;
        GetDistance(orientationA, positionB - positionA, orientationB, prestep.LocalOffsetA, prestep.LocalOffsetB, anchorOffsetA, anchorOffsetB, anchorOffset, distance);
        Vector3Wide::Scale(anchorOffset, kOne / distance, direction);
        ComputeTransforms(inertiaA, inertiaB, anchorOffsetA, anchorOffsetB, distance, direction, dt, prestep.SpringSettings, positionErrorToVelocity, softnessImpulseScale, effectiveMass,
                          angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB);
        VF error = distance - prestep.TargetDistance;
        VF clampedBiasVelocity, maximumImpulse;
        ServoSettingsWide::ComputeClampedBiasVelocity(error, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
        VF linearCSVA, negatedLinearCSVB, angularCSVA, angularCSVB;
        Vector3Wide::Dot(wsvA.Linear, direction, linearCSVA);
        Vector3Wide::Dot(wsvB.Linear, direction, negatedLinearCSVB);
        Vector3Wide::Dot(wsvA.Angular, angularJA, angularCSVA);
        Vector3Wide::Dot(wsvB.Angular, angularJB, angularCSVB);
        VF csi = (clampedBiasVelocity - linearCSVA - angularCSVA + negatedLinearCSVB - angularCSVB) * effectiveMass - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(inertiaA.InverseMass, inertiaB.InverseMass, direction, angularImpulseToVelocityA, angularImpulseToVelocityB, csi, wsvA, wsvB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- DistanceLimit (type id 34)
struct DistanceLimitPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB; VF MinimumDistance, MaximumDistance; SpringSettingsWide SpringSettings; };  // DistanceLimit.cs:94
struct DistanceLimitFunctions {                                                                                                                     // DistanceLimit.cs:103
    typedef DistanceLimitPrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const Vector3Wide& linearJacobianA, const Vector3Wide& angularJacobianA, const Vector3Wide& angularJacobianB, const BodyInertiaWide& inertiaA,
                             const BodyInertiaWide& inertiaB, const VF& csi, BodyVelocityWide& velocityA, BodyVelocityWide& velocityB) {  // :106
        Vector3Wide impulseScaledLinearJacobian = linearJacobianA * csi;
        velocityA.Linear = velocityA.Linear + impulseScaledLinearJacobian * inertiaA.InverseMass;
        velocityB.Linear = velocityB.Linear - impulseScaledLinearJacobian * inertiaB.InverseMass;
        velocityA.Angular = velocityA.Angular + (angularJacobianA * csi) * inertiaA.InverseInertiaTensor;
        velocityB.Angular = velocityB.Angular + (angularJacobianB * csi) * inertiaB.InverseInertiaTensor;
    }
    static void ComputeJacobians(const Vector3Wide& localOffsetA, const Vector3Wide& positionA, const QuaternionWide& orientationA, const Vector3Wide& localOffsetB, const Vector3Wide& positionB,
                                 const QuaternionWide& orientationB, const VF& minimumDistance, const VF& maximumDistance, VI& useMinimum, VF& distance, Vector3Wide& direction,
                                 Vector3Wide& angularJA, Vector3Wide& angularJB) {  // :118
        Vector3Wide offsetA, offsetB;
        QuaternionWide::TransformWithoutOverlap(localOffsetA, orientationA, offsetA);
        QuaternionWide::TransformWithoutOverlap(localOffsetB, orientationB, offsetB);
        Vector3Wide anchorOffset = (offsetB - offsetA) + (positionB - positionA);
        Vector3Wide::Length(anchorOffset, distance);
        useMinimum = LessThan(Abs(distance - minimumDistance), Abs(distance - maximumDistance));
        VF sign = ConditionalSelect(useMinimum, vf(-1.0f), kOne);
        Vector3Wide::Scale(anchorOffset, sign / distance, direction);
        VI needFallback = LessThan(distance, vf(1e-9f));
        direction.X = ConditionalSelect(needFallback, kOne, direction.X);
        direction.Y = ConditionalSelect(needFallback, kZero, direction.Y);
        direction.Z = ConditionalSelect(needFallback, kZero, direction.Z);
        Vector3Wide::CrossWithoutOverlap(offsetA, direction, angularJA);
        Vector3Wide::CrossWithoutOverlap(direction, offsetB, angularJB);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :141
        VI useMinimum;
        VF distance;
        Vector3Wide direction, angularJA, angularJB;
        ComputeJacobians(prestep.LocalOffsetA, positionA, orientationA, prestep.LocalOffsetB, positionB, orientationB, prestep.MinimumDistance, prestep.MaximumDistance, useMinimum, distance,
                         direction, angularJA, angularJB);
        ApplyImpulse(direction, angularJA, angularJB, inertiaA, inertiaB, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :148
        VI useMinimum;
        VF distance;
        Vector3Wide direction, angularJA, angularJB;
        ComputeJacobians(prestep.LocalOffsetA, positionA, orientationA, prestep.LocalOffsetB, positionB, orientationB, prestep.MinimumDistance, prestep.MaximumDistance, useMinimum, distance,
                         direction, angularJA, angularJB);
        VF linearCSVA, negatedLinearCSVB, angularCSVA, angularCSVB;
        Vector3Wide::Dot(wsvA.Linear, direction, linearCSVA);
        Vector3Wide::Dot(wsvB.Linear, direction, negatedLinearCSVB);
        Vector3Wide::Dot(wsvA.Angular, angularJA, angularCSVA);
        Vector3Wide::Dot(wsvB.Angular, angularJB, angularCSVB);
        VF csv = linearCSVA - negatedLinearCSVB + angularCSVA + angularCSVB;
        VF angularContributionA, angularContributionB;
        Symmetric3x3Wide::VectorSandwich(angularJA, inertiaA.InverseInertiaTensor, angularContributionA);
        Symmetric3x3Wide::VectorSandwich(angularJB, inertiaB.InverseInertiaTensor, angularContributionB);
        VF inverseEffectiveMass = inertiaA.InverseMass + inertiaB.InverseMass + angularContributionA + angularContributionB;
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF effectiveMass = effectiveMassCFMScale / inverseEffectiveMass;
        VF error = ConditionalSelect(useMinimum, prestep.MinimumDistance - distance, distance - prestep.MaximumDistance);
        VF biasVelocity;
        InequalityHelpers::ComputeBiasVelocity(error, positionErrorToVelocity, inverseDt, biasVelocity);
        VF csi = neg(accumulatedImpulses) * softnessImpulseScale - effectiveMass * (csv - biasVelocity);
        InequalityHelpers::ClampPositive(accumulatedImpulses, csi);
        ApplyImpulse(direction, angularJA, angularJB, inertiaA, inertiaB, csi, wsvA, wsvB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- LinearAxisServo / Motor / Limit (type ids 38, 39, 40)
struct LinearAxisServoPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB, LocalPlaneNormal; VF TargetOffset; ServoSettingsWide ServoSettings; SpringSettingsWide SpringSettings; };  // LinearAxisServo.cs:79
struct LinearAxisServoFunctions {                                                                                                                                                    // LinearAxisServo.cs:89
    typedef LinearAxisServoPrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const Vector3Wide& linearJA, const Vector3Wide& angularImpulseToVelocityA, const Vector3Wide& angularImpulseToVelocityB, const BodyInertiaWide& inertiaA,
                             const BodyInertiaWide& inertiaB, const VF& csi, BodyVelocityWide& velocityA, BodyVelocityWide& velocityB) {  // :183
        velocityA.Linear = velocityA.Linear + linearJA * (csi * inertiaA.InverseMass);
        velocityB.Linear = velocityB.Linear - linearJA * (csi * inertiaB.InverseMass);
        velocityA.Angular = velocityA.Angular + angularImpulseToVelocityA * csi;
        velocityB.Angular = velocityB.Angular + angularImpulseToVelocityB * csi;
    }
    static void ComputeJacobians(const Vector3Wide& ab, const QuaternionWide& orientationA, const QuaternionWide& orientationB, const Vector3Wide& localPlaneNormalA,
                                 const Vector3Wide& localOffsetA, const Vector3Wide& localOffsetB, VF& planeNormalDot, Vector3Wide& normal, Vector3Wide& angularJA, Vector3Wide& angularJB) {  // :193
        Matrix3x3Wide orientationMatrixA;
        Matrix3x3Wide::CreateFromQuaternion(orientationA, orientationMatrixA);
        Matrix3x3Wide::TransformWithoutOverlap(localPlaneNormalA, orientationMatrixA, normal);
        Vector3Wide anchorA, offsetB;
        Matrix3x3Wide::TransformWithoutOverlap(localOffsetA, orientationMatrixA, anchorA);
        QuaternionWide::TransformWithoutOverlap(localOffsetB, orientationB, offsetB);
        Vector3Wide anchorB = ab + offsetB;
        Vector3Wide::Dot(anchorB - anchorA, normal, planeNormalDot);
        Vector3Wide offsetFromAToClosetPointOnPlaneToB = anchorB - planeNormalDot * normal;
        Vector3Wide::CrossWithoutOverlap(offsetFromAToClosetPointOnPlaneToB, normal, angularJA);
        Vector3Wide::CrossWithoutOverlap(normal, offsetB, angularJB);
    }
    static void ComputeEffectiveMass(const Vector3Wide& angularJA, const Vector3Wide& angularJB, const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const VF& effectiveMassCFMScale,
                                     Vector3Wide& angularImpulseToVelocityA, Vector3Wide& angularImpulseToVelocityB, VF& effectiveMass) {  // :210
        Symmetric3x3Wide::TransformWithoutOverlap(angularJA, inertiaA.InverseInertiaTensor, angularImpulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJB, inertiaB.InverseInertiaTensor, angularImpulseToVelocityB);
        VF angularContributionA, angularContributionB;
        Vector3Wide::Dot(angularJA, angularImpulseToVelocityA, angularContributionA);
        Vector3Wide::Dot(angularJB, angularImpulseToVelocityB, angularContributionB);
        effectiveMass = effectiveMassCFMScale / (inertiaA.InverseMass + inertiaB.InverseMass + angularContributionA + angularContributionB);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :222
        VF planeNormalDot;
        Vector3Wide normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalPlaneNormal, prestep.LocalOffsetA, prestep.LocalOffsetB, planeNormalDot, normal, angularJA, angularJB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJA, inertiaA.InverseInertiaTensor, angularImpulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJB, inertiaB.InverseInertiaTensor, angularImpulseToVelocityB);
        ApplyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, inertiaA, inertiaB, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :230
        VF planeNormalDot, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale, effectiveMass, biasVelocity, maximumImpulse;
        Vector3Wide normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalPlaneNormal, prestep.LocalOffsetA, prestep.LocalOffsetB, planeNormalDot, normal, angularJA, angularJB);
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        ComputeEffectiveMass(angularJA, angularJB, inertiaA, inertiaB, effectiveMassCFMScale, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass);
        ServoSettingsWide::ComputeClampedBiasVelocity(planeNormalDot - prestep.TargetOffset, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, biasVelocity, maximumImpulse);
        VF csv = Vector3Wide::Dot(wsvA.Linear - wsvB.Linear, normal) + Vector3Wide::Dot(wsvA.Angular, angularJA) + Vector3Wide::Dot(wsvB.Angular, angularJB);
        VF csi = effectiveMass * (biasVelocity - csv) - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, inertiaA, inertiaB, csi, wsvA, wsvB);
    }
};

struct LinearAxisMotorPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB, LocalPlaneNormal; VF TargetVelocity; MotorSettingsWide Settings; };  // LinearAxisMotor.cs:73
struct LinearAxisMotorFunctions {                                                                                                              // LinearAxisMotor.cs:82
    typedef LinearAxisMotorPrestepData Prestep;
    typedef VF Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :84
        VF planeNormalDot;
        Vector3Wide normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        LinearAxisServoFunctions::ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalPlaneNormal, prestep.LocalOffsetA, prestep.LocalOffsetB, planeNormalDot, normal,
                                                   angularJA, angularJB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJA, inertiaA.InverseInertiaTensor, angularImpulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJB, inertiaB.InverseInertiaTensor, angularImpulseToVelocityB);
        LinearAxisServoFunctions::ApplyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, inertiaA, inertiaB, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :92
        VF planeNormalDot, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse, effectiveMass;
        Vector3Wide normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        LinearAxisServoFunctions::ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalPlaneNormal, prestep.LocalOffsetA, prestep.LocalOffsetB, planeNormalDot, normal,
                                                   angularJA, angularJB);
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        LinearAxisServoFunctions::ComputeEffectiveMass(angularJA, angularJB, inertiaA, inertiaB, effectiveMassCFMScale, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass);
        VF csv = Vector3Wide::Dot(wsvA.Linear - wsvB.Linear, normal) + Vector3Wide::Dot(wsvA.Angular, angularJA) + Vector3Wide::Dot(wsvB.Angular, angularJB);
        VF csi = effectiveMass * (neg(prestep.TargetVelocity) - csv) - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        LinearAxisServoFunctions::ApplyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, inertiaA, inertiaB, csi, wsvA, wsvB);
    }
};

struct LinearAxisLimitPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB, LocalPlaneNormal; VF MinimumOffset, MaximumOffset; SpringSettingsWide SpringSettings; };  // LinearAxisLimit.cs:80
struct LinearAxisLimitFunctions {                                                                                                                                   // LinearAxisLimit.cs:90
    typedef LinearAxisLimitPrestepData Prestep;
    typedef VF Impulses;
    static void ComputeJacobians(const Vector3Wide& ab, const QuaternionWide& orientationA, const QuaternionWide& orientationB, const Vector3Wide& localPlaneNormal, const Vector3Wide& localOffsetA,
                                 const Vector3Wide& localOffsetB, const VF& minimumOffset, const VF& maximumOffset, VF& error, Vector3Wide& normal, Vector3Wide& angularJA, Vector3Wide& angularJB) {  // :93
        Matrix3x3Wide orientationMatrixA;
        Matrix3x3Wide::CreateFromQuaternion(orientationA, orientationMatrixA);
        Matrix3x3Wide::TransformWithoutOverlap(localPlaneNormal, orientationMatrixA, normal);
        Vector3Wide anchorA, offsetB;
        Matrix3x3Wide::TransformWithoutOverlap(localOffsetA, orientationMatrixA, anchorA);
        QuaternionWide::TransformWithoutOverlap(localOffsetB, orientationB, offsetB);
        Vector3Wide anchorB = ab + offsetB;
        VF planeNormalDot;
        Vector3Wide::Dot(anchorB - anchorA, normal, planeNormalDot);
        VF minimumError = minimumOffset - planeNormalDot;
        VF maximumError = planeNormalDot - maximumOffset;
        VI useMin = LessThan(Abs(minimumError), Abs(maximumError));
        error = ConditionalSelect(useMin, minimumError, maximumError);
        normal.X = ConditionalSelect(useMin, neg(normal.X), normal.X);
        normal.Y = ConditionalSelect(useMin, neg(normal.Y), normal.Y);
        normal.Z = ConditionalSelect(useMin, neg(normal.Z), normal.Z);
        Vector3Wide offsetFromAToClosetPointOnPlaneToB = anchorB - planeNormalDot * normal;
        Vector3Wide::CrossWithoutOverlap(offsetFromAToClosetPointOnPlaneToB, normal, angularJA);
        Vector3Wide::CrossWithoutOverlap(normal, offsetB, angularJB);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :124
        VF error;
        Vector3Wide normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalPlaneNormal, prestep.LocalOffsetA, prestep.LocalOffsetB, prestep.MinimumOffset, prestep.MaximumOffset, error,
                         normal, angularJA, angularJB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJA, inertiaA.InverseInertiaTensor, angularImpulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularJB, inertiaB.InverseInertiaTensor, angularImpulseToVelocityB);
        LinearAxisServoFunctions::ApplyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, inertiaA, inertiaB, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :132
        VF error, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale, effectiveMass, biasVelocity;
        Vector3Wide normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB;
        ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalPlaneNormal, prestep.LocalOffsetA, prestep.LocalOffsetB, prestep.MinimumOffset, prestep.MaximumOffset, error,
                         normal, angularJA, angularJB);
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        LinearAxisServoFunctions::ComputeEffectiveMass(angularJA, angularJB, inertiaA, inertiaB, effectiveMassCFMScale, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass);
        InequalityHelpers::ComputeBiasVelocity(error, positionErrorToVelocity, inverseDt, biasVelocity);
        VF csv = Vector3Wide::Dot(wsvA.Linear - wsvB.Linear, normal) + Vector3Wide::Dot(wsvA.Angular, angularJA) + Vector3Wide::Dot(wsvB.Angular, angularJB);
        VF csi = effectiveMass * (biasVelocity - csv) - accumulatedImpulses * softnessImpulseScale;
        InequalityHelpers::ClampPositive(accumulatedImpulses, csi);
        LinearAxisServoFunctions::ApplyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, inertiaA, inertiaB, csi, wsvA, wsvB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- OneBodyAngularServo (type id 42)
struct OneBodyAngularServoPrestepData { QuaternionWide TargetOrientation; SpringSettingsWide SpringSettings; ServoSettingsWide ServoSettings; };  // OneBodyAngularServo.cs:62
struct OneBodyAngularServoFunctions {                                                                                                           // OneBodyAngularServo.cs:69
    typedef OneBodyAngularServoPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void ApplyImpulse(const Symmetric3x3Wide& inverseInertia, const Vector3Wide& csi, Vector3Wide& angularVelocity) {  // :72
        Vector3Wide velocityChange;
        Symmetric3x3Wide::TransformWithoutOverlap(csi, inverseInertia, velocityChange);
        Vector3Wide::Add(angularVelocity, velocityChange, angularVelocity);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA) {  // :79
        ApplyImpulse(inertiaA.InverseInertiaTensor, accumulatedImpulses, wsvA.Angular);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA) {  // :85
        QuaternionWide inverseOrientation, errorRotation;
        QuaternionWide::Conjugate(orientationA, inverseOrientation);
        QuaternionWide::ConcatenateWithoutOverlap(inverseOrientation, prestep.TargetOrientation, errorRotation);
        Vector3Wide errorAxis;
        VF errorLength;
        GetAxisAngleFromQuaternion(errorRotation, errorAxis, errorLength);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Symmetric3x3Wide effectiveMass;
        Symmetric3x3Wide::Invert(inertiaA.InverseInertiaTensor, effectiveMass);
        Vector3Wide clampedBiasVelocity;
        VF maximumImpulse;
        ServoSettingsMore::ComputeClampedBiasVelocity(errorAxis, errorLength, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
        Vector3Wide csv = clampedBiasVelocity - wsvA.Angular;
        Vector3Wide csi;
        Symmetric3x3Wide::TransformWithoutOverlap(csv, effectiveMass, csi);
        csi = csi * effectiveMassCFMScale - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(inertiaA.InverseInertiaTensor, csi, wsvA.Angular);
    }
};

// ---------------------------------------------------------------------------------------------------------------- OneBodyAngularMotor (type id 43)
struct OneBodyAngularMotorPrestepData { Vector3Wide TargetVelocity; MotorSettingsWide Settings; };  // OneBodyAngularMotor.cs:55
struct OneBodyAngularMotorFunctions {                                                              // OneBodyAngularMotor.cs:61
    typedef OneBodyAngularMotorPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void ApplyImpulse(Vector3Wide& angularVelocity, const Symmetric3x3Wide& impulseToVelocity, const Vector3Wide& csi) {  // :64
        Vector3Wide velocityChange;
        Symmetric3x3Wide::TransformWithoutOverlap(csi, impulseToVelocity, velocityChange);
        Vector3Wide::Add(angularVelocity, velocityChange, angularVelocity);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA) {  // :70
        ApplyImpulse(wsvA.Angular, inertiaA.InverseInertiaTensor, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA) {  // :75
        VF effectiveMassCFMScale, softnessImpulseScale, maximumImpulse;
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        Symmetric3x3Wide unsoftenedEffectiveMass;
        Symmetric3x3Wide::Invert(inertiaA.InverseInertiaTensor, unsoftenedEffectiveMass);
        Vector3Wide csi;
        Symmetric3x3Wide::TransformWithoutOverlap(prestep.TargetVelocity - wsvA.Angular, unsoftenedEffectiveMass, csi);
        csi = csi * effectiveMassCFMScale - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(wsvA.Angular, inertiaA.InverseInertiaTensor, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- OneBodyLinearServo (type id 44)
struct OneBodyLinearServoPrestepData { Vector3Wide LocalOffset, Target; SpringSettingsWide SpringSettings; ServoSettingsWide ServoSettings; };  // OneBodyLinearServo.cs:66
struct OneBodyLinearServoFunctions {                                                                                                          // OneBodyLinearServo.cs:77
    typedef OneBodyLinearServoPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void ApplyImpulse(const Vector3Wide& offset, const BodyInertiaWide& inertia, BodyVelocityWide& velocityA, const Vector3Wide& csi) {  // :96
        Vector3Wide wsi, change;
        Vector3Wide::CrossWithoutOverlap(offset, csi, wsi);
        Symmetric3x3Wide::TransformWithoutOverlap(wsi, inertia.InverseInertiaTensor, change);
        Vector3Wide::Add(velocityA.Angular, change, velocityA.Angular);
        Vector3Wide::Scale(csi, inertia.InverseMass, change);
        Vector3Wide::Add(velocityA.Linear, change, velocityA.Linear);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA) {  // :107
        Vector3Wide offset;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffset, orientationA, offset);
        ApplyImpulse(offset, inertiaA, wsvA, accumulatedImpulses);
    }
    // The effective mass both linear one-body types share (:128-134 here, OneBodyLinearMotor.cs:84-90 there).
    static void ComputeEffectiveMass(const Vector3Wide& offset, const BodyInertiaWide& inertiaA, Symmetric3x3Wide& effectiveMass) {
        Symmetric3x3Wide inverseEffectiveMass;
        Symmetric3x3Wide::SkewSandwichWithoutOverlap(offset, inertiaA.InverseInertiaTensor, inverseEffectiveMass);
        inverseEffectiveMass.XX = inverseEffectiveMass.XX + inertiaA.InverseMass;
        inverseEffectiveMass.YY = inverseEffectiveMass.YY + inertiaA.InverseMass;
        inverseEffectiveMass.ZZ = inverseEffectiveMass.ZZ + inertiaA.InverseMass;
        Symmetric3x3Wide::Invert(inverseEffectiveMass, effectiveMass);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA) {  // :114
        Vector3Wide offset;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffset, orientationA, offset);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Vector3Wide worldGrabPoint, error, biasVelocity;
        Vector3Wide::Add(offset, positionA, worldGrabPoint);
        Vector3Wide::Subtract(prestep.Target, worldGrabPoint, error);
        VF maximumImpulse;
        ServoSettingsMore::ComputeClampedBiasVelocity(error, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, biasVelocity, maximumImpulse);
        Vector3Wide csv = biasVelocity - Vector3Wide::Cross(wsvA.Angular, offset) - wsvA.Linear;
        Symmetric3x3Wide effectiveMass;
        ComputeEffectiveMass(offset, inertiaA, effectiveMass);
        Vector3Wide csi;
        Symmetric3x3Wide::TransformWithoutOverlap(csv, effectiveMass, csi);
        csi = csi * effectiveMassCFMScale - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(offset, inertiaA, wsvA, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- OneBodyLinearMotor (type id 45)
struct OneBodyLinearMotorPrestepData { Vector3Wide LocalOffset, TargetVelocity; MotorSettingsWide Settings; };  // OneBodyLinearMotor.cs:60
struct OneBodyLinearMotorFunctions {                                                                          // OneBodyLinearMotor.cs:67
    typedef OneBodyLinearMotorPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA) {  // :69
        Vector3Wide offset;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffset, orientationA, offset);
        OneBodyLinearServoFunctions::ApplyImpulse(offset, inertiaA, wsvA, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA) {  // :75
        Vector3Wide offset;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffset, orientationA, offset);
        VF effectiveMassCFMScale, softnessImpulseScale, maximumImpulse;
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        Vector3Wide csv = prestep.TargetVelocity - Vector3Wide::Cross(wsvA.Angular, offset) - wsvA.Linear;
        Symmetric3x3Wide effectiveMass;
        OneBodyLinearServoFunctions::ComputeEffectiveMass(offset, inertiaA, effectiveMass);
        Vector3Wide csi;
        Symmetric3x3Wide::TransformWithoutOverlap(csv, effectiveMass, csi);
        csi = csi * effectiveMassCFMScale - accumulatedImpulses * softnessImpulseScale;
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        OneBodyLinearServoFunctions::ApplyImpulse(offset, inertiaA, wsvA, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- Weld (type id 31)
// Symmetric3x3Wide.Multiply(symmetric, matrix) (BepuUtilities/Symmetric3x3Wide.cs:343-356)
static inline void MultiplySymmetricByMatrix(const Symmetric3x3Wide& a, const Matrix3x3Wide& b, Matrix3x3Wide& result) {
    result.X.X = a.XX * b.X.X + a.YX * b.Y.X + a.ZX * b.Z.X;
    result.X.Y = a.XX * b.X.Y + a.YX * b.Y.Y + a.ZX * b.Z.Y;
    result.X.Z = a.XX * b.X.Z + a.YX * b.Y.Z + a.ZX * b.Z.Z;
    result.Y.X = a.YX * b.X.X + a.YY * b.Y.X + a.ZY * b.Z.X;
    result.Y.Y = a.YX * b.X.Y + a.YY * b.Y.Y + a.ZY * b.Z.Y;
    result.Y.Z = a.YX * b.X.Z + a.YY * b.Y.Z + a.ZY * b.Z.Z;
    result.Z.X = a.ZX * b.X.X + a.ZY * b.Y.X + a.ZZ * b.Z.X;
    result.Z.Y = a.ZX * b.X.Y + a.ZY * b.Y.Y + a.ZZ * b.Z.Y;
    result.Z.Z = a.ZX * b.X.Z + a.ZY * b.Y.Z + a.ZZ * b.Z.Z;
}
// Symmetric3x3Wide.CompleteMatrixSandwichTranspose (Symmetric3x3Wide.cs:508-518): a^T * b, known symmetric
static inline void CompleteMatrixSandwichTranspose(const Matrix3x3Wide& a, const Matrix3x3Wide& b, Symmetric3x3Wide& result) {
    result.XX = a.X.X * b.X.X + a.Y.X * b.Y.X + a.Z.X * b.Z.X;
    result.YX = a.X.Y * b.X.X + a.Y.Y * b.Y.X + a.Z.Y * b.Z.X;
    result.YY = a.X.Y * b.X.Y + a.Y.Y * b.Y.Y + a.Z.Y * b.Z.Y;
    result.ZX = a.X.Z * b.X.X + a.Y.Z * b.Y.X + a.Z.Z * b.Z.X;
    result.ZY = a.X.Z * b.X.Y + a.Y.Z * b.Y.Y + a.Z.Z * b.Z.Y;
    result.ZZ = a.X.Z * b.X.Z + a.Y.Z * b.Y.Z + a.Z.Z * b.Z.Z;
}
// QuaternionWide operator * (QuaternionWide.cs:530-538): same expression as ConcatenateWithoutOverlap
static inline QuaternionWide operator*(const QuaternionWide& a, const QuaternionWide& b) {
    QuaternionWide result;
    result.X = a.W * b.X + a.X * b.W + a.Z * b.Y - a.Y * b.Z;
    result.Y = a.W * b.Y + a.Y * b.W + a.X * b.Z - a.Z * b.X;
    result.Z = a.W * b.Z + a.Z * b.W + a.Y * b.X - a.X * b.Y;
    result.W = a.W * b.W - a.X * b.X - a.Y * b.Y - a.Z * b.Z;
    return result;
}
// Symmetric6x6Wide.LDLTSolve (BepuUtilities/Symmetric6x6Wide.cs:84-129): [a b^T; b d] x = [v0; v1] by an LDL^T factorisation without pivoting
static inline void LDLTSolve(const Vector3Wide& v0, const Vector3Wide& v1, const Symmetric3x3Wide& a, const Matrix3x3Wide& b, const Symmetric3x3Wide& d, Vector3Wide& result0, Vector3Wide& result1) {
    VF d1 = a.XX;
    VF inverseD1 = kOne / d1;
    VF l21 = inverseD1 * a.YX;
    VF l31 = inverseD1 * a.ZX;
    VF l41 = inverseD1 * b.X.X;
    VF l51 = inverseD1 * b.X.Y;
    VF l61 = inverseD1 * b.X.Z;
    VF d2 = a.YY - l21 * l21 * d1;
    VF inverseD2 = kOne / d2;
    VF l32 = inverseD2 * (a.ZY - l31 * l21 * d1);
    VF l42 = inverseD2 * (b.Y.X - l41 * l21 * d1);
    VF l52 = inverseD2 * (b.Y.Y - l51 * l21 * d1);
    VF l62 = inverseD2 * (b.Y.Z - l61 * l21 * d1);
    VF d3 = a.ZZ - l31 * l31 * d1 - l32 * l32 * d2;
    VF inverseD3 = kOne / d3;
    VF l43 = inverseD3 * (b.Z.X - l41 * l31 * d1 - l42 * l32 * d2);
    VF l53 = inverseD3 * (b.Z.Y - l51 * l31 * d1 - l52 * l32 * d2);
    VF l63 = inverseD3 * (b.Z.Z - l61 * l31 * d1 - l62 * l32 * d2);
    VF d4 = d.XX - l41 * l41 * d1 - l42 * l42 * d2 - l43 * l43 * d3;
    VF inverseD4 = kOne / d4;
    VF l54 = inverseD4 * (d.YX - l51 * l41 * d1 - l52 * l42 * d2 - l53 * l43 * d3);
    VF l64 = inverseD4 * (d.ZX - l61 * l41 * d1 - l62 * l42 * d2 - l63 * l43 * d3);
    VF d5 = d.YY - l51 * l51 * d1 - l52 * l52 * d2 - l53 * l53 * d3 - l54 * l54 * d4;
    VF inverseD5 = kOne / d5;
    VF l65 = inverseD5 * (d.ZY - l61 * l51 * d1 - l62 * l52 * d2 - l63 * l53 * d3 - l64 * l54 * d4);
    VF d6 = d.ZZ - l61 * l61 * d1 - l62 * l62 * d2 - l63 * l63 * d3 - l64 * l64 * d4 - l65 * l65 * d5;
    VF inverseD6 = kOne / d6;
    result0.X = v0.X;
    result0.Y = v0.Y - l21 * result0.X;
    result0.Z = v0.Z - l31 * result0.X - l32 * result0.Y;
    result1.X = v1.X - l41 * result0.X - l42 * result0.Y - l43 * result0.Z;
    result1.Y = v1.Y - l51 * result0.X - l52 * result0.Y - l53 * result0.Z - l54 * result1.X;
    result1.Z = v1.Z - l61 * result0.X - l62 * result0.Y - l63 * result0.Z - l64 * result1.X - l65 * result1.Y;
    result1.Z = result1.Z * inverseD6;
    result1.Y = result1.Y * inverseD5 - l65 * result1.Z;
    result1.X = result1.X * inverseD4 - l64 * result1.Z - l54 * result1.Y;
    result0.Z = result0.Z * inverseD3 - l63 * result1.Z - l53 * result1.Y - l43 * result1.X;
    result0.Y = result0.Y * inverseD2 - l62 * result1.Z - l52 * result1.Y - l42 * result1.X - l32 * result0.Z;
    result0.X = result0.X * inverseD1 - l61 * result1.Z - l51 * result1.Y - l41 * result1.X - l31 * result0.Z - l21 * result0.Y;
}

struct WeldPrestepData { Vector3Wide LocalOffset; QuaternionWide LocalOrientation; SpringSettingsWide SpringSettings; };  // Weld.cs:70
struct WeldAccumulatedImpulses { Vector3Wide Orientation, Offset; };                                                     // Weld.cs:77
struct WeldFunctions {                                                                                                   // Weld.cs:83
    typedef WeldPrestepData Prestep;
    typedef WeldAccumulatedImpulses Impulses;
    static void ApplyImpulse(const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector3Wide& offset, const Vector3Wide& orientationCSI, const Vector3Wide& offsetCSI,
                             BodyVelocityWide& velocityA, BodyVelocityWide& velocityB) {  // :86
        Vector3Wide linearChangeA, offsetWorldImpulse, angularImpulseA, angularChangeA, negatedLinearChangeB, negatedAngularChangeB;
        Vector3Wide::Scale(offsetCSI, inertiaA.InverseMass, linearChangeA);
        Vector3Wide::Add(velocityA.Linear, linearChangeA, velocityA.Linear);
        Vector3Wide::CrossWithoutOverlap(offset, offsetCSI, offsetWorldImpulse);
        Vector3Wide::Add(offsetWorldImpulse, orientationCSI, angularImpulseA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseA, inertiaA.InverseInertiaTensor, angularChangeA);
        Vector3Wide::Add(velocityA.Angular, angularChangeA, velocityA.Angular);
        Vector3Wide::Scale(offsetCSI, inertiaB.InverseMass, negatedLinearChangeB);
        Vector3Wide::Subtract(velocityB.Linear, negatedLinearChangeB, velocityB.Linear);
        Symmetric3x3Wide::TransformWithoutOverlap(orientationCSI, inertiaB.InverseInertiaTensor, negatedAngularChangeB);
        Vector3Wide::Subtract(velocityB.Angular, negatedAngularChangeB, velocityB.Angular);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :116
        Vector3Wide offset;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffset, orientationA, offset);  // QuaternionWide.Transform(in, in, out) = TransformWithoutOverlap + copy (QuaternionWide.cs:283)
        ApplyImpulse(inertiaA, inertiaB, offset, accumulatedImpulses.Orientation, accumulatedImpulses.Offset, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :123
        Vector3Wide offset;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffset, orientationA, offset);
        Symmetric3x3Wide jmjtA, jmjtD;
        Matrix3x3Wide xAB, jmjtB;
        Symmetric3x3Wide::Add(inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, jmjtA);
        Matrix3x3Wide::CreateCrossProduct(offset, xAB);
        MultiplySymmetricByMatrix(inertiaA.InverseInertiaTensor, xAB, jmjtB);
        CompleteMatrixSandwichTranspose(xAB, jmjtB, jmjtD);
        VF diagonalAdd = inertiaA.InverseMass + inertiaB.InverseMass;
        jmjtD.XX = jmjtD.XX + diagonalAdd;
        jmjtD.YY = jmjtD.YY + diagonalAdd;
        jmjtD.ZZ = jmjtD.ZZ + diagonalAdd;
        Vector3Wide positionError = positionB - positionA - offset;
        QuaternionWide targetOrientationB = prestep.LocalOrientation * orientationA;
        QuaternionWide conjugate, rotationError;
        QuaternionWide::Conjugate(targetOrientationB, conjugate);  // the value-returning overload (:560) has the same body
        QuaternionWide::ConcatenateWithoutOverlap(conjugate, orientationB, rotationError);
        Vector3Wide rotationErrorAxis;
        VF rotationErrorLength;
        GetAxisAngleFromQuaternion(rotationError, rotationErrorAxis, rotationErrorLength);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Vector3Wide orientationBiasVelocity = rotationErrorAxis * (rotationErrorLength * positionErrorToVelocity);
        Vector3Wide offsetBiasVelocity = positionError * positionErrorToVelocity;
        Vector3Wide orientationCSV, offsetCSV;
        orientationCSV.X = orientationBiasVelocity.X - wsvA.Angular.X + wsvB.Angular.X;
        orientationCSV.Y = orientationBiasVelocity.Y - wsvA.Angular.Y + wsvB.Angular.Y;
        orientationCSV.Z = orientationBiasVelocity.Z - wsvA.Angular.Z + wsvB.Angular.Z;
        offsetCSV.X = offsetBiasVelocity.X - wsvA.Linear.X + wsvB.Linear.X - (wsvA.Angular.Y * offset.Z - wsvA.Angular.Z * offset.Y);
        offsetCSV.Y = offsetBiasVelocity.Y - wsvA.Linear.Y + wsvB.Linear.Y - (wsvA.Angular.Z * offset.X - wsvA.Angular.X * offset.Z);
        offsetCSV.Z = offsetBiasVelocity.Z - wsvA.Linear.Z + wsvB.Linear.Z - (wsvA.Angular.X * offset.Y - wsvA.Angular.Y * offset.X);
        Vector3Wide orientationCSI, offsetCSI;
        LDLTSolve(orientationCSV, offsetCSV, jmjtA, jmjtB, jmjtD, orientationCSI, offsetCSI);
        orientationCSI.X = orientationCSI.X * effectiveMassCFMScale - accumulatedImpulses.Orientation.X * softnessImpulseScale;
        orientationCSI.Y = orientationCSI.Y * effectiveMassCFMScale - accumulatedImpulses.Orientation.Y * softnessImpulseScale;
        orientationCSI.Z = orientationCSI.Z * effectiveMassCFMScale - accumulatedImpulses.Orientation.Z * softnessImpulseScale;
        accumulatedImpulses.Orientation.X = accumulatedImpulses.Orientation.X + orientationCSI.X;
        accumulatedImpulses.Orientation.Y = accumulatedImpulses.Orientation.Y + orientationCSI.Y;
        accumulatedImpulses.Orientation.Z = accumulatedImpulses.Orientation.Z + orientationCSI.Z;
        offsetCSI.X = offsetCSI.X * effectiveMassCFMScale - accumulatedImpulses.Offset.X * softnessImpulseScale;
        offsetCSI.Y = offsetCSI.Y * effectiveMassCFMScale - accumulatedImpulses.Offset.Y * softnessImpulseScale;
        offsetCSI.Z = offsetCSI.Z * effectiveMassCFMScale - accumulatedImpulses.Offset.Z * softnessImpulseScale;
        accumulatedImpulses.Offset.X = accumulatedImpulses.Offset.X + offsetCSI.X;
        accumulatedImpulses.Offset.Y = accumulatedImpulses.Offset.Y + offsetCSI.Y;
        accumulatedImpulses.Offset.Z = accumulatedImpulses.Offset.Z + offsetCSI.Z;
        ApplyImpulse(inertiaA, inertiaB, offset, orientationCSI, offsetCSI, wsvA, wsvB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- PointOnLineServo (type id 37)
namespace ServoSettingsMore {  // the Vector2Wide overloads (ServoSettings.cs:88-113, :153-164)
static inline void ComputeClampedBiasVelocity(const Vector2Wide& errorAxis, const VF& errorLength, const VF& positionErrorToBiasVelocity, const ServoSettingsWide& servoSettings, float dt,
                                              float inverseDt, Vector2Wide& clampedBiasVelocity, VF& maximumImpulse) {  // :88
    VF baseSpeed = Min(servoSettings.BaseSpeed, errorLength * vf(inverseDt));
    VF unclampedBiasSpeed = errorLength * positionErrorToBiasVelocity;
    VF targetSpeed = Max(baseSpeed, unclampedBiasSpeed);
    VF scale = Min(kOne, servoSettings.MaximumSpeed / targetSpeed);
    VI useFallback = LessThan(targetSpeed, vf(1e-10f));
    scale = ConditionalSelect(useFallback, kOne, scale);
    Vector2Wide::Scale(errorAxis, scale * unclampedBiasSpeed, clampedBiasVelocity);
    maximumImpulse = servoSettings.MaximumForce * vf(dt);
}
static inline void ComputeClampedBiasVelocity(const Vector2Wide& error, const VF& positionErrorToBiasVelocity, const ServoSettingsWide& servoSettings, float dt, float inverseDt,
                                              Vector2Wide& clampedBiasVelocity, VF& maximumImpulse) {  // :104
    VF errorLength;
    Vector2Wide::Length(error, errorLength);
    Vector2Wide errorAxis;
    Vector2Wide::Scale(error, kOne / errorLength, errorAxis);
    VI useFallback = LessThan(errorLength, vf(1e-10f));
    errorAxis.X = ConditionalSelect(useFallback, kZero, errorAxis.X);
    errorAxis.Y = ConditionalSelect(useFallback, kZero, errorAxis.Y);
    ComputeClampedBiasVelocity(errorAxis, errorLength, positionErrorToBiasVelocity, servoSettings, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
}
static inline void ClampImpulse(const VF& maximumImpulse, Vector2Wide& accumulatedImpulse, Vector2Wide& csi) {  // :153
    Vector2Wide previousImpulse = accumulatedImpulse;
    Vector2Wide unclamped;
    Vector2Wide::Add(accumulatedImpulse, csi, unclamped);
    VF impulseMagnitude;
    Vector2Wide::Length(unclamped, impulseMagnitude);
    VF impulseScale = ConditionalSelect(LessThan(Abs(impulseMagnitude), vf(1e-10f)), kOne, Min(maximumImpulse / impulseMagnitude, kOne));
    Vector2Wide::Scale(unclamped, impulseScale, accumulatedImpulse);
    Vector2Wide::Subtract(accumulatedImpulse, previousImpulse, csi);
}
}  // namespace ServoSettingsMore

// (Symmetric2x2Wide.SandwichScale is in wide_contacts.h, where the friction constraint first needed it.)
struct PointOnLineServoPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB, LocalDirection; ServoSettingsWide ServoSettings; SpringSettingsWide SpringSettings; };  // PointOnLineServo.cs:73
struct PointOnLineServoFunctions {                                                                                                                                  // PointOnLineServo.cs:82
    typedef PointOnLineServoPrestepData Prestep;
    typedef Vector2Wide Impulses;
    static void ApplyImpulse(BodyVelocityWide& velocityA, BodyVelocityWide& velocityB, const Matrix2x3Wide& linearJacobian, const Matrix2x3Wide& angularJacobianA,
                             const Matrix2x3Wide& angularJacobianB, const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, Vector2Wide& csi) {  // :85
        Vector3Wide linearImpulseA, angularImpulseA, angularImpulseB, angularChangeA, angularChangeB, linearChangeA, negatedLinearChangeB;
        Matrix2x3Wide::Transform(csi, linearJacobian, linearImpulseA);
        Matrix2x3Wide::Transform(csi, angularJacobianA, angularImpulseA);
        Matrix2x3Wide::Transform(csi, angularJacobianB, angularImpulseB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseA, inertiaA.InverseInertiaTensor, angularChangeA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseB, inertiaB.InverseInertiaTensor, angularChangeB);
        Vector3Wide::Scale(linearImpulseA, inertiaA.InverseMass, linearChangeA);
        Vector3Wide::Scale(linearImpulseA, inertiaB.InverseMass, negatedLinearChangeB);
        Vector3Wide::Add(linearChangeA, velocityA.Linear, velocityA.Linear);
        Vector3Wide::Add(angularChangeA, velocityA.Angular, velocityA.Angular);
        Vector3Wide::Subtract(velocityB.Linear, negatedLinearChangeB, velocityB.Linear);
        Vector3Wide::Add(angularChangeB, velocityB.Angular, velocityB.Angular);
    }
    static void ComputeJacobians(const Vector3Wide& ab, const QuaternionWide& orientationA, const QuaternionWide& orientationB, const Vector3Wide& localDirection, const Vector3Wide& localOffsetA,
                                 const Vector3Wide& localOffsetB, Vector3Wide& anchorOffset, Matrix2x3Wide& linearJacobian, Matrix2x3Wide& angularJA, Matrix2x3Wide& angularJB) {  // :104
        Vector3Wide localTangentX, localTangentY;
        Helpers::BuildOrthonormalBasis(localDirection, localTangentX, localTangentY);
        Matrix3x3Wide orientationMatrixA;
        Matrix3x3Wide::CreateFromQuaternion(orientationA, orientationMatrixA);
        Vector3Wide anchorA, offsetB, direction, anchorB, lineStartToClosestPointOnLine, offsetA;
        Matrix3x3Wide::TransformWithoutOverlap(localOffsetA, orientationMatrixA, anchorA);
        QuaternionWide::TransformWithoutOverlap(localOffsetB, orientationB, offsetB);
        Matrix3x3Wide::TransformWithoutOverlap(localDirection, orientationMatrixA, direction);
        Vector3Wide::Add(offsetB, ab, anchorB);
        Vector3Wide::Subtract(anchorB, anchorA, anchorOffset);
        VF d;
        Vector3Wide::Dot(anchorOffset, direction, d);
        Vector3Wide::Scale(direction, d, lineStartToClosestPointOnLine);
        Vector3Wide::Add(lineStartToClosestPointOnLine, anchorA, offsetA);
        Matrix3x3Wide::TransformWithoutOverlap(localTangentX, orientationMatrixA, linearJacobian.X);
        Matrix3x3Wide::TransformWithoutOverlap(localTangentY, orientationMatrixA, linearJacobian.Y);
        Vector3Wide::CrossWithoutOverlap(offsetA, linearJacobian.X, angularJA.X);
        Vector3Wide::CrossWithoutOverlap(offsetA, linearJacobian.Y, angularJA.Y);
        Vector3Wide::CrossWithoutOverlap(linearJacobian.X, offsetB, angularJB.X);
        Vector3Wide::CrossWithoutOverlap(linearJacobian.Y, offsetB, angularJB.Y);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :128
        Vector3Wide anchorOffset;
        Matrix2x3Wide linearJacobian, angularJA, angularJB;
        ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalDirection, prestep.LocalOffsetA, prestep.LocalOffsetB, anchorOffset, linearJacobian, angularJA, angularJB);
        ApplyImpulse(wsvA, wsvB, linearJacobian, angularJA, angularJB, inertiaA, inertiaB, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :134
        Vector3Wide anchorOffset;
        Matrix2x3Wide linearJacobian, angularJA, angularJB;
        ComputeJacobians(positionB - positionA, orientationA, orientationB, prestep.LocalDirection, prestep.LocalOffsetA, prestep.LocalOffsetB, anchorOffset, linearJacobian, angularJA, angularJB);
        Symmetric2x2Wide linearContribution, angularContributionA, angularContributionB, inverseEffectiveMass, effectiveMass;
        SandwichScale(linearJacobian, inertiaA.InverseMass + inertiaB.InverseMass, linearContribution);
        Symmetric3x3Wide::MatrixSandwich(angularJA, inertiaA.InverseInertiaTensor, angularContributionA);
        Symmetric3x3Wide::MatrixSandwich(angularJB, inertiaB.InverseInertiaTensor, angularContributionB);
        Symmetric2x2Wide::Add(angularContributionA, angularContributionB, inverseEffectiveMass);
        Symmetric2x2Wide::Add(inverseEffectiveMass, linearContribution, inverseEffectiveMass);
        Symmetric2x2Wide::InvertWithoutOverlap(inverseEffectiveMass, effectiveMass);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Symmetric2x2Wide::Scale(effectiveMass, effectiveMassCFMScale, effectiveMass);
        Vector2Wide linearCSVA, negatedLinearCSVB, angularCSVA, angularCSVB, linearCSV, angularCSV, csv;
        Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvA.Linear, linearJacobian, linearCSVA);
        Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvB.Linear, linearJacobian, negatedLinearCSVB);
        Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvA.Angular, angularJA, angularCSVA);
        Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvB.Angular, angularJB, angularCSVB);
        Vector2Wide::Subtract(linearCSVA, negatedLinearCSVB, linearCSV);
        Vector2Wide::Add(angularCSVA, angularCSVB, angularCSV);
        Vector2Wide::Add(linearCSV, angularCSV, csv);
        Vector2Wide error;
        Vector3Wide::Dot(anchorOffset, linearJacobian.X, error.X);
        Vector3Wide::Dot(anchorOffset, linearJacobian.Y, error.Y);
        Vector2Wide biasVelocity;
        VF maximumImpulse;
        ServoSettingsMore::ComputeClampedBiasVelocity(error, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, biasVelocity, maximumImpulse);
        Vector2Wide::Subtract(biasVelocity, csv, csv);
        Vector2Wide csi, softnessContribution;
        Symmetric2x2Wide::TransformWithoutOverlap(csv, effectiveMass, csi);
        Vector2Wide::Scale(accumulatedImpulses, softnessImpulseScale, softnessContribution);
        Vector2Wide::Subtract(csi, softnessContribution, csi);
        ServoSettingsMore::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        ApplyImpulse(wsvA, wsvB, linearJacobian, angularJA, angularJB, inertiaA, inertiaB, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- nonconvex contact manifolds (type ids 8-10 one body, 15-17 two bodies)
struct NonconvexContactPrestepData { Vector3Wide Offset; VF Depth; Vector3Wide Normal; };  // Contact/ContactNonconvexCommon.cs:11
struct NonconvexAccumulatedImpulses { Vector2Wide Tangent; VF Penetration; };              // ContactNonconvexCommon.cs:165
template <int N> struct ContactNonconvexOneBodyPrestepData { MaterialPropertiesWide MaterialProperties; NonconvexContactPrestepData Contact[N]; };              // ContactNonconvexTypes.cs:161
template <int N> struct ContactNonconvexPrestepData { MaterialPropertiesWide MaterialProperties; Vector3Wide OffsetB; NonconvexContactPrestepData Contact[N]; };  // ContactNonconvexTypes.cs:58
template <int N> struct ContactNonconvexAccumulatedImpulses { NonconvexAccumulatedImpulses Contact[N]; };                                                       // ContactNonconvexTypes.cs:88

template <int N> struct ContactNonconvexOneBodyFunctions {  // ContactNonconvexCommon.cs:171
    typedef ContactNonconvexOneBodyPrestepData<N> Prestep;
    typedef ContactNonconvexAccumulatedImpulses<N> Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA) {  // :187
        for (int i = 0; i < N; ++i) {
            NonconvexContactPrestepData& prestepContact = prestep.Contact[i];
            Vector3Wide x, z;
            Helpers::BuildOrthonormalBasis(prestepContact.Normal, x, z);
            NonconvexAccumulatedImpulses& contactImpulse = accumulatedImpulses.Contact[i];
            TangentFrictionOneBody::WarmStart(x, z, prestepContact.Offset, inertiaA, contactImpulse.Tangent, wsvA);
            PenetrationLimitOneBody::WarmStart(inertiaA, prestepContact.Normal, prestepContact.Offset, contactImpulse.Penetration, wsvA);
        }
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA) {  // :202
        MaterialPropertiesWide& prestepMaterial = prestep.MaterialProperties;
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestepMaterial.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF inverseDtWide = vf(inverseDt);
        for (int i = 0; i < N; ++i) {
            NonconvexContactPrestepData& contact = prestep.Contact[i];
            NonconvexAccumulatedImpulses& contactImpulse = accumulatedImpulses.Contact[i];
            PenetrationLimitOneBody::Solve(inertiaA, contact.Normal, contact.Offset, contact.Depth, positionErrorToVelocity, effectiveMassCFMScale, prestepMaterial.MaximumRecoveryVelocity,
                                           inverseDtWide, softnessImpulseScale, contactImpulse.Penetration, wsvA);
            Vector3Wide x, z;
            Helpers::BuildOrthonormalBasis(contact.Normal, x, z);
            VF maximumTangentImpulse = prestepMaterial.FrictionCoefficient * contactImpulse.Penetration;
            TangentFrictionOneBody::Solve(x, z, contact.Offset, inertiaA, maximumTangentImpulse, contactImpulse.Tangent, wsvA);
        }
    }
    static void IncrementallyUpdateForSubstep(const VF& dt, const BodyVelocityWide& wsvA, Prestep& prestep) {  // :232
        for (int i = 0; i < N; ++i) {
            NonconvexContactPrestepData& prestepContact = prestep.Contact[i];
            PenetrationLimitOneBody::UpdatePenetrationDepth(dt, prestepContact.Offset, prestepContact.Normal, wsvA, prestepContact.Depth);
        }
    }
};

template <int N> struct ContactNonconvexTwoBodyFunctions {  // ContactNonconvexCommon.cs:243
    typedef ContactNonconvexPrestepData<N> Prestep;
    typedef ContactNonconvexAccumulatedImpulses<N> Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :248
        Vector3Wide& prestepOffsetB = prestep.OffsetB;
        for (int i = 0; i < N; ++i) {
            NonconvexContactPrestepData& prestepContact = prestep.Contact[i];
            Vector3Wide x, z, contactOffsetB;
            Helpers::BuildOrthonormalBasis(prestepContact.Normal, x, z);
            Vector3Wide::Subtract(prestepContact.Offset, prestepOffsetB, contactOffsetB);
            NonconvexAccumulatedImpulses& contactImpulse = accumulatedImpulses.Contact[i];
            TangentFriction::WarmStart(x, z, prestepContact.Offset, contactOffsetB, inertiaA, inertiaB, contactImpulse.Tangent, wsvA, wsvB);
            PenetrationLimit::WarmStart(inertiaA, inertiaB, prestepContact.Normal, prestepContact.Offset, contactOffsetB, contactImpulse.Penetration, wsvA, wsvB);
        }
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :265
        Vector3Wide& prestepOffsetB = prestep.OffsetB;
        MaterialPropertiesWide& prestepMaterial = prestep.MaterialProperties;
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestepMaterial.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF inverseDtWide = vf(inverseDt);
        for (int i = 0; i < N; ++i) {
            NonconvexContactPrestepData& contact = prestep.Contact[i];
            NonconvexAccumulatedImpulses& contactImpulse = accumulatedImpulses.Contact[i];
            Vector3Wide contactOffsetB;
            Vector3Wide::Subtract(contact.Offset, prestepOffsetB, contactOffsetB);
            PenetrationLimit::Solve(inertiaA, inertiaB, contact.Normal, contact.Offset, contactOffsetB, contact.Depth, positionErrorToVelocity, effectiveMassCFMScale,
                                    prestepMaterial.MaximumRecoveryVelocity, inverseDtWide, softnessImpulseScale, contactImpulse.Penetration, wsvA, wsvB);
            Vector3Wide x, z;
            Helpers::BuildOrthonormalBasis(contact.Normal, x, z);
            VF maximumTangentImpulse = prestepMaterial.FrictionCoefficient * contactImpulse.Penetration;
            TangentFriction::Solve(x, z, contact.Offset, contactOffsetB, inertiaA, inertiaB, maximumTangentImpulse, contactImpulse.Tangent, wsvA, wsvB);
        }
    }
    static void IncrementallyUpdateForSubstep(const VF& dt, const BodyVelocityWide& wsvA, const BodyVelocityWide& wsvB, Prestep& prestep) {  // :290
        Vector3Wide& prestepOffsetB = prestep.OffsetB;
        for (int i = 0; i < N; ++i) {
            NonconvexContactPrestepData& prestepContact = prestep.Contact[i];
            PenetrationLimit::UpdatePenetrationDepth(dt, prestepContact.Offset, prestepOffsetB, prestepContact.Normal, wsvA, wsvB, prestepContact.Depth);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------- AreaConstraint (type id 36, three bodies)
struct AreaConstraintPrestepData { VF TargetScaledArea; SpringSettingsWide SpringSettings; };  // AreaConstraint.cs:70
struct AreaConstraintFunctions {                                                              // AreaConstraint.cs:76
    typedef AreaConstraintPrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const VF& inverseMassA, const VF& inverseMassB, const VF& inverseMassC, const Vector3Wide& negatedJacobianA, const Vector3Wide& jacobianB,
                             const Vector3Wide& jacobianC, const VF& impulse, BodyVelocityWide& velocityA, BodyVelocityWide& velocityB, BodyVelocityWide& velocityC) {  // :79
        Vector3Wide negativeVelocityChangeA, velocityChangeB, velocityChangeC;
        Vector3Wide::Scale(negatedJacobianA, inverseMassA * impulse, negativeVelocityChangeA);
        Vector3Wide::Scale(jacobianB, inverseMassB * impulse, velocityChangeB);
        Vector3Wide::Scale(jacobianC, inverseMassC * impulse, velocityChangeC);
        Vector3Wide::Subtract(velocityA.Linear, negativeVelocityChangeA, velocityA.Linear);
        Vector3Wide::Add(velocityB.Linear, velocityChangeB, velocityB.Linear);
        Vector3Wide::Add(velocityC.Linear, velocityChangeC, velocityC.Linear);
    }
    static void ComputeJacobian(const Vector3Wide& positionA, const Vector3Wide& positionB, const Vector3Wide& positionC, VF& normalLength, Vector3Wide& negatedJacobianA, Vector3Wide& jacobianB,
                                Vector3Wide& jacobianC, VF& contributionA, VF& contributionB, VF& contributionC, VF& inverseJacobianLength) {  // :92
        Vector3Wide ab = positionB - positionA;
        Vector3Wide ac = positionC - positionA;
        Vector3Wide abxac, normal;
        Vector3Wide::CrossWithoutOverlap(ab, ac, abxac);
        Vector3Wide::Length(abxac, normalLength);
        Vector3Wide::Scale(abxac, ConditionalSelect(GreaterThan(normalLength, vf(1e-10f)), kOne / normalLength, kZero), normal);
        Vector3Wide::CrossWithoutOverlap(ac, normal, jacobianB);
        Vector3Wide::CrossWithoutOverlap(normal, ab, jacobianC);
        Vector3Wide::Add(jacobianB, jacobianC, negatedJacobianA);
        Vector3Wide::Dot(negatedJacobianA, negatedJacobianA, contributionA);
        Vector3Wide::Dot(jacobianB, jacobianB, contributionB);
        Vector3Wide::Dot(jacobianC, jacobianC, contributionC);
        VF jacobianLengthSquared = contributionA + contributionB + contributionC;
        jacobianLengthSquared = Max(vf(1e-14f), jacobianLengthSquared);
        inverseJacobianLength = FastReciprocalSquareRoot(jacobianLengthSquared);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB, const QuaternionWide& orientationB,
                          const BodyInertiaWide& inertiaB, const Vector3Wide& positionC, const QuaternionWide& orientationC, const BodyInertiaWide& inertiaC, Prestep& prestep,
                          Impulses& accumulatedImpulses, BodyVelocityWide& wsvA, BodyVelocityWide& wsvB, BodyVelocityWide& wsvC) {  // :140
        VF normalLength, contributionA, contributionB, contributionC, inverseJacobianLength;
        Vector3Wide negatedJacobianA, jacobianB, jacobianC;
        ComputeJacobian(positionA, positionB, positionC, normalLength, negatedJacobianA, jacobianB, jacobianC, contributionA, contributionB, contributionC, inverseJacobianLength);
        ApplyImpulse(inertiaA.InverseMass, inertiaB.InverseMass, inertiaC.InverseMass, negatedJacobianA, jacobianB, jacobianC, inverseJacobianLength * accumulatedImpulses, wsvA, wsvB, wsvC);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB, const QuaternionWide& orientationB,
                      const BodyInertiaWide& inertiaB, const Vector3Wide& positionC, const QuaternionWide& orientationC, const BodyInertiaWide& inertiaC, float dt, float inverseDt,
                      Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA, BodyVelocityWide& wsvB, BodyVelocityWide& wsvC) {  // :152
        VF normalLength, contributionA, contributionB, contributionC, inverseJacobianLength;
        Vector3Wide negatedJacobianA, jacobianB, jacobianC;
        ComputeJacobian(positionA, positionB, positionC, normalLength, negatedJacobianA, jacobianB, jacobianC, contributionA, contributionB, contributionC, inverseJacobianLength);
        VF inverseJacobianLengthSquared = inverseJacobianLength * inverseJacobianLength;
        VF inverseEffectiveMass =
            Max(vf(1e-14f), inverseJacobianLengthSquared * (contributionA * inertiaA.InverseMass + contributionB * inertiaB.InverseMass + contributionC * inertiaC.InverseMass));
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF effectiveMass = effectiveMassCFMScale / inverseEffectiveMass;
        VF biasVelocity = (prestep.TargetScaledArea - normalLength) * inverseJacobianLength * positionErrorToVelocity;
        VF negatedVelocityContributionA, velocityContributionB, velocityContributionC;
        Vector3Wide::Dot(negatedJacobianA, wsvA.Linear, negatedVelocityContributionA);
        Vector3Wide::Dot(jacobianB, wsvB.Linear, velocityContributionB);
        Vector3Wide::Dot(jacobianC, wsvC.Linear, velocityContributionC);
        VF csv = inverseJacobianLength * (velocityContributionB + velocityContributionC - negatedVelocityContributionA);
        VF csi = (biasVelocity - csv) * effectiveMass - accumulatedImpulses * softnessImpulseScale;
        accumulatedImpulses = accumulatedImpulses + csi;
        ApplyImpulse(inertiaA.InverseMass, inertiaB.InverseMass, inertiaC.InverseMass, negatedJacobianA, jacobianB, jacobianC, inverseJacobianLength * csi, wsvA, wsvB, wsvC);
    }
};

// ---------------------------------------------------------------------------------------------------------------- VolumeConstraint (type id 32, four bodies)
struct VolumeConstraintPrestepData { VF TargetScaledVolume; SpringSettingsWide SpringSettings; };  // VolumeConstraint.cs:70
struct VolumeConstraintFunctions {                                                                // VolumeConstraint.cs:76
    typedef VolumeConstraintPrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const VF& inverseMassA, const VF& inverseMassB, const VF& inverseMassC, const VF& inverseMassD, const Vector3Wide& negatedJacobianA, const Vector3Wide& jacobianB,
                             const Vector3Wide& jacobianC, const Vector3Wide& jacobianD, const VF& impulse, BodyVelocityWide& velocityA, BodyVelocityWide& velocityB, BodyVelocityWide& velocityC,
                             BodyVelocityWide& velocityD) {  // :79
        Vector3Wide negativeVelocityChangeA, velocityChangeB, velocityChangeC, velocityChangeD;
        Vector3Wide::Scale(negatedJacobianA, inverseMassA * impulse, negativeVelocityChangeA);
        Vector3Wide::Scale(jacobianB, inverseMassB * impulse, velocityChangeB);
        Vector3Wide::Scale(jacobianC, inverseMassC * impulse, velocityChangeC);
        Vector3Wide::Scale(jacobianD, inverseMassD * impulse, velocityChangeD);
        Vector3Wide::Subtract(velocityA.Linear, negativeVelocityChangeA, velocityA.Linear);
        Vector3Wide::Add(velocityB.Linear, velocityChangeB, velocityB.Linear);
        Vector3Wide::Add(velocityC.Linear, velocityChangeC, velocityC.Linear);
        Vector3Wide::Add(velocityD.Linear, velocityChangeD, velocityD.Linear);
    }
    static void ComputeJacobian(const Vector3Wide& positionA, const Vector3Wide& positionB, const Vector3Wide& positionC, const Vector3Wide& positionD, Vector3Wide& ad, Vector3Wide& negatedJA,
                                Vector3Wide& jacobianB, Vector3Wide& jacobianC, Vector3Wide& jacobianD, VF& contributionA, VF& contributionB, VF& contributionC, VF& contributionD,
                                VF& inverseJacobianLength) {  // :95
        Vector3Wide ab = positionB - positionA;
        Vector3Wide ac = positionC - positionA;
        ad = positionD - positionA;
        Vector3Wide::CrossWithoutOverlap(ac, ad, jacobianB);
        Vector3Wide::CrossWithoutOverlap(ad, ab, jacobianC);
        Vector3Wide::CrossWithoutOverlap(ab, ac, jacobianD);
        Vector3Wide::Add(jacobianB, jacobianC, negatedJA);
        Vector3Wide::Add(jacobianD, negatedJA, negatedJA);
        Vector3Wide::Dot(negatedJA, negatedJA, contributionA);
        Vector3Wide::Dot(jacobianB, jacobianB, contributionB);
        Vector3Wide::Dot(jacobianC, jacobianC, contributionC);
        Vector3Wide::Dot(jacobianD, jacobianD, contributionD);
        VF jacobianLengthSquared = contributionA + contributionB + contributionC + contributionD;
        jacobianLengthSquared = Max(vf(1e-14f), jacobianLengthSquared);
        inverseJacobianLength = FastReciprocalSquareRoot(jacobianLengthSquared);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB, const QuaternionWide& orientationB,
                          const BodyInertiaWide& inertiaB, const Vector3Wide& positionC, const QuaternionWide& orientationC, const BodyInertiaWide& inertiaC, const Vector3Wide& positionD,
                          const QuaternionWide& orientationD, const BodyInertiaWide& inertiaD, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA, BodyVelocityWide& wsvB,
                          BodyVelocityWide& wsvC, BodyVelocityWide& wsvD) {  // :125
        Vector3Wide ad, negatedJA, jacobianB, jacobianC, jacobianD;
        VF contributionA, contributionB, contributionC, contributionD, inverseJacobianLength;
        ComputeJacobian(positionA, positionB, positionC, positionD, ad, negatedJA, jacobianB, jacobianC, jacobianD, contributionA, contributionB, contributionC, contributionD,
                        inverseJacobianLength);
        ApplyImpulse(inertiaA.InverseMass, inertiaB.InverseMass, inertiaC.InverseMass, inertiaD.InverseMass, negatedJA, jacobianB, jacobianC, jacobianD,
                     inverseJacobianLength * accumulatedImpulses, wsvA, wsvB, wsvC, wsvD);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB, const QuaternionWide& orientationB,
                      const BodyInertiaWide& inertiaB, const Vector3Wide& positionC, const QuaternionWide& orientationC, const BodyInertiaWide& inertiaC, const Vector3Wide& positionD,
                      const QuaternionWide& orientationD, const BodyInertiaWide& inertiaD, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                      BodyVelocityWide& wsvB, BodyVelocityWide& wsvC, BodyVelocityWide& wsvD) {  // :133
        Vector3Wide ad, negatedJA, jacobianB, jacobianC, jacobianD;
        VF contributionA, contributionB, contributionC, contributionD, inverseJacobianLength;
        ComputeJacobian(positionA, positionB, positionC, positionD, ad, negatedJA, jacobianB, jacobianC, jacobianD, contributionA, contributionB, contributionC, contributionD,
                        inverseJacobianLength);
        VF inverseJacobianLengthSquared = inverseJacobianLength * inverseJacobianLength;
        VF inverseEffectiveMass = Max(vf(1e-14f), inverseJacobianLengthSquared * (contributionA * inertiaA.InverseMass + contributionB * inertiaB.InverseMass +
                                                                                contributionC * inertiaC.InverseMass + contributionD * inertiaD.InverseMass));
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF effectiveMass = effectiveMassCFMScale / inverseEffectiveMass;
        VF volume;
        Vector3Wide::Dot(jacobianD, ad, volume);
        VF biasVelocity = (prestep.TargetScaledVolume - volume) * inverseJacobianLength * positionErrorToVelocity;
        VF negatedVelocityContributionA, velocityContributionB, velocityContributionC, velocityContributionD;
        Vector3Wide::Dot(negatedJA, wsvA.Linear, negatedVelocityContributionA);
        Vector3Wide::Dot(jacobianB, wsvB.Linear, velocityContributionB);
        Vector3Wide::Dot(jacobianC, wsvC.Linear, velocityContributionC);
        Vector3Wide::Dot(jacobianD, wsvD.Linear, velocityContributionD);
        VF csv = inverseJacobianLength * (velocityContributionB + velocityContributionC + velocityContributionD - negatedVelocityContributionA);
        VF csi = (biasVelocity - csv) * effectiveMass - accumulatedImpulses * softnessImpulseScale;
        accumulatedImpulses = accumulatedImpulses + csi;
        ApplyImpulse(inertiaA.InverseMass, inertiaB.InverseMass, inertiaC.InverseMass, inertiaD.InverseMass, negatedJA, jacobianB, jacobianC, jacobianD, inverseJacobianLength * csi, wsvA,
                     wsvB, wsvC, wsvD);
    }
};

}  // namespace wide
