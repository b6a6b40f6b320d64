// oracle/wide — TEST INFRASTRUCTURE (parity unpinned, see wide_vec.h). The joint constraint functions of SURVEY.md 8(a) rows a8-a13, transcribed bundle-for-bundle from
// BepuPhysics/Constraints/*.cs (file:line cited per function). Statement order and association follow the C#.
#pragma once
#include "wide_math.h"

namespace wide {

struct BodyVelocityWide { Vector3Wide Linear, Angular; };                       // BepuPhysics/BodyProperties.cs (BodyVelocityWide)
struct BodyInertiaWide { Symmetric3x3Wide InverseInertiaTensor; VF InverseMass; };  // BodyInertiaWide

struct SpringSettingsWide {  // BepuPhysics/Constraints/SpringSettings.cs:9-55
    VF AngularFrequency, TwiceDampingRatio;
    static void ComputeSpringiness(const SpringSettingsWide& settings, float dt, VF& positionErrorToVelocity, VF& effectiveMassCFMScale, VF& softnessImpulseScale) {  // :37
        VF angularFrequencyDt = settings.AngularFrequency * vf(dt);
        positionErrorToVelocity = settings.AngularFrequency / (angularFrequencyDt + settings.TwiceDampingRatio);
        VF extra = kOne / (angularFrequencyDt * (angularFrequencyDt + settings.TwiceDampingRatio));
        effectiveMassCFMScale = kOne / (kOne + extra);
        softnessImpulseScale = extra * effectiveMassCFMScale;
    }
};

struct MotorSettingsWide {  // BepuPhysics/Constraints/MotorSettings.cs:50-99
    VF MaximumForce, Damping;
    static void ComputeSoftness(const MotorSettingsWide& settings, float dt, VF& effectiveMassCFMScale, VF& softnessImpulseScale, VF& maximumImpulse) {  // :70
        VF dtWide = vf(dt);
        VF dtd = dtWide * settings.Damping;
        maximumImpulse = settings.MaximumForce * dtWide;
        softnessImpulseScale = kOne / (dtd + kOne);
        effectiveMassCFMScale = dtd * softnessImpulseScale;
    }
};

struct ServoSettingsWide {  // BepuPhysics/Constraints/ServoSettings.cs:68-178
    VF MaximumSpeed, BaseSpeed, MaximumForce;
    static void ComputeClampedBiasVelocity(const VF& error, const VF& positionErrorToVelocity, const ServoSettingsWide& servoSettings, float dt, float inverseDt,
                                           VF& clampedBiasVelocity, VF& maximumImpulse) {  // :75
        VF baseSpeed = Min(servoSettings.BaseSpeed, Abs(error) * vf(inverseDt));
        VF biasVelocity = error * positionErrorToVelocity;
        clampedBiasVelocity = ConditionalSelect(LessThan(biasVelocity, kZero),
                                                Max(neg(servoSettings.MaximumSpeed), Min(neg(baseSpeed), biasVelocity)),
                                                Min(servoSettings.MaximumSpeed, Max(baseSpeed, biasVelocity)));
        maximumImpulse = servoSettings.MaximumForce * vf(dt);
    }
    static void ClampImpulse(const VF& maximumImpulse, VF& accumulatedImpulse, VF& csi) {  // :145
        VF previousImpulse = accumulatedImpulse;
        accumulatedImpulse = Max(neg(maximumImpulse), Min(maximumImpulse, accumulatedImpulse + csi));
        csi = accumulatedImpulse - previousImpulse;
    }
    static void ClampImpulse(const VF& maximumImpulse, Vector3Wide& accumulatedImpulse, Vector3Wide& csi) {  // :167
        Vector3Wide previousAccumulatedImpulse = accumulatedImpulse;
        Vector3Wide::Add(accumulatedImpulse, csi, accumulatedImpulse);
        VF impulseMagnitude;
        Vector3Wide::Length(accumulatedImpulse, impulseMagnitude);
        VF impulseScale = ConditionalSelect(LessThan(Abs(impulseMagnitude), vf(1e-10f)), kOne, Min(maximumImpulse / impulseMagnitude, kOne));
        Vector3Wide::Scale(accumulatedImpulse, impulseScale, accumulatedImpulse);
        Vector3Wide::Subtract(accumulatedImpulse, previousAccumulatedImpulse, csi);
    }
};

namespace InequalityHelpers {  // BepuPhysics/Constraints/InequalityHelpers.cs
static inline void ClampPositive(VF& accumulatedImpulse, VF& impulse) {  // :15
    VF previous = accumulatedImpulse;
    accumulatedImpulse = Max(kZero, accumulatedImpulse + impulse);
    impulse = accumulatedImpulse - previous;
}
}  // namespace InequalityHelpers

// ---------------------------------------------------------------------------------------------------------------- BallSocket (type id 22)
namespace BallSocketShared {  // BepuPhysics/Constraints/BallSocketShared.cs
static inline void ComputeEffectiveMass(const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector3Wide& offsetA, const Vector3Wide& offsetB,
                                        const VF& effectiveMassCFMScale, Symmetric3x3Wide& effectiveMass) {  // :19
    Symmetric3x3Wide inverseEffectiveMass, angularBContribution;
    Symmetric3x3Wide::SkewSandwichWithoutOverlap(offsetA, inertiaA.InverseInertiaTensor, inverseEffectiveMass);
    Symmetric3x3Wide::SkewSandwichWithoutOverlap(offsetB, inertiaB.InverseInertiaTensor, angularBContribution);
    Symmetric3x3Wide::Add(inverseEffectiveMass, angularBContribution, inverseEffectiveMass);
    VF linearContribution = inertiaA.InverseMass + inertiaB.InverseMass;
    inverseEffectiveMass.XX += linearContribution;
    inverseEffectiveMass.YY += linearContribution;
    inverseEffectiveMass.ZZ += linearContribution;
    Symmetric3x3Wide::Invert(inverseEffectiveMass, effectiveMass);
    Symmetric3x3Wide::Scale(effectiveMass, effectiveMassCFMScale, effectiveMass);
}
static inline void ApplyImpulse(BodyVelocityWide& velocityA, BodyVelocityWide& velocityB, const Vector3Wide& offsetA, const Vector3Wide& offsetB,
                                const BodyInertiaWide& inertiaA, const BodyInertiaWide& inertiaB, const Vector3Wide& constraintSpaceImpulse) {  // :77
    Vector3Wide wsi, change;
    Vector3Wide::CrossWithoutOverlap(offsetA, constraintSpaceImpulse, wsi);
    Symmetric3x3Wide::TransformWithoutOverlap(wsi, inertiaA.InverseInertiaTensor, change);
    Vector3Wide::Add(velocityA.Angular, change, velocityA.Angular);
    Vector3Wide::Scale(constraintSpaceImpulse, inertiaA.InverseMass, change);
    Vector3Wide::Add(velocityA.Linear, change, velocityA.Linear);
    Vector3Wide::CrossWithoutOverlap(constraintSpaceImpulse, offsetB, wsi);
    Symmetric3x3Wide::TransformWithoutOverlap(wsi, inertiaB.InverseInertiaTensor, change);
    Vector3Wide::Add(velocityB.Angular, change, velocityB.Angular);
    Vector3Wide::Scale(constraintSpaceImpulse, inertiaB.InverseMass, change);
    Vector3Wide::Subtract(velocityB.Linear, change, velocityB.Linear);
}
static inline void ComputeCorrectiveImpulse(BodyVelocityWide& velocityA, BodyVelocityWide& velocityB, const Vector3Wide& offsetA, const Vector3Wide& offsetB,
                                            const Vector3Wide& biasVelocity, const Symmetric3x3Wide& effectiveMass, const VF& softnessImpulseScale,
                                            const Vector3Wide& accumulatedImpulse, Vector3Wide& correctiveImpulse) {  // :96
    Vector3Wide csv, angularCSV;
    Vector3Wide::Subtract(velocityA.Linear, velocityB.Linear, csv);
    Vector3Wide::CrossWithoutOverlap(velocityA.Angular, offsetA, angularCSV);
    Vector3Wide::Add(csv, angularCSV, csv);
    Vector3Wide::CrossWithoutOverlap(offsetB, velocityB.Angular, angularCSV);
    Vector3Wide::Add(csv, angularCSV, csv);
    Vector3Wide::Subtract(biasVelocity, csv, csv);
    Symmetric3x3Wide::TransformWithoutOverlap(csv, effectiveMass, correctiveImpulse);
    Vector3Wide softness;
    Vector3Wide::Scale(accumulatedImpulse, softnessImpulseScale, softness);
    Vector3Wide::Subtract(correctiveImpulse, softness, correctiveImpulse);
}
static inline void Solve(BodyVelocityWide& velocityA, BodyVelocityWide& velocityB, const Vector3Wide& offsetA, const Vector3Wide& offsetB, const Vector3Wide& biasVelocity,
                         const Symmetric3x3Wide& effectiveMass, const VF& softnessImpulseScale, Vector3Wide& accumulatedImpulse, const BodyInertiaWide& inertiaA,
                         const BodyInertiaWide& inertiaB) {  // :115
    Vector3Wide correctiveImpulse;
    ComputeCorrectiveImpulse(velocityA, velocityB, offsetA, offsetB, biasVelocity, effectiveMass, softnessImpulseScale, accumulatedImpulse, correctiveImpulse);
    Vector3Wide::Add(accumulatedImpulse, correctiveImpulse, accumulatedImpulse);
    ApplyImpulse(velocityA, velocityB, offsetA, offsetB, inertiaA, inertiaB, correctiveImpulse);
}
}  // namespace BallSocketShared

struct BallSocketPrestepData { Vector3Wide LocalOffsetA, LocalOffsetB; SpringSettingsWide SpringSettings; };  // BallSocket.cs:60
struct BallSocketFunctions {                                                                                   // BallSocket.cs:66
    typedef BallSocketPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :68
        Vector3Wide offsetA, offsetB;
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetA, orientationA, offsetA);
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetB, orientationB, offsetB);
        BallSocketShared::ApplyImpulse(wsvA, wsvB, offsetA, offsetB, inertiaA, inertiaB, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :76
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
        Vector3Wide::Scale(error, positionErrorToVelocity, biasVelocity);
        BallSocketShared::Solve(wsvA, wsvB, offsetA, offsetB, biasVelocity, effectiveMass, softnessImpulseScale, accumulatedImpulses, inertiaA, inertiaB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- AngularHinge (type id 23)
struct AngularHingePrestepData { Vector3Wide LocalHingeAxisA, LocalHingeAxisB; SpringSettingsWide SpringSettings; };  // AngularHinge.cs:64
struct AngularHingeFunctions {                                                                                         // AngularHinge.cs:71
    typedef AngularHingePrestepData Prestep;
    typedef Vector2Wide Impulses;
    static void GetErrorAngles(const Vector3Wide& hingeAxisA, const Vector3Wide& hingeAxisB, const Matrix2x3Wide& jacobianA, Vector2Wide& errorAngles) {  // :74
        VF hingeAxisBDotX, hingeAxisBDotY;
        Vector3Wide::Dot(hingeAxisB, jacobianA.X, hingeAxisBDotX);
        Vector3Wide::Dot(hingeAxisB, jacobianA.Y, hingeAxisBDotY);
        Vector3Wide toRemoveX, toRemoveY, hingeAxisBOnPlaneX, hingeAxisBOnPlaneY;
        Vector3Wide::Scale(jacobianA.X, hingeAxisBDotX, toRemoveX);
        Vector3Wide::Scale(jacobianA.Y, hingeAxisBDotY, toRemoveY);
        Vector3Wide::Subtract(hingeAxisB, toRemoveX, hingeAxisBOnPlaneX);
        Vector3Wide::Subtract(hingeAxisB, toRemoveY, hingeAxisBOnPlaneY);
        VF xLength, yLength;
        Vector3Wide::Length(hingeAxisBOnPlaneX, xLength);
        Vector3Wide::Length(hingeAxisBOnPlaneY, yLength);
        VF scaleX = kOne / xLength;
        VF scaleY = kOne / yLength;
        Vector3Wide::Scale(hingeAxisBOnPlaneX, scaleX, hingeAxisBOnPlaneX);
        Vector3Wide::Scale(hingeAxisBOnPlaneY, scaleY, hingeAxisBOnPlaneY);
        VF epsilon = vf(1e-7f);
        VI useFallbackX = LessThan(xLength, epsilon);
        VI useFallbackY = LessThan(yLength, epsilon);
        Vector3Wide::ConditionalSelect(useFallbackX, hingeAxisA, hingeAxisBOnPlaneX, hingeAxisBOnPlaneX);
        Vector3Wide::ConditionalSelect(useFallbackY, hingeAxisA, hingeAxisBOnPlaneY, hingeAxisBOnPlaneY);
        VF hbxha, hbyha;
        Vector3Wide::Dot(hingeAxisBOnPlaneX, hingeAxisA, hbxha);
        Vector3Wide::Dot(hingeAxisBOnPlaneY, hingeAxisA, hbyha);
        errorAngles.X = MathHelper::Acos(hbxha);
        errorAngles.Y = MathHelper::Acos(hbyha);
        VF hbxay, hbyax;
        Vector3Wide::Dot(hingeAxisBOnPlaneX, jacobianA.Y, hbxay);
        Vector3Wide::Dot(hingeAxisBOnPlaneY, jacobianA.X, hbyax);
        errorAngles.X = ConditionalSelect(LessThan(hbxay, kZero), errorAngles.X, neg(errorAngles.X));
        errorAngles.Y = ConditionalSelect(LessThan(hbyax, kZero), neg(errorAngles.Y), errorAngles.Y);
    }
    static void ApplyImpulse(const Matrix2x3Wide& impulseToVelocityA, const Matrix2x3Wide& negatedImpulseToVelocityB, const Vector2Wide& csi, Vector3Wide& angularVelocityA,
                             Vector3Wide& angularVelocityB) {  // :114
        Vector3Wide velocityChangeA, negatedVelocityChangeB;
        Matrix2x3Wide::Transform(csi, impulseToVelocityA, velocityChangeA);
        Vector3Wide::Add(angularVelocityA, velocityChangeA, angularVelocityA);
        Matrix2x3Wide::Transform(csi, negatedImpulseToVelocityB, negatedVelocityChangeB);
        Vector3Wide::Subtract(angularVelocityB, negatedVelocityChangeB, angularVelocityB);
    }
    static void ComputeJacobians(const Vector3Wide& localHingeAxisA, const QuaternionWide& orientationA, Vector3Wide& hingeAxisA, Matrix2x3Wide& jacobianA) {  // :123
        Vector3Wide localAX, localAY;
        Helpers::BuildOrthonormalBasis(localHingeAxisA, localAX, localAY);
        Matrix3x3Wide orientationMatrixA;
        Matrix3x3Wide::CreateFromQuaternion(orientationA, orientationMatrixA);
        Matrix3x3Wide::TransformWithoutOverlap(localHingeAxisA, orientationMatrixA, hingeAxisA);
        Matrix3x3Wide::TransformWithoutOverlap(localAX, orientationMatrixA, jacobianA.X);
        Matrix3x3Wide::TransformWithoutOverlap(localAY, orientationMatrixA, jacobianA.Y);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :133
        Vector3Wide unusedAxis;
        Matrix2x3Wide jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobians(prestep.LocalHingeAxisA, orientationA, unusedAxis, jacobianA);
        Symmetric3x3Wide::MultiplyWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::MultiplyWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses, wsvA.Angular, wsvB.Angular);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :141
        Vector3Wide hingeAxisA, hingeAxisB;
        Matrix2x3Wide jacobianA;
        ComputeJacobians(prestep.LocalHingeAxisA, orientationA, hingeAxisA, jacobianA);
        QuaternionWide::TransformWithoutOverlap(prestep.LocalHingeAxisB, orientationB, hingeAxisB);
        Matrix2x3Wide impulseToVelocityA, negatedImpulseToVelocityB;
        Symmetric3x3Wide::MultiplyWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::MultiplyWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        Symmetric2x2Wide angularA, angularB, inverseEffectiveMass, effectiveMass;
        Symmetric2x2Wide::CompleteMatrixSandwich(impulseToVelocityA, jacobianA, angularA);
        Symmetric2x2Wide::CompleteMatrixSandwich(negatedImpulseToVelocityB, jacobianA, angularB);
        Symmetric2x2Wide::Add(angularA, angularB, inverseEffectiveMass);
        Symmetric2x2Wide::InvertWithoutOverlap(inverseEffectiveMass, effectiveMass);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Vector2Wide errorAngle;
        GetErrorAngles(hingeAxisA, hingeAxisB, jacobianA, errorAngle);
        Vector2Wide biasVelocity, biasImpulse;
        Vector2Wide::Scale(errorAngle, neg(positionErrorToVelocity), biasVelocity);
        Symmetric2x2Wide::TransformWithoutOverlap(biasVelocity, effectiveMass, biasImpulse);
        Vector3Wide difference;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, difference);
        Vector2Wide csv, csi;
        Matrix2x3Wide::TransformByTransposeWithoutOverlap(difference, jacobianA, csv);
        Symmetric2x2Wide::TransformWithoutOverlap(csv, effectiveMass, csi);
        Vector2Wide::Scale(csi, effectiveMassCFMScale, csi);
        Vector2Wide softnessContribution;
        Vector2Wide::Scale(accumulatedImpulses, softnessImpulseScale, softnessContribution);
        Vector2Wide::Add(softnessContribution, csi, csi);
        Vector2Wide::Subtract(biasImpulse, csi, csi);
        Vector2Wide::Add(accumulatedImpulses, csi, accumulatedImpulses);
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, csi, wsvA.Angular, wsvB.Angular);
    }
};

// ---------------------------------------------------------------------------------------------------------------- SwingLimit (type id 25)
struct SwingLimitPrestepData { Vector3Wide AxisLocalA, AxisLocalB; VF MinimumDot; SpringSettingsWide SpringSettings; };  // SwingLimit.cs:84
struct SwingLimitFunctions {                                                                                              // SwingLimit.cs:92
    typedef SwingLimitPrestepData Prestep;
    typedef VF Impulses;
    static void ApplyImpulse(const Vector3Wide& impulseToVelocityA, const Vector3Wide& negatedImpulseToVelocityB, const VF& csi, Vector3Wide& angularVelocityA,
                             Vector3Wide& angularVelocityB) {  // :95
        Vector3Wide velocityChangeA, negatedVelocityChangeB;
        Vector3Wide::Scale(impulseToVelocityA, csi, velocityChangeA);
        Vector3Wide::Add(angularVelocityA, velocityChangeA, angularVelocityA);
        Vector3Wide::Scale(negatedImpulseToVelocityB, csi, negatedVelocityChangeB);
        Vector3Wide::Subtract(angularVelocityB, negatedVelocityChangeB, angularVelocityB);
    }
    static void ComputeJacobian(const Vector3Wide& axisLocalA, const Vector3Wide& axisLocalB, const QuaternionWide& orientationA, const QuaternionWide& orientationB,
                                Vector3Wide& axisA, Vector3Wide& axisB, Vector3Wide& jacobianA) {  // :104
        QuaternionWide::TransformWithoutOverlap(axisLocalA, orientationA, axisA);
        QuaternionWide::TransformWithoutOverlap(axisLocalB, orientationB, axisB);
        Vector3Wide::CrossWithoutOverlap(axisA, axisB, jacobianA);
        Vector3Wide fallbackJacobian;
        Helpers::FindPerpendicular(axisA, fallbackJacobian);
        VF jacobianLengthSquared;
        Vector3Wide::Dot(jacobianA, jacobianA, jacobianLengthSquared);
        VI useFallback = LessThan(jacobianLengthSquared, vf(1e-7f));
        Vector3Wide::ConditionalSelect(useFallback, fallbackJacobian, jacobianA, jacobianA);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :116
        Vector3Wide axisA, axisB, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(prestep.AxisLocalA, prestep.AxisLocalB, orientationA, orientationB, axisA, axisB, jacobianA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses, wsvA.Angular, wsvB.Angular);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :124
        Vector3Wide axisA, axisB, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(prestep.AxisLocalA, prestep.AxisLocalB, orientationA, orientationB, axisA, axisB, jacobianA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        VF angularContributionA, angularContributionB;
        Vector3Wide::Dot(impulseToVelocityA, jacobianA, angularContributionA);
        Vector3Wide::Dot(negatedImpulseToVelocityB, jacobianA, angularContributionB);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        VF effectiveMass = effectiveMassCFMScale / (angularContributionA + angularContributionB);
        VF axisDot;
        Vector3Wide::Dot(axisA, axisB, axisDot);
        VF error = axisDot - prestep.MinimumDot;
        VF biasVelocity = neg(Min(error * vf(inverseDt), error * positionErrorToVelocity));
        Vector3Wide difference;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, difference);
        VF csv;
        Vector3Wide::Dot(difference, jacobianA, csv);
        VF csi = effectiveMass * (biasVelocity - csv) - accumulatedImpulses * softnessImpulseScale;
        InequalityHelpers::ClampPositive(accumulatedImpulses, csi);
        ApplyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, csi, wsvA.Angular, wsvB.Angular);
    }
};

// ---------------------------------------------------------------------------------------------------------------- TwistServo (type id 26)
struct TwistServoPrestepData {  // TwistServo.cs:77
    QuaternionWide LocalBasisA, LocalBasisB;
    VF TargetAngle;
    SpringSettingsWide SpringSettings;
    ServoSettingsWide ServoSettings;
};
struct TwistServoFunctions {  // TwistServo.cs:86
    typedef TwistServoPrestepData Prestep;
    typedef VF Impulses;
    static void ComputeJacobian(const QuaternionWide& orientationA, const QuaternionWide& orientationB, const QuaternionWide& localBasisA, const QuaternionWide& localBasisB,
                                Vector3Wide& basisBX, Vector3Wide& basisBZ, Matrix3x3Wide& basisA, Vector3Wide& jacobianA) {  // :89
        QuaternionWide basisQuaternionA, basisQuaternionB;
        QuaternionWide::ConcatenateWithoutOverlap(localBasisA, orientationA, basisQuaternionA);
        QuaternionWide::ConcatenateWithoutOverlap(localBasisB, orientationB, basisQuaternionB);
        QuaternionWide::TransformUnitXZ(basisQuaternionB, basisBX, basisBZ);
        Matrix3x3Wide::CreateFromQuaternion(basisQuaternionA, basisA);
        Vector3Wide::Add(basisA.Z, basisBZ, jacobianA);
        VF length;
        Vector3Wide::Length(jacobianA, length);
        Vector3Wide::Scale(jacobianA, kOne / length, jacobianA);
        Vector3Wide::ConditionalSelect(LessThan(length, vf(1e-10f)), basisA.Z, jacobianA, jacobianA);
    }
    static void ComputeCurrentAngle(const Vector3Wide& basisBX, const Vector3Wide& basisBZ, const Matrix3x3Wide& basisA, VF& angle) {  // :117
        QuaternionWide aligningRotation;
        QuaternionWide::GetQuaternionBetweenNormalizedVectors(basisBZ, basisA.Z, aligningRotation);
        Vector3Wide alignedBasisBX;
        QuaternionWide::TransformWithoutOverlap(basisBX, aligningRotation, alignedBasisBX);
        VF x, y;
        Vector3Wide::Dot(alignedBasisBX, basisA.X, x);
        Vector3Wide::Dot(alignedBasisBX, basisA.Y, y);
        VF absAngle = MathHelper::Acos(x);
        angle = ConditionalSelect(LessThan(y, kZero), neg(absAngle), absAngle);
    }
    static void ComputeEffectiveMassContributions(const Symmetric3x3Wide& inverseInertiaA, const Symmetric3x3Wide& inverseInertiaB, const Vector3Wide& jacobianA,
                                                  Vector3Wide& impulseToVelocityA, Vector3Wide& negatedImpulseToVelocityB, VF& unsoftenedInverseEffectiveMass) {  // :133
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inverseInertiaA, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inverseInertiaB, negatedImpulseToVelocityB);
        VF angularA, angularB;
        Vector3Wide::Dot(impulseToVelocityA, jacobianA, angularA);
        Vector3Wide::Dot(negatedImpulseToVelocityB, jacobianA, angularB);
        unsoftenedInverseEffectiveMass = angularA + angularB;
    }
    static void ComputeEffectiveMass(float dt, const SpringSettingsWide& springSettings, const Symmetric3x3Wide& inverseInertiaA, const Symmetric3x3Wide& inverseInertiaB,
                                     const Vector3Wide& jacobianA, Vector3Wide& impulseToVelocityA, Vector3Wide& negatedImpulseToVelocityB, VF& positionErrorToVelocity,
                                     VF& softnessImpulseScale, VF& effectiveMass, Vector3Wide& velocityToImpulseA) {  // :147
        VF unsoftenedInverseEffectiveMass;
        ComputeEffectiveMassContributions(inverseInertiaA, inverseInertiaB, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB, unsoftenedInverseEffectiveMass);
        VF effectiveMassCFMScale;
        SpringSettingsWide::ComputeSpringiness(springSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        effectiveMass = effectiveMassCFMScale / unsoftenedInverseEffectiveMass;
        Vector3Wide::Scale(jacobianA, effectiveMass, velocityToImpulseA);
    }
    static void ApplyImpulse(Vector3Wide& angularVelocityA, Vector3Wide& angularVelocityB, const Vector3Wide& impulseToVelocityA, const Vector3Wide& negatedImpulseToVelocityB,
                             const VF& csi) {  // :161
        Vector3Wide velocityChangeA, negatedVelocityChangeB;
        Vector3Wide::Scale(impulseToVelocityA, csi, velocityChangeA);
        Vector3Wide::Add(angularVelocityA, velocityChangeA, angularVelocityA);
        Vector3Wide::Scale(negatedImpulseToVelocityB, csi, negatedVelocityChangeB);
        Vector3Wide::Subtract(angularVelocityB, negatedVelocityChangeB, angularVelocityB);
    }
    static void ComputeJacobian(const QuaternionWide& orientationA, const QuaternionWide& orientationB, const QuaternionWide& localBasisA, const QuaternionWide& localBasisB,
                                Vector3Wide& jacobianA) {  // :170
        QuaternionWide basisQuaternionA, basisQuaternionB;
        QuaternionWide::ConcatenateWithoutOverlap(localBasisA, orientationA, basisQuaternionA);
        QuaternionWide::ConcatenateWithoutOverlap(localBasisB, orientationB, basisQuaternionB);
        Vector3Wide basisAZ = QuaternionWide::TransformUnitZ(basisQuaternionA);
        Vector3Wide basisBZ = QuaternionWide::TransformUnitZ(basisQuaternionB);
        Vector3Wide::Add(basisAZ, basisBZ, jacobianA);
        VF length;
        Vector3Wide::Length(jacobianA, length);
        Vector3Wide::Scale(jacobianA, kOne / length, jacobianA);
        Vector3Wide::ConditionalSelect(LessThan(length, vf(1e-10f)), basisAZ, jacobianA, jacobianA);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :184
        Vector3Wide jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(orientationA, orientationB, prestep.LocalBasisA, prestep.LocalBasisB, jacobianA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        ApplyImpulse(wsvA.Angular, wsvB.Angular, impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :192
        Vector3Wide basisBX, basisBZ, jacobianA;
        Matrix3x3Wide basisA;
        ComputeJacobian(orientationA, orientationB, prestep.LocalBasisA, prestep.LocalBasisB, basisBX, basisBZ, basisA, jacobianA);
        Vector3Wide impulseToVelocityA, negatedImpulseToVelocityB, velocityToImpulseA;
        VF positionErrorToVelocity, softnessImpulseScale, effectiveMass;
        ComputeEffectiveMass(dt, prestep.SpringSettings, inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB,
                             positionErrorToVelocity, softnessImpulseScale, effectiveMass, velocityToImpulseA);
        VF angle;
        ComputeCurrentAngle(basisBX, basisBZ, basisA, angle);
        VF error;
        MathHelper::GetSignedAngleDifference(prestep.TargetAngle, angle, error);
        VF clampedBiasVelocity, maximumImpulse;
        ServoSettingsWide::ComputeClampedBiasVelocity(error, positionErrorToVelocity, prestep.ServoSettings, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
        VF biasImpulse = clampedBiasVelocity * effectiveMass;
        Vector3Wide netVelocity;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, netVelocity);
        VF csiVelocityComponent;
        Vector3Wide::Dot(netVelocity, velocityToImpulseA, csiVelocityComponent);
        VF csi = biasImpulse - accumulatedImpulses * softnessImpulseScale - csiVelocityComponent;
        VF previousAccumulatedImpulse = accumulatedImpulses;
        accumulatedImpulses = Min(Max(accumulatedImpulses + csi, neg(maximumImpulse)), maximumImpulse);
        csi = accumulatedImpulses - previousAccumulatedImpulse;
        ApplyImpulse(wsvA.Angular, wsvB.Angular, impulseToVelocityA, negatedImpulseToVelocityB, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- TwistLimit (type id 27)
struct TwistLimitPrestepData {  // TwistLimit.cs:77
    QuaternionWide LocalBasisA, LocalBasisB;
    VF MinimumAngle, MaximumAngle;
    SpringSettingsWide SpringSettings;
};
struct TwistLimitFunctions {  // TwistLimit.cs:86
    typedef TwistLimitPrestepData Prestep;
    typedef VF Impulses;
    static void ComputeJacobian(const QuaternionWide& orientationA, const QuaternionWide& orientationB, const QuaternionWide& localBasisA, const QuaternionWide& localBasisB,
                                const VF& minimumAngle, const VF& maximumAngle, VF& error, Vector3Wide& jacobianA) {  // :89
        Vector3Wide basisBX, basisBZ;
        Matrix3x3Wide basisA;
        TwistServoFunctions::ComputeJacobian(orientationA, orientationB, localBasisA, localBasisB, basisBX, basisBZ, basisA, jacobianA);
        VF angle;
        TwistServoFunctions::ComputeCurrentAngle(basisBX, basisBZ, basisA, angle);
        VF minError, maxError;
        MathHelper::GetSignedAngleDifference(minimumAngle, angle, minError);
        MathHelper::GetSignedAngleDifference(maximumAngle, angle, maxError);
        VI useMin = LessThan(Abs(minError), Abs(maxError));
        error = ConditionalSelect(useMin, neg(minError), maxError);
        Vector3Wide negatedJacobianA;
        Vector3Wide::Negate(jacobianA, negatedJacobianA);
        Vector3Wide::ConditionalSelect(useMin, negatedJacobianA, jacobianA, jacobianA);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :105
        VF error;
        Vector3Wide jacobianA, impulseToVelocityA, negatedImpulseToVelocityB;
        ComputeJacobian(orientationA, orientationB, prestep.LocalBasisA, prestep.LocalBasisB, prestep.MinimumAngle, prestep.MaximumAngle, error, jacobianA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaA.InverseInertiaTensor, impulseToVelocityA);
        Symmetric3x3Wide::TransformWithoutOverlap(jacobianA, inertiaB.InverseInertiaTensor, negatedImpulseToVelocityB);
        TwistServoFunctions::ApplyImpulse(wsvA.Angular, wsvB.Angular, impulseToVelocityA, negatedImpulseToVelocityB, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :113
        VF error;
        Vector3Wide jacobianA;
        ComputeJacobian(orientationA, orientationB, prestep.LocalBasisA, prestep.LocalBasisB, prestep.MinimumAngle, prestep.MaximumAngle, error, jacobianA);
        Vector3Wide impulseToVelocityA, negatedImpulseToVelocityB, velocityToImpulseA;
        VF positionErrorToVelocity, softnessImpulseScale, effectiveMass;
        TwistServoFunctions::ComputeEffectiveMass(dt, prestep.SpringSettings, inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, jacobianA, impulseToVelocityA,
                                                  negatedImpulseToVelocityB, positionErrorToVelocity, softnessImpulseScale, effectiveMass, velocityToImpulseA);
        VF biasVelocity = ConditionalSelect(LessThan(error, kZero), error * vf(inverseDt), error * positionErrorToVelocity);
        VF biasImpulse = biasVelocity * effectiveMass;
        Vector3Wide netVelocity;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, netVelocity);
        VF csiVelocityComponent;
        Vector3Wide::Dot(netVelocity, velocityToImpulseA, csiVelocityComponent);
        VF csi = biasImpulse - accumulatedImpulses * softnessImpulseScale - csiVelocityComponent;
        InequalityHelpers::ClampPositive(accumulatedImpulses, csi);
        TwistServoFunctions::ApplyImpulse(wsvA.Angular, wsvB.Angular, impulseToVelocityA, negatedImpulseToVelocityB, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- AngularMotor (type id 30)
namespace AngularServoFunctions {  // AngularServo.cs:73
static inline void ApplyImpulse(Vector3Wide& angularVelocityA, Vector3Wide& angularVelocityB, const Symmetric3x3Wide& impulseToVelocityA,
                                const Symmetric3x3Wide& negatedImpulseToVelocityB, const Vector3Wide& csi) {
    Vector3Wide velocityChangeA, negatedVelocityChangeB;
    Symmetric3x3Wide::TransformWithoutOverlap(csi, impulseToVelocityA, velocityChangeA);
    Vector3Wide::Add(angularVelocityA, velocityChangeA, angularVelocityA);
    Symmetric3x3Wide::TransformWithoutOverlap(csi, negatedImpulseToVelocityB, negatedVelocityChangeB);
    Vector3Wide::Subtract(angularVelocityB, negatedVelocityChangeB, angularVelocityB);
}
}  // namespace AngularServoFunctions

struct AngularMotorPrestepData { Vector3Wide TargetVelocityLocalA; MotorSettingsWide Settings; };  // AngularMotor.cs:55
struct AngularMotorFunctions {                                                                     // AngularMotor.cs:61
    typedef AngularMotorPrestepData Prestep;
    typedef Vector3Wide Impulses;
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :63
        AngularServoFunctions::ApplyImpulse(wsvA.Angular, wsvB.Angular, inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, accumulatedImpulses);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :68
        VF effectiveMassCFMScale, softnessImpulseScale, maximumImpulse;
        MotorSettingsWide::ComputeSoftness(prestep.Settings, dt, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        Symmetric3x3Wide unsoftenedInverseEffectiveMass, unsoftenedEffectiveMass;
        Symmetric3x3Wide::Add(inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, unsoftenedInverseEffectiveMass);
        Symmetric3x3Wide::Invert(unsoftenedInverseEffectiveMass, unsoftenedEffectiveMass);
        Vector3Wide biasVelocity;
        QuaternionWide::TransformWithoutOverlap(prestep.TargetVelocityLocalA, orientationA, biasVelocity);
        Vector3Wide csv, csi;
        Vector3Wide::Subtract(wsvA.Angular, wsvB.Angular, csv);
        Vector3Wide::Subtract(biasVelocity, csv, csv);
        Symmetric3x3Wide::TransformWithoutOverlap(csv, unsoftenedEffectiveMass, csi);
        csi = csi * effectiveMassCFMScale;
        Vector3Wide softnessComponent;
        Vector3Wide::Scale(accumulatedImpulses, softnessImpulseScale, softnessComponent);
        Vector3Wide::Subtract(csi, softnessComponent, csi);
        ServoSettingsWide::ClampImpulse(maximumImpulse, accumulatedImpulses, csi);
        AngularServoFunctions::ApplyImpulse(wsvA.Angular, wsvB.Angular, inertiaA.InverseInertiaTensor, inertiaB.InverseInertiaTensor, csi);
    }
};

// ---------------------------------------------------------------------------------------------------------------- SwivelHinge (type id 46)
struct SwivelHingePrestepData {  // SwivelHinge.cs:74
    Vector3Wide LocalOffsetA, LocalSwivelAxisA, LocalOffsetB, LocalHingeAxisB;
    SpringSettingsWide SpringSettings;
};
struct SwivelHingeFunctions {  // SwivelHinge.cs:83
    typedef SwivelHingePrestepData Prestep;
    typedef Vector4Wide Impulses;
    static void ApplyImpulse(const Vector3Wide& offsetA, const Vector3Wide& offsetB, const Vector3Wide& swivelHingeJacobian, const BodyInertiaWide& inertiaA,
                             const BodyInertiaWide& inertiaB, Vector4Wide& csi, BodyVelocityWide& velocityA, BodyVelocityWide& velocityB) {  // :86
        Vector3Wide& ballSocketCSI = *reinterpret_cast<Vector3Wide*>(&csi.X);  // Unsafe.As<Vector<float>, Vector3Wide>(ref csi.X)
        Vector3Wide linearChangeA;
        Vector3Wide::Scale(ballSocketCSI, inertiaA.InverseMass, linearChangeA);
        Vector3Wide::Add(velocityA.Linear, linearChangeA, velocityA.Linear);
        Vector3Wide ballSocketAngularImpulseA, swivelHingeAngularImpulseA, angularImpulseA, angularChangeA;
        Vector3Wide::CrossWithoutOverlap(offsetA, ballSocketCSI, ballSocketAngularImpulseA);
        Vector3Wide::Scale(swivelHingeJacobian, csi.W, swivelHingeAngularImpulseA);
        Vector3Wide::Add(ballSocketAngularImpulseA, swivelHingeAngularImpulseA, angularImpulseA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseA, inertiaA.InverseInertiaTensor, angularChangeA);
        Vector3Wide::Add(velocityA.Angular, angularChangeA, velocityA.Angular);
        Vector3Wide negatedLinearChangeB;
        Vector3Wide::Scale(ballSocketCSI, inertiaB.InverseMass, negatedLinearChangeB);
        Vector3Wide::Subtract(velocityB.Linear, negatedLinearChangeB, velocityB.Linear);
        Vector3Wide ballSocketAngularImpulseB, angularImpulseB, angularChangeB;
        Vector3Wide::CrossWithoutOverlap(ballSocketCSI, offsetB, ballSocketAngularImpulseB);
        Vector3Wide::Subtract(ballSocketAngularImpulseB, swivelHingeAngularImpulseA, angularImpulseB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseB, inertiaB.InverseInertiaTensor, angularChangeB);
        Vector3Wide::Add(velocityB.Angular, angularChangeB, velocityB.Angular);
    }
    static void ComputeJacobian(const Vector3Wide& localOffsetA, const Vector3Wide& localSwivelAxisA, const Vector3Wide& localOffsetB, const Vector3Wide& localHingeAxisB,
                                const QuaternionWide& orientationA, const QuaternionWide& orientationB, Vector3Wide& swivelAxis, Vector3Wide& hingeAxis, Vector3Wide& offsetA,
                                Vector3Wide& offsetB, Vector3Wide& swivelHingeJacobian) {  // :110
        Matrix3x3Wide orientationMatrixA, orientationMatrixB;
        Matrix3x3Wide::CreateFromQuaternion(orientationA, orientationMatrixA);
        Matrix3x3Wide::CreateFromQuaternion(orientationB, orientationMatrixB);
        Matrix3x3Wide::TransformWithoutOverlap(localOffsetA, orientationMatrixA, offsetA);
        Matrix3x3Wide::TransformWithoutOverlap(localSwivelAxisA, orientationMatrixA, swivelAxis);
        Matrix3x3Wide::TransformWithoutOverlap(localOffsetB, orientationMatrixB, offsetB);
        Matrix3x3Wide::TransformWithoutOverlap(localHingeAxisB, orientationMatrixB, hingeAxis);
        Vector3Wide::CrossWithoutOverlap(swivelAxis, hingeAxis, swivelHingeJacobian);
        VF lengthSquared = swivelHingeJacobian.X * swivelHingeJacobian.X + swivelHingeJacobian.Y * swivelHingeJacobian.Y + swivelHingeJacobian.Z * swivelHingeJacobian.Z;  // Vector3Wide.cs:605
        VI useFallbackJacobian = LessThan(lengthSquared, vf(1e-3f));
        Vector3Wide selected;
        Vector3Wide::ConditionalSelect(useFallbackJacobian, hingeAxis, swivelHingeJacobian, selected);
        swivelHingeJacobian = selected;
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :127
        Vector3Wide swivelAxis, hingeAxis, offsetA, offsetB, swivelHingeJacobian;
        ComputeJacobian(prestep.LocalOffsetA, prestep.LocalSwivelAxisA, prestep.LocalOffsetB, prestep.LocalHingeAxisB, orientationA, orientationB, swivelAxis, hingeAxis, offsetA,
                        offsetB, swivelHingeJacobian);
        ApplyImpulse(offsetA, offsetB, swivelHingeJacobian, inertiaA, inertiaB, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :134
        Vector3Wide swivelAxis, hingeAxis, offsetA, offsetB, swivelHingeJacobian;
        ComputeJacobian(prestep.LocalOffsetA, prestep.LocalSwivelAxisA, prestep.LocalOffsetB, prestep.LocalHingeAxisB, orientationA, orientationB, swivelAxis, hingeAxis, offsetA,
                        offsetB, swivelHingeJacobian);
        Symmetric3x3Wide ballSocketContributionAngularA, ballSocketContributionAngularB;
        Symmetric3x3Wide::SkewSandwichWithoutOverlap(offsetA, inertiaA.InverseInertiaTensor, ballSocketContributionAngularA);
        Symmetric3x3Wide::SkewSandwichWithoutOverlap(offsetB, inertiaB.InverseInertiaTensor, ballSocketContributionAngularB);
        Symmetric4x4Wide inverseEffectiveMass;
        Symmetric3x3Wide& upperLeft = *reinterpret_cast<Symmetric3x3Wide*>(&inverseEffectiveMass.XX);  // Symmetric4x4Wide.GetUpperLeft3x3Block
        Symmetric3x3Wide::Add(ballSocketContributionAngularA, ballSocketContributionAngularB, upperLeft);
        VF linearContribution = inertiaA.InverseMass + inertiaB.InverseMass;
        upperLeft.XX += linearContribution;
        upperLeft.YY += linearContribution;
        upperLeft.ZZ += linearContribution;
        Vector3Wide swivelHingeInertiaA, swivelHingeInertiaB;
        Symmetric3x3Wide::TransformWithoutOverlap(swivelHingeJacobian, inertiaA.InverseInertiaTensor, swivelHingeInertiaA);
        Symmetric3x3Wide::TransformWithoutOverlap(swivelHingeJacobian, inertiaB.InverseInertiaTensor, swivelHingeInertiaB);
        VF swivelHingeContributionAngularA, swivelHingeContributionAngularB;
        Vector3Wide::Dot(swivelHingeInertiaA, swivelHingeJacobian, swivelHingeContributionAngularA);
        Vector3Wide::Dot(swivelHingeInertiaB, swivelHingeJacobian, swivelHingeContributionAngularB);
        inverseEffectiveMass.WW = swivelHingeContributionAngularA + swivelHingeContributionAngularB;
        Vector3Wide offDiagonalContributionA, offDiagonalContributionB;
        Vector3Wide::CrossWithoutOverlap(swivelHingeInertiaA, offsetA, offDiagonalContributionA);
        Vector3Wide::CrossWithoutOverlap(swivelHingeInertiaB, offsetB, offDiagonalContributionB);
        Vector3Wide::Add(offDiagonalContributionA, offDiagonalContributionB, *reinterpret_cast<Vector3Wide*>(&inverseEffectiveMass.WX));  // GetUpperRight3x1Block
        Symmetric4x4Wide effectiveMass;
        Symmetric4x4Wide::InvertWithoutOverlap(inverseEffectiveMass, effectiveMass);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Vector3Wide anchorB, ballSocketError;
        Vector3Wide::Add(positionB - positionA, offsetB, anchorB);
        Vector3Wide::Subtract(anchorB, offsetA, ballSocketError);
        Vector4Wide biasVelocity;
        biasVelocity.X = ballSocketError.X * positionErrorToVelocity;
        biasVelocity.Y = ballSocketError.Y * positionErrorToVelocity;
        biasVelocity.Z = ballSocketError.Z * positionErrorToVelocity;
        VF error;
        Vector3Wide::Dot(hingeAxis, swivelAxis, error);
        biasVelocity.W = positionErrorToVelocity * neg(error);
        Vector3Wide ballSocketAngularCSVA, ballSocketAngularCSVB;
        VF swivelHingeCSVA, negatedSwivelHingeCSVB;
        Vector3Wide::CrossWithoutOverlap(wsvA.Angular, offsetA, ballSocketAngularCSVA);
        Vector3Wide::Dot(swivelHingeJacobian, wsvA.Angular, swivelHingeCSVA);
        Vector3Wide::CrossWithoutOverlap(offsetB, wsvB.Angular, ballSocketAngularCSVB);
        Vector3Wide::Dot(swivelHingeJacobian, wsvB.Angular, negatedSwivelHingeCSVB);
        Vector3Wide ballSocketAngularCSV, ballSocketLinearCSV;
        Vector3Wide::Add(ballSocketAngularCSVA, ballSocketAngularCSVB, ballSocketAngularCSV);
        Vector3Wide::Subtract(wsvA.Linear, wsvB.Linear, ballSocketLinearCSV);
        Vector4Wide csv;
        csv.X = ballSocketAngularCSV.X + ballSocketLinearCSV.X;
        csv.Y = ballSocketAngularCSV.Y + ballSocketLinearCSV.Y;
        csv.Z = ballSocketAngularCSV.Z + ballSocketLinearCSV.Z;
        csv.W = swivelHingeCSVA - negatedSwivelHingeCSVB;
        Vector4Wide::Subtract(biasVelocity, csv, csv);
        Vector4Wide csi, softnessContribution;
        Symmetric4x4Wide::TransformWithoutOverlap(csv, effectiveMass, csi);
        Vector4Wide::Scale(csi, effectiveMassCFMScale, csi);
        Vector4Wide::Scale(accumulatedImpulses, softnessImpulseScale, softnessContribution);
        Vector4Wide::Subtract(csi, softnessContribution, csi);
        accumulatedImpulses = accumulatedImpulses + csi;
        ApplyImpulse(offsetA, offsetB, swivelHingeJacobian, inertiaA, inertiaB, csi, wsvA, wsvB);
    }
};

// ---------------------------------------------------------------------------------------------------------------- Hinge (type id 47)
struct HingePrestepData {  // Hinge.cs:74
    Vector3Wide LocalOffsetA, LocalHingeAxisA, LocalOffsetB, LocalHingeAxisB;
    SpringSettingsWide SpringSettings;
};
struct HingeAccumulatedImpulses { Vector3Wide BallSocket; Vector2Wide Hinge; };  // Hinge.cs:83
struct HingeFunctions {                                                            // Hinge.cs:89
    typedef HingePrestepData Prestep;
    typedef HingeAccumulatedImpulses Impulses;
    static void ApplyImpulse(const Vector3Wide& offsetA, const Vector3Wide& offsetB, const Matrix2x3Wide& hingeJacobian, const BodyInertiaWide& inertiaA,
                             const BodyInertiaWide& inertiaB, const HingeAccumulatedImpulses& csi, BodyVelocityWide& velocityA, BodyVelocityWide& velocityB) {  // :92
        Vector3Wide linearChangeA;
        Vector3Wide::Scale(csi.BallSocket, inertiaA.InverseMass, linearChangeA);
        Vector3Wide::Add(velocityA.Linear, linearChangeA, velocityA.Linear);
        Vector3Wide ballSocketAngularImpulseA, hingeAngularImpulseA, angularImpulseA, angularChangeA;
        Vector3Wide::CrossWithoutOverlap(offsetA, csi.BallSocket, ballSocketAngularImpulseA);
        Matrix2x3Wide::Transform(csi.Hinge, hingeJacobian, hingeAngularImpulseA);
        Vector3Wide::Add(ballSocketAngularImpulseA, hingeAngularImpulseA, angularImpulseA);
        Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseA, inertiaA.InverseInertiaTensor, angularChangeA);
        Vector3Wide::Add(velocityA.Angular, angularChangeA, velocityA.Angular);
        Vector3Wide negatedLinearChangeB;
        Vector3Wide::Scale(csi.BallSocket, inertiaB.InverseMass, negatedLinearChangeB);
        Vector3Wide::Subtract(velocityB.Linear, negatedLinearChangeB, velocityB.Linear);
        Vector3Wide ballSocketAngularImpulseB, angularImpulseB, angularChangeB;
        Vector3Wide::CrossWithoutOverlap(csi.BallSocket, offsetB, ballSocketAngularImpulseB);
        Vector3Wide::Subtract(ballSocketAngularImpulseB, hingeAngularImpulseA, angularImpulseB);
        Symmetric3x3Wide::TransformWithoutOverlap(angularImpulseB, inertiaB.InverseInertiaTensor, angularChangeB);
        Vector3Wide::Add(velocityB.Angular, angularChangeB, velocityB.Angular);
    }
    static void WarmStart(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                          const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, Prestep& prestep, Impulses& accumulatedImpulses, BodyVelocityWide& wsvA,
                          BodyVelocityWide& wsvB) {  // :116
        Matrix3x3Wide orientationMatrixA;
        Matrix3x3Wide::CreateFromQuaternion(orientationA, orientationMatrixA);
        Vector3Wide offsetA, offsetB, localAX, localAY;
        Matrix3x3Wide::TransformWithoutOverlap(prestep.LocalOffsetA, orientationMatrixA, offsetA);
        QuaternionWide::TransformWithoutOverlap(prestep.LocalOffsetB, orientationB, offsetB);
        Helpers::BuildOrthonormalBasis(prestep.LocalHingeAxisA, localAX, localAY);
        Matrix2x3Wide hingeJacobian;
        Matrix3x3Wide::TransformWithoutOverlap(localAX, orientationMatrixA, hingeJacobian.X);
        Matrix3x3Wide::TransformWithoutOverlap(localAY, orientationMatrixA, hingeJacobian.Y);
        ApplyImpulse(offsetA, offsetB, hingeJacobian, inertiaA, inertiaB, accumulatedImpulses, wsvA, wsvB);
    }
    static void Solve(const Vector3Wide& positionA, const QuaternionWide& orientationA, const BodyInertiaWide& inertiaA, const Vector3Wide& positionB,
                      const QuaternionWide& orientationB, const BodyInertiaWide& inertiaB, float dt, float inverseDt, Prestep& prestep, Impulses& accumulatedImpulses,
                      BodyVelocityWide& wsvA, BodyVelocityWide& wsvB) {  // :128
        Matrix3x3Wide orientationMatrixA, orientationMatrixB;
        Matrix3x3Wide::CreateFromQuaternion(orientationA, orientationMatrixA);
        Matrix3x3Wide::CreateFromQuaternion(orientationB, orientationMatrixB);
        Vector3Wide offsetA, hingeAxisA, offsetB, hingeAxisB, localAX, localAY;
        Matrix3x3Wide::TransformWithoutOverlap(prestep.LocalOffsetA, orientationMatrixA, offsetA);
        Matrix3x3Wide::TransformWithoutOverlap(prestep.LocalHingeAxisA, orientationMatrixA, hingeAxisA);
        Matrix3x3Wide::TransformWithoutOverlap(prestep.LocalOffsetB, orientationMatrixB, offsetB);
        Matrix3x3Wide::TransformWithoutOverlap(prestep.LocalHingeAxisB, orientationMatrixB, hingeAxisB);
        Helpers::BuildOrthonormalBasis(prestep.LocalHingeAxisA, localAX, localAY);
        Matrix2x3Wide hingeJacobian;
        Matrix3x3Wide::TransformWithoutOverlap(localAX, orientationMatrixA, hingeJacobian.X);
        Matrix3x3Wide::TransformWithoutOverlap(localAY, orientationMatrixA, hingeJacobian.Y);
        Symmetric3x3Wide ballSocketContributionAngularA, ballSocketContributionAngularB;
        Symmetric3x3Wide::SkewSandwichWithoutOverlap(offsetA, inertiaA.InverseInertiaTensor, ballSocketContributionAngularA);
        Symmetric3x3Wide::SkewSandwichWithoutOverlap(offsetB, inertiaB.InverseInertiaTensor, ballSocketContributionAngularB);
        Symmetric5x5Wide inverseEffectiveMass;
        Symmetric3x3Wide::Add(ballSocketContributionAngularA, ballSocketContributionAngularB, inverseEffectiveMass.A);
        VF linearContribution = inertiaA.InverseMass + inertiaB.InverseMass;
        inverseEffectiveMass.A.XX += linearContribution;
        inverseEffectiveMass.A.YY += linearContribution;
        inverseEffectiveMass.A.ZZ += linearContribution;
        Matrix2x3Wide hingeInertiaA, hingeInertiaB;
        Symmetric3x3Wide::MultiplyWithoutOverlap(hingeJacobian, inertiaA.InverseInertiaTensor, hingeInertiaA);
        Symmetric3x3Wide::MultiplyWithoutOverlap(hingeJacobian, inertiaB.InverseInertiaTensor, hingeInertiaB);
        Symmetric2x2Wide hingeContributionAngularA, hingeContributionAngularB;
        Symmetric2x2Wide::CompleteMatrixSandwich(hingeInertiaA, hingeJacobian, hingeContributionAngularA);
        Symmetric2x2Wide::CompleteMatrixSandwich(hingeInertiaB, hingeJacobian, hingeContributionAngularB);
        Symmetric2x2Wide::Add(hingeContributionAngularA, hingeContributionAngularB, inverseEffectiveMass.D);
        Vector3Wide offDiagonalContributionAX, offDiagonalContributionAY, offDiagonalContributionBX, offDiagonalContributionBY;
        Vector3Wide::CrossWithoutOverlap(hingeInertiaA.X, offsetA, offDiagonalContributionAX);
        Vector3Wide::CrossWithoutOverlap(hingeInertiaA.Y, offsetA, offDiagonalContributionAY);
        Vector3Wide::CrossWithoutOverlap(hingeInertiaB.X, offsetB, offDiagonalContributionBX);
        Vector3Wide::CrossWithoutOverlap(hingeInertiaB.Y, offsetB, offDiagonalContributionBY);
        Vector3Wide::Add(offDiagonalContributionAX, offDiagonalContributionBX, inverseEffectiveMass.B.X);
        Vector3Wide::Add(offDiagonalContributionAY, offDiagonalContributionBY, inverseEffectiveMass.B.Y);
        Symmetric5x5Wide effectiveMass;
        Symmetric5x5Wide::InvertWithoutOverlap(inverseEffectiveMass, effectiveMass);
        VF positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale;
        SpringSettingsWide::ComputeSpringiness(prestep.SpringSettings, dt, positionErrorToVelocity, effectiveMassCFMScale, softnessImpulseScale);
        Vector3Wide anchorB, ballSocketError, ballSocketBiasVelocity;
        Vector3Wide::Add(positionB - positionA, offsetB, anchorB);
        Vector3Wide::Subtract(anchorB, offsetA, ballSocketError);
        Vector3Wide::Scale(ballSocketError, positionErrorToVelocity, ballSocketBiasVelocity);
        Vector2Wide errorAngles;
        AngularHingeFunctions::GetErrorAngles(hingeAxisA, hingeAxisB, hingeJacobian, errorAngles);
        Vector2Wide hingeBiasVelocity;
        Vector2Wide::Scale(errorAngles, neg(positionErrorToVelocity), hingeBiasVelocity);
        Vector3Wide ballSocketAngularCSVA, ballSocketAngularCSVB;
        Vector2Wide hingeCSVA, negatedHingeCSVB;
        Vector3Wide::CrossWithoutOverlap(wsvA.Angular, offsetA, ballSocketAngularCSVA);
        Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvA.Angular, hingeJacobian, hingeCSVA);
        Vector3Wide::CrossWithoutOverlap(offsetB, wsvB.Angular, ballSocketAngularCSVB);
        Matrix2x3Wide::TransformByTransposeWithoutOverlap(wsvB.Angular, hingeJacobian, negatedHingeCSVB);
        Vector3Wide ballSocketAngularCSV, ballSocketLinearCSV, ballSocketCSV;
        Vector3Wide::Add(ballSocketAngularCSVA, ballSocketAngularCSVB, ballSocketAngularCSV);
        Vector3Wide::Subtract(wsvA.Linear, wsvB.Linear, ballSocketLinearCSV);
        Vector3Wide::Add(ballSocketAngularCSV, ballSocketLinearCSV, ballSocketCSV);
        Vector3Wide::Subtract(ballSocketBiasVelocity, ballSocketCSV, ballSocketCSV);
        Vector2Wide hingeCSV;
        Vector2Wide::Subtract(hingeCSVA, negatedHingeCSVB, hingeCSV);
        Vector2Wide::Subtract(hingeBiasVelocity, hingeCSV, hingeCSV);
        HingeAccumulatedImpulses csi;
        Symmetric5x5Wide::TransformWithoutOverlap(ballSocketCSV, hingeCSV, effectiveMass, csi.BallSocket, csi.Hinge);
        csi.BallSocket = csi.BallSocket * effectiveMassCFMScale;
        csi.Hinge = csi.Hinge * effectiveMassCFMScale;
        Vector3Wide ballSocketSoftnessContribution;
        Vector3Wide::Scale(accumulatedImpulses.BallSocket, softnessImpulseScale, ballSocketSoftnessContribution);
        Vector3Wide::Subtract(csi.BallSocket, ballSocketSoftnessContribution, csi.BallSocket);
        Vector2Wide hingeSoftnessContribution;
        Vector2Wide::Scale(accumulatedImpulses.Hinge, softnessImpulseScale, hingeSoftnessContribution);
        Vector2Wide::Subtract(csi.Hinge, hingeSoftnessContribution, csi.Hinge);
        accumulatedImpulses.BallSocket = accumulatedImpulses.BallSocket + csi.BallSocket;
        accumulatedImpulses.Hinge = accumulatedImpulses.Hinge + csi.Hinge;
        ApplyImpulse(offsetA, offsetB, hingeJacobian, inertiaA, inertiaB, csi, wsvA, wsvB);
    }
};

}  // namespace wide
