// Device-side constraint functions for the gfx950 solver kernels: one GPU lane evaluates one constraint
// (one lane of the reference's wide WarmStart / Solve / IncrementallyUpdateForSubstep). `p` and `a` are the lane's
// prestep data and accumulated impulses held in registers (loaded from the SoA slabs up front so every load is in
// flight together); operation order follows the reference file:line cited at each function.
// Access masks mirror the reference's IBodyAccessFilter choices (BepuPhysics/Constraints/IBodyAccessFilter.cs) so
// that only the body fields a type needs are gathered/scattered.
#pragma once

#include "bepu_device_math.h"

namespace bd {

// Type ids: BepuPhysics/Constraints/Contact/ContactConvexTypes.cs:338..1523 (one-body N -> N-1, two-body N -> 3+N),
// BepuPhysics/Constraints/*.cs BatchTypeId constants.
// Body field access bits (IBodyAccessFilter equivalents).
enum Access { kPos = 1, kOri = 2, kLin = 4, kAng = 8, kInertia = 16,
              kAccessAll = 31, kAccessNoOrientation = 29, kAccessNoPosition = 30, kAccessNoPose = 28, kAccessOnlyAngular = 26, kAccessOnlyAngularWithoutPose = 24, kAccessOnlyVelocity = 12,
              kAccessOnlyLinear = 21 /* position, inverse mass (gathered with the tensor), linear velocity */ };
enum TypeId {
    kContact1OneBody = 0, kContact2OneBody = 1, kContact3OneBody = 2, kContact4OneBody = 3,
    kContact1 = 4, kContact2 = 5, kContact3 = 6, kContact4 = 7,
    kBallSocket = 22, kAngularHinge = 23, kSwingLimit = 25, kTwistServo = 26, kTwistLimit = 27,
    kAngularMotor = 30, kWeld = 31, kSwivelHinge = 46, kHinge = 47,
    kAngularSwivelHinge = 24, kTwistMotor = 28, kAngularServo = 29, kDistanceServo = 33, kDistanceLimit = 34, kAngularAxisMotor = 41,
    kOneBodyAngularServo = 42, kOneBodyAngularMotor = 43, kOneBodyLinearServo = 44, kOneBodyLinearMotor = 45, kBallSocketMotor = 52, kBallSocketServo = 53,
    kPointOnLineServo = 37, kLinearAxisServo = 38, kLinearAxisMotor = 39, kLinearAxisLimit = 40, kAngularAxisGearMotor = 54,
    kVolumeConstraint = 32, kCenterDistanceConstraint = 35, kAreaConstraint = 36, kCenterDistanceLimit = 55,
    kContact2NonconvexOneBody = 8, kContact3NonconvexOneBody = 9, kContact4NonconvexOneBody = 10, kContact2Nonconvex = 15, kContact3Nonconvex = 16, kContact4Nonconvex = 17,
};
// Contact manifolds are the types with RequiresIncrementalSubstepUpdates (their depths advance every substep).
BD_FN bool isContactType(int id) { return id <= kContact4Nonconvex; }

// ======================================================================================
// Convex contact manifolds, N = 1..4 contacts, one or two bodies.
// Prestep layout (ContactConvexTypes.cs:1418-1430 / .tt:174-187): N x {OffsetA xyz, Depth}, [OffsetB xyz], Normal xyz,
// {FrictionCoefficient, AngularFrequency, TwiceDampingRatio, MaximumRecoveryVelocity}.
// Accumulated impulses (:11-99): Tangent xy, Penetration0..N-1, Twist.
// ======================================================================================
#ifndef BEPU_LATE_IMPULSES
#define BEPU_LATE_IMPULSES 0  // 1 (the 128-VGPR cluster units, bepu_cluster_variant.inc): see Contact::lateImpulses
#endif
// Gates that can fetch an accumulated impulse when the tail asks for it say so (ClusterGate in the 128-VGPR units); every other gate:
template <class G, class = void> struct GateFetchesImpulses { static constexpr bool value = false; };
template <class G> struct GateFetchesImpulses<G, std::void_t<decltype(G::kLateImpulses)>> { static constexpr bool value = G::kLateImpulses; };
template <int N, bool TwoBody>
struct Contact {
    // A two-body manifold with two to four contacts does not fit a 128-register wave: the compiler spills across the gate and reloads in the velocity-dependent tail —
    // in the committed round-4 build the twist impulse and the second body's local reference, each a memory round trip on the cluster's dependency chain
    // (profiles/r05_s16_headline_trace*.txt: the Contact4 item that ends every pass of the bench scene spends 7.5 k clocks in its tail with 128 registers, 2.3 k with
    // 256). The tangent and twist impulses are first used behind all penetration rows: such a type's Solve leaves them in memory until the gate, which asks for them
    // under the wait for the predecessors (address from wave-uniform values and the lane id: nothing of it is live in front of the gate).
    static constexpr bool lateImpulses = BEPU_LATE_IMPULSES != 0 && TwoBody && N >= 2;
    static constexpr int bodies = TwoBody ? 2 : 1;
    static constexpr int prestepFloats = 4 * N + (TwoBody ? 10 : 7);
    static constexpr int impulseFloats = N + 3;
    static constexpr int typeId = TwoBody ? 3 + N : N - 1;
    static constexpr bool incremental = true;
    static constexpr int contacts = N;
    static constexpr int depthRow(int c) { return 4 * c + 3; }  // the rows IncrementallyUpdateForSubstep rewrites
    // TwoBodyContactTypeProcessor / OneBodyContactTypeProcessor: AccessNoPose everywhere (TwoBodyTypeProcessor.cs:244-250).
    static constexpr int wsA = kAccessNoPose, wsB = kAccessNoPose, svA = kAccessNoPose, svB = kAccessNoPose;

    BD_FN V3 offsetA(const float* p, int i) { return {p[4 * i], p[4 * i + 1], p[4 * i + 2]}; }
    BD_FN float& depth(float* p, int i) { return p[4 * i + 3]; }
    BD_FN V3 offsetB(const float* p) { return {p[4 * N], p[4 * N + 1], p[4 * N + 2]}; }
    static constexpr int nOff = 4 * N + (TwoBody ? 3 : 0);
    BD_FN V3 normal(const float* p) { return {p[nOff], p[nOff + 1], p[nOff + 2]}; }
    BD_FN float friction(const float* p) { return p[nOff + 3]; }
    BD_FN float springFreq(const float* p) { return p[nOff + 4]; }
    BD_FN float springDamp(const float* p) { return p[nOff + 5]; }
    BD_FN float maxRecovery(const float* p) { return p[nOff + 6]; }

    // FrictionHelpers.ComputeFrictionCenter, ContactConvexTypes.cs:121-196.
    BD_FN V3 frictionCenter(float* p) {
        float w[N];
        float weightSum = 0;
        _Pragma("unroll") for (int i = 0; i < N; ++i) w[i] = sel(depth(p, i) < 0.0f, 0.0f, 1.0f);
        if (N == 2) weightSum = w[0] + w[1];
        if (N == 3) weightSum = w[0] + w[1] + w[2];
        if (N == 4) weightSum = w[0] + w[1] + w[2] + w[3];
        bool useFallback = weightSum == 0.0f;
        weightSum = sel(useFallback, (float)N, weightSum);
        float inverseWeightSum = 1.0f / weightSum;
        V3 c[N];
        _Pragma("unroll") for (int i = 0; i < N; ++i) {
            w[i] = sel(useFallback, inverseWeightSum, w[i] * inverseWeightSum);
            c[i] = scale(offsetA(p, i), w[i]);
        }
        if (N == 2) return add(c[0], c[1]);
        if (N == 3) return add(add(c[0], c[1]), c[2]);
        return add(add(c[0], c[1]), add(c[2], c[N - 1]));  // N==4: (a0+a1)+(a2+a3)
    }

    // ---- PenetrationLimit.cs / PenetrationLimitOneBody.cs ----
    BD_FN void penApply(const Inertia& iA, const Inertia& iB, V3 n, V3 angularA, V3 angularB, float csi, BodyVel& vA, BodyVel& vB) {
        // PenetrationLimit.cs:46-66 (two-body), PenetrationLimitOneBody.cs ApplyImpulse (one-body)
        float linearVelocityChangeA = csi * iA.invMass;
        V3 corrALin = scale(n, linearVelocityChangeA);
        V3 corrAngImpA = scale(angularA, csi);
        V3 corrAAng = transform(corrAngImpA, iA.t);
        if (TwoBody) {
            float linearVelocityChangeB = csi * iB.invMass;
            V3 corrBLin = scale(n, linearVelocityChangeB);
            V3 corrAngImpB = scale(angularB, csi);
            V3 corrBAng = transform(corrAngImpB, iB.t);
            vA.lin = add(vA.lin, corrALin);
            vA.ang = add(vA.ang, corrAAng);
            vB.lin = sub(vB.lin, corrBLin);
            vB.ang = add(vB.ang, corrBAng);
        } else {
            vA.lin = add(vA.lin, corrALin);
            vA.ang = add(vA.ang, corrAAng);
        }
    }
    // The velocity-independent half of every row (jacobians, effective mass, bias) is separated from the velocity-dependent half
    // (corrective impulse, application) so that a caller can evaluate the former before the bodies' velocities are available.
    // Every value is produced by the same operations in the same order as the reference's fused form.
    struct PenRow { V3 angularA, angularB; float effectiveMass, biasVelocity; };
    BD_FN PenRow penSetup(const Inertia& iA, const Inertia& iB, V3 n, V3 offA, V3 offB, float dep, float posErrToVel, float effMassCFMScale,
                          float maxRecoveryVelocity, float inverseDt) {
        // PenetrationLimit.cs:79-131 up to the bias velocity; PenetrationLimitOneBody.cs Solve
        PenRow r;
        r.angularA = cross(offA, n);
        r.angularB = TwoBody ? cross(n, offB) : V3{0, 0, 0};
        float angularA0 = vectorSandwich(r.angularA, iA.t);
        if (TwoBody) {
            float angularB0 = vectorSandwich(r.angularB, iB.t);
            float linear = iA.invMass + iB.invMass;
            r.effectiveMass = effMassCFMScale / (linear + angularA0 + angularB0);
        } else {
            r.effectiveMass = effMassCFMScale / (iA.invMass + angularA0);
        }
        r.biasVelocity = vmin(dep * inverseDt, vmin(dep * posErrToVel, maxRecoveryVelocity));
        return r;
    }
    BD_FN void penIterate(const PenRow& r, const Inertia& iA, const Inertia& iB, V3 n, float softnessImpulseScale, float& acc, BodyVel& vA, BodyVel& vB) {
        // ComputeCorrectiveImpulse, PenetrationLimit.cs:9-26
        float csvaLinear = dot(vA.lin, n);
        float csvaAngular = dot(vA.ang, r.angularA);
        float negatedCSI;
        if (TwoBody) {
            float negatedCSVBLinear = dot(vB.lin, n);
            float csvbAngular = dot(vB.ang, r.angularB);
            negatedCSI = acc * softnessImpulseScale + (csvaLinear - negatedCSVBLinear + csvaAngular + csvbAngular - r.biasVelocity) * r.effectiveMass;
        } else {
            negatedCSI = acc * softnessImpulseScale + (csvaLinear + csvaAngular - r.biasVelocity) * r.effectiveMass;
        }
        float previousAccumulated = acc;
        acc = vmax(0.0f, acc - negatedCSI);
        float correctiveCSI = acc - previousAccumulated;
        penApply(iA, iB, n, r.angularA, r.angularB, correctiveCSI, vA, vB);
    }

    // ---- TangentFriction.cs / TangentFrictionOneBody.cs ----
    struct TJ { M23 linearA, angularA, angularB; };
    BD_FN TJ tangentJacobians(V3 tX, V3 tY, V3 offA, V3 offB) {  // TangentFriction.cs:16-50
        TJ j;
        j.linearA.X = tX; j.linearA.Y = tY;
        j.angularA.X = cross(offA, tX); j.angularA.Y = cross(offA, tY);
        if (TwoBody) { j.angularB.X = cross(tX, offB); j.angularB.Y = cross(tY, offB); }
        else { j.angularB.X = {0, 0, 0}; j.angularB.Y = {0, 0, 0}; }
        return j;
    }
    BD_FN void tangentApply(const TJ& j, const Inertia& iA, const Inertia& iB, V2 csi, BodyVel& vA, BodyVel& vB) {  // :53-69
        V3 linearImpulseA = transform(csi, j.linearA);
        V3 angularImpulseA = transform(csi, j.angularA);
        V3 corrALin = scale(linearImpulseA, iA.invMass);
        V3 corrAAng = transform(angularImpulseA, iA.t);
        if (TwoBody) {
            V3 angularImpulseB = transform(csi, j.angularB);
            V3 corrBLin = scale(linearImpulseA, iB.invMass);
            V3 corrBAng = transform(angularImpulseB, iB.t);
            vA.lin = add(vA.lin, corrALin);
            vA.ang = add(vA.ang, corrAAng);
            vB.lin = sub(vB.lin, corrBLin);
            vB.ang = add(vB.ang, corrBAng);
        } else {
            vA.lin = add(vA.lin, corrALin);
            vA.ang = add(vA.ang, corrAAng);
        }
    }
    struct TangentSetup { TJ j; Sym2 effectiveMass; };
    BD_FN TangentSetup tangentSetup(V3 tX, V3 tY, V3 offA, V3 offB, const Inertia& iA, const Inertia& iB) {
        // TangentFriction.cs:117-137 up to the effective mass; TangentFrictionOneBody.cs Solve
        TangentSetup t;
        t.j = tangentJacobians(tX, tY, offA, offB);
        Sym2 inverseEffectiveMass;
        if (TwoBody) {
            Sym2 linearContributionA = sandwichScale(t.j.linearA, iA.invMass);
            Sym2 linearContributionB = sandwichScale(t.j.linearA, iB.invMass);
            Sym2 angularContributionA = matrixSandwich(t.j.angularA, iA.t);
            Sym2 angularContributionB = matrixSandwich(t.j.angularB, iB.t);
            Sym2 linear = add(linearContributionA, linearContributionB);
            Sym2 angular = add(angularContributionA, angularContributionB);
            inverseEffectiveMass = add(linear, angular);
        } else {
            Sym2 linearContributionA = sandwichScale(t.j.linearA, iA.invMass);
            Sym2 angularContributionA = matrixSandwich(t.j.angularA, iA.t);
            inverseEffectiveMass = add(linearContributionA, angularContributionA);
        }
        t.effectiveMass = invert(inverseEffectiveMass);
        return t;
    }
    BD_FN void tangentIterate(const TangentSetup& t, const Inertia& iA, const Inertia& iB, float maximumImpulse, V2& acc, BodyVel& vA, BodyVel& vB) {
        // ComputeCorrectiveImpulse, TangentFriction.cs:72-101 / OneBody variant
        const TJ& j = t.j;
        V2 previousAccumulated = acc;
        if (TwoBody) {
            V2 csvaLinear = transformByTranspose(vA.lin, j.linearA);
            V2 csvaAngular = transformByTranspose(vA.ang, j.angularA);
            V2 csvbLinear = transformByTranspose(vB.lin, j.linearA);
            V2 csvbAngular = transformByTranspose(vB.ang, j.angularB);
            V2 csvLinear = sub(csvbLinear, csvaLinear);
            V2 csvAngular = add(csvaAngular, csvbAngular);
            V2 csv = sub(csvLinear, csvAngular);
            V2 csi = transform(csv, t.effectiveMass);
            acc = add(acc, csi);
        } else {
            V2 csvaLinear = transformByTranspose(vA.lin, j.linearA);
            V2 csvaAngular = transformByTranspose(vA.ang, j.angularA);
            V2 csv = add(csvaLinear, csvaAngular);
            V2 negativeCSI = transform(csv, t.effectiveMass);
            acc = sub(acc, negativeCSI);
        }
        float accumulatedMagnitude = length(acc);
        float sc = vmin(1.0f, maximumImpulse / vmax(1e-16f, accumulatedMagnitude));
        acc = scale(acc, sc);
        V2 correctiveCSI = sub(acc, previousAccumulated);
        tangentApply(j, iA, iB, correctiveCSI, vA, vB);
    }

    // ---- TwistFriction.cs / TwistFrictionOneBody.cs ----
    BD_FN void twistApply(V3 angularJacobianA, const Inertia& iA, const Inertia& iB, float csi, BodyVel& vA, BodyVel& vB) {  // :10-19
        V3 worldCorrectiveImpulseA = scale(angularJacobianA, csi);
        V3 worldCorrectiveVelocityA = transform(worldCorrectiveImpulseA, iA.t);
        if (TwoBody) {
            V3 worldCorrectiveVelocityB = transform(worldCorrectiveImpulseA, iB.t);
            vA.ang = add(vA.ang, worldCorrectiveVelocityA);
            vB.ang = sub(vB.ang, worldCorrectiveVelocityB);
        } else {
            vA.ang = add(vA.ang, worldCorrectiveVelocityA);
        }
    }
    BD_FN float twistEffectiveMass(V3 angularJacobianA, const Inertia& iA, const Inertia& iB) {  // :46-71 up to the effective mass
        float angularA = vectorSandwich(angularJacobianA, iA.t);
        float inverseEffectiveMass = angularA;
        if (TwoBody) {
            float angularB = vectorSandwich(angularJacobianA, iB.t);
            inverseEffectiveMass = angularA + angularB;
        }
        bool inverseIsZero = 0.0f == inverseEffectiveMass;
        return sel(inverseIsZero, 0.0f, 1.0f / inverseEffectiveMass);
    }
    BD_FN void twistIterate(V3 angularJacobianA, float effectiveMass, const Inertia& iA, const Inertia& iB, float maximumImpulse, float& acc, BodyVel& vA, BodyVel& vB) {
        // ComputeCorrectiveImpulse :22-36
        float csvA = dot(vA.ang, angularJacobianA);
        float negatedCSI;
        if (TwoBody) {
            float negatedCSVB = dot(vB.ang, angularJacobianA);
            negatedCSI = (csvA - negatedCSVB) * effectiveMass;
        } else {
            negatedCSI = csvA * effectiveMass;
        }
        float previousAccumulated = acc;
        acc = vmin(maximumImpulse, vmax(-maximumImpulse, acc - negatedCSI));
        float correctiveCSI = acc - previousAccumulated;
        twistApply(angularJacobianA, iA, iB, correctiveCSI, vA, vB);
    }

    // ---- ContactNFunctions (ContactConvexTypes.cs:1461-1514, template .tt:220-289) ----
    BD_FN void incrementalUpdate(float dt, const BodyVel& vA, const BodyVel& vB, float* p) {
        // PenetrationLimit.UpdatePenetrationDepth, PenetrationLimit.cs:29-43; OneBody: PenetrationLimitOneBody.cs
        V3 n = normal(p);
        _Pragma("unroll") for (int i = 0; i < N; ++i) {
            V3 contactOffsetA = offsetA(p, i);
            V3 wxra = cross(vA.ang, contactOffsetA);
            V3 contactVelocityA = add(wxra, vA.lin);
            float estimatedDepthChangeVelocity;
            if (TwoBody) {
                V3 contactOffsetB = sub(contactOffsetA, offsetB(p));
                V3 wxrb = cross(vB.ang, contactOffsetB);
                V3 contactVelocityB = add(wxrb, vB.lin);
                V3 contactVelocityDifference = sub(contactVelocityA, contactVelocityB);
                estimatedDepthChangeVelocity = dot(n, contactVelocityDifference);
            } else {
                estimatedDepthChangeVelocity = dot(n, contactVelocityA);
            }
            depth(p, i) = depth(p, i) - estimatedDepthChangeVelocity * dt;
        }
    }
    template <class G> BD_FN void warmStart(V3, Q, const Inertia& iA, V3, Q, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {
        V3 n = normal(p);
        V3 x, z;
        buildOrthonormalBasis(n, x, z);
        V3 centerA = (N > 1) ? frictionCenter(p) : offsetA(p, 0);
        V3 centerB = TwoBody ? sub(centerA, offsetB(p)) : V3{0, 0, 0};
        TJ j = tangentJacobians(x, z, centerA, centerB);
        V3 angularA[N], angularB[N];  // PenetrationLimit.cs:69-76 jacobians
        _Pragma("unroll") for (int i = 0; i < N; ++i) {
            V3 oA = offsetA(p, i);
            V3 oB = TwoBody ? sub(oA, offsetB(p)) : V3{0, 0, 0};
            angularA[i] = cross(oA, n);
            angularB[i] = TwoBody ? cross(n, oB) : V3{0, 0, 0};
        }
        BD_GATE(vA, vB, j, angularA, angularB);
        tangentApply(j, iA, iB, V2{a[0], a[1]}, vA, vB);
        _Pragma("unroll") for (int i = 0; i < N; ++i) penApply(iA, iB, n, angularA[i], angularB[i], a[2 + i], vA, vB);
        twistApply(n, iA, iB, a[2 + N], vA, vB);
    }
    template <class G> BD_FN void solve(V3, Q, const Inertia& iA, V3, Q, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(springFreq(p), springDamp(p), dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 n = normal(p);
        PenRow rows[N];
        _Pragma("unroll") for (int i = 0; i < N; ++i) {
            V3 oA = offsetA(p, i);
            V3 oB = TwoBody ? sub(oA, offsetB(p)) : V3{0, 0, 0};
            rows[i] = penSetup(iA, iB, n, oA, oB, depth(p, i), posErrToVel, effMassCFMScale, maxRecovery(p), inverseDt);
        }
        V3 x, z;
        buildOrthonormalBasis(n, x, z);
        float premultipliedFrictionCoefficient = (N > 1) ? (1.0f / (float)N) * friction(p) : friction(p);
        V3 centerA = (N > 1) ? frictionCenter(p) : offsetA(p, 0);
        V3 centerB = TwoBody ? sub(centerA, offsetB(p)) : V3{0, 0, 0};
        TangentSetup tangentSetupData = tangentSetup(x, z, centerA, centerB, iA, iB);
        float twistMass = twistEffectiveMass(n, iA, iB);
        float leverArm[N];
        if (N > 1) { _Pragma("unroll") for (int i = 0; i < N; ++i) leverArm[i] = distance(centerA, offsetA(p, i)); }
        BD_GATE(vA, vB, rows, tangentSetupData, twistMass, leverArm, premultipliedFrictionCoefficient, softnessImpulseScale);
        _Pragma("unroll") for (int i = 0; i < N; ++i) penIterate(rows[i], iA, iB, n, softnessImpulseScale, a[2 + i], vA, vB);
        float penSum = a[2];
        _Pragma("unroll") for (int i = 1; i < N; ++i) penSum = penSum + a[2 + i];
        float maximumTangentImpulse = premultipliedFrictionCoefficient * penSum;
        constexpr bool fetched = lateImpulses && GateFetchesImpulses<std::remove_reference_t<G>>::value;  // then a[0], a[1], a[2 + N] arrive through the gate
        V2 tangent{a[0], a[1]};
        if constexpr (fetched) tangent = V2{gate.lateImpulse(0), gate.lateImpulse(1)};
        tangentIterate(tangentSetupData, iA, iB, maximumTangentImpulse, tangent, vA, vB);
        a[0] = tangent.x; a[1] = tangent.y;
        float maximumTwistImpulse;
        if (N == 1) {
            maximumTwistImpulse = friction(p) * a[2] * vmax(0.0f, depth(p, 0));
        } else {
            float s = a[2] * leverArm[0];
            _Pragma("unroll") for (int i = 1; i < N; ++i) s = s + a[2 + i] * leverArm[i];
            maximumTwistImpulse = premultipliedFrictionCoefficient * s;
        }
        if constexpr (fetched) a[2 + N] = gate.lateImpulse(2);
        twistIterate(n, twistMass, iA, iB, maximumTwistImpulse, a[2 + N], vA, vB);
    }
};

// ======================================================================================
// One manifold function for a PER-LANE contact count (round 5, the merged manifold work item of the island schedule: lanes of Contact1..4 type batches of one
// batch share a wave). The reference generates Contact1..4 from one template (ContactConvexTypes.tt; ContactConvexTypes.cs:901-1514 are its four expansions):
// the row functions are the same, what depends on N is how many penetration rows there are, how the friction centre is summed and two constants. `p` and `a`
// are held in Contact4's layout (contact c at p[4c..4c+3], the common block at 16.., penetration impulses at a[2..5], twist at a[6]); `count` is the lane's N.
// Every lane evaluates, operation for operation, what Contact<count, TwoBody> evaluates: rows beyond the lane's count are skipped under the lane mask, sums
// that the template writes per N are selected per lane, never re-associated. (tests/test_contact_fused_host.py checks the bits on the host for all four counts.)
// ======================================================================================
template <bool TwoBody>
struct ContactFused {
    using C = Contact<4, TwoBody>;
    static constexpr int bodies = TwoBody ? 2 : 1;
    static constexpr int prestepFloats = C::prestepFloats;
    static constexpr int impulseFloats = C::impulseFloats;
    static constexpr int commonFloats = TwoBody ? 10 : 7;  // [OffsetB], Normal, material: the rows behind the contacts
    static constexpr int wsA = kAccessNoPose, wsB = kAccessNoPose, svA = kAccessNoPose, svB = kAccessNoPose;
    using PenRow = typename C::PenRow;
    using TJ = typename C::TJ;
    using TangentSetup = typename C::TangentSetup;

    // FrictionHelpers.ComputeFrictionCenter for count = 2, 3, 4 (Contact<N>::frictionCenter above, per lane).
    BD_FN V3 frictionCenter(float* p, int count) {
        float w[4];
        _Pragma("unroll") for (int i = 0; i < 4; ++i) w[i] = (i < count) ? sel(C::depth(p, i) < 0.0f, 0.0f, 1.0f) : 0.0f;
        float weightSum = w[0] + w[1] + w[2] + w[3];  // left to right: the weights are 0 or 1, adding the absent ones' +0 changes no bit
        bool useFallback = weightSum == 0.0f;
        weightSum = sel(useFallback, (float)count, weightSum);
        float inverseWeightSum = 1.0f / weightSum;
        V3 c[4];
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {
            w[i] = sel(useFallback, inverseWeightSum, w[i] * inverseWeightSum);
            c[i] = scale(C::offsetA(p, i), w[i]);
        }
        const V3 s01 = add(c[0], c[1]);
        const V3 s23 = add(c[2], c[3]);
        const V3 tail = sel3(count == 4, s23, c[2]);  // N == 4: (c0 + c1) + (c2 + c3); N == 3: (c0 + c1) + c2
        const V3 three = add(s01, tail);
        return sel3(count == 2, s01, three);
    }
    BD_FN V3 centerOf(float* p, int count) {
        const V3 many = frictionCenter(p, count);
        return sel3(count > 1, many, C::offsetA(p, 0));
    }
    BD_FN void incrementalUpdate(float dt, const BodyVel& vA, const BodyVel& vB, float* p, int count) {
        V3 n = C::normal(p);
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {
            if (i < count) {
                V3 contactOffsetA = C::offsetA(p, i);
                V3 wxra = cross(vA.ang, contactOffsetA);
                V3 contactVelocityA = add(wxra, vA.lin);
                float estimatedDepthChangeVelocity;
                if (TwoBody) {
                    V3 contactOffsetB = sub(contactOffsetA, C::offsetB(p));
                    V3 wxrb = cross(vB.ang, contactOffsetB);
                    V3 contactVelocityB = add(wxrb, vB.lin);
                    V3 contactVelocityDifference = sub(contactVelocityA, contactVelocityB);
                    estimatedDepthChangeVelocity = dot(n, contactVelocityDifference);
                } else {
                    estimatedDepthChangeVelocity = dot(n, contactVelocityA);
                }
                C::depth(p, i) = C::depth(p, i) - estimatedDepthChangeVelocity * dt;
            }
        }
    }
    template <class G> BD_FN void warmStart(const Inertia& iA, const Inertia& iB, float* p, float* a, int count, BodyVel& vA, BodyVel& vB, G&& gate) {
        V3 n = C::normal(p);
        V3 x, z;
        buildOrthonormalBasis(n, x, z);
        V3 centerA = centerOf(p, count);
        V3 centerB = TwoBody ? sub(centerA, C::offsetB(p)) : V3{0, 0, 0};
        TJ j = C::tangentJacobians(x, z, centerA, centerB);
        V3 angularA[4], angularB[4];
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {
            V3 oA = C::offsetA(p, i);
            V3 oB = TwoBody ? sub(oA, C::offsetB(p)) : V3{0, 0, 0};
            angularA[i] = cross(oA, n);
            angularB[i] = TwoBody ? cross(n, oB) : V3{0, 0, 0};
        }
        BD_GATE(vA, vB, j, angularA, angularB);
        C::tangentApply(j, iA, iB, V2{a[0], a[1]}, vA, vB);
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { if (i < count) C::penApply(iA, iB, n, angularA[i], angularB[i], a[2 + i], vA, vB); }
        C::twistApply(n, iA, iB, a[6], vA, vB);
    }
    template <class G> BD_FN void solve(const Inertia& iA, const Inertia& iB, float dt, float inverseDt, float* p, float* a, int count, BodyVel& vA, BodyVel& vB, G&& gate) {
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(C::springFreq(p), C::springDamp(p), dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 n = C::normal(p);
        PenRow rows[4];
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {
            rows[i] = PenRow{{0, 0, 0}, {0, 0, 0}, 0.0f, 0.0f};
            if (i < count) {
                V3 oA = C::offsetA(p, i);
                V3 oB = TwoBody ? sub(oA, C::offsetB(p)) : V3{0, 0, 0};
                rows[i] = C::penSetup(iA, iB, n, oA, oB, C::depth(p, i), posErrToVel, effMassCFMScale, C::maxRecovery(p), inverseDt);
            }
        }
        V3 x, z;
        buildOrthonormalBasis(n, x, z);
        // (1.0f / (float)N) * friction with N = 1 is friction itself, bit for bit: one expression serves all four counts
        const float inverseCount = count == 1 ? 1.0f : (count == 2 ? (1.0f / 2.0f) : (count == 3 ? (1.0f / 3.0f) : (1.0f / 4.0f)));
        float premultipliedFrictionCoefficient = inverseCount * C::friction(p);
        V3 centerA = centerOf(p, count);
        V3 centerB = TwoBody ? sub(centerA, C::offsetB(p)) : V3{0, 0, 0};
        TangentSetup tangentSetupData = C::tangentSetup(x, z, centerA, centerB, iA, iB);
        float twistMass = C::twistEffectiveMass(n, iA, iB);
        float leverArm[4];
        _Pragma("unroll") for (int i = 0; i < 4; ++i) leverArm[i] = (i < count) ? distance(centerA, C::offsetA(p, i)) : 0.0f;
        BD_GATE(vA, vB, rows, tangentSetupData, twistMass, leverArm, premultipliedFrictionCoefficient, softnessImpulseScale);
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { if (i < count) C::penIterate(rows[i], iA, iB, n, softnessImpulseScale, a[2 + i], vA, vB); }
        float penSum = a[2];
        _Pragma("unroll") for (int i = 1; i < 4; ++i) penSum = (i < count) ? penSum + a[2 + i] : penSum;
        float maximumTangentImpulse = premultipliedFrictionCoefficient * penSum;
        V2 tangent{a[0], a[1]};
        C::tangentIterate(tangentSetupData, iA, iB, maximumTangentImpulse, tangent, vA, vB);
        a[0] = tangent.x; a[1] = tangent.y;
        const float single = C::friction(p) * a[2] * vmax(0.0f, C::depth(p, 0));  // N == 1
        float s = a[2] * leverArm[0];
        _Pragma("unroll") for (int i = 1; i < 4; ++i) s = (i < count) ? s + a[2 + i] * leverArm[i] : s;
        const float maximumTwistImpulse = count == 1 ? single : premultipliedFrictionCoefficient * s;
        C::twistIterate(n, twistMass, iA, iB, maximumTwistImpulse, a[6], vA, vB);
    }
};

// ======================================================================================
// Nonconvex contact manifolds, N = 2..4 contacts, one or two bodies — ContactNonconvexCommon.cs:177-299.
// Prestep layout (ContactNonconvexTypes.cs:58-66 two-body, :161-167 one-body): {FrictionCoefficient, AngularFrequency, TwiceDampingRatio,
// MaximumRecoveryVelocity}, [OffsetB xyz], N x {Offset xyz, Depth, Normal xyz} (NonconvexContactPrestepData, ContactNonconvexCommon.cs:11-16).
// Accumulated impulses: N x {Tangent xy, Penetration} (NonconvexAccumulatedImpulses :171-175).
// Every contact carries its own normal and friction: the rows are solved contact by contact, penetration then tangent (:274-283), with
// the tangent basis of that contact's normal and maximum friction = FrictionCoefficient * that contact's penetration impulse.
// ======================================================================================
template <int N, bool TwoBody>
struct NonconvexContact {
    using R = Contact<1, TwoBody>;  // the penetration / tangent row functions are shared with the convex manifolds
    static constexpr int bodies = TwoBody ? 2 : 1;
    static constexpr int cOff = TwoBody ? 7 : 4;
    static constexpr int prestepFloats = cOff + 7 * N;
    static constexpr int impulseFloats = 3 * N;
    static constexpr int typeId = TwoBody ? 13 + N : 6 + N;  // ContactNonconvexTypes.cs:109,192,300,385,496,583
    static constexpr bool incremental = true;
    static constexpr int contacts = N;
    static constexpr int depthRow(int c) { return cOff + 7 * c + 3; }
    static constexpr int wsA = kAccessNoPose, wsB = kAccessNoPose, svA = kAccessNoPose, svB = kAccessNoPose;  // contact type processors: TwoBodyTypeProcessor.cs:244-250

    BD_FN float friction(const float* p) { return p[0]; }
    BD_FN float springFreq(const float* p) { return p[1]; }
    BD_FN float springDamp(const float* p) { return p[2]; }
    BD_FN float maxRecovery(const float* p) { return p[3]; }
    BD_FN V3 offsetB(const float* p) { return {p[4], p[5], p[6]}; }
    BD_FN V3 offset(const float* p, int i) { return {p[cOff + 7 * i], p[cOff + 7 * i + 1], p[cOff + 7 * i + 2]}; }
    BD_FN float& depth(float* p, int i) { return p[cOff + 7 * i + 3]; }
    BD_FN V3 normal(const float* p, int i) { return {p[cOff + 7 * i + 4], p[cOff + 7 * i + 5], p[cOff + 7 * i + 6]}; }

    BD_FN void incrementalUpdate(float dt, const BodyVel& vA, const BodyVel& vB, float* p) {  // :238-247, :289-298 -> PenetrationLimit.cs:29-43
        _Pragma("unroll") for (int i = 0; i < N; ++i) {
            V3 n = normal(p, i);
            V3 contactOffsetA = offset(p, i);
            V3 wxra = cross(vA.ang, contactOffsetA);
            V3 contactVelocityA = add(wxra, vA.lin);
            float estimatedDepthChangeVelocity;
            if (TwoBody) {
                V3 contactOffsetB = sub(contactOffsetA, offsetB(p));
                V3 wxrb = cross(vB.ang, contactOffsetB);
                V3 contactVelocityB = add(wxrb, vB.lin);
                V3 contactVelocityDifference = sub(contactVelocityA, contactVelocityB);
                estimatedDepthChangeVelocity = dot(n, contactVelocityDifference);
            } else {
                estimatedDepthChangeVelocity = dot(n, contactVelocityA);
            }
            depth(p, i) = depth(p, i) - estimatedDepthChangeVelocity * dt;
        }
    }
    // A waiting gate (island schedule) wants every contact's velocity-independent rows evaluated ahead of it; the launch-per-batch kernels have
    // nothing to wait for and keep one contact's rows live at a time instead (same values either way: the rows do not depend on velocities).
    template <class G> BD_FN void warmStart(V3, Q, const Inertia& iA, V3, Q, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :192-206, :254-270
        if constexpr (std::remove_reference_t<G>::kPin) {
            typename R::TJ j[N];
            V3 angularA[N], angularB[N];
            _Pragma("unroll") for (int i = 0; i < N; ++i) warmStartRows(p, i, j[i], angularA[i], angularB[i]);
            BD_GATE(vA, vB, j, angularA, angularB);
            _Pragma("unroll") for (int i = 0; i < N; ++i) {
                R::tangentApply(j[i], iA, iB, V2{a[3 * i], a[3 * i + 1]}, vA, vB);
                R::penApply(iA, iB, normal(p, i), angularA[i], angularB[i], a[3 * i + 2], vA, vB);
            }
        } else {
            gate(vA, vB);
            _Pragma("unroll") for (int i = 0; i < N; ++i) {
                if (i > 0) orderAfterPreviousContact(p, i, vA, vB);
                typename R::TJ j; V3 angularA, angularB;
                warmStartRows(p, i, j, angularA, angularB);
                R::tangentApply(j, iA, iB, V2{a[3 * i], a[3 * i + 1]}, vA, vB);
                R::penApply(iA, iB, normal(p, i), angularA, angularB, a[3 * i + 2], vA, vB);
            }
        }
    }
    // Launch-per-batch kernels: keep the compiler from evaluating every contact's rows up front (it would, they are independent of the velocities, and
    // the register count of the whole kernel follows); contact i's inputs become available only after contact i-1 has updated the velocities.
    BD_FN void orderAfterPreviousContact(float* p, int i, BodyVel& vA, BodyVel& vB) {
        pin(vA, vB);
        _Pragma("unroll") for (int f = 0; f < 7; ++f) pin_one(p[cOff + 7 * i + f]);
    }
    BD_FN void warmStartRows(const float* p, int i, typename R::TJ& j, V3& angularA, V3& angularB) {
        V3 n = normal(p, i);
        V3 x, z;
        buildOrthonormalBasis(n, x, z);
        V3 oA = offset(p, i);
        V3 oB = TwoBody ? sub(oA, offsetB(p)) : V3{0, 0, 0};
        j = R::tangentJacobians(x, z, oA, oB);
        angularA = cross(oA, n);                          // PenetrationLimit.cs:69-76
        angularB = TwoBody ? cross(n, oB) : V3{0, 0, 0};
    }
    BD_FN void solveRows(const Inertia& iA, const Inertia& iB, float* p, int i, float posErrToVel, float effMassCFMScale, float inverseDt,
                         typename R::PenRow& row, typename R::TangentSetup& tangent) {
        V3 n = normal(p, i);
        V3 oA = offset(p, i);
        V3 oB = TwoBody ? sub(oA, offsetB(p)) : V3{0, 0, 0};
        row = R::penSetup(iA, iB, n, oA, oB, depth(p, i), posErrToVel, effMassCFMScale, maxRecovery(p), inverseDt);
        V3 x, z;
        buildOrthonormalBasis(n, x, z);
        tangent = R::tangentSetup(x, z, oA, oB, iA, iB);
    }
    BD_FN void solveContact(const typename R::PenRow& row, const typename R::TangentSetup& tangentRows, const Inertia& iA, const Inertia& iB, float* p, float* a, int i,
                            float softnessImpulseScale, BodyVel& vA, BodyVel& vB) {
        R::penIterate(row, iA, iB, normal(p, i), softnessImpulseScale, a[3 * i + 2], vA, vB);
        float maximumTangentImpulse = friction(p) * a[3 * i + 2];
        V2 tangent{a[3 * i], a[3 * i + 1]};
        R::tangentIterate(tangentRows, iA, iB, maximumTangentImpulse, tangent, vA, vB);
        a[3 * i] = tangent.x; a[3 * i + 1] = tangent.y;
    }
    template <class G> BD_FN void solve(V3, Q, const Inertia& iA, V3, Q, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :208-228, :272-285
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(springFreq(p), springDamp(p), dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        if constexpr (std::remove_reference_t<G>::kPin) {
            typename R::PenRow rows[N];
            typename R::TangentSetup tangents[N];
            _Pragma("unroll") for (int i = 0; i < N; ++i) solveRows(iA, iB, p, i, posErrToVel, effMassCFMScale, inverseDt, rows[i], tangents[i]);
            BD_GATE(vA, vB, rows, tangents, softnessImpulseScale);
            _Pragma("unroll") for (int i = 0; i < N; ++i) solveContact(rows[i], tangents[i], iA, iB, p, a, i, softnessImpulseScale, vA, vB);
        } else {
            gate(vA, vB);
            _Pragma("unroll") for (int i = 0; i < N; ++i) {
                if (i > 0) orderAfterPreviousContact(p, i, vA, vB);
                typename R::PenRow row; typename R::TangentSetup tangent;
                solveRows(iA, iB, p, i, posErrToVel, effMassCFMScale, inverseDt, row, tangent);
                solveContact(row, tangent, iA, iB, p, a, i, softnessImpulseScale, vA, vB);
            }
        }
    }
};

// ======================================================================================
// BallSocket — BepuPhysics/Constraints/BallSocket.cs:60-103, BallSocketShared.cs:19-135.
// Prestep: LocalOffsetA xyz, LocalOffsetB xyz, {AngularFrequency, TwiceDampingRatio}. Impulses: xyz.
// ======================================================================================
struct BallSocketShared {
    BD_FN Sym3 computeEffectiveMass(const Inertia& iA, const Inertia& iB, V3 offsetA, V3 offsetB, float effectiveMassCFMScale) {  // :19-46
        Sym3 inverseEffectiveMass = skewSandwich(offsetA, iA.t);
        Sym3 angularBContribution = skewSandwich(offsetB, iB.t);
        inverseEffectiveMass = add(inverseEffectiveMass, angularBContribution);
        float linearContribution = iA.invMass + iB.invMass;
        inverseEffectiveMass.xx += linearContribution;
        inverseEffectiveMass.yy += linearContribution;
        inverseEffectiveMass.zz += linearContribution;
        Sym3 effectiveMass = invert(inverseEffectiveMass);
        return scale(effectiveMass, effectiveMassCFMScale);
    }
    BD_FN void applyImpulse(BodyVel& vA, BodyVel& vB, V3 offsetA, V3 offsetB, const Inertia& iA, const Inertia& iB, V3 csi) {  // :49-66
        V3 wsi = cross(offsetA, csi);
        V3 change = transform(wsi, iA.t);
        vA.ang = add(vA.ang, change);
        change = scale(csi, iA.invMass);
        vA.lin = add(vA.lin, change);
        wsi = cross(csi, offsetB);
        change = transform(wsi, iB.t);
        vB.ang = add(vB.ang, change);
        change = scale(csi, iB.invMass);
        vB.lin = sub(vB.lin, change);
    }
    BD_FN V3 computeCorrectiveImpulse(const BodyVel& vA, const BodyVel& vB, V3 offsetA, V3 offsetB, V3 biasVelocity, const Sym3& effectiveMass,
                                       float softnessImpulseScale, V3 accumulatedImpulse) {  // :69-98
        V3 csv = sub(vA.lin, vB.lin);
        V3 angularCSV = cross(vA.ang, offsetA);
        csv = add(csv, angularCSV);
        angularCSV = cross(offsetB, vB.ang);
        csv = add(csv, angularCSV);
        csv = sub(biasVelocity, csv);
        V3 correctiveImpulse = transform(csv, effectiveMass);
        V3 softness = scale(accumulatedImpulse, softnessImpulseScale);
        return sub(correctiveImpulse, softness);
    }
};
struct BallSocket {
    static constexpr int bodies = 2, prestepFloats = 8, impulseFloats = 3, typeId = kBallSocket;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessNoPosition, wsB = kAccessNoPosition, svA = kAccessAll, svB = kAccessAll;  // BallSocket.cs:100-103
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :68-74
        V3 offsetA = transform(V3{p[0], p[1], p[2]}, oA);
        V3 offsetB = transform(V3{p[3], p[4], p[5]}, oB);
        BD_GATE(vA, vB, offsetA, offsetB);
        BallSocketShared::applyImpulse(vA, vB, offsetA, offsetB, iA, iB, V3{a[0], a[1], a[2]});
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :76-91
        V3 offsetA = transform(V3{p[0], p[1], p[2]}, oA);
        V3 offsetB = transform(V3{p[3], p[4], p[5]}, oB);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[6], p[7], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        Sym3 effectiveMass = BallSocketShared::computeEffectiveMass(iA, iB, offsetA, offsetB, effMassCFMScale);
        V3 ab = sub(pB, pA);
        V3 anchorB = add(ab, offsetB);
        V3 error = sub(anchorB, offsetA);
        V3 biasVelocity = scale(error, posErrToVel);
        BD_GATE(vA, vB, offsetA, offsetB, effectiveMass, biasVelocity, softnessImpulseScale);
        // BallSocketShared.Solve :101-108
        V3 acc{a[0], a[1], a[2]};
        V3 correctiveImpulse = BallSocketShared::computeCorrectiveImpulse(vA, vB, offsetA, offsetB, biasVelocity, effectiveMass, softnessImpulseScale, acc);
        acc = add(acc, correctiveImpulse);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        BallSocketShared::applyImpulse(vA, vB, offsetA, offsetB, iA, iB, correctiveImpulse);
    }
};

// ======================================================================================
// AngularHinge — BepuPhysics/Constraints/AngularHinge.cs:52-228.
// Prestep: LocalHingeAxisA xyz, LocalHingeAxisB xyz, spring{freq, 2*damp}. Impulses: xy.
// ======================================================================================
struct AngularHinge {
    static constexpr int bodies = 2, prestepFloats = 8, impulseFloats = 2, typeId = kAngularHinge;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngularWithoutPose, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // AngularHinge.cs:225
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN V2 getErrorAngles(V3 hingeAxisA, V3 hingeAxisB, const M23& jacobianA) {  // :74-111
        float hingeAxisBDotX = dot(hingeAxisB, jacobianA.X);
        float hingeAxisBDotY = dot(hingeAxisB, jacobianA.Y);
        V3 toRemoveX = scale(jacobianA.X, hingeAxisBDotX);
        V3 toRemoveY = scale(jacobianA.Y, hingeAxisBDotY);
        V3 hingeAxisBOnPlaneX = sub(hingeAxisB, toRemoveX);
        V3 hingeAxisBOnPlaneY = sub(hingeAxisB, toRemoveY);
        float xLength = length(hingeAxisBOnPlaneX);
        float yLength = length(hingeAxisBOnPlaneY);
        float scaleX = 1.0f / xLength;
        float scaleY = 1.0f / yLength;
        hingeAxisBOnPlaneX = scale(hingeAxisBOnPlaneX, scaleX);
        hingeAxisBOnPlaneY = scale(hingeAxisBOnPlaneY, scaleY);
        const float epsilon = 1e-7f;
        bool useFallbackX = xLength < epsilon;
        bool useFallbackY = yLength < epsilon;
        hingeAxisBOnPlaneX = sel3(useFallbackX, hingeAxisA, hingeAxisBOnPlaneX);
        hingeAxisBOnPlaneY = sel3(useFallbackY, hingeAxisA, hingeAxisBOnPlaneY);
        float hbxha = dot(hingeAxisBOnPlaneX, hingeAxisA);
        float hbyha = dot(hingeAxisBOnPlaneY, hingeAxisA);
        V2 errorAngles;
        errorAngles.x = bacos(hbxha);
        errorAngles.y = bacos(hbyha);
        float hbxay = dot(hingeAxisBOnPlaneX, jacobianA.Y);
        float hbyax = dot(hingeAxisBOnPlaneY, jacobianA.X);
        errorAngles.x = sel(hbxay < 0.0f, errorAngles.x, -errorAngles.x);
        errorAngles.y = sel(hbyax < 0.0f, -errorAngles.y, errorAngles.y);
        return errorAngles;
    }
    BD_FN void applyImpulse(const M23& impulseToVelocityA, const M23& negatedImpulseToVelocityB, V2 csi, V3& angA, V3& angB) {  // :114-120
        V3 velocityChangeA = transform(csi, impulseToVelocityA);
        angA = add(angA, velocityChangeA);
        V3 negatedVelocityChangeB = transform(csi, negatedImpulseToVelocityB);
        angB = sub(angB, negatedVelocityChangeB);
    }
    BD_FN void computeJacobians(V3 localHingeAxisA, Q orientationA, V3& hingeAxisA, M23& jacobianA) {  // :123-130
        V3 localAX, localAY;
        buildOrthonormalBasis(localHingeAxisA, localAX, localAY);
        M3 orientationMatrixA = createFromQuaternion(orientationA);
        hingeAxisA = transform(localHingeAxisA, orientationMatrixA);
        jacobianA.X = transform(localAX, orientationMatrixA);
        jacobianA.Y = transform(localAY, orientationMatrixA);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :132-138
        V3 hingeAxisA; M23 jacobianA;
        computeJacobians(V3{p[0], p[1], p[2]}, oA, hingeAxisA, jacobianA);
        M23 impulseToVelocityA = multiply(jacobianA, iA.t);
        M23 negatedImpulseToVelocityB = multiply(jacobianA, iB.t);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB);
        applyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, V2{a[0], a[1]}, vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :140-217
        V3 hingeAxisA; M23 jacobianA;
        computeJacobians(V3{p[0], p[1], p[2]}, oA, hingeAxisA, jacobianA);
        V3 hingeAxisB = transform(V3{p[3], p[4], p[5]}, oB);
        M23 impulseToVelocityA = multiply(jacobianA, iA.t);
        M23 negatedImpulseToVelocityB = multiply(jacobianA, iB.t);
        Sym2 angularA = completeMatrixSandwich2(impulseToVelocityA, jacobianA);
        Sym2 angularB = completeMatrixSandwich2(negatedImpulseToVelocityB, jacobianA);
        Sym2 inverseEffectiveMass = add(angularA, angularB);
        Sym2 effectiveMass = invert(inverseEffectiveMass);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[6], p[7], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V2 errorAngle = getErrorAngles(hingeAxisA, hingeAxisB, jacobianA);
        V2 biasVelocity = scale(errorAngle, -posErrToVel);
        V2 biasImpulse = transform(biasVelocity, effectiveMass);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB, jacobianA, effectiveMass, biasImpulse, effMassCFMScale, softnessImpulseScale);
        V3 difference = sub(vA.ang, vB.ang);
        V2 csv = transformByTranspose(difference, jacobianA);
        V2 csi = transform(csv, effectiveMass);
        csi = scale(csi, effMassCFMScale);
        V2 acc{a[0], a[1]};
        V2 softnessContribution = scale(acc, softnessImpulseScale);
        csi = add(softnessContribution, csi);
        csi = sub(biasImpulse, csi);
        acc = add(acc, csi);
        a[0] = acc.x; a[1] = acc.y;
        applyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, csi, vA.ang, vB.ang);
    }
};

// ======================================================================================
// SwingLimit — BepuPhysics/Constraints/SwingLimit.cs:84-171; InequalityHelpers.cs:15-20.
// Prestep: AxisLocalA xyz, AxisLocalB xyz, MinimumDot, spring{freq, 2*damp}. Impulse: scalar.
// ======================================================================================
BD_FN void clampPositive(float& accumulatedImpulse, float& impulse) {  // InequalityHelpers.cs:15-20
    float previous = accumulatedImpulse;
    accumulatedImpulse = vmax(0.0f, accumulatedImpulse + impulse);
    impulse = accumulatedImpulse - previous;
}
// Shared 1-DOF angular impulse application: SwingLimit.cs:95-101 == TwistServo.cs:161-167.
BD_FN void applyAngularImpulse1(V3 impulseToVelocityA, V3 negatedImpulseToVelocityB, float csi, V3& angA, V3& angB) {
    V3 velocityChangeA = scale(impulseToVelocityA, csi);
    angA = add(angA, velocityChangeA);
    V3 negatedVelocityChangeB = scale(negatedImpulseToVelocityB, csi);
    angB = sub(angB, negatedVelocityChangeB);
}
struct SwingLimit {
    static constexpr int bodies = 2, prestepFloats = 9, impulseFloats = 1, typeId = kSwingLimit;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngular, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // SwingLimit.cs:171
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void computeJacobian(V3 axisLocalA, V3 axisLocalB, Q oA, Q oB, V3& axisA, V3& axisB, V3& jacobianA) {  // :104-113
        axisA = transform(axisLocalA, oA);
        axisB = transform(axisLocalB, oB);
        jacobianA = cross(axisA, axisB);
        V3 fallbackJacobian = findPerpendicular(axisA);
        float jacobianLengthSquared = dot(jacobianA, jacobianA);
        bool useFallback = jacobianLengthSquared < 1e-7f;
        jacobianA = sel3(useFallback, fallbackJacobian, jacobianA);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :114-120
        V3 axisA, axisB, jacobianA;
        computeJacobian(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, oA, oB, axisA, axisB, jacobianA);
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB);
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, a[0], vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :122-163
        V3 axisA, axisB, jacobianA;
        computeJacobian(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, oA, oB, axisA, axisB, jacobianA);
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        float angularContributionA = dot(impulseToVelocityA, jacobianA);
        float angularContributionB = dot(negatedImpulseToVelocityB, jacobianA);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[7], p[8], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / (angularContributionA + angularContributionB);
        float axisDot = dot(axisA, axisB);
        float error = axisDot - p[6];
        float biasVelocity = -vmin(error * inverseDt, error * posErrToVel);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB, jacobianA, effectiveMass, biasVelocity, softnessImpulseScale);
        V3 difference = sub(vA.ang, vB.ang);
        float csv = dot(difference, jacobianA);
        float csi = effectiveMass * (biasVelocity - csv) - a[0] * softnessImpulseScale;
        clampPositive(a[0], csi);
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, csi, vA.ang, vB.ang);
    }
};

// ======================================================================================
// TwistServo / TwistLimit — BepuPhysics/Constraints/TwistServo.cs:77-228, TwistLimit.cs:77-141.
// TwistServo prestep: LocalBasisA xyzw, LocalBasisB xyzw, TargetAngle, spring{2}, servo{MaximumSpeed, BaseSpeed, MaximumForce}.
// TwistLimit prestep: LocalBasisA xyzw, LocalBasisB xyzw, MinimumAngle, MaximumAngle, spring{2}. Impulse: scalar.
// ======================================================================================
struct TwistShared {
    BD_FN void computeJacobianFull(Q oA, Q oB, Q localBasisA, Q localBasisB, V3& basisBX, V3& basisBZ, M3& basisA, V3& jacobianA) {  // TwistServo.cs:89-114
        Q basisQuaternionA = concatenate(localBasisA, oA);
        Q basisQuaternionB = concatenate(localBasisB, oB);
        transformUnitXZ(basisQuaternionB, basisBX, basisBZ);
        basisA = createFromQuaternion(basisQuaternionA);
        jacobianA = add(basisA.Z, basisBZ);
        float len = length(jacobianA);
        jacobianA = scale(jacobianA, 1.0f / len);
        jacobianA = sel3(len < 1e-10f, basisA.Z, jacobianA);
    }
    BD_FN void computeJacobianOnly(Q oA, Q oB, Q localBasisA, Q localBasisB, V3& jacobianA) {  // TwistServo.cs:170-182
        Q basisQuaternionA = concatenate(localBasisA, oA);
        Q basisQuaternionB = concatenate(localBasisB, oB);
        V3 basisAZ = transformUnitZ(basisQuaternionA);
        V3 basisBZ = transformUnitZ(basisQuaternionB);
        jacobianA = add(basisAZ, basisBZ);
        float len = length(jacobianA);
        jacobianA = scale(jacobianA, 1.0f / len);
        jacobianA = sel3(len < 1e-10f, basisAZ, jacobianA);
    }
    BD_FN float computeCurrentAngle(V3 basisBX, V3 basisBZ, const M3& basisA) {  // TwistServo.cs:117-128
        Q aligningRotation = quaternionBetweenNormalizedVectors(basisBZ, basisA.Z);
        V3 alignedBasisBX = transform(basisBX, aligningRotation);
        float x = dot(alignedBasisBX, basisA.X);
        float y = dot(alignedBasisBX, basisA.Y);
        float absAngle = bacos(x);
        return sel(y < 0.0f, -absAngle, absAngle);
    }
    BD_FN void computeEffectiveMass(float dt, float springFreq, float springDamp, const Sym3& invA, const Sym3& invB, V3 jacobianA,
                                     V3& impulseToVelocityA, V3& negatedImpulseToVelocityB, float& posErrToVel, float& softnessImpulseScale,
                                     float& effectiveMass, V3& velocityToImpulseA) {  // TwistServo.cs:131-158
        impulseToVelocityA = transform(jacobianA, invA);
        negatedImpulseToVelocityB = transform(jacobianA, invB);
        float angularA = dot(impulseToVelocityA, jacobianA);
        float angularB = dot(negatedImpulseToVelocityB, jacobianA);
        float unsoftenedInverseEffectiveMass = angularA + angularB;
        float effMassCFMScale;
        computeSpringiness(springFreq, springDamp, dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        effectiveMass = effMassCFMScale / unsoftenedInverseEffectiveMass;
        velocityToImpulseA = scale(jacobianA, effectiveMass);
    }
};
struct TwistServo {
    static constexpr int bodies = 2, prestepFloats = 14, impulseFloats = 1, typeId = kTwistServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngular, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // TwistServo.cs:224
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :184-190
        V3 jacobianA;
        TwistShared::computeJacobianOnly(oA, oB, Q{p[0], p[1], p[2], p[3]}, Q{p[4], p[5], p[6], p[7]}, jacobianA);
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB);
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, a[0], vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :192-222
        V3 basisBX, basisBZ, jacobianA; M3 basisA;
        TwistShared::computeJacobianFull(oA, oB, Q{p[0], p[1], p[2], p[3]}, Q{p[4], p[5], p[6], p[7]}, basisBX, basisBZ, basisA, jacobianA);
        V3 impulseToVelocityA, negatedImpulseToVelocityB, velocityToImpulseA;
        float posErrToVel, softnessImpulseScale, effectiveMass;
        TwistShared::computeEffectiveMass(dt, p[9], p[10], iA.t, iB.t, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB,
                                          posErrToVel, softnessImpulseScale, effectiveMass, velocityToImpulseA);
        float angle = TwistShared::computeCurrentAngle(basisBX, basisBZ, basisA);
        float error = signedAngleDifference(p[8], angle);
        // ServoSettingsWide.ComputeClampedBiasVelocity (scalar-error form), ServoSettings.cs:75-85
        float maximumSpeed = p[11], baseSpeedSetting = p[12], maximumForce = p[13];
        float baseSpeed = vmin(baseSpeedSetting, vabs(error) * inverseDt);
        float biasVelocity = error * posErrToVel;
        float clampedBiasVelocity = sel(biasVelocity < 0.0f,
                                        vmax(-maximumSpeed, vmin(-baseSpeed, biasVelocity)),
                                        vmin(maximumSpeed, vmax(baseSpeed, biasVelocity)));
        float maximumImpulse = maximumForce * dt;
        float biasImpulse = clampedBiasVelocity * effectiveMass;
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB, velocityToImpulseA, biasImpulse, softnessImpulseScale, maximumImpulse);
        V3 netVelocity = sub(vA.ang, vB.ang);
        float csiVelocityComponent = dot(netVelocity, velocityToImpulseA);
        float csi = biasImpulse - a[0] * softnessImpulseScale - csiVelocityComponent;
        float previousAccumulatedImpulse = a[0];
        a[0] = vmin(vmax(a[0] + csi, -maximumImpulse), maximumImpulse);
        csi = a[0] - previousAccumulatedImpulse;
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, csi, vA.ang, vB.ang);
    }
};
struct TwistLimit {
    static constexpr int bodies = 2, prestepFloats = 12, impulseFloats = 1, typeId = kTwistLimit;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngular, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // TwistLimit.cs:137
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void computeJacobian(Q oA, Q oB, Q localBasisA, Q localBasisB, float minimumAngle, float maximumAngle, float& error, V3& jacobianA) {  // :88-103
        V3 basisBX, basisBZ; M3 basisA;
        TwistShared::computeJacobianFull(oA, oB, localBasisA, localBasisB, basisBX, basisBZ, basisA, jacobianA);
        float angle = TwistShared::computeCurrentAngle(basisBX, basisBZ, basisA);
        float minError = signedAngleDifference(minimumAngle, angle);
        float maxError = signedAngleDifference(maximumAngle, angle);
        bool useMin = vabs(minError) < vabs(maxError);
        error = sel(useMin, -minError, maxError);
        V3 negatedJacobianA = neg(jacobianA);
        jacobianA = sel3(useMin, negatedJacobianA, jacobianA);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :104-110
        float error; V3 jacobianA;
        computeJacobian(oA, oB, Q{p[0], p[1], p[2], p[3]}, Q{p[4], p[5], p[6], p[7]}, p[8], p[9], error, jacobianA);
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB);
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, a[0], vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :112-131
        float error; V3 jacobianA;
        computeJacobian(oA, oB, Q{p[0], p[1], p[2], p[3]}, Q{p[4], p[5], p[6], p[7]}, p[8], p[9], error, jacobianA);
        V3 impulseToVelocityA, negatedImpulseToVelocityB, velocityToImpulseA;
        float posErrToVel, softnessImpulseScale, effectiveMass;
        TwistShared::computeEffectiveMass(dt, p[10], p[11], iA.t, iB.t, jacobianA, impulseToVelocityA, negatedImpulseToVelocityB,
                                          posErrToVel, softnessImpulseScale, effectiveMass, velocityToImpulseA);
        float biasVelocity = sel(error < 0.0f, error * inverseDt, error * posErrToVel);
        float biasImpulse = biasVelocity * effectiveMass;
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB, velocityToImpulseA, biasImpulse, softnessImpulseScale);
        V3 netVelocity = sub(vA.ang, vB.ang);
        float csiVelocityComponent = dot(netVelocity, velocityToImpulseA);
        float csi = biasImpulse - a[0] * softnessImpulseScale - csiVelocityComponent;
        clampPositive(a[0], csi);
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, csi, vA.ang, vB.ang);
    }
};

// ======================================================================================
// AngularMotor — BepuPhysics/Constraints/AngularMotor.cs:55-97; AngularServo.cs:73-79 (ApplyImpulse);
// MotorSettings.cs:70-99 (ComputeSoftness); ServoSettings.cs:167-178 (ClampImpulse, Vector3Wide).
// Prestep: TargetVelocityLocalA xyz, {MaximumForce, Damping}. Impulses: xyz.
// ======================================================================================
struct AngularMotor {
    static constexpr int bodies = 2, prestepFloats = 5, impulseFloats = 3, typeId = kAngularMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngularWithoutPose, wsB = kAccessOnlyAngularWithoutPose, svA = kAccessOnlyAngular, svB = kAccessOnlyAngularWithoutPose;  // AngularMotor.cs:96
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(V3& angA, V3& angB, const Sym3& impulseToVelocityA, const Sym3& negatedImpulseToVelocityB, V3 csi) {  // AngularServo.cs:73-79
        V3 velocityChangeA = transform(csi, impulseToVelocityA);
        angA = add(angA, velocityChangeA);
        V3 negatedVelocityChangeB = transform(csi, negatedImpulseToVelocityB);
        angB = sub(angB, negatedVelocityChangeB);
    }
    template <class G> BD_FN void warmStart(V3, Q, const Inertia& iA, V3, Q, const Inertia& iB, float*, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :63-66
        gate(vA, vB);
        applyImpulse(vA.ang, vB.ang, iA.t, iB.t, V3{a[0], a[1], a[2]});
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :68-90
        // MotorSettingsWide.ComputeSoftness, MotorSettings.cs:70-99
        float dtd = dt * p[4];
        float maximumImpulse = p[3] * dt;
        float softnessImpulseScale = 1.0f / (dtd + 1.0f);
        float effectiveMassCFMScale = dtd * softnessImpulseScale;
        Sym3 unsoftenedInverseEffectiveMass = add(iA.t, iB.t);
        Sym3 unsoftenedEffectiveMass = invert(unsoftenedInverseEffectiveMass);
        V3 biasVelocity = transform(V3{p[0], p[1], p[2]}, oA);
        BD_GATE(vA, vB, unsoftenedEffectiveMass, biasVelocity, effectiveMassCFMScale, softnessImpulseScale, maximumImpulse);
        V3 csv = sub(vA.ang, vB.ang);
        csv = sub(biasVelocity, csv);
        V3 csi = transform(csv, unsoftenedEffectiveMass);
        csi = scale(csi, effectiveMassCFMScale);
        V3 acc{a[0], a[1], a[2]};
        V3 softnessComponent = scale(acc, softnessImpulseScale);
        csi = sub(csi, softnessComponent);
        // ServoSettingsWide.ClampImpulse(Vector3Wide), ServoSettings.cs:167-178
        V3 previousAccumulatedImpulse = acc;
        acc = add(acc, csi);
        float impulseMagnitude = length(acc);
        float impulseScale = sel(vabs(impulseMagnitude) < 1e-10f, 1.0f, vmin(maximumImpulse / impulseMagnitude, 1.0f));
        acc = scale(acc, impulseScale);
        csi = sub(acc, previousAccumulatedImpulse);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        applyImpulse(vA.ang, vB.ang, iA.t, iB.t, csi);
    }
};

// ======================================================================================
// SwivelHinge — BepuPhysics/Constraints/SwivelHinge.cs:74-218 (4x4 effective mass).
// Prestep: LocalOffsetA, LocalSwivelAxisA, LocalOffsetB, LocalHingeAxisB (xyz each), spring{2}. Impulses: xyzw.
// ======================================================================================
struct SwivelHinge {
    static constexpr int bodies = 2, prestepFloats = 14, impulseFloats = 4, typeId = kSwivelHinge;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessNoPosition, wsB = kAccessNoPosition, svA = kAccessAll, svB = kAccessAll;  // SwivelHinge.cs:216
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(V3 offsetA, V3 offsetB, V3 swivelHingeJacobian, const Inertia& iA, const Inertia& iB, V4 csi, BodyVel& vA, BodyVel& vB) {  // :86-105
        V3 ballSocketCSI{csi.x, csi.y, csi.z};
        V3 linearChangeA = scale(ballSocketCSI, iA.invMass);
        vA.lin = add(vA.lin, linearChangeA);
        V3 ballSocketAngularImpulseA = cross(offsetA, ballSocketCSI);
        V3 swivelHingeAngularImpulseA = scale(swivelHingeJacobian, csi.w);
        V3 angularImpulseA = add(ballSocketAngularImpulseA, swivelHingeAngularImpulseA);
        V3 angularChangeA = transform(angularImpulseA, iA.t);
        vA.ang = add(vA.ang, angularChangeA);
        V3 negatedLinearChangeB = scale(ballSocketCSI, iB.invMass);
        vB.lin = sub(vB.lin, negatedLinearChangeB);
        V3 ballSocketAngularImpulseB = cross(ballSocketCSI, offsetB);
        V3 angularImpulseB = sub(ballSocketAngularImpulseB, swivelHingeAngularImpulseA);
        V3 angularChangeB = transform(angularImpulseB, iB.t);
        vB.ang = add(vB.ang, angularChangeB);
    }
    BD_FN void computeJacobian(const float* p, Q oA, Q oB, V3& swivelAxis, V3& hingeAxis, V3& offsetA, V3& offsetB, V3& swivelHingeJacobian) {  // :108-122
        M3 orientationMatrixA = createFromQuaternion(oA);
        M3 orientationMatrixB = createFromQuaternion(oB);
        offsetA = transform(V3{p[0], p[1], p[2]}, orientationMatrixA);
        swivelAxis = transform(V3{p[3], p[4], p[5]}, orientationMatrixA);
        offsetB = transform(V3{p[6], p[7], p[8]}, orientationMatrixB);
        hingeAxis = transform(V3{p[9], p[10], p[11]}, orientationMatrixB);
        swivelHingeJacobian = cross(swivelAxis, hingeAxis);
        float lenSq = lengthSquared(swivelHingeJacobian);
        bool useFallbackJacobian = lenSq < 1e-3f;
        swivelHingeJacobian = sel3(useFallbackJacobian, hingeAxis, swivelHingeJacobian);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :124-129
        V3 swivelAxis, hingeAxis, offsetA, offsetB, jac;
        computeJacobian(p, oA, oB, swivelAxis, hingeAxis, offsetA, offsetB, jac);
        BD_GATE(vA, vB, offsetA, offsetB, jac);
        applyImpulse(offsetA, offsetB, jac, iA, iB, V4{a[0], a[1], a[2], a[3]}, vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :131-208
        V3 swivelAxis, hingeAxis, offsetA, offsetB, jac;
        computeJacobian(p, oA, oB, swivelAxis, hingeAxis, offsetA, offsetB, jac);
        Sym3 ballSocketContributionAngularA = skewSandwich(offsetA, iA.t);
        Sym3 ballSocketContributionAngularB = skewSandwich(offsetB, iB.t);
        Sym3 upperLeft = add(ballSocketContributionAngularA, ballSocketContributionAngularB);
        float linearContribution = iA.invMass + iB.invMass;
        upperLeft.xx += linearContribution;
        upperLeft.yy += linearContribution;
        upperLeft.zz += linearContribution;
        V3 swivelHingeInertiaA = transform(jac, iA.t);
        V3 swivelHingeInertiaB = transform(jac, iB.t);
        float swivelHingeContributionAngularA = dot(swivelHingeInertiaA, jac);
        float swivelHingeContributionAngularB = dot(swivelHingeInertiaB, jac);
        Sym4 inverseEffectiveMass;
        inverseEffectiveMass.xx = upperLeft.xx; inverseEffectiveMass.yx = upperLeft.yx; inverseEffectiveMass.yy = upperLeft.yy;
        inverseEffectiveMass.zx = upperLeft.zx; inverseEffectiveMass.zy = upperLeft.zy; inverseEffectiveMass.zz = upperLeft.zz;
        inverseEffectiveMass.ww = swivelHingeContributionAngularA + swivelHingeContributionAngularB;
        V3 offDiagonalContributionA = cross(swivelHingeInertiaA, offsetA);
        V3 offDiagonalContributionB = cross(swivelHingeInertiaB, offsetB);
        V3 upperRight = add(offDiagonalContributionA, offDiagonalContributionB);
        inverseEffectiveMass.wx = upperRight.x; inverseEffectiveMass.wy = upperRight.y; inverseEffectiveMass.wz = upperRight.z;
        Sym4 effectiveMass = invert(inverseEffectiveMass);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[12], p[13], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 anchorB = add(sub(pB, pA), offsetB);
        V3 ballSocketError = sub(anchorB, offsetA);
        V4 biasVelocity;
        biasVelocity.x = ballSocketError.x * posErrToVel;
        biasVelocity.y = ballSocketError.y * posErrToVel;
        biasVelocity.z = ballSocketError.z * posErrToVel;
        float error = dot(hingeAxis, swivelAxis);
        biasVelocity.w = posErrToVel * -error;
        BD_GATE(vA, vB, offsetA, offsetB, jac, effectiveMass, biasVelocity, effMassCFMScale, softnessImpulseScale);
        V3 ballSocketAngularCSVA = cross(vA.ang, offsetA);
        float swivelHingeCSVA = dot(jac, vA.ang);
        V3 ballSocketAngularCSVB = cross(offsetB, vB.ang);
        float negatedSwivelHingeCSVB = dot(jac, vB.ang);
        V3 ballSocketAngularCSV = add(ballSocketAngularCSVA, ballSocketAngularCSVB);
        V3 ballSocketLinearCSV = sub(vA.lin, vB.lin);
        V4 csv;
        csv.x = ballSocketAngularCSV.x + ballSocketLinearCSV.x;
        csv.y = ballSocketAngularCSV.y + ballSocketLinearCSV.y;
        csv.z = ballSocketAngularCSV.z + ballSocketLinearCSV.z;
        csv.w = swivelHingeCSVA - negatedSwivelHingeCSVB;
        csv = V4{biasVelocity.x - csv.x, biasVelocity.y - csv.y, biasVelocity.z - csv.z, biasVelocity.w - csv.w};
        V4 csi = transform(csv, effectiveMass);
        csi = V4{csi.x * effMassCFMScale, csi.y * effMassCFMScale, csi.z * effMassCFMScale, csi.w * effMassCFMScale};
        V4 soft{a[0] * softnessImpulseScale, a[1] * softnessImpulseScale, a[2] * softnessImpulseScale, a[3] * softnessImpulseScale};
        csi = V4{csi.x - soft.x, csi.y - soft.y, csi.z - soft.z, csi.w - soft.w};
        a[0] = a[0] + csi.x; a[1] = a[1] + csi.y; a[2] = a[2] + csi.z; a[3] = a[3] + csi.w;
        applyImpulse(offsetA, offsetB, jac, iA, iB, csi, vA, vB);
    }
};

// ======================================================================================
// Hinge — BepuPhysics/Constraints/Hinge.cs:74-226 (5x5 effective mass via Schur complement).
// Prestep: LocalOffsetA, LocalHingeAxisA, LocalOffsetB, LocalHingeAxisB (xyz each), spring{2}.
// Impulses: BallSocket xyz, Hinge xy.
// ======================================================================================
struct Hinge {
    static constexpr int bodies = 2, prestepFloats = 14, impulseFloats = 5, typeId = kHinge;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessNoPosition, wsB = kAccessNoPosition, svA = kAccessAll, svB = kAccessAll;  // Hinge.cs:224
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(V3 offsetA, V3 offsetB, const M23& hingeJacobian, const Inertia& iA, const Inertia& iB, V3 csiBall, V2 csiHinge, BodyVel& vA, BodyVel& vB) {  // :91-110
        V3 linearChangeA = scale(csiBall, iA.invMass);
        vA.lin = add(vA.lin, linearChangeA);
        V3 ballSocketAngularImpulseA = cross(offsetA, csiBall);
        V3 hingeAngularImpulseA = transform(csiHinge, hingeJacobian);
        V3 angularImpulseA = add(ballSocketAngularImpulseA, hingeAngularImpulseA);
        V3 angularChangeA = transform(angularImpulseA, iA.t);
        vA.ang = add(vA.ang, angularChangeA);
        V3 negatedLinearChangeB = scale(csiBall, iB.invMass);
        vB.lin = sub(vB.lin, negatedLinearChangeB);
        V3 ballSocketAngularImpulseB = cross(csiBall, offsetB);
        V3 angularImpulseB = sub(ballSocketAngularImpulseB, hingeAngularImpulseA);
        V3 angularChangeB = transform(angularImpulseB, iB.t);
        vB.ang = add(vB.ang, angularChangeB);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :112-122
        M3 orientationMatrixA = createFromQuaternion(oA);
        V3 offsetA = transform(V3{p[0], p[1], p[2]}, orientationMatrixA);
        V3 offsetB = transform(V3{p[6], p[7], p[8]}, oB);
        V3 localAX, localAY;
        buildOrthonormalBasis(V3{p[3], p[4], p[5]}, localAX, localAY);
        M23 hingeJacobian;
        hingeJacobian.X = transform(localAX, orientationMatrixA);
        hingeJacobian.Y = transform(localAY, orientationMatrixA);
        BD_GATE(vA, vB, offsetA, offsetB, hingeJacobian);
        applyImpulse(offsetA, offsetB, hingeJacobian, iA, iB, V3{a[0], a[1], a[2]}, V2{a[3], a[4]}, vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :124-216
        M3 orientationMatrixA = createFromQuaternion(oA);
        M3 orientationMatrixB = createFromQuaternion(oB);
        V3 offsetA = transform(V3{p[0], p[1], p[2]}, orientationMatrixA);
        V3 hingeAxisA = transform(V3{p[3], p[4], p[5]}, orientationMatrixA);
        V3 offsetB = transform(V3{p[6], p[7], p[8]}, orientationMatrixB);
        V3 hingeAxisB = transform(V3{p[9], p[10], p[11]}, orientationMatrixB);
        V3 localAX, localAY;
        buildOrthonormalBasis(V3{p[3], p[4], p[5]}, localAX, localAY);
        M23 hingeJacobian;
        hingeJacobian.X = transform(localAX, orientationMatrixA);
        hingeJacobian.Y = transform(localAY, orientationMatrixA);
        Sym3 ballSocketContributionAngularA = skewSandwich(offsetA, iA.t);
        Sym3 ballSocketContributionAngularB = skewSandwich(offsetB, iB.t);
        Sym3 mA = add(ballSocketContributionAngularA, ballSocketContributionAngularB);
        float linearContribution = iA.invMass + iB.invMass;
        mA.xx += linearContribution;
        mA.yy += linearContribution;
        mA.zz += linearContribution;
        M23 hingeInertiaA = multiply(hingeJacobian, iA.t);
        M23 hingeInertiaB = multiply(hingeJacobian, iB.t);
        Sym2 hingeContributionAngularA = completeMatrixSandwich2(hingeInertiaA, hingeJacobian);
        Sym2 hingeContributionAngularB = completeMatrixSandwich2(hingeInertiaB, hingeJacobian);
        Sym2 mD = add(hingeContributionAngularA, hingeContributionAngularB);
        V3 offDiagonalContributionAX = cross(hingeInertiaA.X, offsetA);
        V3 offDiagonalContributionAY = cross(hingeInertiaA.Y, offsetA);
        V3 offDiagonalContributionBX = cross(hingeInertiaB.X, offsetB);
        V3 offDiagonalContributionBY = cross(hingeInertiaB.Y, offsetB);
        M23 mB;
        mB.X = add(offDiagonalContributionAX, offDiagonalContributionBX);
        mB.Y = add(offDiagonalContributionAY, offDiagonalContributionBY);
        Sym5 effectiveMass = invert5(mA, mB, mD);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[12], p[13], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 anchorB = add(sub(pB, pA), offsetB);
        V3 ballSocketError = sub(anchorB, offsetA);
        V3 ballSocketBiasVelocity = scale(ballSocketError, posErrToVel);
        V2 errorAngles = AngularHinge::getErrorAngles(hingeAxisA, hingeAxisB, hingeJacobian);
        V2 hingeBiasVelocity = scale(errorAngles, -posErrToVel);
        BD_GATE(vA, vB, offsetA, offsetB, hingeJacobian, effectiveMass, ballSocketBiasVelocity, hingeBiasVelocity, effMassCFMScale, softnessImpulseScale);
        V3 ballSocketAngularCSVA = cross(vA.ang, offsetA);
        V2 hingeCSVA = transformByTranspose(vA.ang, hingeJacobian);
        V3 ballSocketAngularCSVB = cross(offsetB, vB.ang);
        V2 negatedHingeCSVB = transformByTranspose(vB.ang, hingeJacobian);
        V3 ballSocketAngularCSV = add(ballSocketAngularCSVA, ballSocketAngularCSVB);
        V3 ballSocketLinearCSV = sub(vA.lin, vB.lin);
        V3 ballSocketCSV = add(ballSocketAngularCSV, ballSocketLinearCSV);
        ballSocketCSV = sub(ballSocketBiasVelocity, ballSocketCSV);
        V2 hingeCSV = sub(hingeCSVA, negatedHingeCSVB);
        hingeCSV = sub(hingeBiasVelocity, hingeCSV);
        V3 csiBall; V2 csiHinge;
        transform5(ballSocketCSV, hingeCSV, effectiveMass, csiBall, csiHinge);
        csiBall = scale(csiBall, effMassCFMScale);
        csiHinge = scale(csiHinge, effMassCFMScale);
        V3 accBall{a[0], a[1], a[2]}; V2 accHinge{a[3], a[4]};
        V3 ballSocketSoftnessContribution = scale(accBall, softnessImpulseScale);
        csiBall = sub(csiBall, ballSocketSoftnessContribution);
        V2 hingeSoftnessContribution = scale(accHinge, softnessImpulseScale);
        csiHinge = sub(csiHinge, hingeSoftnessContribution);
        accBall = add(accBall, csiBall);
        accHinge = add(accHinge, csiHinge);
        a[0] = accBall.x; a[1] = accBall.y; a[2] = accBall.z; a[3] = accHinge.x; a[4] = accHinge.y;
        applyImpulse(offsetA, offsetB, hingeJacobian, iA, iB, csiBall, csiHinge, vA, vB);
    }
};

// ======================================================================================
// Weld — BepuPhysics/Constraints/Weld.cs:70-215 (6 DOF; Symmetric6x6Wide.LDLTSolve, BepuUtilities/Symmetric6x6Wide.cs:84-129).
// Prestep: LocalOffset xyz, LocalOrientation xyzw, spring{freq, 2*damp}. Impulses: Orientation xyz, Offset xyz.
// ======================================================================================
struct Weld {
    static constexpr int bodies = 2, prestepFloats = 9, impulseFloats = 6, typeId = kWeld;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessNoPosition, wsB = kAccessNoPose, svA = kAccessAll, svB = kAccessAll;  // Weld.cs:212
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(const Inertia& iA, const Inertia& iB, V3 offset, V3 orientationCSI, V3 offsetCSI, BodyVel& vA, BodyVel& vB) {  // :85-113
        V3 linearChangeA = scale(offsetCSI, iA.invMass);
        vA.lin = add(vA.lin, linearChangeA);
        V3 offsetWorldImpulse = cross(offset, offsetCSI);
        V3 angularImpulseA = add(offsetWorldImpulse, orientationCSI);
        V3 angularChangeA = transform(angularImpulseA, iA.t);
        vA.ang = add(vA.ang, angularChangeA);
        V3 negatedLinearChangeB = scale(offsetCSI, iB.invMass);
        vB.lin = sub(vB.lin, negatedLinearChangeB);
        V3 negatedAngularChangeB = transform(orientationCSI, iB.t);
        vB.ang = sub(vB.ang, negatedAngularChangeB);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :116-121
        V3 offset = transform(V3{p[0], p[1], p[2]}, oA);
        BD_GATE(vA, vB, offset);
        applyImpulse(iA, iB, offset, V3{a[0], a[1], a[2]}, V3{a[3], a[4], a[5]}, vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :123-204
        V3 offset = transform(V3{p[0], p[1], p[2]}, oA);
        Sym3 jmjtA = add(iA.t, iB.t);
        M3 xAB = createCrossProduct(offset);
        M3 jmjtB = multiply(iA.t, xAB);
        Sym3 jmjtD = completeMatrixSandwichTranspose(xAB, jmjtB);
        float diagonalAdd = iA.invMass + iB.invMass;
        jmjtD.xx += diagonalAdd;
        jmjtD.yy += diagonalAdd;
        jmjtD.zz += diagonalAdd;
        V3 positionError = sub(sub(pB, pA), offset);
        Q targetOrientationB = concatenate(Q{p[3], p[4], p[5], p[6]}, oA);
        Q rotationError = concatenate(conjugate(targetOrientationB), oB);
        V3 rotationErrorAxis; float rotationErrorLength;
        getAxisAngleFromQuaternion(rotationError, rotationErrorAxis, rotationErrorLength);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[7], p[8], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 orientationBiasVelocity = scale(rotationErrorAxis, rotationErrorLength * posErrToVel);
        V3 offsetBiasVelocity = scale(positionError, posErrToVel);
        LDLT6 factor = ldltFactor(jmjtA, jmjtB, jmjtD);  // the factorisation half of LDLTSolve needs no velocity
        BD_GATE(vA, vB, offset, factor, orientationBiasVelocity, offsetBiasVelocity, effMassCFMScale, softnessImpulseScale);
        V3 orientationCSV, offsetCSV;
        orientationCSV.x = orientationBiasVelocity.x - vA.ang.x + vB.ang.x;
        orientationCSV.y = orientationBiasVelocity.y - vA.ang.y + vB.ang.y;
        orientationCSV.z = orientationBiasVelocity.z - vA.ang.z + vB.ang.z;
        offsetCSV.x = offsetBiasVelocity.x - vA.lin.x + vB.lin.x - (vA.ang.y * offset.z - vA.ang.z * offset.y);
        offsetCSV.y = offsetBiasVelocity.y - vA.lin.y + vB.lin.y - (vA.ang.z * offset.x - vA.ang.x * offset.z);
        offsetCSV.z = offsetBiasVelocity.z - vA.lin.z + vB.lin.z - (vA.ang.x * offset.y - vA.ang.y * offset.x);
        V3 orientationCSI, offsetCSI;
        ldltSubstitute(factor, orientationCSV, offsetCSV, orientationCSI, offsetCSI);
        orientationCSI.x = orientationCSI.x * effMassCFMScale - a[0] * softnessImpulseScale;
        orientationCSI.y = orientationCSI.y * effMassCFMScale - a[1] * softnessImpulseScale;
        orientationCSI.z = orientationCSI.z * effMassCFMScale - a[2] * softnessImpulseScale;
        a[0] += orientationCSI.x; a[1] += orientationCSI.y; a[2] += orientationCSI.z;
        offsetCSI.x = offsetCSI.x * effMassCFMScale - a[3] * softnessImpulseScale;
        offsetCSI.y = offsetCSI.y * effMassCFMScale - a[4] * softnessImpulseScale;
        offsetCSI.z = offsetCSI.z * effMassCFMScale - a[5] * softnessImpulseScale;
        a[3] += offsetCSI.x; a[4] += offsetCSI.y; a[5] += offsetCSI.z;
        applyImpulse(iA, iB, offset, orientationCSI, offsetCSI, vA, vB);
    }
};

// ======================================================================================
// SURVEY.md 8(f) widening — further constraint types, same template as above (velocity gate, pinned setup).
// (CenterDistanceConstraint / CenterDistanceLimit / AreaConstraint / VolumeConstraint, which go through MathHelper.FastReciprocal*, are further below,
// restated on that helper's portable branch.)
// ======================================================================================
// ServoSettingsWide (BepuPhysics/Constraints/ServoSettings.cs): prestep order {MaximumSpeed, BaseSpeed, MaximumForce}.
BD_FN void servoClampedBiasVelocity(float error, float positionErrorToVelocity, float maximumSpeed, float baseSpeedSetting, float maximumForce, float dt, float inverseDt,
                                    float& clampedBiasVelocity, float& maximumImpulse) {  // :75-85
    float baseSpeed = vmin(baseSpeedSetting, vabs(error) * inverseDt);
    float biasVelocity = error * positionErrorToVelocity;
    clampedBiasVelocity = sel(biasVelocity < 0.0f, vmax(-maximumSpeed, vmin(-baseSpeed, biasVelocity)), vmin(maximumSpeed, vmax(baseSpeed, biasVelocity)));
    maximumImpulse = maximumForce * dt;
}
BD_FN void servoClampedBiasVelocity(V3 errorAxis, float errorLength, float positionErrorToBiasVelocity, float maximumSpeed, float baseSpeedSetting, float maximumForce,
                                    float dt, float inverseDt, V3& clampedBiasVelocity, float& maximumImpulse) {  // :116-130
    float baseSpeed = vmin(baseSpeedSetting, errorLength * inverseDt);
    float unclampedBiasSpeed = errorLength * positionErrorToBiasVelocity;
    float targetSpeed = vmax(baseSpeed, unclampedBiasSpeed);
    float sc = vmin(1.0f, maximumSpeed / targetSpeed);
    bool useFallback = targetSpeed < 1e-10f;
    sc = sel(useFallback, 1.0f, sc);
    clampedBiasVelocity = scale(errorAxis, sc * unclampedBiasSpeed);
    maximumImpulse = maximumForce * dt;
}
BD_FN void servoClampedBiasVelocityFromError(V3 error, float positionErrorToBiasVelocity, float maximumSpeed, float baseSpeedSetting, float maximumForce,
                                             float dt, float inverseDt, V3& clampedBiasVelocity, float& maximumImpulse) {  // :132-143
    float errorLength = length(error);
    V3 errorAxis = scale(error, 1.0f / errorLength);
    bool useFallback = errorLength < 1e-10f;
    errorAxis = {sel(useFallback, 0.0f, errorAxis.x), sel(useFallback, 0.0f, errorAxis.y), sel(useFallback, 0.0f, errorAxis.z)};
    servoClampedBiasVelocity(errorAxis, errorLength, positionErrorToBiasVelocity, maximumSpeed, baseSpeedSetting, maximumForce, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
}
BD_FN void servoClampImpulse(float maximumImpulse, float& accumulatedImpulse, float& csi) {  // :145-151
    float previousImpulse = accumulatedImpulse;
    accumulatedImpulse = vmax(-maximumImpulse, vmin(maximumImpulse, accumulatedImpulse + csi));
    csi = accumulatedImpulse - previousImpulse;
}
BD_FN void servoClampImpulse(float maximumImpulse, V3& accumulatedImpulse, V3& csi) {  // :167-178
    V3 previousAccumulatedImpulse = accumulatedImpulse;
    accumulatedImpulse = add(accumulatedImpulse, csi);
    float impulseMagnitude = length(accumulatedImpulse);
    float impulseScale = sel(vabs(impulseMagnitude) < 1e-10f, 1.0f, vmin(maximumImpulse / impulseMagnitude, 1.0f));
    accumulatedImpulse = scale(accumulatedImpulse, impulseScale);
    csi = sub(accumulatedImpulse, previousAccumulatedImpulse);
}
// MotorSettingsWide.ComputeSoftness (BepuPhysics/Constraints/MotorSettings.cs:70-99): prestep order {MaximumForce, Damping}.
BD_FN void motorSoftness(float maximumForce, float damping, float dt, float& effectiveMassCFMScale, float& softnessImpulseScale, float& maximumImpulse) {
    float dtd = dt * damping;
    maximumImpulse = maximumForce * dt;
    softnessImpulseScale = 1.0f / (dtd + 1.0f);
    effectiveMassCFMScale = dtd * softnessImpulseScale;
}

// AngularSwivelHinge — AngularSwivelHinge.cs:53-155. Prestep: LocalSwivelAxisA xyz, LocalHingeAxisB xyz, spring{2}. Impulse: scalar.
struct AngularSwivelHinge {
    static constexpr int bodies = 2, prestepFloats = 8, impulseFloats = 1, typeId = kAngularSwivelHinge;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngular, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // :152
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void computeJacobian(V3 localSwivelAxisA, V3 localHingeAxisB, Q oA, Q oB, V3& swivelAxis, V3& hingeAxis, V3& jacobianA) {  // :73-86
        swivelAxis = transform(localSwivelAxisA, oA);
        hingeAxis = transform(localHingeAxisB, oB);
        jacobianA = cross(swivelAxis, hingeAxis);
        V3 fallbackJacobian = findPerpendicular(swivelAxis);
        float jacobianLengthSquared = dot(jacobianA, jacobianA);
        bool useFallback = jacobianLengthSquared < 1e-3f;
        jacobianA = sel3(useFallback, fallbackJacobian, jacobianA);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :88-94
        V3 swivelAxis, hingeAxis, jacobianA;
        computeJacobian(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, oA, oB, swivelAxis, hingeAxis, jacobianA);
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB);
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, a[0], vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :96-138
        V3 swivelAxis, hingeAxis, jacobianA;
        computeJacobian(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, oA, oB, swivelAxis, hingeAxis, jacobianA);
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        float angularA = dot(impulseToVelocityA, jacobianA);
        float angularB = dot(negatedImpulseToVelocityB, jacobianA);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[6], p[7], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / (angularA + angularB);
        float error = dot(hingeAxis, swivelAxis);
        float biasVelocity = -(posErrToVel * error);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB, jacobianA, effectiveMass, biasVelocity, softnessImpulseScale);
        V3 difference = sub(vA.ang, vB.ang);
        float csv = dot(difference, jacobianA);
        float csi = effectiveMass * (biasVelocity - csv) - a[0] * softnessImpulseScale;
        a[0] += csi;
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, csi, vA.ang, vB.ang);
    }
};

// TwistMotor — TwistMotor.cs:45-125. Prestep: LocalAxisA xyz, LocalAxisB xyz, TargetVelocity, motor{MaximumForce, Damping}. Impulse: scalar.
struct TwistMotor {
    static constexpr int bodies = 2, prestepFloats = 9, impulseFloats = 1, typeId = kTwistMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngular, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // :122
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN V3 computeJacobian(Q oA, Q oB, V3 localAxisA, V3 localAxisB) {  // :59-68
        V3 axisA = transform(localAxisA, oA);
        V3 axisB = transform(localAxisB, oB);
        V3 jacobianA = add(axisA, axisB);
        float len = length(jacobianA);
        jacobianA = scale(jacobianA, 1.0f / len);
        return sel3(len < 1e-10f, axisA, jacobianA);
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :70-76
        V3 jacobianA = computeJacobian(oA, oB, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]});
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB);
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, a[0], vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :78-103
        V3 jacobianA = computeJacobian(oA, oB, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]});
        // TwistServoFunctions.ComputeEffectiveMassContributions, TwistServo.cs:132-144
        V3 impulseToVelocityA = transform(jacobianA, iA.t);
        V3 negatedImpulseToVelocityB = transform(jacobianA, iB.t);
        float angularA = dot(impulseToVelocityA, jacobianA);
        float angularB = dot(negatedImpulseToVelocityB, jacobianA);
        float unsoftenedInverseEffectiveMass = angularA + angularB;
        float effMassCFMScale, softnessImpulseScale, maximumImpulse;
        motorSoftness(p[7], p[8], dt, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        float effectiveMass = effMassCFMScale / unsoftenedInverseEffectiveMass;
        V3 velocityToImpulseA = scale(jacobianA, effectiveMass);
        float biasImpulse = p[6] * effectiveMass;
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB, velocityToImpulseA, biasImpulse, softnessImpulseScale, maximumImpulse);
        V3 netVelocity = sub(vA.ang, vB.ang);
        float csiVelocityComponent = dot(netVelocity, velocityToImpulseA);
        float csi = biasImpulse - a[0] * softnessImpulseScale - csiVelocityComponent;
        float previousAccumulatedImpulse = a[0];
        a[0] = vmax(vmin(a[0] + csi, maximumImpulse), -maximumImpulse);
        csi = a[0] - previousAccumulatedImpulse;
        applyAngularImpulse1(impulseToVelocityA, negatedImpulseToVelocityB, csi, vA.ang, vB.ang);
    }
};

// AngularServo — AngularServo.cs:53-145. Prestep: TargetRelativeRotationLocalA xyzw, spring{2}, servo{3}. Impulses: xyz.
struct AngularServo {
    static constexpr int bodies = 2, prestepFloats = 9, impulseFloats = 3, typeId = kAngularServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngularWithoutPose, wsB = kAccessOnlyAngularWithoutPose, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // :142
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q, const Inertia& iA, V3, Q, const Inertia& iB, float*, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :100-103
        gate(vA, vB);
        AngularMotor::applyImpulse(vA.ang, vB.ang, iA.t, iB.t, V3{a[0], a[1], a[2]});
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :105-134
        Q targetOrientationB = concatenate(Q{p[0], p[1], p[2], p[3]}, oA);
        Q inverseTarget = conjugate(targetOrientationB);
        Q errorRotation = concatenate(inverseTarget, oB);
        V3 errorAxis; float errorLength;
        getAxisAngleFromQuaternion(errorRotation, errorAxis, errorLength);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[4], p[5], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        Sym3 unsoftenedInverseEffectiveMass = add(iA.t, iB.t);
        Sym3 unsoftenedEffectiveMass = invert(unsoftenedInverseEffectiveMass);
        V3 clampedBiasVelocity; float maximumImpulse;
        servoClampedBiasVelocity(errorAxis, errorLength, posErrToVel, p[6], p[7], p[8], dt, inverseDt, clampedBiasVelocity, maximumImpulse);
        BD_GATE(vA, vB, unsoftenedEffectiveMass, clampedBiasVelocity, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        V3 csv = sub(vA.ang, vB.ang);
        csv = sub(clampedBiasVelocity, csv);
        V3 csi = transform(csv, unsoftenedEffectiveMass);
        csi = scale(csi, effMassCFMScale);
        V3 acc{a[0], a[1], a[2]};
        V3 softnessComponent = scale(acc, softnessImpulseScale);
        csi = sub(csi, softnessComponent);
        servoClampImpulse(maximumImpulse, acc, csi);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        AngularMotor::applyImpulse(vA.ang, vB.ang, iA.t, iB.t, csi);
    }
};

// DistanceServo — DistanceServo.cs:82-232. Prestep: LocalOffsetA xyz, LocalOffsetB xyz, TargetDistance, servo{3}, spring{2}. Impulse: scalar.
struct DistanceServo {
    static constexpr int bodies = 2, prestepFloats = 12, impulseFloats = 1, typeId = kDistanceServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessAll, wsB = kAccessAll, svA = kAccessAll, svB = kAccessAll;  // :229
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void getDistance(Q oA, V3 ab, Q oB, V3 localOffsetA, V3 localOffsetB, V3& anchorOffsetA, V3& anchorOffsetB, V3& anchorOffset, float& dist) {  // :93-102
        anchorOffsetA = transform(localOffsetA, oA);
        anchorOffsetB = transform(localOffsetB, oB);
        V3 anchorB = add(anchorOffsetB, ab);
        anchorOffset = sub(anchorB, anchorOffsetA);
        dist = length(anchorOffset);
    }
    BD_FN void computeJacobian(float dist, V3 anchorOffsetA, V3 anchorOffsetB, V3& direction, V3& angularJA, V3& angularJB) {  // :104-114
        bool needFallback = dist < 1e-9f;
        direction = {sel(needFallback, 1.0f, direction.x), sel(needFallback, 0.0f, direction.y), sel(needFallback, 0.0f, direction.z)};
        angularJA = cross(anchorOffsetA, direction);
        angularJB = cross(direction, anchorOffsetB);
    }
    BD_FN void applyImpulse(float inverseMassA, float inverseMassB, V3 direction, V3 angularImpulseToVelocityA, V3 angularImpulseToVelocityB, float csi, BodyVel& vA, BodyVel& vB) {  // :139-154
        V3 linearVelocityChangeA = scale(direction, csi * inverseMassA);
        V3 angularVelocityChangeA = scale(angularImpulseToVelocityA, csi);
        vA.lin = add(linearVelocityChangeA, vA.lin);
        vA.ang = add(angularVelocityChangeA, vA.ang);
        V3 negatedLinearVelocityChangeB = scale(direction, csi * inverseMassB);
        V3 angularVelocityChangeB = scale(angularImpulseToVelocityB, csi);
        vB.lin = sub(vB.lin, negatedLinearVelocityChangeB);
        vB.ang = add(angularVelocityChangeB, vB.ang);
    }
    template <class G> BD_FN void warmStart(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :156-165
        V3 anchorOffsetA, anchorOffsetB, anchorOffset; float dist;
        getDistance(oA, sub(pB, pA), oB, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, anchorOffsetA, anchorOffsetB, anchorOffset, dist);
        V3 direction = scale(anchorOffset, 1.0f / dist);
        V3 angularJA, angularJB;
        computeJacobian(dist, anchorOffsetA, anchorOffsetB, direction, angularJA, angularJB);
        V3 angularImpulseToVelocityA = transform(angularJA, iA.t);
        V3 angularImpulseToVelocityB = transform(angularJB, iB.t);
        BD_GATE(vA, vB, direction, angularImpulseToVelocityA, angularImpulseToVelocityB);
        applyImpulse(iA.invMass, iB.invMass, direction, angularImpulseToVelocityA, angularImpulseToVelocityB, a[0], vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :167-216
        V3 anchorOffsetA, anchorOffsetB, anchorOffset; float dist;
        getDistance(oA, sub(pB, pA), oB, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, anchorOffsetA, anchorOffsetB, anchorOffset, dist);
        V3 direction = scale(anchorOffset, 1.0f / dist);
        // ComputeTransforms :116-137
        V3 angularJA, angularJB;
        computeJacobian(dist, anchorOffsetA, anchorOffsetB, direction, angularJA, angularJB);
        V3 angularImpulseToVelocityA = transform(angularJA, iA.t);
        V3 angularImpulseToVelocityB = transform(angularJB, iB.t);
        float angularContributionA = dot(angularJA, angularImpulseToVelocityA);
        float angularContributionB = dot(angularJB, angularImpulseToVelocityB);
        float inverseEffectiveMass = iA.invMass + iB.invMass + angularContributionA + angularContributionB;
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[10], p[11], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / inverseEffectiveMass;
        float error = dist - p[6];
        float clampedBiasVelocity, maximumImpulse;
        servoClampedBiasVelocity(error, posErrToVel, p[7], p[8], p[9], dt, inverseDt, clampedBiasVelocity, maximumImpulse);
        BD_GATE(vA, vB, direction, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass, clampedBiasVelocity, softnessImpulseScale, maximumImpulse);
        float linearCSVA = dot(vA.lin, direction);
        float negatedLinearCSVB = dot(vB.lin, direction);
        float angularCSVA = dot(vA.ang, angularJA);
        float angularCSVB = dot(vB.ang, angularJB);
        float csi = (clampedBiasVelocity - linearCSVA - angularCSVA + negatedLinearCSVB - angularCSVB) * effectiveMass - a[0] * softnessImpulseScale;
        servoClampImpulse(maximumImpulse, a[0], csi);
        applyImpulse(iA.invMass, iB.invMass, direction, angularImpulseToVelocityA, angularImpulseToVelocityB, csi, vA, vB);
    }
};

// DistanceLimit — DistanceLimit.cs:72-187. Prestep: LocalOffsetA xyz, LocalOffsetB xyz, MinimumDistance, MaximumDistance, spring{2}. Impulse: scalar.
struct DistanceLimit {
    static constexpr int bodies = 2, prestepFloats = 10, impulseFloats = 1, typeId = kDistanceLimit;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessAll, wsB = kAccessAll, svA = kAccessAll, svB = kAccessAll;  // :184
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(V3 linearJacobianA, V3 angularJacobianA, V3 angularJacobianB, const Inertia& iA, const Inertia& iB, float csi, BodyVel& vA, BodyVel& vB) {  // :84-93
        V3 impulseScaledLinearJacobian = scale(linearJacobianA, csi);
        vA.lin = add(vA.lin, scale(impulseScaledLinearJacobian, iA.invMass));
        vB.lin = sub(vB.lin, scale(impulseScaledLinearJacobian, iB.invMass));
        vA.ang = add(vA.ang, transform(scale(angularJacobianA, csi), iA.t));
        vB.ang = add(vB.ang, transform(scale(angularJacobianB, csi), iB.t));
    }
    BD_FN void computeJacobians(V3 localOffsetA, V3 pA, Q oA, V3 localOffsetB, V3 pB, Q oB, float minimumDistance, float maximumDistance,
                                bool& useMinimum, float& dist, V3& direction, V3& angularJA, V3& angularJB) {  // :95-117
        V3 offsetA = transform(localOffsetA, oA);
        V3 offsetB = transform(localOffsetB, oB);
        V3 anchorOffset = add(sub(offsetB, offsetA), sub(pB, pA));
        dist = length(anchorOffset);
        useMinimum = vabs(dist - minimumDistance) < vabs(dist - maximumDistance);
        float sign = sel(useMinimum, -1.0f, 1.0f);
        direction = scale(anchorOffset, sign / dist);
        bool needFallback = dist < 1e-9f;
        direction = {sel(needFallback, 1.0f, direction.x), sel(needFallback, 0.0f, direction.y), sel(needFallback, 0.0f, direction.z)};
        angularJA = cross(offsetA, direction);
        angularJB = cross(direction, offsetB);
    }
    template <class G> BD_FN void warmStart(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :119-124
        bool useMinimum; float dist; V3 direction, angularJA, angularJB;
        computeJacobians(V3{p[0], p[1], p[2]}, pA, oA, V3{p[3], p[4], p[5]}, pB, oB, p[6], p[7], useMinimum, dist, direction, angularJA, angularJB);
        BD_GATE(vA, vB, direction, angularJA, angularJB);
        applyImpulse(direction, angularJA, angularJB, iA, iB, a[0], vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :126-156
        bool useMinimum; float dist; V3 direction, angularJA, angularJB;
        computeJacobians(V3{p[0], p[1], p[2]}, pA, oA, V3{p[3], p[4], p[5]}, pB, oB, p[6], p[7], useMinimum, dist, direction, angularJA, angularJB);
        float angularContributionA = vectorSandwich(angularJA, iA.t);
        float angularContributionB = vectorSandwich(angularJB, iB.t);
        float inverseEffectiveMass = iA.invMass + iB.invMass + angularContributionA + angularContributionB;
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[8], p[9], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / inverseEffectiveMass;
        float error = sel(useMinimum, p[6] - dist, dist - p[7]);
        float biasVelocity = vmin(error * inverseDt, error * posErrToVel);  // InequalityHelpers.ComputeBiasVelocity, InequalityHelpers.cs:9-12
        BD_GATE(vA, vB, direction, angularJA, angularJB, effectiveMass, biasVelocity, softnessImpulseScale);
        float linearCSVA = dot(vA.lin, direction);
        float negatedLinearCSVB = dot(vB.lin, direction);
        float angularCSVA = dot(vA.ang, angularJA);
        float angularCSVB = dot(vB.ang, angularJB);
        float csv = linearCSVA - negatedLinearCSVB + angularCSVA + angularCSVB;
        float csi = -a[0] * softnessImpulseScale - effectiveMass * (csv - biasVelocity);
        clampPositive(a[0], csi);
        applyImpulse(direction, angularJA, angularJB, iA, iB, csi, vA, vB);
    }
};

// AngularAxisMotor — AngularAxisMotor.cs:43-113. Prestep: LocalAxisA xyz, TargetVelocity, motor{2}. Impulse: scalar.
struct AngularAxisMotor {
    static constexpr int bodies = 2, prestepFloats = 6, impulseFloats = 1, typeId = kAngularAxisMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngularWithoutPose, svA = kAccessOnlyAngular, svB = kAccessOnlyAngular;  // :110
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(V3 impulseToVelocityA, V3 negatedImpulseToVelocityB, float csi, V3& angA, V3& angB) {  // :56-60
        angA = add(angA, scale(impulseToVelocityA, csi));
        angB = sub(angB, scale(negatedImpulseToVelocityB, csi));
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :62-69
        V3 axis = transform(V3{p[0], p[1], p[2]}, oA);
        V3 jIA = transform(axis, iA.t);
        V3 jIB = transform(axis, iB.t);
        BD_GATE(vA, vB, jIA, jIB);
        applyImpulse(jIA, jIB, a[0], vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :71-98
        V3 jA = transform(V3{p[0], p[1], p[2]}, oA);
        V3 jIA = transform(jA, iA.t);
        float contributionA = dot(jA, jIA);
        V3 jIB = transform(jA, iB.t);
        float contributionB = dot(jA, jIB);
        float effMassCFMScale, softnessImpulseScale, maximumImpulse;
        motorSoftness(p[4], p[5], dt, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        float inverseEffectiveMass = contributionA + contributionB;
        BD_GATE(vA, vB, jA, jIA, jIB, inverseEffectiveMass, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        float csi = (p[3] + dot(vB.ang, jA) - dot(vA.ang, jA)) * effMassCFMScale / inverseEffectiveMass - a[0] * softnessImpulseScale;
        servoClampImpulse(maximumImpulse, a[0], csi);
        applyImpulse(jIA, jIB, csi, vA.ang, vB.ang);
    }
};

// OneBodyAngularServo — OneBodyAngularServo.cs:44-116. Prestep: TargetOrientation xyzw, spring{2}, servo{3}. Impulses: xyz.
struct OneBodyAngularServo {
    static constexpr int bodies = 1, prestepFloats = 9, impulseFloats = 3, typeId = kOneBodyAngularServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = 0, svA = kAccessOnlyAngular, svB = 0;  // :113
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q, const Inertia& iA, V3, Q, const Inertia&, float*, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :66-69
        gate(vA, vB);
        vA.ang = add(vA.ang, transform(V3{a[0], a[1], a[2]}, iA.t));
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q, const Inertia&, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :71-101
        Q inverseOrientation = conjugate(oA);
        Q errorRotation = concatenate(inverseOrientation, Q{p[0], p[1], p[2], p[3]});
        V3 errorAxis; float errorLength;
        getAxisAngleFromQuaternion(errorRotation, errorAxis, errorLength);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[4], p[5], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        Sym3 effectiveMass = invert(iA.t);
        V3 clampedBiasVelocity; float maximumImpulse;
        servoClampedBiasVelocity(errorAxis, errorLength, posErrToVel, p[6], p[7], p[8], dt, inverseDt, clampedBiasVelocity, maximumImpulse);
        BD_GATE(vA, vB, effectiveMass, clampedBiasVelocity, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        V3 csv = sub(clampedBiasVelocity, vA.ang);
        V3 csi = transform(csv, effectiveMass);
        V3 acc{a[0], a[1], a[2]};
        csi = sub(scale(csi, effMassCFMScale), scale(acc, softnessImpulseScale));
        servoClampImpulse(maximumImpulse, acc, csi);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        vA.ang = add(vA.ang, transform(csi, iA.t));
    }
};

// OneBodyAngularMotor — OneBodyAngularMotor.cs:41-100. Prestep: TargetVelocity xyz, motor{2}. Impulses: xyz.
struct OneBodyAngularMotor {
    static constexpr int bodies = 1, prestepFloats = 5, impulseFloats = 3, typeId = kOneBodyAngularMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngularWithoutPose, wsB = 0, svA = kAccessOnlyAngular, svB = 0;  // :97
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q, const Inertia& iA, V3, Q, const Inertia&, float*, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :59-62
        gate(vA, vB);
        vA.ang = add(vA.ang, transform(V3{a[0], a[1], a[2]}, iA.t));
    }
    template <class G> BD_FN void solve(V3, Q, const Inertia& iA, V3, Q, const Inertia&, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :64-86
        float effMassCFMScale, softnessImpulseScale, maximumImpulse;
        motorSoftness(p[3], p[4], dt, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        Sym3 unsoftenedEffectiveMass = invert(iA.t);
        BD_GATE(vA, vB, unsoftenedEffectiveMass, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        V3 csi = transform(sub(V3{p[0], p[1], p[2]}, vA.ang), unsoftenedEffectiveMass);
        V3 acc{a[0], a[1], a[2]};
        csi = sub(scale(csi, effMassCFMScale), scale(acc, softnessImpulseScale));
        servoClampImpulse(maximumImpulse, acc, csi);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        vA.ang = add(vA.ang, transform(csi, iA.t));
    }
};

// OneBodyLinearServo / OneBodyLinearMotor — OneBodyLinearServo.cs:51-152, OneBodyLinearMotor.cs:44-106.
// Servo prestep: LocalOffset xyz, Target xyz, spring{2}, servo{3}. Motor prestep: LocalOffset xyz, TargetVelocity xyz, motor{2}. Impulses: xyz.
struct OneBodyLinearShared {
    BD_FN void applyImpulse(V3 offset, const Inertia& inertia, BodyVel& vA, V3 csi) {  // OneBodyLinearServo.cs:85-93
        V3 wsi = cross(offset, csi);
        V3 change = transform(wsi, inertia.t);
        vA.ang = add(vA.ang, change);
        change = scale(csi, inertia.invMass);
        vA.lin = add(vA.lin, change);
    }
    BD_FN Sym3 effectiveMass(V3 offset, const Inertia& inertia) {  // :117-121 / OneBodyLinearMotor.cs:77-81
        Sym3 inverseEffectiveMass = skewSandwich(offset, inertia.t);
        inverseEffectiveMass.xx += inertia.invMass;
        inverseEffectiveMass.yy += inertia.invMass;
        inverseEffectiveMass.zz += inertia.invMass;
        return invert(inverseEffectiveMass);
    }
};
struct OneBodyLinearServo {
    static constexpr int bodies = 1, prestepFloats = 11, impulseFloats = 3, typeId = kOneBodyLinearServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessAll, wsB = 0, svA = kAccessAll, svB = 0;  // :149
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q, const Inertia&, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :95-100
        V3 offset = transform(V3{p[0], p[1], p[2]}, oA);
        BD_GATE(vA, vB, offset);
        OneBodyLinearShared::applyImpulse(offset, iA, vA, V3{a[0], a[1], a[2]});
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3, Q, const Inertia&, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :102-133
        V3 offset = transform(V3{p[0], p[1], p[2]}, oA);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[6], p[7], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 worldGrabPoint = add(offset, pA);
        V3 error = sub(V3{p[3], p[4], p[5]}, worldGrabPoint);
        V3 biasVelocity; float maximumImpulse;
        servoClampedBiasVelocityFromError(error, posErrToVel, p[8], p[9], p[10], dt, inverseDt, biasVelocity, maximumImpulse);
        Sym3 effectiveMass = OneBodyLinearShared::effectiveMass(offset, iA);
        BD_GATE(vA, vB, offset, biasVelocity, effectiveMass, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        V3 csv = sub(sub(biasVelocity, cross(vA.ang, offset)), vA.lin);
        V3 csi = transform(csv, effectiveMass);
        V3 acc{a[0], a[1], a[2]};
        csi = sub(scale(csi, effMassCFMScale), scale(acc, softnessImpulseScale));
        servoClampImpulse(maximumImpulse, acc, csi);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        OneBodyLinearShared::applyImpulse(offset, iA, vA, csi);
    }
};
struct OneBodyLinearMotor {
    static constexpr int bodies = 1, prestepFloats = 8, impulseFloats = 3, typeId = kOneBodyLinearMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessNoPosition, wsB = 0, svA = kAccessNoPosition, svB = 0;  // :103
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q, const Inertia&, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :57-61
        V3 offset = transform(V3{p[0], p[1], p[2]}, oA);
        BD_GATE(vA, vB, offset);
        OneBodyLinearShared::applyImpulse(offset, iA, vA, V3{a[0], a[1], a[2]});
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q, const Inertia&, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :63-91
        V3 offset = transform(V3{p[0], p[1], p[2]}, oA);
        float effMassCFMScale, softnessImpulseScale, maximumImpulse;
        motorSoftness(p[6], p[7], dt, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        Sym3 effectiveMass = OneBodyLinearShared::effectiveMass(offset, iA);
        BD_GATE(vA, vB, offset, effectiveMass, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        V3 csv = sub(sub(V3{p[3], p[4], p[5]}, cross(vA.ang, offset)), vA.lin);
        V3 csi = transform(csv, effectiveMass);
        V3 acc{a[0], a[1], a[2]};
        csi = sub(scale(csi, effMassCFMScale), scale(acc, softnessImpulseScale));
        servoClampImpulse(maximumImpulse, acc, csi);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        OneBodyLinearShared::applyImpulse(offset, iA, vA, csi);
    }
};

// BallSocketMotor / BallSocketServo — BallSocketMotor.cs:47-103, BallSocketServo.cs:48-113, BallSocketShared.cs:100-135.
// Motor prestep: LocalOffsetB xyz, TargetVelocityLocalA xyz, motor{2}. Servo prestep: LocalOffsetA xyz, LocalOffsetB xyz, spring{2}, servo{3}. Impulses: xyz.
struct BallSocketMotor {
    static constexpr int bodies = 2, prestepFloats = 8, impulseFloats = 3, typeId = kBallSocketMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessNoOrientation, wsB = kAccessAll, svA = kAccessAll, svB = kAccessAll;  // :100
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3 pA, Q, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :60-64
        V3 targetOffsetB = transform(V3{p[0], p[1], p[2]}, oB);
        V3 offsetA = add(sub(pB, pA), targetOffsetB);
        BD_GATE(vA, vB, targetOffsetB, offsetA);
        BallSocketShared::applyImpulse(vA, vB, offsetA, targetOffsetB, iA, iB, V3{a[0], a[1], a[2]});
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :66-88
        V3 targetOffsetB = transform(V3{p[0], p[1], p[2]}, oB);
        V3 offsetA = add(sub(pB, pA), targetOffsetB);
        float effMassCFMScale, softnessImpulseScale, maximumImpulse;
        motorSoftness(p[6], p[7], dt, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        Sym3 effectiveMass = BallSocketShared::computeEffectiveMass(iA, iB, offsetA, targetOffsetB, effMassCFMScale);
        V3 biasVelocity = neg(transform(V3{p[3], p[4], p[5]}, oA));
        BD_GATE(vA, vB, targetOffsetB, offsetA, effectiveMass, biasVelocity, softnessImpulseScale, maximumImpulse);
        V3 acc{a[0], a[1], a[2]};  // BallSocketShared.Solve with maximum impulse, BallSocketShared.cs:118-126
        V3 correctiveImpulse = BallSocketShared::computeCorrectiveImpulse(vA, vB, offsetA, targetOffsetB, biasVelocity, effectiveMass, softnessImpulseScale, acc);
        servoClampImpulse(maximumImpulse, acc, correctiveImpulse);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        BallSocketShared::applyImpulse(vA, vB, offsetA, targetOffsetB, iA, iB, correctiveImpulse);
    }
};
struct BallSocketServo {
    static constexpr int bodies = 2, prestepFloats = 11, impulseFloats = 3, typeId = kBallSocketServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessNoPosition, wsB = kAccessNoPosition, svA = kAccessAll, svB = kAccessAll;  // :110
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :62-67
        V3 offsetA = transform(V3{p[0], p[1], p[2]}, oA);
        V3 offsetB = transform(V3{p[3], p[4], p[5]}, oB);
        BD_GATE(vA, vB, offsetA, offsetB);
        BallSocketShared::applyImpulse(vA, vB, offsetA, offsetB, iA, iB, V3{a[0], a[1], a[2]});
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :69-98
        V3 offsetA = transform(V3{p[0], p[1], p[2]}, oA);
        V3 offsetB = transform(V3{p[3], p[4], p[5]}, oB);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[6], p[7], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        Sym3 effectiveMass = BallSocketShared::computeEffectiveMass(iA, iB, offsetA, offsetB, effMassCFMScale);
        V3 ab = sub(pB, pA);
        V3 anchorB = add(ab, offsetB);
        V3 error = sub(anchorB, offsetA);
        V3 biasVelocity; float maximumImpulse;
        servoClampedBiasVelocityFromError(error, posErrToVel, p[8], p[9], p[10], dt, inverseDt, biasVelocity, maximumImpulse);
        BD_GATE(vA, vB, offsetA, offsetB, effectiveMass, biasVelocity, softnessImpulseScale, maximumImpulse);
        V3 acc{a[0], a[1], a[2]};
        V3 correctiveImpulse = BallSocketShared::computeCorrectiveImpulse(vA, vB, offsetA, offsetB, biasVelocity, effectiveMass, softnessImpulseScale, acc);
        servoClampImpulse(maximumImpulse, acc, correctiveImpulse);
        a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
        BallSocketShared::applyImpulse(vA, vB, offsetA, offsetB, iA, iB, correctiveImpulse);
    }
};

// ---- ServoSettingsWide, two-component forms (ServoSettings.cs:87-114, :153-165) ----
BD_FN void servoClampedBiasVelocity(V2 errorAxis, float errorLength, float positionErrorToBiasVelocity, float maximumSpeed, float baseSpeedSetting, float maximumForce,
                                    float dt, float inverseDt, V2& clampedBiasVelocity, float& maximumImpulse) {  // :87-101
    float baseSpeed = vmin(baseSpeedSetting, errorLength * inverseDt);
    float unclampedBiasSpeed = errorLength * positionErrorToBiasVelocity;
    float targetSpeed = vmax(baseSpeed, unclampedBiasSpeed);
    float sc = vmin(1.0f, maximumSpeed / targetSpeed);
    bool useFallback = targetSpeed < 1e-10f;
    sc = sel(useFallback, 1.0f, sc);
    clampedBiasVelocity = scale(errorAxis, sc * unclampedBiasSpeed);
    maximumImpulse = maximumForce * dt;
}
BD_FN void servoClampedBiasVelocityFromError(V2 error, float positionErrorToBiasVelocity, float maximumSpeed, float baseSpeedSetting, float maximumForce,
                                             float dt, float inverseDt, V2& clampedBiasVelocity, float& maximumImpulse) {  // :103-113
    float errorLength = length(error);
    V2 errorAxis = scale(error, 1.0f / errorLength);
    bool useFallback = errorLength < 1e-10f;
    errorAxis = {sel(useFallback, 0.0f, errorAxis.x), sel(useFallback, 0.0f, errorAxis.y)};
    servoClampedBiasVelocity(errorAxis, errorLength, positionErrorToBiasVelocity, maximumSpeed, baseSpeedSetting, maximumForce, dt, inverseDt, clampedBiasVelocity, maximumImpulse);
}
BD_FN void servoClampImpulse(float maximumImpulse, V2& accumulatedImpulse, V2& csi) {  // :153-165
    V2 previousImpulse = accumulatedImpulse;
    V2 unclamped = add(accumulatedImpulse, csi);
    float impulseMagnitude = length(unclamped);
    float impulseScale = sel(vabs(impulseMagnitude) < 1e-10f, 1.0f, vmin(maximumImpulse / impulseMagnitude, 1.0f));
    accumulatedImpulse = scale(unclamped, impulseScale);
    csi = sub(accumulatedImpulse, previousImpulse);
}

// ---- Linear axis constraints: a point of B held relative to a plane attached to A (LinearAxisServo.cs:88-224 shared functions) ----
struct LinearAxisShared {
    BD_FN void computeJacobians(V3 ab, Q orientationA, Q orientationB, V3 localPlaneNormalA, V3 localOffsetA, V3 localOffsetB,
                                float& planeNormalDot, V3& normal, V3& angularJA, V3& angularJB) {  // :197-212
        M3 orientationMatrixA = createFromQuaternion(orientationA);
        normal = transform(localPlaneNormalA, orientationMatrixA);
        V3 anchorA = transform(localOffsetA, orientationMatrixA);
        V3 offsetB = transform(localOffsetB, orientationB);
        V3 anchorB = add(ab, offsetB);
        planeNormalDot = dot(sub(anchorB, anchorA), normal);
        V3 offsetFromAToClosestPointOnPlaneToB = sub(anchorB, scale(normal, planeNormalDot));
        angularJA = cross(offsetFromAToClosestPointOnPlaneToB, normal);
        angularJB = cross(normal, offsetB);
    }
    BD_FN void computeEffectiveMass(V3 angularJA, V3 angularJB, const Inertia& iA, const Inertia& iB, float effectiveMassCFMScale,
                                    V3& angularImpulseToVelocityA, V3& angularImpulseToVelocityB, float& effectiveMass) {  // :214-224
        angularImpulseToVelocityA = transform(angularJA, iA.t);
        angularImpulseToVelocityB = transform(angularJB, iB.t);
        float angularContributionA = dot(angularJA, angularImpulseToVelocityA);
        float angularContributionB = dot(angularJB, angularImpulseToVelocityB);
        effectiveMass = effectiveMassCFMScale / (iA.invMass + iB.invMass + angularContributionA + angularContributionB);
    }
    BD_FN void applyImpulse(V3 linearJA, V3 angularImpulseToVelocityA, V3 angularImpulseToVelocityB, const Inertia& iA, const Inertia& iB, float csi, BodyVel& vA, BodyVel& vB) {  // :186-195
        vA.lin = add(vA.lin, scale(linearJA, csi * iA.invMass));
        vB.lin = sub(vB.lin, scale(linearJA, csi * iB.invMass));
        vA.ang = add(vA.ang, scale(angularImpulseToVelocityA, csi));
        vB.ang = add(vB.ang, scale(angularImpulseToVelocityB, csi));
    }
    BD_FN float constraintSpaceVelocity(const BodyVel& vA, const BodyVel& vB, V3 normal, V3 angularJA, V3 angularJB) {  // :246, LinearAxisMotor.cs:99, LinearAxisLimit.cs:139
        return dot(sub(vA.lin, vB.lin), normal) + dot(vA.ang, angularJA) + dot(vB.ang, angularJB);
    }
};

// LinearAxisServo — LinearAxisServo.cs:78-253. Prestep: LocalOffsetA xyz, LocalOffsetB xyz, LocalPlaneNormal xyz, TargetOffset, servo{3}, spring{2}. Impulse: scalar.
struct LinearAxisServo {
    static constexpr int bodies = 2, prestepFloats = 15, impulseFloats = 1, typeId = kLinearAxisServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessAll, wsB = kAccessAll, svA = kAccessAll, svB = kAccessAll;  // :250
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :226-232
        float planeNormalDot; V3 normal, angularJA, angularJB;
        LinearAxisShared::computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, planeNormalDot, normal, angularJA, angularJB);
        V3 angularImpulseToVelocityA = transform(angularJA, iA.t);
        V3 angularImpulseToVelocityB = transform(angularJB, iB.t);
        BD_GATE(vA, vB, normal, angularImpulseToVelocityA, angularImpulseToVelocityB);
        LinearAxisShared::applyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, iA, iB, a[0], vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :234-253
        float planeNormalDot; V3 normal, angularJA, angularJB;
        LinearAxisShared::computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, planeNormalDot, normal, angularJA, angularJB);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[13], p[14], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 angularImpulseToVelocityA, angularImpulseToVelocityB; float effectiveMass;
        LinearAxisShared::computeEffectiveMass(angularJA, angularJB, iA, iB, effMassCFMScale, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass);
        float biasVelocity, maximumImpulse;
        servoClampedBiasVelocity(planeNormalDot - p[9], posErrToVel, p[10], p[11], p[12], dt, inverseDt, biasVelocity, maximumImpulse);
        BD_GATE(vA, vB, normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass, biasVelocity, softnessImpulseScale, maximumImpulse);
        float csv = LinearAxisShared::constraintSpaceVelocity(vA, vB, normal, angularJA, angularJB);
        float csi = effectiveMass * (biasVelocity - csv) - a[0] * softnessImpulseScale;
        servoClampImpulse(maximumImpulse, a[0], csi);
        LinearAxisShared::applyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, iA, iB, csi, vA, vB);
    }
};

// LinearAxisMotor — LinearAxisMotor.cs:73-114. Prestep: LocalOffsetA xyz, LocalOffsetB xyz, LocalPlaneNormal xyz, TargetVelocity, motor{MaximumForce, Damping}. Impulse: scalar.
struct LinearAxisMotor {
    static constexpr int bodies = 2, prestepFloats = 12, impulseFloats = 1, typeId = kLinearAxisMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessAll, wsB = kAccessAll, svA = kAccessAll, svB = kAccessAll;  // :112
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    template <class G> BD_FN void warmStart(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :84-90
        float planeNormalDot; V3 normal, angularJA, angularJB;
        LinearAxisShared::computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, planeNormalDot, normal, angularJA, angularJB);
        V3 angularImpulseToVelocityA = transform(angularJA, iA.t);
        V3 angularImpulseToVelocityB = transform(angularJB, iB.t);
        BD_GATE(vA, vB, normal, angularImpulseToVelocityA, angularImpulseToVelocityB);
        LinearAxisShared::applyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, iA, iB, a[0], vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :92-105
        float planeNormalDot; V3 normal, angularJA, angularJB;
        LinearAxisShared::computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, planeNormalDot, normal, angularJA, angularJB);
        float effMassCFMScale, softnessImpulseScale, maximumImpulse;
        motorSoftness(p[10], p[11], dt, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        V3 angularImpulseToVelocityA, angularImpulseToVelocityB; float effectiveMass;
        LinearAxisShared::computeEffectiveMass(angularJA, angularJB, iA, iB, effMassCFMScale, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass);
        BD_GATE(vA, vB, normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass, softnessImpulseScale, maximumImpulse);
        float csv = LinearAxisShared::constraintSpaceVelocity(vA, vB, normal, angularJA, angularJB);
        float csi = effectiveMass * (-p[9] - csv) - a[0] * softnessImpulseScale;
        servoClampImpulse(maximumImpulse, a[0], csi);
        LinearAxisShared::applyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, iA, iB, csi, vA, vB);
    }
};

// LinearAxisLimit — LinearAxisLimit.cs:80-156. Prestep: LocalOffsetA xyz, LocalOffsetB xyz, LocalPlaneNormal xyz, MinimumOffset, MaximumOffset, spring{2}. Impulse: scalar.
struct LinearAxisLimit {
    static constexpr int bodies = 2, prestepFloats = 13, impulseFloats = 1, typeId = kLinearAxisLimit;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessAll, wsB = kAccessAll, svA = kAccessAll, svB = kAccessAll;  // :153
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void computeJacobians(V3 ab, Q orientationA, Q orientationB, V3 localPlaneNormal, V3 localOffsetA, V3 localOffsetB, float minimumOffset, float maximumOffset,
                                float& error, V3& normal, V3& angularJA, V3& angularJB) {  // :92-118
        M3 orientationMatrixA = createFromQuaternion(orientationA);
        normal = transform(localPlaneNormal, orientationMatrixA);
        V3 anchorA = transform(localOffsetA, orientationMatrixA);
        V3 offsetB = transform(localOffsetB, orientationB);
        V3 anchorB = add(ab, offsetB);
        float planeNormalDot = dot(sub(anchorB, anchorA), normal);
        // The limit chooses the normal's sign depending on which limit is closer.
        float minimumError = minimumOffset - planeNormalDot;
        float maximumError = planeNormalDot - maximumOffset;
        bool useMin = vabs(minimumError) < vabs(maximumError);
        error = sel(useMin, minimumError, maximumError);
        normal = {sel(useMin, -normal.x, normal.x), sel(useMin, -normal.y, normal.y), sel(useMin, -normal.z, normal.z)};
        // as in the reference, the (possibly negated) normal is scaled by the un-negated plane distance here
        V3 offsetFromAToClosestPointOnPlaneToB = sub(anchorB, scale(normal, planeNormalDot));
        angularJA = cross(offsetFromAToClosestPointOnPlaneToB, normal);
        angularJB = cross(normal, offsetB);
    }
    template <class G> BD_FN void warmStart(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :120-126
        float error; V3 normal, angularJA, angularJB;
        computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, p[9], p[10], error, normal, angularJA, angularJB);
        V3 angularImpulseToVelocityA = transform(angularJA, iA.t);
        V3 angularImpulseToVelocityB = transform(angularJB, iB.t);
        BD_GATE(vA, vB, normal, angularImpulseToVelocityA, angularImpulseToVelocityB);
        LinearAxisShared::applyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, iA, iB, a[0], vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :128-146
        float error; V3 normal, angularJA, angularJB;
        computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, p[9], p[10], error, normal, angularJA, angularJB);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[11], p[12], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        V3 angularImpulseToVelocityA, angularImpulseToVelocityB; float effectiveMass;
        LinearAxisShared::computeEffectiveMass(angularJA, angularJB, iA, iB, effMassCFMScale, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass);
        float biasVelocity = vmin(error * inverseDt, error * posErrToVel);  // InequalityHelpers.ComputeBiasVelocity, InequalityHelpers.cs:9-12
        BD_GATE(vA, vB, normal, angularJA, angularJB, angularImpulseToVelocityA, angularImpulseToVelocityB, effectiveMass, biasVelocity, softnessImpulseScale);
        float csv = LinearAxisShared::constraintSpaceVelocity(vA, vB, normal, angularJA, angularJB);
        float csi = effectiveMass * (biasVelocity - csv) - a[0] * softnessImpulseScale;
        clampPositive(a[0], csi);
        LinearAxisShared::applyImpulse(normal, angularImpulseToVelocityA, angularImpulseToVelocityB, iA, iB, csi, vA, vB);
    }
};

// PointOnLineServo — PointOnLineServo.cs:72-198. Prestep: LocalOffsetA xyz, LocalOffsetB xyz, LocalDirection xyz, servo{3}, spring{2}. Impulses: xy.
struct PointOnLineServo {
    static constexpr int bodies = 2, prestepFloats = 14, impulseFloats = 2, typeId = kPointOnLineServo;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessAll, wsB = kAccessAll, svA = kAccessAll, svB = kAccessAll;  // :195
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(BodyVel& vA, BodyVel& vB, const M23& linearJacobian, const M23& angularJacobianA, const M23& angularJacobianB, const Inertia& iA, const Inertia& iB, V2 csi) {  // :83-99
        V3 linearImpulseA = transform(csi, linearJacobian);
        V3 angularImpulseA = transform(csi, angularJacobianA);
        V3 angularImpulseB = transform(csi, angularJacobianB);
        V3 angularChangeA = transform(angularImpulseA, iA.t);
        V3 angularChangeB = transform(angularImpulseB, iB.t);
        V3 linearChangeA = scale(linearImpulseA, iA.invMass);
        V3 negatedLinearChangeB = scale(linearImpulseA, iB.invMass);
        vA.lin = add(linearChangeA, vA.lin);
        vA.ang = add(angularChangeA, vA.ang);
        vB.lin = sub(vB.lin, negatedLinearChangeB);
        vB.ang = add(angularChangeB, vB.ang);
    }
    BD_FN void computeJacobians(V3 ab, Q orientationA, Q orientationB, V3 localDirection, V3 localOffsetA, V3 localOffsetB,
                                V3& anchorOffset, M23& linearJacobian, M23& angularJA, M23& angularJB) {  // :102-127
        V3 localTangentX, localTangentY;
        buildOrthonormalBasis(localDirection, localTangentX, localTangentY);
        M3 orientationMatrixA = createFromQuaternion(orientationA);
        V3 anchorA = transform(localOffsetA, orientationMatrixA);
        V3 offsetB = transform(localOffsetB, orientationB);
        // offsetA: the closest point on the line to anchorB
        V3 direction = transform(localDirection, orientationMatrixA);
        V3 anchorB = add(offsetB, ab);
        anchorOffset = sub(anchorB, anchorA);
        float d = dot(anchorOffset, direction);
        V3 lineStartToClosestPointOnLine = scale(direction, d);
        V3 offsetA = add(lineStartToClosestPointOnLine, anchorA);
        linearJacobian.X = transform(localTangentX, orientationMatrixA);
        linearJacobian.Y = transform(localTangentY, orientationMatrixA);
        angularJA.X = cross(offsetA, linearJacobian.X);
        angularJA.Y = cross(offsetA, linearJacobian.Y);
        angularJB.X = cross(linearJacobian.X, offsetB);
        angularJB.Y = cross(linearJacobian.Y, offsetB);
    }
    template <class G> BD_FN void warmStart(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :128-132
        V3 anchorOffset; M23 linearJacobian, angularJA, angularJB;
        computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, anchorOffset, linearJacobian, angularJA, angularJB);
        BD_GATE(vA, vB, linearJacobian, angularJA, angularJB);
        applyImpulse(vA, vB, linearJacobian, angularJA, angularJB, iA, iB, V2{a[0], a[1]});
    }
    template <class G> BD_FN void solve(V3 pA, Q oA, const Inertia& iA, V3 pB, Q oB, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :134-187
        V3 anchorOffset; M23 linearJacobian, angularJA, angularJB;
        computeJacobians(sub(pB, pA), oA, oB, V3{p[6], p[7], p[8]}, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, anchorOffset, linearJacobian, angularJA, angularJB);
        Sym2 linearContribution = sandwichScale(linearJacobian, iA.invMass + iB.invMass);
        Sym2 angularContributionA = matrixSandwich(angularJA, iA.t);
        Sym2 angularContributionB = matrixSandwich(angularJB, iB.t);
        Sym2 inverseEffectiveMass = add(angularContributionA, angularContributionB);
        inverseEffectiveMass = add(inverseEffectiveMass, linearContribution);
        Sym2 effectiveMass = invert(inverseEffectiveMass);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[12], p[13], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        effectiveMass = {effectiveMass.xx * effMassCFMScale, effectiveMass.yx * effMassCFMScale, effectiveMass.yy * effMassCFMScale};  // Symmetric2x2Wide.Scale :31-36
        // position error and bias velocity (independent of the velocities: evaluated ahead of the constraint space velocity)
        V2 error = {dot(anchorOffset, linearJacobian.X), dot(anchorOffset, linearJacobian.Y)};
        V2 biasVelocity; float maximumImpulse;
        servoClampedBiasVelocityFromError(error, posErrToVel, p[9], p[10], p[11], dt, inverseDt, biasVelocity, maximumImpulse);
        BD_GATE(vA, vB, linearJacobian, angularJA, angularJB, effectiveMass, biasVelocity, softnessImpulseScale, maximumImpulse);
        V2 linearCSVA = transformByTranspose(vA.lin, linearJacobian);
        V2 negatedLinearCSVB = transformByTranspose(vB.lin, linearJacobian);
        V2 angularCSVA = transformByTranspose(vA.ang, angularJA);
        V2 angularCSVB = transformByTranspose(vB.ang, angularJB);
        V2 linearCSV = sub(linearCSVA, negatedLinearCSVB);
        V2 angularCSV = add(angularCSVA, angularCSVB);
        V2 csv = add(linearCSV, angularCSV);
        csv = sub(biasVelocity, csv);
        V2 csi = transform(csv, effectiveMass);
        V2 softnessContribution = scale(V2{a[0], a[1]}, softnessImpulseScale);
        csi = sub(csi, softnessContribution);
        V2 acc{a[0], a[1]};
        servoClampImpulse(maximumImpulse, acc, csi);
        a[0] = acc.x; a[1] = acc.y;
        applyImpulse(vA, vB, linearJacobian, angularJA, angularJB, iA, iB, csi);
    }
};

// AngularAxisGearMotor — AngularAxisGearMotor.cs:63-119. Prestep: LocalAxisA xyz, VelocityScale, motor{MaximumForce, Damping}. Impulse: scalar.
struct AngularAxisGearMotor {
    static constexpr int bodies = 2, prestepFloats = 6, impulseFloats = 1, typeId = kAngularAxisGearMotor;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyAngular, wsB = kAccessOnlyAngularWithoutPose, svA = kAccessOnlyAngular, svB = kAccessOnlyAngularWithoutPose;  // :116
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(V3 impulseToVelocityA, V3 negatedImpulseToVelocityB, float csi, V3& angularVelocityA, V3& angularVelocityB) {  // :73-77
        angularVelocityA = add(angularVelocityA, scale(impulseToVelocityA, csi));
        angularVelocityB = sub(angularVelocityB, scale(negatedImpulseToVelocityB, csi));
    }
    template <class G> BD_FN void warmStart(V3, Q oA, const Inertia& iA, V3, Q, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :80-87
        V3 axis = transform(V3{p[0], p[1], p[2]}, oA);
        V3 jA = scale(axis, p[3]);
        V3 impulseToVelocityA = transform(jA, iA.t);
        V3 negatedImpulseToVelocityB = transform(axis, iB.t);
        BD_GATE(vA, vB, impulseToVelocityA, negatedImpulseToVelocityB);
        applyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, a[0], vA.ang, vB.ang);
    }
    template <class G> BD_FN void solve(V3, Q oA, const Inertia& iA, V3, Q, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :90-109
        V3 axis = transform(V3{p[0], p[1], p[2]}, oA);
        V3 jA = scale(axis, p[3]);
        V3 impulseToVelocityA = transform(jA, iA.t);
        float contributionA = dot(jA, impulseToVelocityA);
        V3 negatedImpulseToVelocityB = transform(axis, iB.t);
        float contributionB = dot(axis, negatedImpulseToVelocityB);
        float effMassCFMScale, softnessImpulseScale, maximumImpulse;
        motorSoftness(p[4], p[5], dt, effMassCFMScale, softnessImpulseScale, maximumImpulse);
        float effectiveMass = effMassCFMScale / (contributionA + contributionB);
        BD_GATE(vA, vB, axis, jA, impulseToVelocityA, negatedImpulseToVelocityB, effectiveMass, softnessImpulseScale, maximumImpulse);
        float unscaledCSVA = dot(vA.ang, jA);
        float negatedCSVB = dot(vB.ang, axis);
        float csi = (negatedCSVB - unscaledCSVA) * effectiveMass - a[0] * softnessImpulseScale;
        servoClampImpulse(maximumImpulse, a[0], csi);
        // The reference applies the ACCUMULATED impulse here, not the corrective one (AngularAxisGearMotor.cs:108); reproduced as is.
        applyImpulse(impulseToVelocityA, negatedImpulseToVelocityB, a[0], vA.ang, vB.ang);
    }
};

// ======================================================================================
// The four types built on MathHelper.FastReciprocal / FastReciprocalSquareRoot (BepuUtilities/MathHelper.cs:380-412). On x86 hosts the reference evaluates
// those with rcpps / rsqrtps, approximations whose bits differ between CPU vendors; everywhere else it takes the portable branch, 1 / v and 1 / sqrt(v).
// These restate the PORTABLE branch: parity for the four types is defined against the reference on hosts without those intrinsics (the oracle does the
// same); against an x86 run the results agree to the approximation's ~1.5e-4 relative error in the reciprocal, not bit for bit.
// All four use AccessOnlyLinear (IBodyAccessFilter.cs:118-126): position, inverse mass and linear velocity only.
// ======================================================================================
BD_FN float fastReciprocal(float v) { return 1.0f / v; }                      // MathHelper.cs:392
BD_FN float fastReciprocalSquareRoot(float v) { return 1.0f / sqrtf(v); }     // MathHelper.cs:409

// CenterDistanceConstraint — CenterDistanceConstraint.cs:58-138. Prestep: TargetDistance, spring{2}. Impulse: scalar.
struct CenterDistanceConstraint {
    static constexpr int bodies = 2, prestepFloats = 3, impulseFloats = 1, typeId = kCenterDistanceConstraint;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyLinear, wsB = kAccessOnlyLinear, svA = kAccessOnlyLinear, svB = kAccessOnlyLinear;  // :135
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void applyImpulse(V3 jacobianA, float inverseMassA, float inverseMassB, float impulse, BodyVel& a, BodyVel& b) {  // :68-74
        V3 changeA = scale(jacobianA, impulse * inverseMassA);
        V3 negatedChangeB = scale(jacobianA, impulse * inverseMassB);
        a.lin = add(a.lin, changeA);
        b.lin = sub(b.lin, negatedChangeB);
    }
    template <class G> BD_FN void warmStart(V3 pA, Q, const Inertia& iA, V3 pB, Q, const Inertia& iB, float*, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :78-91
        V3 ab = sub(pB, pA);
        float lengthSq = lengthSquared(ab);
        float inverseDistance = fastReciprocalSquareRoot(lengthSq);
        bool useFallback = lengthSq < 1e-10f;
        V3 jacobianA = scale(ab, inverseDistance);
        jacobianA = {sel(useFallback, 1.0f, jacobianA.x), sel(useFallback, 0.0f, jacobianA.y), sel(useFallback, 0.0f, jacobianA.z)};
        BD_GATE(vA, vB, jacobianA);
        applyImpulse(jacobianA, iA.invMass, iB.invMass, a[0], vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q, const Inertia& iA, V3 pB, Q, const Inertia& iB, float dt, float, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :93-121
        V3 ab = sub(pB, pA);
        float dist = length(ab);
        float inverseDistance = fastReciprocal(dist);
        bool useFallback = dist < 1e-5f;
        V3 jacobianA = scale(ab, inverseDistance);
        jacobianA = {sel(useFallback, 1.0f, jacobianA.x), sel(useFallback, 0.0f, jacobianA.y), sel(useFallback, 0.0f, jacobianA.z)};
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[1], p[2], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / (iA.invMass + iB.invMass);
        float biasVelocity = (dist - p[0]) * posErrToVel;
        BD_GATE(vA, vB, jacobianA, effectiveMass, biasVelocity, softnessImpulseScale);
        float linearCSVA = dot(vA.lin, jacobianA);
        float negatedCSVB = dot(vB.lin, jacobianA);
        float csi = (biasVelocity - (linearCSVA - negatedCSVB)) * effectiveMass - a[0] * softnessImpulseScale;
        a[0] = a[0] + csi;
        applyImpulse(jacobianA, iA.invMass, iB.invMass, csi, vA, vB);
    }
};

// CenterDistanceLimit — CenterDistanceLimit.cs:73-137. Prestep: MinimumDistance, MaximumDistance, spring{2}. Impulse: scalar.
struct CenterDistanceLimit {
    static constexpr int bodies = 2, prestepFloats = 4, impulseFloats = 1, typeId = kCenterDistanceLimit;
    static constexpr bool incremental = false;
    static constexpr int wsA = kAccessOnlyLinear, wsB = kAccessOnlyLinear, svA = kAccessOnlyLinear, svB = kAccessOnlyLinear;  // :134
    BD_FN void incrementalUpdate(float, const BodyVel&, const BodyVel&, float*) {}
    BD_FN void computeJacobian(float minimumDistance, float maximumDistance, V3 positionA, V3 positionB, V3& jacobianA, float& dist, bool& useMinimum) {  // :82-98
        V3 ab = sub(positionB, positionA);
        dist = length(ab);
        float inverseDistance = fastReciprocal(dist);
        bool useFallback = dist < 1e-5f;
        jacobianA = scale(ab, inverseDistance);
        jacobianA = {sel(useFallback, 1.0f, jacobianA.x), sel(useFallback, 0.0f, jacobianA.y), sel(useFallback, 0.0f, jacobianA.z)};
        // closer to the minimum: calibrate for the minimum, otherwise for the maximum
        useMinimum = vabs(dist - minimumDistance) < vabs(dist - maximumDistance);
        jacobianA = {sel(useMinimum, -jacobianA.x, jacobianA.x), sel(useMinimum, -jacobianA.y, jacobianA.y), sel(useMinimum, -jacobianA.z, jacobianA.z)};
    }
    template <class G> BD_FN void warmStart(V3 pA, Q, const Inertia& iA, V3 pB, Q, const Inertia& iB, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :100-105
        V3 jacobianA; float dist; bool useMinimum;
        computeJacobian(p[0], p[1], pA, pB, jacobianA, dist, useMinimum);
        BD_GATE(vA, vB, jacobianA);
        CenterDistanceConstraint::applyImpulse(jacobianA, iA.invMass, iB.invMass, a[0], vA, vB);
    }
    template <class G> BD_FN void solve(V3 pA, Q, const Inertia& iA, V3 pB, Q, const Inertia& iB, float dt, float inverseDt, float* p, float* a, BodyVel& vA, BodyVel& vB, G&& gate) {  // :107-124
        V3 jacobianA; float dist; bool useMinimum;
        computeJacobian(p[0], p[1], pA, pB, jacobianA, dist, useMinimum);
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[2], p[3], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / (iA.invMass + iB.invMass);
        float error = sel(useMinimum, p[0] - dist, dist - p[1]);
        float biasVelocity = vmin(error * inverseDt, error * posErrToVel);  // InequalityHelpers.ComputeBiasVelocity, InequalityHelpers.cs:9-12
        BD_GATE(vA, vB, jacobianA, effectiveMass, biasVelocity, softnessImpulseScale);
        float csv = dot(vA.lin, jacobianA) - dot(vB.lin, jacobianA);
        float csi = -a[0] * softnessImpulseScale - effectiveMass * (csv - biasVelocity);
        clampPositive(a[0], csi);
        CenterDistanceConstraint::applyImpulse(jacobianA, iA.invMass, iB.invMass, csi, vA, vB);
    }
};

// ---- Three- and four-body constraints (ThreeBodyTypeProcessor.cs / FourBodyTypeProcessor.cs): the functions take the bodies as arrays. ----
// AreaConstraint — AreaConstraint.cs:69-202. Bodies A, B, C are the triangle's vertices. Prestep: TargetScaledArea, spring{2}. Impulse: scalar.
struct AreaConstraint {
    static constexpr int bodies = 3, prestepFloats = 3, impulseFloats = 1, typeId = kAreaConstraint;
    static constexpr bool incremental = false;
    static constexpr int access = kAccessOnlyLinear;  // every body, warm start and solve (:199)
    BD_FN void applyImpulse(const float* inverseMass, V3 negatedJacobianA, V3 jacobianB, V3 jacobianC, float impulse, BodyVel* v) {  // :78-88
        V3 negativeVelocityChangeA = scale(negatedJacobianA, inverseMass[0] * impulse);
        V3 velocityChangeB = scale(jacobianB, inverseMass[1] * impulse);
        V3 velocityChangeC = scale(jacobianC, inverseMass[2] * impulse);
        v[0].lin = sub(v[0].lin, negativeVelocityChangeA);
        v[1].lin = add(v[1].lin, velocityChangeB);
        v[2].lin = add(v[2].lin, velocityChangeC);
    }
    BD_FN void computeJacobian(const V3* pos, float& normalLength, V3& negatedJacobianA, V3& jacobianB, V3& jacobianC,
                               float& contributionA, float& contributionB, float& contributionC, float& inverseJacobianLength) {  // :91-139
        V3 ab = sub(pos[1], pos[0]);
        V3 ac = sub(pos[2], pos[0]);
        V3 abxac = cross(ab, ac);
        normalLength = length(abxac);
        // parallel / antiparallel edges: no normal
        V3 normal = scale(abxac, sel(normalLength > 1e-10f, 1.0f / normalLength, 0.0f));
        jacobianB = cross(ac, normal);
        jacobianC = cross(normal, ab);
        negatedJacobianA = add(jacobianB, jacobianC);
        contributionA = dot(negatedJacobianA, negatedJacobianA);
        contributionB = dot(jacobianB, jacobianB);
        contributionC = dot(jacobianC, jacobianC);
        float jacobianLengthSquared = contributionA + contributionB + contributionC;
        jacobianLengthSquared = vmax(1e-14f, jacobianLengthSquared);
        inverseJacobianLength = fastReciprocalSquareRoot(jacobianLengthSquared);
    }
    template <class G> BD_FN void warmStartN(const V3* pos, const float* inverseMass, float*, float* a, BodyVel* v, G&& gate) {  // :141-151
        float normalLength, cA, cB, cC, inverseJacobianLength; V3 negatedJacobianA, jacobianB, jacobianC;
        computeJacobian(pos, normalLength, negatedJacobianA, jacobianB, jacobianC, cA, cB, cC, inverseJacobianLength);
        float impulse = inverseJacobianLength * a[0];
        BD_GATE_N(v, negatedJacobianA, jacobianB, jacobianC, impulse);
        applyImpulse(inverseMass, negatedJacobianA, jacobianB, jacobianC, impulse, v);
    }
    template <class G> BD_FN void solveN(const V3* pos, const float* inverseMass, float dt, float, float* p, float* a, BodyVel* v, G&& gate) {  // :153-192
        float normalLength, contributionA, contributionB, contributionC, inverseJacobianLength; V3 negatedJacobianA, jacobianB, jacobianC;
        computeJacobian(pos, normalLength, negatedJacobianA, jacobianB, jacobianC, contributionA, contributionB, contributionC, inverseJacobianLength);
        float inverseJacobianLengthSquared = inverseJacobianLength * inverseJacobianLength;
        float inverseEffectiveMass = vmax(1e-14f, inverseJacobianLengthSquared * (contributionA * inverseMass[0] + contributionB * inverseMass[1] + contributionC * inverseMass[2]));
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[1], p[2], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / inverseEffectiveMass;
        float biasVelocity = (p[0] - normalLength) * inverseJacobianLength * posErrToVel;
        BD_GATE_N(v, negatedJacobianA, jacobianB, jacobianC, inverseJacobianLength, effectiveMass, biasVelocity, softnessImpulseScale);
        float negatedVelocityContributionA = dot(negatedJacobianA, v[0].lin);
        float velocityContributionB = dot(jacobianB, v[1].lin);
        float velocityContributionC = dot(jacobianC, v[2].lin);
        float csv = inverseJacobianLength * (velocityContributionB + velocityContributionC - negatedVelocityContributionA);
        float csi = (biasVelocity - csv) * effectiveMass - a[0] * softnessImpulseScale;
        a[0] = a[0] + csi;
        applyImpulse(inverseMass, negatedJacobianA, jacobianB, jacobianC, inverseJacobianLength * csi, v);
    }
};

// VolumeConstraint — VolumeConstraint.cs:69-191. Bodies A, B, C, D are the tetrahedron's vertices. Prestep: TargetScaledVolume, spring{2}. Impulse: scalar.
struct VolumeConstraint {
    static constexpr int bodies = 4, prestepFloats = 3, impulseFloats = 1, typeId = kVolumeConstraint;
    static constexpr bool incremental = false;
    static constexpr int access = kAccessOnlyLinear;  // :188
    BD_FN void applyImpulse(const float* inverseMass, V3 negatedJacobianA, V3 jacobianB, V3 jacobianC, V3 jacobianD, float impulse, BodyVel* v) {  // :78-91
        V3 negativeVelocityChangeA = scale(negatedJacobianA, inverseMass[0] * impulse);
        V3 velocityChangeB = scale(jacobianB, inverseMass[1] * impulse);
        V3 velocityChangeC = scale(jacobianC, inverseMass[2] * impulse);
        V3 velocityChangeD = scale(jacobianD, inverseMass[3] * impulse);
        v[0].lin = sub(v[0].lin, negativeVelocityChangeA);
        v[1].lin = add(v[1].lin, velocityChangeB);
        v[2].lin = add(v[2].lin, velocityChangeC);
        v[3].lin = add(v[3].lin, velocityChangeD);
    }
    BD_FN void computeJacobian(const V3* pos, V3& ad, V3& negatedJA, V3& jacobianB, V3& jacobianC, V3& jacobianD,
                               float& contributionA, float& contributionB, float& contributionC, float& contributionD, float& inverseJacobianLength) {  // :94-123
        V3 ab = sub(pos[1], pos[0]);
        V3 ac = sub(pos[2], pos[0]);
        ad = sub(pos[3], pos[0]);
        jacobianB = cross(ac, ad);
        jacobianC = cross(ad, ab);
        jacobianD = cross(ab, ac);
        negatedJA = add(jacobianB, jacobianC);
        negatedJA = add(jacobianD, negatedJA);
        contributionA = dot(negatedJA, negatedJA);
        contributionB = dot(jacobianB, jacobianB);
        contributionC = dot(jacobianC, jacobianC);
        contributionD = dot(jacobianD, jacobianD);
        float jacobianLengthSquared = contributionA + contributionB + contributionC + contributionD;
        jacobianLengthSquared = vmax(1e-14f, jacobianLengthSquared);
        inverseJacobianLength = fastReciprocalSquareRoot(jacobianLengthSquared);
    }
    template <class G> BD_FN void warmStartN(const V3* pos, const float* inverseMass, float*, float* a, BodyVel* v, G&& gate) {  // :125-129
        V3 ad, negatedJA, jacobianB, jacobianC, jacobianD; float cA, cB, cC, cD, inverseJacobianLength;
        computeJacobian(pos, ad, negatedJA, jacobianB, jacobianC, jacobianD, cA, cB, cC, cD, inverseJacobianLength);
        float impulse = inverseJacobianLength * a[0];
        BD_GATE_N(v, negatedJA, jacobianB, jacobianC, jacobianD, impulse);
        applyImpulse(inverseMass, negatedJA, jacobianB, jacobianC, jacobianD, impulse, v);
    }
    template <class G> BD_FN void solveN(const V3* pos, const float* inverseMass, float dt, float, float* p, float* a, BodyVel* v, G&& gate) {  // :131-157
        V3 ad, negatedJA, jacobianB, jacobianC, jacobianD; float contributionA, contributionB, contributionC, contributionD, inverseJacobianLength;
        computeJacobian(pos, ad, negatedJA, jacobianB, jacobianC, jacobianD, contributionA, contributionB, contributionC, contributionD, inverseJacobianLength);
        float inverseJacobianLengthSquared = inverseJacobianLength * inverseJacobianLength;
        float inverseEffectiveMass = vmax(1e-14f, inverseJacobianLengthSquared * (contributionA * inverseMass[0] + contributionB * inverseMass[1] + contributionC * inverseMass[2] + contributionD * inverseMass[3]));
        float posErrToVel, effMassCFMScale, softnessImpulseScale;
        computeSpringiness(p[1], p[2], dt, posErrToVel, effMassCFMScale, softnessImpulseScale);
        float effectiveMass = effMassCFMScale / inverseEffectiveMass;
        float volume = dot(jacobianD, ad);
        float biasVelocity = (p[0] - volume) * inverseJacobianLength * posErrToVel;
        BD_GATE_N(v, negatedJA, jacobianB, jacobianC, jacobianD, inverseJacobianLength, effectiveMass, biasVelocity, softnessImpulseScale);
        float negatedVelocityContributionA = dot(negatedJA, v[0].lin);
        float velocityContributionB = dot(jacobianB, v[1].lin);
        float velocityContributionC = dot(jacobianC, v[2].lin);
        float velocityContributionD = dot(jacobianD, v[3].lin);
        float csv = inverseJacobianLength * (velocityContributionB + velocityContributionC + velocityContributionD - negatedVelocityContributionA);
        float csi = (biasVelocity - csv) * effectiveMass - a[0] * softnessImpulseScale;
        a[0] = a[0] + csi;
        applyImpulse(inverseMass, negatedJA, jacobianB, jacobianC, jacobianD, inverseJacobianLength * csi, v);
    }
};

// The non-contact types for the dispatch switches, X(type id, struct): SURVEY 8(a)'s rows a8-a13, and the 8(f) widening.
#define BD_HOT_JOINT_TYPES(X)                                                                                                       \
    X(kBallSocket, BallSocket) X(kAngularHinge, AngularHinge) X(kSwingLimit, SwingLimit) X(kTwistServo, TwistServo)                 \
    X(kTwistLimit, TwistLimit) X(kAngularMotor, AngularMotor) X(kSwivelHinge, SwivelHinge) X(kHinge, Hinge)
#define BD_WIDENED_JOINT_TYPES(X)                                                                                                   \
    X(kWeld, Weld) X(kAngularSwivelHinge, AngularSwivelHinge) X(kTwistMotor, TwistMotor) X(kAngularServo, AngularServo)             \
    X(kDistanceServo, DistanceServo) X(kDistanceLimit, DistanceLimit) X(kAngularAxisMotor, AngularAxisMotor)                        \
    X(kOneBodyAngularServo, OneBodyAngularServo) X(kOneBodyAngularMotor, OneBodyAngularMotor) X(kOneBodyLinearServo, OneBodyLinearServo) \
    X(kOneBodyLinearMotor, OneBodyLinearMotor) X(kBallSocketMotor, BallSocketMotor) X(kBallSocketServo, BallSocketServo) \
    X(kPointOnLineServo, PointOnLineServo) X(kLinearAxisServo, LinearAxisServo) X(kLinearAxisMotor, LinearAxisMotor) X(kLinearAxisLimit, LinearAxisLimit) \
    X(kAngularAxisGearMotor, AngularAxisGearMotor) X(kCenterDistanceConstraint, CenterDistanceConstraint) X(kCenterDistanceLimit, CenterDistanceLimit)
#define BD_JOINT_TYPES(X) BD_HOT_JOINT_TYPES(X) BD_WIDENED_JOINT_TYPES(X)
// constraints over three and four bodies take body arrays (warmStartN / solveN)
#define BD_MANY_BODY_TYPES(X) X(kAreaConstraint, AreaConstraint) X(kVolumeConstraint, VolumeConstraint)
using NC2O = NonconvexContact<2, false>; using NC3O = NonconvexContact<3, false>; using NC4O = NonconvexContact<4, false>;
using NC2T = NonconvexContact<2, true>; using NC3T = NonconvexContact<3, true>; using NC4T = NonconvexContact<4, true>;
#define BD_NONCONVEX_CONTACT_TYPES(X)                                                                                              \
    X(kContact2NonconvexOneBody, NC2O) X(kContact3NonconvexOneBody, NC3O) X(kContact4NonconvexOneBody, NC4O)                       \
    X(kContact2Nonconvex, NC2T) X(kContact3Nonconvex, NC3T) X(kContact4Nonconvex, NC4T)

}  // namespace bd
