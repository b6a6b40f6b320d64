// Synthetic scene recipes for the solver hot path, mined from the reference's demos/benchmarks (inputs only — collision
// detection is out of scope, so contact manifolds are synthesised):
//   pyramid       Demos/Demos/PyramidDemo.cs:26-47 (unit boxes, rows of N..1), material Demos/DemoCallbacks.cs:117-131
//   pile          DemoBenchmarks/ShapePileBenchmark.cs:98-229-style dense pile as a jittered lattice contact graph
//   ragdoll_tube  DemoBenchmarks/RagdollTubeBenchmark.cs:185-592 (16 bodies + 58 joints per ragdoll, kinematic tube, static ground)
// Constraints are added through Solver::Add in the reference's order, so the greedy batch colouring matches what the C# would build.
#include <cmath>
#include <cstring>
#include <random>

#include "bepu_host.h"

namespace bepu {
namespace {

constexpr float kPi = 3.141592653589793239f;
constexpr float kFloatMax = 3.402823466e+38f;

struct Rng {
    std::mt19937 gen;
    explicit Rng(uint32_t seed) : gen(seed) {}
    float uniform(float lo, float hi) { return lo + (hi - lo) * (float)((gen() >> 8) * (1.0 / 16777216.0)); }
    Vector3 unit() {
        for (;;) {
            Vector3 v{uniform(-1, 1), uniform(-1, 1), uniform(-1, 1)};
            float l2 = v.X * v.X + v.Y * v.Y + v.Z * v.Z;
            if (l2 > 1e-3f && l2 <= 1) { float s = 1 / std::sqrt(l2); return {v.X * s, v.Y * s, v.Z * s}; }
        }
    }
};

Vector3 operator+(Vector3 a, Vector3 b) { return {a.X + b.X, a.Y + b.Y, a.Z + b.Z}; }
Vector3 operator-(Vector3 a, Vector3 b) { return {a.X - b.X, a.Y - b.Y, a.Z - b.Z}; }
Vector3 operator*(Vector3 a, float s) { return {a.X * s, a.Y * s, a.Z * s}; }
Vector3 cross(Vector3 a, Vector3 b) { return {a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X}; }
float length(Vector3 a) { return std::sqrt(a.X * a.X + a.Y * a.Y + a.Z * a.Z); }
Vector3 normalize(Vector3 a) { float l = length(a); return a * (1 / l); }

Quaternion conjugate(Quaternion q) { return {-q.X, -q.Y, -q.Z, q.W}; }
Quaternion concatenate(Quaternion a, Quaternion b) {  // QuaternionEx.ConcatenateWithoutOverlap: apply a, then b
    return {a.W * b.X + a.X * b.W + a.Z * b.Y - a.Y * b.Z, a.W * b.Y + a.Y * b.W + a.X * b.Z - a.Z * b.X,
            a.W * b.Z + a.Z * b.W + a.Y * b.X - a.X * b.Y, a.W * b.W - a.X * b.X - a.Y * b.Y - a.Z * b.Z};
}
Vector3 transform(Vector3 v, Quaternion r) {
    float x2 = r.X + r.X, y2 = r.Y + r.Y, z2 = r.Z + r.Z;
    float xx2 = r.X * x2, xy2 = r.X * y2, xz2 = r.X * z2, yy2 = r.Y * y2, yz2 = r.Y * z2, zz2 = r.Z * z2;
    float wx2 = r.W * x2, wy2 = r.W * y2, wz2 = r.W * z2;
    return {v.X * (1 - yy2 - zz2) + v.Y * (xy2 - wz2) + v.Z * (xz2 + wy2), v.X * (xy2 + wz2) + v.Y * (1 - xx2 - zz2) + v.Z * (yz2 - wx2),
            v.X * (xz2 - wy2) + v.Y * (yz2 + wx2) + v.Z * (1 - xx2 - yy2)};
}
Quaternion fromAxisAngle(Vector3 axis, float angle) {
    float h = angle * 0.5f, s = std::sin(h);
    return {axis.X * s, axis.Y * s, axis.Z * s, std::cos(h)};
}
Quaternion fromBasis(Vector3 X, Vector3 Y, Vector3 Z) {  // rotation taking unit axes to X, Y, Z
    float t = X.X + Y.Y + Z.Z;
    Quaternion q;
    if (t > 0) {
        float s = std::sqrt(t + 1) * 2;
        q = {(Y.Z - Z.Y) / s, (Z.X - X.Z) / s, (X.Y - Y.X) / s, 0.25f * s};
    } else if (X.X > Y.Y && X.X > Z.Z) {
        float s = std::sqrt(1 + X.X - Y.Y - Z.Z) * 2;
        q = {0.25f * s, (Y.X + X.Y) / s, (Z.X + X.Z) / s, (Y.Z - Z.Y) / s};
    } else if (Y.Y > Z.Z) {
        float s = std::sqrt(1 + Y.Y - X.X - Z.Z) * 2;
        q = {(Y.X + X.Y) / s, 0.25f * s, (Z.Y + Y.Z) / s, (Z.X - X.Z) / s};
    } else {
        float s = std::sqrt(1 + Z.Z - X.X - Y.Y) * 2;
        q = {(Z.X + X.Z) / s, (Z.Y + Y.Z) / s, 0.25f * s, (X.Y - Y.X) / s};
    }
    float n = 1 / std::sqrt(q.X * q.X + q.Y * q.Y + q.Z * q.Z + q.W * q.W);
    return {q.X * n, q.Y * n, q.Z * n, q.W * n};
}
Quaternion createBasis(Vector3 z, Vector3 x) {  // RagdollTubeBenchmark.cs:166-175
    Vector3 Z = normalize(z);
    Vector3 Y = normalize(cross(Z, x));
    Vector3 X = cross(Y, Z);
    return fromBasis(X, Y, Z);
}

BodyInertia capsuleInertia(float radius, float lengthFull, float mass) {  // Collidables/Capsule.cs:159-180
    BodyInertia in;
    float halfLength = lengthFull * 0.5f;
    in.InverseMass = 1 / mass;
    float r2 = radius * radius, h2 = halfLength * halfLength;
    float cylinderVolume = 2 * halfLength * r2 * kPi;
    float sphereVolume = (4.f / 3.f) * r2 * radius * kPi;
    float inverseTotal = 1 / (cylinderVolume + sphereVolume);
    cylinderVolume *= inverseTotal;
    sphereVolume *= inverseTotal;
    in.InverseInertiaTensor.XX = in.InverseMass / (cylinderVolume * ((3.f / 12.f) * r2 + (4.f / 12.f) * h2) + sphereVolume * ((2.f / 5.f) * r2 + (6.f / 8.f) * radius * halfLength + h2));
    in.InverseInertiaTensor.YY = in.InverseMass / (cylinderVolume * (1.f / 2.f) * r2 + sphereVolume * (2.f / 5.f) * r2);
    in.InverseInertiaTensor.ZZ = in.InverseInertiaTensor.XX;
    return in;
}
BodyInertia boxInertia(float w, float h, float l, float mass) {  // Collidables/Box.cs:149-163
    BodyInertia in;
    in.InverseMass = 1 / mass;
    float x2 = 0.25f * w * w, y2 = 0.25f * h * h, z2 = 0.25f * l * l;
    in.InverseInertiaTensor.XX = in.InverseMass * 3 / (y2 + z2);
    in.InverseInertiaTensor.YY = in.InverseMass * 3 / (x2 + z2);
    in.InverseInertiaTensor.ZZ = in.InverseMass * 3 / (x2 + y2);
    return in;
}
BodyInertia sphereInertia(float r, float mass) {  // Collidables/Sphere.cs:95-106
    BodyInertia in;
    in.InverseMass = 1 / mass;
    in.InverseInertiaTensor.XX = in.InverseInertiaTensor.YY = in.InverseInertiaTensor.ZZ = in.InverseMass / ((2.f / 5.f) * r * r);
    return in;
}

// ---- constraint description helpers: prestep lanes in the reference struct's field order ----
struct Material { float friction; SpringSettings spring; float maxRecovery; };

struct ContactPoint { Vector3 offsetA; float depth; };
int addContact(Solver& s, int n, int32_t a, int32_t b /* -1 => one body */, const ContactPoint* pts, Vector3 offsetB, Vector3 normal, const Material& m) {
    float lane[26];
    int o = 0;
    for (int i = 0; i < n; ++i) { lane[o++] = pts[i].offsetA.X; lane[o++] = pts[i].offsetA.Y; lane[o++] = pts[i].offsetA.Z; lane[o++] = pts[i].depth; }
    if (b >= 0) { lane[o++] = offsetB.X; lane[o++] = offsetB.Y; lane[o++] = offsetB.Z; }
    lane[o++] = normal.X; lane[o++] = normal.Y; lane[o++] = normal.Z;
    lane[o++] = m.friction; lane[o++] = m.spring.AngularFrequency; lane[o++] = m.spring.TwiceDampingRatio; lane[o++] = m.maxRecovery;
    int32_t hs[2] = {a, b};
    return s.Add(hs, b >= 0 ? 2 : 1, b >= 0 ? 3 + n : n - 1, lane);
}
int addBallSocket(Solver& s, int32_t a, int32_t b, Vector3 la, Vector3 lb, SpringSettings sp) {
    float lane[8] = {la.X, la.Y, la.Z, lb.X, lb.Y, lb.Z, sp.AngularFrequency, sp.TwiceDampingRatio};
    int32_t hs[2] = {a, b};
    return s.Add(hs, 2, 22, lane);
}
int addSwingLimit(Solver& s, int32_t a, int32_t b, Vector3 axA, Vector3 axB, float maximumSwingAngle, SpringSettings sp) {
    float lane[9] = {axA.X, axA.Y, axA.Z, axB.X, axB.Y, axB.Z, (float)std::cos((double)maximumSwingAngle), sp.AngularFrequency, sp.TwiceDampingRatio};  // SwingLimit.cs:36
    int32_t hs[2] = {a, b};
    return s.Add(hs, 2, 25, lane);
}
int addTwistLimit(Solver& s, int32_t a, int32_t b, Quaternion ba, Quaternion bb, float mn, float mx, SpringSettings sp) {
    float lane[12] = {ba.X, ba.Y, ba.Z, ba.W, bb.X, bb.Y, bb.Z, bb.W, mn, mx, sp.AngularFrequency, sp.TwiceDampingRatio};
    int32_t hs[2] = {a, b};
    return s.Add(hs, 2, 27, lane);
}
int addTwistServo(Solver& s, int32_t a, int32_t b, Quaternion ba, Quaternion bb, float target, SpringSettings sp, ServoSettings sv) {
    float lane[14] = {ba.X, ba.Y, ba.Z, ba.W, bb.X, bb.Y, bb.Z, bb.W, target, sp.AngularFrequency, sp.TwiceDampingRatio, sv.MaximumSpeed, sv.BaseSpeed, sv.MaximumForce};
    int32_t hs[2] = {a, b};
    return s.Add(hs, 2, 26, lane);
}
int addAngularMotor(Solver& s, int32_t a, int32_t b) {  // BuildAngularMotor, RagdollTubeBenchmark.cs:177-183
    MotorSettings ms(kFloatMax, 0.01f);
    float lane[5] = {0, 0, 0, ms.MaximumForce, ms.Damping};
    int32_t hs[2] = {a, b};
    return s.Add(hs, 2, 30, lane);
}
int addSwivelHinge(Solver& s, int32_t a, int32_t b, Vector3 oa, Vector3 swivelA, Vector3 ob, Vector3 hingeB, SpringSettings sp) {
    float lane[14] = {oa.X, oa.Y, oa.Z, swivelA.X, swivelA.Y, swivelA.Z, ob.X, ob.Y, ob.Z, hingeB.X, hingeB.Y, hingeB.Z, sp.AngularFrequency, sp.TwiceDampingRatio};
    int32_t hs[2] = {a, b};
    return s.Add(hs, 2, 46, lane);
}
int addHinge(Solver& s, int32_t a, int32_t b, Vector3 oa, Vector3 axA, Vector3 ob, Vector3 axB, SpringSettings sp) {
    float lane[14] = {oa.X, oa.Y, oa.Z, axA.X, axA.Y, axA.Z, ob.X, ob.Y, ob.Z, axB.X, axB.Y, axB.Z, sp.AngularFrequency, sp.TwiceDampingRatio};
    int32_t hs[2] = {a, b};
    return s.Add(hs, 2, 47, lane);
}

// ---- ragdoll (RagdollTubeBenchmark.cs:185-520) ----
struct Ragdoll {
    int32_t hips, abdomen, chest, head;
    int32_t upperArm[2], lowerArm[2], hand[2];  // [0] right (sign +1), [1] left
    int32_t upperLeg[2], lowerLeg[2], foot[2];  // [0] right (x = -0.17), [1] left
};

void capsuleForLineSegment(Vector3 start, Vector3 end, float& lengthOut, Vector3& position, Quaternion& orientation) {  // :152-164
    position = (start + end) * 0.5f;
    Vector3 offset = end - start;
    lengthOut = length(offset);
    Vector3 c = cross(offset * (1 / lengthOut), Vector3{0, 1, 0});
    float cl = length(c);
    orientation = cl > 1e-8f ? fromAxisAngle(c * (1 / cl), (float)std::asin((double)cl)) : Quaternion{};
}

int32_t addBody(Simulation& sim, const BodyInertia& inertia, Vector3 localPosition, Quaternion localOrientation, const RigidPose& ragdollPose, Rng& rng, float jitter) {
    RigidPose world;  // GetWorldPose :145-151
    world.Position = transform(localPosition, ragdollPose.Orientation) + ragdollPose.Position;
    world.Orientation = concatenate(localOrientation, ragdollPose.Orientation);
    BodyDescription d = BodyDescription::CreateDynamic(world, inertia);
    d.Velocity.Linear = {rng.uniform(-jitter, jitter), rng.uniform(-jitter, jitter), rng.uniform(-jitter, jitter)};
    d.Velocity.Angular = {rng.uniform(-jitter, jitter), rng.uniform(-jitter, jitter), rng.uniform(-jitter, jitter)};
    return sim.bodies.Add(d);
}

void addArm(Simulation& sim, Ragdoll& r, int side, float sign, Vector3 localShoulder, RigidPose localChestPose, const RigidPose& ragdollPose, SpringSettings sp, Rng& rng, float jitter) {  // :185-289
    Solver& s = sim.solver;
    Vector3 localElbow = localShoulder + Vector3{sign * 0.45f, 0, 0};
    Vector3 localWrist = localElbow + Vector3{sign * 0.45f, 0, 0};
    Vector3 handPosition = localWrist + Vector3{sign * 0.1f, 0, 0};
    float len; Vector3 upperArmPosition, lowerArmPosition; Quaternion upperArmOrientation, lowerArmOrientation;
    capsuleForLineSegment(localShoulder, localElbow, len, upperArmPosition, upperArmOrientation);
    r.upperArm[side] = addBody(sim, capsuleInertia(0.1f, len, 5), upperArmPosition, upperArmOrientation, ragdollPose, rng, jitter);
    capsuleForLineSegment(localElbow, localWrist, len, lowerArmPosition, lowerArmOrientation);
    r.lowerArm[side] = addBody(sim, capsuleInertia(0.09f, len, 5), lowerArmPosition, lowerArmOrientation, ragdollPose, rng, jitter);
    r.hand[side] = addBody(sim, boxInertia(0.2f, 0.1f, 0.2f, 2), handPosition, Quaternion{}, ragdollPose, rng, jitter);
    Quaternion chestInv = conjugate(localChestPose.Orientation), upperInv = conjugate(upperArmOrientation), lowerInv = conjugate(lowerArmOrientation);
    // Chest - upper arm
    addBallSocket(s, r.chest, r.upperArm[side], transform(localShoulder - localChestPose.Position, chestInv), transform(localShoulder - upperArmPosition, upperInv), sp);
    addSwingLimit(s, r.chest, r.upperArm[side], transform(normalize(Vector3{sign, 0, 1}), chestInv), transform(Vector3{sign, 0, 0}, upperInv), kPi * 0.56f, sp);
    addTwistLimit(s, r.chest, r.upperArm[side], concatenate(createBasis({1, 0, 0}, {0, 0, -1}), chestInv), concatenate(createBasis({1, 0, 0}, {0, 0, -1}), upperInv), kPi * -0.55f, kPi * 0.55f, sp);
    addAngularMotor(s, r.chest, r.upperArm[side]);
    // Upper arm - lower arm
    addSwivelHinge(s, r.upperArm[side], r.lowerArm[side], transform(localElbow - upperArmPosition, upperInv), {1, 0, 0}, transform(localElbow - lowerArmPosition, lowerInv), {0, 1, 0}, sp);
    addSwingLimit(s, r.upperArm[side], r.lowerArm[side], {0, 1, 0}, {sign, 0, 0}, kPi * 0.5f, sp);
    addTwistLimit(s, r.upperArm[side], r.lowerArm[side], concatenate(createBasis({1, 0, 0}, {0, 0, -1}), upperInv), concatenate(createBasis({1, 0, 0}, {0, 0, -1}), lowerInv), kPi * -0.55f, kPi * 0.55f, sp);
    addAngularMotor(s, r.upperArm[side], r.lowerArm[side]);
    // Lower arm - hand
    addBallSocket(s, r.lowerArm[side], r.hand[side], transform(localWrist - lowerArmPosition, lowerInv), localWrist - handPosition, sp);
    addSwingLimit(s, r.lowerArm[side], r.hand[side], transform(Vector3{sign, 0, 0}, lowerInv), {sign, 0, 0}, kPi * 0.5f, sp);
    addTwistServo(s, r.lowerArm[side], r.hand[side], concatenate(createBasis({1, 0, 0}, {0, 0, 1}), lowerInv), createBasis({1, 0, 0}, {0, 0, 1}), 0, sp, ServoSettings{kFloatMax, 0, kFloatMax});
    addAngularMotor(s, r.lowerArm[side], r.hand[side]);
}

void addLeg(Simulation& sim, Ragdoll& r, int side, Vector3 localHip, RigidPose localHipsPose, const RigidPose& ragdollPose, SpringSettings sp, Rng& rng, float jitter) {  // :291-388
    Solver& s = sim.solver;
    Vector3 localKnee = localHip - Vector3{0, 0.5f, 0};
    Vector3 localAnkle = localKnee - Vector3{0, 0.5f, 0};
    Vector3 localFoot = localAnkle + Vector3{0, -0.075f, 0.05f};
    float len; Vector3 upperLegPosition, lowerLegPosition; Quaternion upperLegOrientation, lowerLegOrientation;
    capsuleForLineSegment(localHip, localKnee, len, upperLegPosition, upperLegOrientation);
    r.upperLeg[side] = addBody(sim, capsuleInertia(0.12f, len, 5), upperLegPosition, upperLegOrientation, ragdollPose, rng, jitter);
    capsuleForLineSegment(localKnee, localAnkle, len, lowerLegPosition, lowerLegOrientation);
    r.lowerLeg[side] = addBody(sim, capsuleInertia(0.11f, len, 5), lowerLegPosition, lowerLegOrientation, ragdollPose, rng, jitter);
    r.foot[side] = addBody(sim, boxInertia(0.2f, 0.15f, 0.3f, 2), localFoot, Quaternion{}, ragdollPose, rng, jitter);
    Quaternion hipsInv = conjugate(localHipsPose.Orientation), upperInv = conjugate(upperLegOrientation), lowerInv = conjugate(lowerLegOrientation);
    float signX = localHip.X > 0 ? 1.f : (localHip.X < 0 ? -1.f : 0.f);
    // Hips - upper leg
    addBallSocket(s, r.hips, r.upperLeg[side], transform(localHip - localHipsPose.Position, hipsInv), transform(localHip - upperLegPosition, upperInv), sp);
    addSwingLimit(s, r.hips, r.upperLeg[side], transform(normalize(Vector3{signX, -1, 0}), hipsInv), transform(Vector3{0, -1, 0}, upperInv), kPi * 0.5f, sp);
    addTwistLimit(s, r.hips, r.upperLeg[side], concatenate(createBasis({0, -1, 0}, {0, 0, 1}), hipsInv), concatenate(createBasis({0, -1, 0}, {0, 0, 1}), upperInv),
                  localHip.X < 0 ? kPi * -0.05f : kPi * -0.55f, localHip.X < 0 ? kPi * 0.55f : kPi * 0.05f, sp);
    addAngularMotor(s, r.hips, r.upperLeg[side]);
    // Upper leg - lower leg
    addHinge(s, r.upperLeg[side], r.lowerLeg[side], transform(localKnee - upperLegPosition, upperInv), transform(Vector3{1, 0, 0}, upperInv), transform(localKnee - lowerLegPosition, lowerInv),
             transform(Vector3{1, 0, 0}, lowerInv), sp);
    addSwingLimit(s, r.upperLeg[side], r.lowerLeg[side], transform(Vector3{0, 0, 1}, upperInv), transform(Vector3{0, 1, 0}, lowerInv), kPi * 0.5f, sp);
    addAngularMotor(s, r.upperLeg[side], r.lowerLeg[side]);
    // Lower leg - foot
    addBallSocket(s, r.lowerLeg[side], r.foot[side], transform(localAnkle - lowerLegPosition, lowerInv), localAnkle - localFoot, sp);
    addSwingLimit(s, r.lowerLeg[side], r.foot[side], transform(Vector3{0, 1, 0}, lowerInv), {0, 1, 0}, 1, sp);
    addTwistServo(s, r.lowerLeg[side], r.foot[side], concatenate(createBasis({0, 1, 0}, {0, 0, 1}), lowerInv), createBasis({0, 1, 0}, {0, 0, 1}), 0, sp, ServoSettings{kFloatMax, 0, kFloatMax});
    addAngularMotor(s, r.lowerLeg[side], r.foot[side]);
}

Ragdoll addRagdoll(Simulation& sim, Vector3 position, Quaternion orientation, Rng& rng, float jitter) {  // :390-520
    Solver& s = sim.solver;
    Ragdoll r;
    RigidPose ragdollPose{position, orientation};
    Quaternion horizontal = fromAxisAngle({0, 0, 1}, kPi * 0.5f);
    RigidPose hipsPose{{0, 1.1f, 0}, horizontal}, abdomenPose{{0, 1.3f, 0}, horizontal}, chestPose{{0, 1.6f, 0}, horizontal}, headPose{{0, 2.05f, 0}, Quaternion{}};
    r.hips = addBody(sim, capsuleInertia(0.17f, 0.25f, 8), hipsPose.Position, hipsPose.Orientation, ragdollPose, rng, jitter);
    r.abdomen = addBody(sim, capsuleInertia(0.17f, 0.22f, 7), abdomenPose.Position, abdomenPose.Orientation, ragdollPose, rng, jitter);
    r.chest = addBody(sim, capsuleInertia(0.21f, 0.3f, 10), chestPose.Position, chestPose.Orientation, ragdollPose, rng, jitter);
    r.head = addBody(sim, sphereInertia(0.2f, 5), headPose.Position, headPose.Orientation, ragdollPose, rng, jitter);
    SpringSettings sp(15.f, 1.f);
    Quaternion hipsInv = conjugate(hipsPose.Orientation), abdomenInv = conjugate(abdomenPose.Orientation), chestInv = conjugate(chestPose.Orientation), headInv = conjugate(headPose.Orientation);
    Vector3 lowerSpine = (hipsPose.Position + abdomenPose.Position) * 0.5f;
    addBallSocket(s, r.hips, r.abdomen, transform(lowerSpine - hipsPose.Position, hipsInv), transform(lowerSpine - abdomenPose.Position, abdomenInv), sp);
    addSwingLimit(s, r.hips, r.abdomen, transform(Vector3{0, 1, 0}, hipsInv), transform(Vector3{0, 1, 0}, abdomenInv), kPi * 0.27f, sp);
    addTwistLimit(s, r.hips, r.abdomen, concatenate(createBasis({0, 1, 0}, {1, 0, 0}), hipsInv), concatenate(createBasis({0, 1, 0}, {1, 0, 0}), abdomenInv), kPi * -0.2f, kPi * 0.2f, sp);
    addAngularMotor(s, r.hips, r.abdomen);
    Vector3 upperSpine = (abdomenPose.Position + chestPose.Position) * 0.5f;
    addBallSocket(s, r.abdomen, r.chest, transform(upperSpine - abdomenPose.Position, abdomenInv), transform(upperSpine - chestPose.Position, chestInv), sp);
    addSwingLimit(s, r.abdomen, r.chest, transform(Vector3{0, 1, 0}, abdomenInv), transform(Vector3{0, 1, 0}, chestInv), kPi * 0.27f, sp);
    addTwistLimit(s, r.abdomen, r.chest, concatenate(createBasis({0, 1, 0}, {1, 0, 0}), abdomenInv), concatenate(createBasis({0, 1, 0}, {1, 0, 0}), chestInv), kPi * -0.2f, kPi * 0.2f, sp);
    addAngularMotor(s, r.abdomen, r.chest);
    Vector3 neck = (headPose.Position + chestPose.Position) * 0.5f;
    addBallSocket(s, r.chest, r.head, transform(neck - chestPose.Position, chestInv), neck - headPose.Position, sp);
    addSwingLimit(s, r.chest, r.head, transform(Vector3{0, 1, 0}, chestInv), {0, 1, 0}, kPi * 0.5f * 0.9f, sp);
    addTwistLimit(s, r.chest, r.head, concatenate(createBasis({0, 1, 0}, {1, 0, 0}), chestInv), concatenate(createBasis({0, 1, 0}, {1, 0, 0}), headInv), kPi * -0.5f, kPi * 0.5f, sp);
    addAngularMotor(s, r.chest, r.head);
    addArm(sim, r, 0, 1, chestPose.Position + Vector3{0.4f, 0.1f, 0}, chestPose, ragdollPose, sp, rng, jitter);
    addArm(sim, r, 1, -1, chestPose.Position + Vector3{-0.4f, 0.1f, 0}, chestPose, ragdollPose, sp, rng, jitter);
    addLeg(sim, r, 0, hipsPose.Position + Vector3{-0.17f, -0.2f, 0}, hipsPose, ragdollPose, sp, rng, jitter);
    addLeg(sim, r, 1, hipsPose.Position + Vector3{0.17f, -0.2f, 0}, hipsPose, ragdollPose, sp, rng, jitter);
    return r;
}

Vector3 bodyPosition(const Simulation& sim, int32_t handle) {
    const float* f = sim.bodies.DynamicsState[sim.bodies.HandleToIndex[handle]].f;
    return {f[4], f[5], f[6]};
}

Quaternion bodyOrientation(const Simulation& sim, int32_t handle) {
    const float* f = sim.bodies.DynamicsState[sim.bodies.HandleToIndex[handle]].f;
    return {f[0], f[1], f[2], f[3]};
}

// Synthetic two-body manifold between bodies a and b (what the narrow phase would have emitted).
void addSyntheticPairContact(Simulation& sim, Rng& rng, int n, int32_t a, int32_t b, const Material& m, bool bIsFarKinematic) {
    Vector3 pa = bodyPosition(sim, a), pb = bodyPosition(sim, b);
    Vector3 offsetB = pb - pa;
    Vector3 normal, center;
    if (bIsFarKinematic) {
        normal = rng.unit();
        center = normal * -0.15f;
    } else {
        normal = normalize(pa - pb);  // calibrated to point from B to A (PenetrationLimit.cs:31)
        center = offsetB * 0.5f;
    }
    ContactPoint pts[4];
    for (int i = 0; i < n; ++i) {
        pts[i].offsetA = center + Vector3{rng.uniform(-0.05f, 0.05f), rng.uniform(-0.05f, 0.05f), rng.uniform(-0.05f, 0.05f)};
        pts[i].depth = rng.uniform(-0.01f, 0.004f);
    }
    addContact(sim.solver, n, a, b, pts, offsetB, normal, m);
}
void addSyntheticGroundContact(Simulation& sim, Rng& rng, int n, int32_t a, float hx, float hy, float hz, const Material& m) {
    ContactPoint pts[4];
    const float sx[4] = {-1, 1, -1, 1}, sz[4] = {-1, -1, 1, 1};
    for (int i = 0; i < n; ++i) {
        pts[i].offsetA = {sx[i] * hx, -hy, sz[i] * hz};
        pts[i].depth = rng.uniform(-0.005f, 0.01f);
    }
    addContact(sim.solver, n, a, -1, pts, {}, {0, 1, 0}, m);
}

Simulation* buildRagdollTube(int64_t ragdollCount, int64_t withContacts, int64_t lattice, uint32_t seed) {
    PoseIntegratorCallbacks cb;  // DemoPoseIntegratorCallbacks(new Vector3(0, -10, 0)) :534
    Simulation* sim = new Simulation(cb, SolveDescription(1, 4));  // config 3: 4 substeps x 1 iteration
    Rng rng(seed);
    // grid: spacing (1.7, 1.8, 0.5), origin formula :538-541, yaw 0.05*pi :548
    int width = (int)std::ceil(std::cbrt((double)ragdollCount * 2)), length = (width + 1) / 2;
    if (width < 1) width = 1;
    if (length < 1) length = 1;
    int height = (int)((ragdollCount + (int64_t)width * length - 1) / ((int64_t)width * length));
    Vector3 spacing{1.7f, 1.8f, 0.5f};
    Vector3 origin = Vector3{-0.5f * spacing.X * (width - 1), 5.f, -0.5f * spacing.Z * (length - 1)};
    Quaternion yaw = fromAxisAngle({0, 1, 0}, kPi * 0.05f);
    std::vector<Ragdoll> ragdolls;
    ragdolls.reserve(ragdollCount);
    int64_t made = 0;
    for (int i = 0; i < width && made < ragdollCount; ++i)
        for (int j = 0; j < height && made < ragdollCount; ++j)
            for (int k = 0; k < length && made < ragdollCount; ++k, ++made)
                ragdolls.push_back(addRagdoll(*sim, origin + Vector3{spacing.X * i, spacing.Y * j, spacing.Z * k}, yaw, rng, 0.05f));
    // Kinematic tube: BodyDescription.CreateKinematic(tubeCenter, (default, (0,0,.25))) :566
    int32_t tube = sim->bodies.Add(BodyDescription::CreateKinematic(RigidPose{{0, 8, 0}, Quaternion{}}, BodyVelocity{{0, 0, 0}, {0, 0, 0.25f}}));
    if (lattice == 1) {
        // config 5: one connected lattice — chain neighbouring ragdolls hand-to-hand (x) and head-to-foot (y) with ball sockets.
        SpringSettings sp(15.f, 1.f);
        // The anchor of every link is the midpoint between the two bodies' rest positions, expressed in each body's local frame: the lattice
        // starts (almost) at rest instead of being yanked together.
        auto link = [&](int32_t a, int32_t b) {
            Vector3 pa = bodyPosition(*sim, a), pb = bodyPosition(*sim, b);
            Vector3 mid = (pa + pb) * 0.5f;
            addBallSocket(sim->solver, a, b, transform(mid - pa, conjugate(bodyOrientation(*sim, a))), transform(mid - pb, conjugate(bodyOrientation(*sim, b))), sp);
        };
        for (int64_t r = 0; r + 1 < (int64_t)ragdolls.size(); ++r) {
            link(ragdolls[r].hand[0], ragdolls[r + 1].hand[1]);
            int64_t up = r + (int64_t)length * height;
            if (up < (int64_t)ragdolls.size()) link(ragdolls[r].head, ragdolls[up].foot[0]);
        }
    }
    if (withContacts) {
        // What the narrow phase would add on the first frame (after all joints): PairMaterialProperties(2, float.MaxValue, SpringSettings(10, 1)) :534
        Material m{2.f, SpringSettings(10.f, 1.f), kFloatMax};
        for (const Ragdoll& r : ragdolls) {
            addSyntheticGroundContact(*sim, rng, 4, r.foot[0], 0.1f, 0.075f, 0.15f, m);
            addSyntheticGroundContact(*sim, rng, 4, r.foot[1], 0.1f, 0.075f, 0.15f, m);
            addSyntheticPairContact(*sim, rng, 2, r.hips, tube, m, true);
            addSyntheticPairContact(*sim, rng, 1, r.head, tube, m, true);
            addSyntheticPairContact(*sim, rng, 1, r.hand[0], tube, m, true);
            addSyntheticPairContact(*sim, rng, 1, r.hand[1], r.upperLeg[0], m, false);
            addSyntheticPairContact(*sim, rng, 2, r.lowerArm[0], r.abdomen, m, false);
            addSyntheticPairContact(*sim, rng, 3, r.lowerLeg[1], r.lowerLeg[0], m, false);
            addSyntheticPairContact(*sim, rng, 4, r.upperArm[1], r.hips, m, false);
        }
        if (lattice == 2) {
            // "Crowd": what the reference's benchmark really turns into once the ragdolls have dropped into the rotating tube and lie on each other
            // (RagdollTubeBenchmark.cs:536-569): contact manifolds BETWEEN neighbouring ragdolls, added after every ragdoll's own constraints, as the
            // narrow phase would. Neighbours along the three grid axes touch limb to limb, so the ragdolls form one connected island (no workgroup's LDS
            // holds it: the general-topology schedule has to take it).
            const int64_t strideY = length, strideX = (int64_t)length * height;
            for (int64_t r = 0; r < (int64_t)ragdolls.size(); ++r) {
                const Ragdoll& a = ragdolls[r];
                if (r + 1 < (int64_t)ragdolls.size() && (r + 1) % length != 0) {       // next along z (0.5 apart): chest to chest, arm to arm
                    const Ragdoll& b = ragdolls[r + 1];
                    addSyntheticPairContact(*sim, rng, 4, a.chest, b.chest, m, false);
                    addSyntheticPairContact(*sim, rng, 2, a.upperArm[0], b.upperArm[0], m, false);
                    addSyntheticPairContact(*sim, rng, 2, a.upperLeg[1], b.upperLeg[1], m, false);
                }
                if (r + strideY < (int64_t)ragdolls.size() && (r / strideY + 1) % height != 0) {  // the one above (1.8 up): its feet on this one's shoulders
                    const Ragdoll& b = ragdolls[r + strideY];
                    addSyntheticPairContact(*sim, rng, 3, b.foot[0], a.upperArm[0], m, false);
                    addSyntheticPairContact(*sim, rng, 3, b.foot[1], a.upperArm[1], m, false);
                }
                if (r + strideX < (int64_t)ragdolls.size()) {                          // next along x (1.7 apart): hand in hand
                    const Ragdoll& b = ragdolls[r + strideX];
                    addSyntheticPairContact(*sim, rng, 1, a.hand[0], b.hand[1], m, false);
                }
            }
        }
    }
    return sim;
}

Simulation* buildPyramids(int64_t pyramidCount, int64_t rowCount) {  // PyramidDemo.cs:26-47
    PoseIntegratorCallbacks cb;
    Simulation* sim = new Simulation(cb, SolveDescription(4, 1));  // config 1: 1 substep x 4 velocity iterations
    Material m{1.f, SpringSettings(30.f, 1.f), 2.f};               // DemoCallbacks.cs:117-131
    BodyInertia boxI = boxInertia(1, 1, 1, 1);
    for (int64_t p = 0; p < pyramidCount; ++p) {
        std::vector<std::vector<int32_t>> rows(rowCount);
        for (int r = 0; r < rowCount; ++r) {
            int columnCount = (int)rowCount - r;
            for (int c = 0; c < columnCount; ++c) {
                RigidPose pose{{(-columnCount * 0.5f + c) * 1.f, (r + 0.5f) * 1.f, (p - pyramidCount * 0.5f) * 6.f}, Quaternion{}};
                rows[r].push_back(sim->bodies.Add(BodyDescription::CreateDynamic(pose, boxI)));
            }
        }
        for (int32_t h : rows[0]) {  // ground manifolds: 4 corners of the bottom face, normal +Y, depth 0
            ContactPoint pts[4] = {{{-0.5f, -0.5f, -0.5f}, 0}, {{0.5f, -0.5f, -0.5f}, 0}, {{-0.5f, -0.5f, 0.5f}, 0}, {{0.5f, -0.5f, 0.5f}, 0}};
            addContact(sim->solver, 4, h, -1, pts, {}, {0, 1, 0}, m);
        }
        for (int r = 1; r < rowCount; ++r)
            for (size_t c = 0; c < rows[r].size(); ++c)
                for (int side = 0; side < 2; ++side) {
                    int32_t a = rows[r][c], b = rows[r - 1][c + side];
                    Vector3 pa = bodyPosition(*sim, a), pb = bodyPosition(*sim, b);
                    float x0 = std::max(pa.X, pb.X) - 0.5f, x1 = std::min(pa.X, pb.X) + 0.5f;
                    ContactPoint pts[4] = {{{x0 - pa.X, -0.5f, -0.5f}, 0}, {{x1 - pa.X, -0.5f, -0.5f}, 0}, {{x0 - pa.X, -0.5f, 0.5f}, 0}, {{x1 - pa.X, -0.5f, 0.5f}, 0}};
                    addContact(sim->solver, 4, a, b, pts, pb - pa, {0, 1, 0}, m);
                }
    }
    return sim;
}

Simulation* buildPile(int64_t bodyTarget, uint32_t seed) {  // ShapePileBenchmark-style (config 2)
    PoseIntegratorCallbacks cb;
    Simulation* sim = new Simulation(cb, SolveDescription(2, 4));  // 4 substeps x 2 iterations
    Material m{1.f, SpringSettings(30.f, 1.f), 2.f};
    Rng rng(seed);
    int n = (int)std::ceil(std::cbrt((double)bodyTarget));
    int nx = n, nz = n, ny = (int)((bodyTarget + (int64_t)nx * nz - 1) / ((int64_t)nx * nz));
    BodyInertia boxI = boxInertia(1, 3, 2, 1);  // ShapePileBenchmark.cs:111,160
    auto index = [&](int x, int y, int z) { return ((int64_t)y * nz + z) * nx + x; };
    std::vector<int32_t> handles((size_t)nx * ny * nz, -1);
    int64_t made = 0;
    for (int y = 0; y < ny; ++y)
        for (int z = 0; z < nz; ++z)
            for (int x = 0; x < nx && made < bodyTarget; ++x, ++made) {
                RigidPose pose{{x * 1.1f + rng.uniform(-0.05f, 0.05f), 0.5f + y * 1.1f + rng.uniform(-0.05f, 0.05f), z * 1.1f + rng.uniform(-0.05f, 0.05f)},
                               fromAxisAngle(rng.unit(), rng.uniform(0, 0.3f))};
                BodyDescription d = BodyDescription::CreateDynamic(pose, boxI);
                d.Velocity.Linear = {rng.uniform(-0.2f, 0.2f), rng.uniform(-0.2f, 0.2f), rng.uniform(-0.2f, 0.2f)};
                d.Velocity.Angular = {rng.uniform(-0.2f, 0.2f), rng.uniform(-0.2f, 0.2f), rng.uniform(-0.2f, 0.2f)};
                handles[index(x, y, z)] = sim->bodies.Add(d);
            }
    auto contactCount = [&]() {  // manifold size distribution {1:10%, 2:20%, 3:10%, 4:60%}
        float u = rng.uniform(0, 1);
        return u < 0.1f ? 1 : (u < 0.3f ? 2 : (u < 0.4f ? 3 : 4));
    };
    for (int y = 0; y < ny; ++y)
        for (int z = 0; z < nz; ++z)
            for (int x = 0; x < nx; ++x) {
                int32_t a = handles[index(x, y, z)];
                if (a < 0) continue;
                if (y == 0) addSyntheticGroundContact(*sim, rng, 4, a, 0.5f, 0.5f, 0.5f, m);
                const int dx[3] = {1, 0, 0}, dy[3] = {0, 1, 0}, dz[3] = {0, 0, 1};
                for (int k = 0; k < 3; ++k) {
                    int X = x + dx[k], Y = y + dy[k], Z = z + dz[k];
                    if (X >= nx || Y >= ny || Z >= nz) continue;
                    int32_t b = handles[index(X, Y, Z)];
                    if (b < 0) continue;
                    addSyntheticPairContact(*sim, rng, contactCount(), a, b, m, false);
                }
            }
    return sim;
}

}  // namespace

Simulation* BuildScene(const char* name, int64_t a, int64_t b, int64_t c, uint32_t seed) {
    std::string n = name;
    if (n == "pyramid") return buildPyramids(a > 0 ? a : 5, b > 0 ? b : 20);
    if (n == "pile") return buildPile(a > 0 ? a : 100000, seed);
    if (n == "ragdoll_tube") return buildRagdollTube(a > 0 ? a : 15000, b, c, seed);
    throw std::invalid_argument("unknown scene '" + n + "'");
}

}  // namespace bepu
