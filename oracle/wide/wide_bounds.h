// oracle/wide — TEST INFRASTRUCTURE (parity unpinned: no run of the reference behind it). PoseIntegrator.PredictBoundingBoxes with the BoundingBoxBatcher, transcribed from the C# alone like the rest of this directory
// (no text shared with oracle/bepu_bounds.h or the device's bepu_device_bounds.h): bodies walked in bundles of Vector<float>.Count, the velocity callback on the bundle,
// collidables accumulated per shape type and flushed sixteen at a time, TShapeWide.GetBounds on eight shapes at once, compound children re-entering the batcher as convex
// shapes with a merge continuation, meshes through the scalar path. Included by wide_solver.cpp after Bodies and PoseIntegratorCallbacks.
#pragma once

namespace wide {
namespace bounds {

// ---------------------------------------------------------------------------------------------------------------- System.Numerics scalars (the narrow paths)
struct Vector3 {
    float X, Y, Z;
    float LengthSquared() const { return X * X + Y * Y + Z * Z; }  // Vector3.Dot(v, v): (x*x + y*y) + z*z
    float Length() const { return sqrtf(LengthSquared()); }
};
static inline Vector3 operator+(Vector3 a, Vector3 b) { return {a.X + b.X, a.Y + b.Y, a.Z + b.Z}; }
static inline Vector3 operator-(Vector3 a, Vector3 b) { return {a.X - b.X, a.Y - b.Y, a.Z - b.Z}; }
static inline Vector3 operator-(Vector3 a) { return {-a.X, -a.Y, -a.Z}; }
static inline Vector3 operator*(Vector3 a, Vector3 b) { return {a.X * b.X, a.Y * b.Y, a.Z * b.Z}; }
static inline Vector3 operator*(Vector3 a, float s) { return {a.X * s, a.Y * s, a.Z * s}; }
static inline float MinF(float a, float b) { return a < b ? a : b; }  // minps / maxps lane semantics of Vector3.Min / Vector3.Max on the reference's x86 hosts
static inline float MaxF(float a, float b) { return a > b ? a : b; }
static inline Vector3 Min(Vector3 a, Vector3 b) { return {MinF(a.X, b.X), MinF(a.Y, b.Y), MinF(a.Z, b.Z)}; }
static inline Vector3 Max(Vector3 a, Vector3 b) { return {MaxF(a.X, b.X), MaxF(a.Y, b.Y), MaxF(a.Z, b.Z)}; }
static inline Vector3 AbsV(Vector3 a) { return {fabsf(a.X), fabsf(a.Y), fabsf(a.Z)}; }
static inline Vector3 Cross(Vector3 a, Vector3 b) { return {a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X}; }
static inline Vector3 Broadcast3(float s) { return {s, s, s}; }
struct Quaternion { float X, Y, Z, W; };
static inline float MathFMax(float a, float b) {  // System.MathF.Max
    if (a != a) return a;
    if (b != b) return b;
    if (a == b) return std::signbit(a) ? b : a;
    return a > b ? a : b;
}
static inline float MathFMin(float a, float b) {  // System.MathF.Min
    if (a != a) return a;
    if (b != b) return b;
    if (a == b) return std::signbit(a) ? a : b;
    return a < b ? a : b;
}
namespace QuaternionEx {
static inline void ConcatenateWithoutOverlap(Quaternion a, Quaternion b, Quaternion& result) {  // BepuUtilities/QuaternionEx.cs:50-56
    result.X = a.W * b.X + a.X * b.W + a.Z * b.Y - a.Y * b.Z;
    result.Y = a.W * b.Y + a.Y * b.W + a.X * b.Z - a.Z * b.X;
    result.Z = a.W * b.Z + a.Z * b.W + a.Y * b.X - a.X * b.Y;
    result.W = a.W * b.W - a.X * b.X - a.Y * b.Y - a.Z * b.Z;
}
static inline void Transform(Vector3 v, Quaternion rotation, Vector3& result) {  // QuaternionEx.cs:373-395 (TransformWithoutOverlap, copied out by :405-409)
    float x2 = rotation.X + rotation.X;
    float y2 = rotation.Y + rotation.Y;
    float z2 = rotation.Z + rotation.Z;
    float xx2 = rotation.X * x2;
    float xy2 = rotation.X * y2;
    float xz2 = rotation.X * z2;
    float yy2 = rotation.Y * y2;
    float yz2 = rotation.Y * z2;
    float zz2 = rotation.Z * z2;
    float wx2 = rotation.W * x2;
    float wy2 = rotation.W * y2;
    float wz2 = rotation.W * z2;
    result.X = v.X * (1.0f - yy2 - zz2) + v.Y * (xy2 - wz2) + v.Z * (xz2 + wy2);
    result.Y = v.X * (xy2 + wz2) + v.Y * (1.0f - xx2 - zz2) + v.Z * (yz2 - wx2);
    result.Z = v.X * (xz2 - wy2) + v.Y * (yz2 + wx2) + v.Z * (1.0f - xx2 - yy2);
}
}  // namespace QuaternionEx
struct Matrix3x3 {  // BepuUtilities/Matrix3x3.cs
    Vector3 X, Y, Z;
    static void CreateFromQuaternion(Quaternion quaternion, Matrix3x3& result) {  // :306-335
        float qX2 = quaternion.X + quaternion.X;
        float qY2 = quaternion.Y + quaternion.Y;
        float qZ2 = quaternion.Z + quaternion.Z;
        float XX = qX2 * quaternion.X;
        float YY = qY2 * quaternion.Y;
        float ZZ = qZ2 * quaternion.Z;
        float XY = qX2 * quaternion.Y;
        float XZ = qX2 * quaternion.Z;
        float XW = qX2 * quaternion.W;
        float YZ = qY2 * quaternion.Z;
        float YW = qY2 * quaternion.W;
        float ZW = qZ2 * quaternion.W;
        result.X = {1 - YY - ZZ, XY + ZW, XZ - YW};
        result.Y = {XY - ZW, 1 - XX - ZZ, YZ + XW};
        result.Z = {XZ + YW, YZ - XW, 1 - XX - YY};
    }
    static void Transform(Vector3 v, const Matrix3x3& m, Vector3& result) {  // :200-206
        Vector3 x = Broadcast3(v.X), y = Broadcast3(v.Y), z = Broadcast3(v.Z);
        result = m.X * x + m.Y * y + m.Z * z;
    }
};
struct RigidPose { Vector3 Position; Quaternion Orientation; };
struct BodyVelocity { Vector3 Linear, Angular; };
struct MotionState { RigidPose Pose; BodyVelocity Velocity; };

// ---------------------------------------------------------------------------------------------------------------- BoundingBoxHelpers.cs
namespace BoundingBoxHelpers {
static inline VF GetAngularBoundsExpansion(VF angularSpeed, VF vectorDt, VF maximumRadius, VF maximumAngularExpansion) {  // :12-47
    VF a = wide::Min(angularSpeed * vectorDt, vf(3.14159274f / 3.0f));  // MathHelper.Pi / 3f
    VF a2 = a * a;
    VF a4 = a2 * a2;
    VF a6 = a4 * a2;
    VF cosAngleMinusOne = a2 * vf(-1.0f / 2.0f) + a4 * vf(1.0f / 24.0f) - a6 * vf(1.0f / 720.0f);
    return wide::Min(maximumAngularExpansion, SquareRoot(vf(-2.0f) * maximumRadius * maximumRadius * cosAngleMinusOne));
}
static inline void GetBoundsExpansion(Vector3Wide linearVelocity, VF dtWide, VF angularExpansion, Vector3Wide& minExpansion, Vector3Wide& maxExpansion) {  // :51-60
    Vector3Wide linearDisplacement = linearVelocity * dtWide;
    VF zero = kZero;
    minExpansion = Vector3Wide{wide::Min(zero, linearDisplacement.X), wide::Min(zero, linearDisplacement.Y), wide::Min(zero, linearDisplacement.Z)};
    maxExpansion = Vector3Wide{wide::Max(zero, linearDisplacement.X), wide::Max(zero, linearDisplacement.Y), wide::Max(zero, linearDisplacement.Z)};
    minExpansion = Vector3Wide{minExpansion.X - angularExpansion, minExpansion.Y - angularExpansion, minExpansion.Z - angularExpansion};  // Vector3Wide.Subtract(v, s)
    maxExpansion = Vector3Wide{maxExpansion.X + angularExpansion, maxExpansion.Y + angularExpansion, maxExpansion.Z + angularExpansion};  // Vector3Wide.Add(v, s)
}
static inline float GetAngularBoundsExpansion(float angularVelocityMagnitude, float dt, float maximumRadius, float maximumAngularExpansion) {  // :129-138
    float a = MinF(angularVelocityMagnitude * dt, 3.14159274f / 3.0f);  // MathHelper.Min
    float a2 = a * a;
    float a4 = a2 * a2;
    float a6 = a4 * a2;
    float cosAngleMinusOne = a2 * (-1.0f / 2.0f) + a4 * (1.0f / 24.0f) - a6 * (1.0f / 720.0f);
    return MinF(maximumAngularExpansion, (float)sqrt((double)(-2.0f * maximumRadius * maximumRadius * cosAngleMinusOne)));  // (float)Math.Sqrt(float)
}
static inline void GetBoundsExpansion(Vector3 linearVelocity, float dt, float angularExpansion, Vector3& minExpansion, Vector3& maxExpansion) {  // :141-148
    Vector3 linearDisplacement = linearVelocity * dt;
    Vector3 zero = {0, 0, 0};
    Vector3 broadcastExpansion = Broadcast3(angularExpansion);
    minExpansion = Min(zero, linearDisplacement) - broadcastExpansion;
    maxExpansion = Max(zero, linearDisplacement) + broadcastExpansion;
}
}  // namespace BoundingBoxHelpers

// ---------------------------------------------------------------------------------------------------------------- the shapes and their wide forms
enum { SphereId = 0, CapsuleId = 1, BoxId = 2, TriangleId = 3, CylinderId = 4, ConvexHullId = 5, CompoundId = 6, BigCompoundId = 7, MeshId = 8, RegisteredTypeSpan = 9 };
struct Sphere { float Radius; };
struct Capsule { float Radius, HalfLength; };
struct Box { float HalfWidth, HalfHeight, HalfLength; };
struct Triangle { Vector3 A, B, C; };
struct Cylinder { float Radius, HalfLength; };
struct ConvexHull { std::vector<Vector3Wide> Points; };  // ConvexHull.cs:30; the last bundle's unused lanes repeat a real point

struct SphereWide {  // Sphere.cs:127-160
    VF Radius;
    void WriteSlot(int index, const Sphere& source) { Radius[index] = source.Radius; }
    void GetBounds(QuaternionWide& orientations, int countInBundle, VF& maximumRadius, VF& maximumAngularExpansion, Vector3Wide& min, Vector3Wide& max) {  // :149
        maximumRadius = kZero;
        maximumAngularExpansion = kZero;
        VF negatedRadius = neg(Radius);
        max = Vector3Wide{Radius, Radius, Radius};
        min = Vector3Wide{negatedRadius, negatedRadius, negatedRadius};
    }
};
struct CapsuleWide {  // Capsule.cs:202-239
    VF Radius, HalfLength;
    void WriteSlot(int index, const Capsule& source) { Radius[index] = source.Radius; HalfLength[index] = source.HalfLength; }
    void GetBounds(QuaternionWide& orientations, int countInBundle, VF& maximumRadius, VF& maximumAngularExpansion, Vector3Wide& min, Vector3Wide& max) {  // :226
        Vector3Wide segmentOffset = QuaternionWide::TransformUnitY(orientations);
        Vector3Wide::Scale(segmentOffset, HalfLength, segmentOffset);
        segmentOffset = Vector3Wide{Abs(segmentOffset.X), Abs(segmentOffset.Y), Abs(segmentOffset.Z)};
        max = Vector3Wide{segmentOffset.X + Radius, segmentOffset.Y + Radius, segmentOffset.Z + Radius};
        Vector3Wide::Negate(max, min);
        maximumRadius = HalfLength + Radius;
        maximumAngularExpansion = HalfLength;
    }
};
struct BoxWide {  // Box.cs:184-222
    VF HalfWidth, HalfHeight, HalfLength;
    void WriteSlot(int index, const Box& source) { HalfWidth[index] = source.HalfWidth; HalfHeight[index] = source.HalfHeight; HalfLength[index] = source.HalfLength; }
    void GetBounds(QuaternionWide& orientations, int countInBundle, VF& maximumRadius, VF& maximumAngularExpansion, Vector3Wide& min, Vector3Wide& max) {  // :211
        Matrix3x3Wide basis;
        Matrix3x3Wide::CreateFromQuaternion(orientations, basis);
        max.X = Abs(HalfWidth * basis.X.X) + Abs(HalfHeight * basis.Y.X) + Abs(HalfLength * basis.Z.X);
        max.Y = Abs(HalfWidth * basis.X.Y) + Abs(HalfHeight * basis.Y.Y) + Abs(HalfLength * basis.Z.Y);
        max.Z = Abs(HalfWidth * basis.X.Z) + Abs(HalfHeight * basis.Y.Z) + Abs(HalfLength * basis.Z.Z);
        Vector3Wide::Negate(max, min);
        maximumRadius = SquareRoot(HalfWidth * HalfWidth + HalfHeight * HalfHeight + HalfLength * HalfLength);
        maximumAngularExpansion = maximumRadius - wide::Min(HalfLength, wide::Min(HalfHeight, HalfLength));  // as written (:221)
    }
};
struct TriangleWide {  // Triangle.cs:143-221
    Vector3Wide A, B, C;
    static void WriteFirst(Vector3 source, int index, Vector3Wide& target) { target.X[index] = source.X; target.Y[index] = source.Y; target.Z[index] = source.Z; }
    void WriteSlot(int index, const Triangle& source) { WriteFirst(source.A, index, A); WriteFirst(source.B, index, B); WriteFirst(source.C, index, C); }
    void GetBounds(QuaternionWide& orientations, int countInBundle, VF& maximumRadius, VF& maximumAngularExpansion, Vector3Wide& min, Vector3Wide& max) {  // :203
        Matrix3x3Wide basis;
        Matrix3x3Wide::CreateFromQuaternion(orientations, basis);
        Vector3Wide worldA, worldB, worldC;
        Matrix3x3Wide::TransformWithoutOverlap(A, basis, worldA);
        Matrix3x3Wide::TransformWithoutOverlap(B, basis, worldB);
        Matrix3x3Wide::TransformWithoutOverlap(C, basis, worldC);
        min.X = wide::Min(worldA.X, wide::Min(worldB.X, worldC.X));
        min.Y = wide::Min(worldA.Y, wide::Min(worldB.Y, worldC.Y));
        min.Z = wide::Min(worldA.Z, wide::Min(worldB.Z, worldC.Z));
        max.X = wide::Max(worldA.X, wide::Max(worldB.X, worldC.X));
        max.Y = wide::Max(worldA.Y, wide::Max(worldB.Y, worldC.Y));
        max.Z = wide::Max(worldA.Z, wide::Max(worldB.Z, worldC.Z));
        VF aLengthSquared, bLengthSquared, cLengthSquared;
        Vector3Wide::LengthSquared(A, aLengthSquared);
        Vector3Wide::LengthSquared(B, bLengthSquared);
        Vector3Wide::LengthSquared(C, cLengthSquared);
        maximumRadius = SquareRoot(wide::Max(aLengthSquared, wide::Max(bLengthSquared, cLengthSquared)));
        maximumAngularExpansion = maximumRadius;
    }
};
struct CylinderWide {  // Cylinder.cs:198-235
    VF Radius, HalfLength;
    void WriteSlot(int index, const Cylinder& source) { Radius[index] = source.Radius; HalfLength[index] = source.HalfLength; }
    void GetBounds(QuaternionWide& orientations, int countInBundle, VF& maximumRadius, VF& maximumAngularExpansion, Vector3Wide& min, Vector3Wide& max) {  // :222
        Vector3Wide y = QuaternionWide::TransformUnitY(orientations);
        Vector3Wide yy = Vector3Wide{y.X * y.X, y.Y * y.Y, y.Z * y.Z};
        Vector3Wide squared = Vector3Wide{kOne - yy.X, kOne - yy.Y, kOne - yy.Z};
        max.X = Abs(HalfLength * y.X) + SquareRoot(wide::Max(kZero, squared.X)) * Radius;
        max.Y = Abs(HalfLength * y.Y) + SquareRoot(wide::Max(kZero, squared.Y)) * Radius;
        max.Z = Abs(HalfLength * y.Z) + SquareRoot(wide::Max(kZero, squared.Z)) * Radius;
        Vector3Wide::Negate(max, min);
        maximumRadius = SquareRoot(HalfLength * HalfLength + Radius * Radius);
        maximumAngularExpansion = maximumRadius - wide::Min(HalfLength, Radius);
    }
};
struct ConvexHullWide {  // ConvexHull.cs:296-364
    const ConvexHull* Hulls[W];
    void WriteSlot(int index, const ConvexHull& source) { Hulls[index] = &source; }
    void GetBounds(QuaternionWide& orientations, int countInBundle, VF& maximumRadius, VF& maximumAngularExpansion, Vector3Wide& min, Vector3Wide& max) {  // :319
        maximumRadius = kZero;  // (Unsafe.SkipInit there)
        min = max = Vector3Wide{kZero, kZero, kZero};
        for (int i = 0; i < countInBundle; ++i) {
            Vector3Wide minWide = Vector3Wide::Broadcast(3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f);
            Vector3Wide maxWide = Vector3Wide::Broadcast(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);  // float.MinValue
            QuaternionWide orientationWide = {vf(orientations.X[i]), vf(orientations.Y[i]), vf(orientations.Z[i]), vf(orientations.W[i])};  // QuaternionWide.Rebroadcast
            Matrix3x3Wide orientationMatrix;
            Matrix3x3Wide::CreateFromQuaternion(orientationWide, orientationMatrix);
            VF maximumRadiusSquaredWide = kZero;
            const ConvexHull& hull = *Hulls[i];
            for (size_t j = 0; j < hull.Points.size(); ++j) {
                const Vector3Wide& localPoint = hull.Points[j];
                Vector3Wide p;
                Matrix3x3Wide::TransformWithoutOverlap(localPoint, orientationMatrix, p);
                VF lengthSquared;
                Vector3Wide::LengthSquared(localPoint, lengthSquared);
                maximumRadiusSquaredWide = wide::Max(lengthSquared, maximumRadiusSquaredWide);
                minWide = Vector3Wide{wide::Min(minWide.X, p.X), wide::Min(minWide.Y, p.Y), wide::Min(minWide.Z, p.Z)};
                maxWide = Vector3Wide{wide::Max(maxWide.X, p.X), wide::Max(maxWide.Y, p.Y), wide::Max(maxWide.Z, p.Z)};
            }
            Vector3 minNarrow = {minWide.X[0], minWide.Y[0], minWide.Z[0]};
            Vector3 maxNarrow = {maxWide.X[0], maxWide.Y[0], maxWide.Z[0]};
            float maximumRadiusSquared = maximumRadiusSquaredWide[0];
            for (int j = 1; j < W; ++j) {
                Vector3 minCandidate = {minWide.X[j], minWide.Y[j], minWide.Z[j]};
                Vector3 maxCandidate = {maxWide.X[j], maxWide.Y[j], maxWide.Z[j]};
                minNarrow = Min(minCandidate, minNarrow);
                maxNarrow = Max(maxCandidate, maxNarrow);
                float maxRadiusCandidate = maximumRadiusSquaredWide[j];
                if (maxRadiusCandidate > maximumRadiusSquared) maximumRadiusSquared = maxRadiusCandidate;
            }
            maximumRadius[i] = maximumRadiusSquared;
            min.X[i] = minNarrow.X; min.Y[i] = minNarrow.Y; min.Z[i] = minNarrow.Z;
            max.X[i] = maxNarrow.X; max.Y[i] = maxNarrow.Y; max.Z[i] = maxNarrow.Z;
        }
        maximumRadius = SquareRoot(maximumRadius);
        maximumAngularExpansion = maximumRadius;
    }
};

struct CompoundChild { int ShapeType, ShapeIndex; Vector3 LocalPosition; Quaternion LocalOrientation; };  // Compound.cs:13-40 (TypedIndex split in two)
struct Compound { std::vector<CompoundChild> Children; };
struct Mesh {  // Mesh.cs
    std::vector<Triangle> Triangles;
    Vector3 scale;
    void ComputeBounds(Quaternion orientation, Vector3& min, Vector3& max) const {  // :232-255
        Matrix3x3 r;
        Matrix3x3::CreateFromQuaternion(orientation, r);
        min = Broadcast3(3.402823466e+38f);
        max = Broadcast3(-3.402823466e+38f);
        for (size_t i = 0; i < Triangles.size(); ++i) {
            const Triangle& triangle = Triangles[i];
            Vector3 a, b, c;
            Matrix3x3::Transform(scale * triangle.A, r, a);
            Matrix3x3::Transform(scale * triangle.B, r, b);
            Matrix3x3::Transform(scale * triangle.C, r, c);
            Vector3 min0 = Min(a, b);
            Vector3 min1 = Min(c, min);
            Vector3 max0 = Max(a, b);
            Vector3 max1 = Max(c, max);
            min = Min(min0, min1);
            max = Max(max0, max1);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------- what the batcher writes into
struct Collidable {  // Collidables/Collidable.cs (the fields the bounds update touches)
    int ShapeType, ShapeIndex;  // Shape (TypedIndex); ShapeType < 0: Shape.Exists == false
    float MinimumSpeculativeMargin, MaximumSpeculativeMargin, SpeculativeMargin;
    bool AllowExpansionBeyondSpeculativeMargin;  // Continuity.AllowExpansionBeyondSpeculativeMargin
};
struct BodyActivity { float SleepThreshold; int MinimumTimestepsUnderThreshold; int TimestepsUnderThresholdCount; bool SleepCandidate; };  // BodyActivity (BodyProperties.cs); the count is a byte
struct Shapes {
    std::vector<Sphere> spheres; std::vector<Capsule> capsules; std::vector<Box> boxes; std::vector<Triangle> triangles; std::vector<Cylinder> cylinders;
    std::vector<ConvexHull> hulls; std::vector<Compound> compounds; std::vector<Mesh> meshes;
};
struct World {
    Shapes shapes;
    std::vector<Collidable> collidables;  // bodies.ActiveSet.Collidables
    std::vector<BodyActivity> activities;  // bodies.ActiveSet.Activity
    std::vector<Vector3> boundsMin, boundsMax;  // broadPhase.GetActiveBoundsPointers(collidable.BroadPhaseIndex, ...): one leaf per body here
};

struct BoundsContinuation {  // BoundingBoxBatcher.cs:12-49
    uint32_t packed;
    int BodyIndex() const { return (int)(packed & 0x7FFFFFFF); }
    bool CompoundChild() const { return (packed & (1u << 31)) > 0; }
    static BoundsContinuation CreateContinuation(int bodyIndex) { return {(uint32_t)bodyIndex}; }
    static BoundsContinuation CreateCompoundChildContinuation(int compoundBodyIndex) { return {(1u << 31) | (uint32_t)compoundBodyIndex}; }
};
struct BoundingBoxBatch {  // :57-108
    std::vector<int> ShapeIndices;
    std::vector<BoundsContinuation> Continuations;
    std::vector<MotionState> MotionStates;
    int Count = 0;
    bool Allocated = false;
    void Allocate(int capacity) { ShapeIndices.resize(capacity); Continuations.resize(capacity); MotionStates.resize(capacity); Count = 0; Allocated = true; }
    void Add(int shapeIndex, const RigidPose& pose, const BodyVelocity& velocity, BoundsContinuation continuation) {
        ShapeIndices[Count] = shapeIndex;
        Continuations[Count] = continuation;
        MotionStates[Count].Pose = pose;
        MotionStates[Count].Velocity = velocity;
        Count++;
    }
};

struct BoundingBoxBatcher {  // :110-375
    World& world;
    float dt;
    int minimumBatchIndex, maximumBatchIndex;
    BoundingBoxBatch batches[RegisteredTypeSpan];
    static constexpr int CollidablesPerFlush = 16;  // :126
    BoundingBoxBatcher(World& world, float dt) : world(world), dt(dt), minimumBatchIndex(RegisteredTypeSpan), maximumBatchIndex(-1) {}

    template <typename TShape, typename TShapeWide> void ExecuteConvexBatch(int typeId, const std::vector<TShape>& shapes) {  // :142-223
        TShapeWide shapeWide = {};  // Unsafe.SkipInit in the reference; unused lanes never reach a result
        BoundingBoxBatch& batch = batches[typeId];
        VF dtWide = vf(dt);
        for (int bundleStartIndex = 0; bundleStartIndex < batch.Count; bundleStartIndex += W) {
            int countInBundle = batch.Count - bundleStartIndex;
            if (countInBundle > W) countInBundle = W;
            VF minimumSpeculativeMargin = kZero, maximumSpeculativeMargin = kZero;
            VI allowExpansionBeyondSpeculativeMargin = vi(0);
            Vector3Wide positions = {kZero, kZero, kZero};
            QuaternionWide orientations = {kZero, kZero, kZero, kZero};
            BodyVelocityWide velocities = {{kZero, kZero, kZero}, {kZero, kZero, kZero}};
            for (int innerIndex = 0; innerIndex < countInBundle; ++innerIndex) {
                int indexInBatch = bundleStartIndex + innerIndex;
                int shapeIndex = batch.ShapeIndices[indexInBatch];
                shapeWide.WriteSlot(innerIndex, shapes[shapeIndex]);
                const Collidable& collidable = world.collidables[batch.Continuations[indexInBatch].BodyIndex()];
                minimumSpeculativeMargin[innerIndex] = collidable.MinimumSpeculativeMargin;
                maximumSpeculativeMargin[innerIndex] = collidable.MaximumSpeculativeMargin;
                allowExpansionBeyondSpeculativeMargin[innerIndex] = collidable.AllowExpansionBeyondSpeculativeMargin ? -1 : 0;
                const MotionState& state = batch.MotionStates[indexInBatch];  // Bodies.TransposeMotionStates(:169)
                positions.X[innerIndex] = state.Pose.Position.X; positions.Y[innerIndex] = state.Pose.Position.Y; positions.Z[innerIndex] = state.Pose.Position.Z;
                orientations.X[innerIndex] = state.Pose.Orientation.X; orientations.Y[innerIndex] = state.Pose.Orientation.Y;
                orientations.Z[innerIndex] = state.Pose.Orientation.Z; orientations.W[innerIndex] = state.Pose.Orientation.W;
                velocities.Linear.X[innerIndex] = state.Velocity.Linear.X; velocities.Linear.Y[innerIndex] = state.Velocity.Linear.Y; velocities.Linear.Z[innerIndex] = state.Velocity.Linear.Z;
                velocities.Angular.X[innerIndex] = state.Velocity.Angular.X; velocities.Angular.Y[innerIndex] = state.Velocity.Angular.Y; velocities.Angular.Z[innerIndex] = state.Velocity.Angular.Z;
            }
            VF maximumRadius, maximumAngularExpansion;
            Vector3Wide bundleMin, bundleMax;
            shapeWide.GetBounds(orientations, countInBundle, maximumRadius, maximumAngularExpansion, bundleMin, bundleMax);
            VF angularBoundsExpansion = BoundingBoxHelpers::GetAngularBoundsExpansion(Vector3Wide::Length(velocities.Angular), dtWide, maximumRadius, maximumAngularExpansion);
            VF speculativeMargin = Vector3Wide::Length(velocities.Linear) * dtWide + angularBoundsExpansion;
            speculativeMargin = wide::Max(minimumSpeculativeMargin, wide::Min(maximumSpeculativeMargin, speculativeMargin));
            VF maximumBoundsExpansion = ConditionalSelect(allowExpansionBeyondSpeculativeMargin, vf(3.402823466e+38f), speculativeMargin);
            Vector3Wide minExpansion, maxExpansion;
            BoundingBoxHelpers::GetBoundsExpansion(velocities.Linear, dtWide, angularBoundsExpansion, minExpansion, maxExpansion);
            VF negatedMaximumBoundsExpansion = neg(maximumBoundsExpansion);
            minExpansion = Vector3Wide{wide::Max(negatedMaximumBoundsExpansion, minExpansion.X), wide::Max(negatedMaximumBoundsExpansion, minExpansion.Y),
                                       wide::Max(negatedMaximumBoundsExpansion, minExpansion.Z)};
            maxExpansion = Vector3Wide{wide::Min(maximumBoundsExpansion, maxExpansion.X), wide::Min(maximumBoundsExpansion, maxExpansion.Y), wide::Min(maximumBoundsExpansion, maxExpansion.Z)};
            bundleMin = positions + (bundleMin + minExpansion);
            bundleMax = positions + (bundleMax + maxExpansion);
            for (int innerIndex = 0; innerIndex < countInBundle; ++innerIndex) {
                BoundsContinuation continuation = batch.Continuations[bundleStartIndex + innerIndex];
                Collidable& collidable = world.collidables[continuation.BodyIndex()];
                Vector3& minPointer = world.boundsMin[continuation.BodyIndex()];
                Vector3& maxPointer = world.boundsMax[continuation.BodyIndex()];
                if (continuation.CompoundChild()) {
                    collidable.SpeculativeMargin = MathFMax(collidable.SpeculativeMargin, speculativeMargin[innerIndex]);
                    Vector3 min = {bundleMin.X[innerIndex], bundleMin.Y[innerIndex], bundleMin.Z[innerIndex]};
                    Vector3 max = {bundleMax.X[innerIndex], bundleMax.Y[innerIndex], bundleMax.Z[innerIndex]};
                    minPointer = Min(minPointer, min);  // BoundingBox.CreateMerged (BepuUtilities/BoundingBox.cs:173-177)
                    maxPointer = Max(maxPointer, max);
                } else {
                    collidable.SpeculativeMargin = speculativeMargin[innerIndex];
                    minPointer = {bundleMin.X[innerIndex], bundleMin.Y[innerIndex], bundleMin.Z[innerIndex]};
                    maxPointer = {bundleMax.X[innerIndex], bundleMax.Y[innerIndex], bundleMax.Z[innerIndex]};
                }
            }
        }
    }

    void ExecuteHomogeneousCompoundBatch(int typeId) {  // :225-266 (Mesh is the one homogeneous compound)
        BoundingBoxBatch& batch = batches[typeId];
        for (int i = 0; i < batch.Count; ++i) {
            int shapeIndex = batch.ShapeIndices[i];
            const MotionState& motionState = batch.MotionStates[i];
            int bodyIndex = batch.Continuations[i].BodyIndex();
            Collidable& collidable = world.collidables[bodyIndex];
            Vector3 min, max;
            world.shapes.meshes[shapeIndex].ComputeBounds(motionState.Pose.Orientation, min, max);
            Vector3 absMin = AbsV(min);
            Vector3 absMax = AbsV(max);
            float maximumRadius = Max(absMin, absMax).Length();
            Vector3 minimumComponents = Min(absMin, absMax);
            float minimumRadius = MinF(minimumComponents.X, MinF(minimumComponents.Y, minimumComponents.Z));
            float maximumAngularExpansion = maximumRadius - minimumRadius;
            float angularBoundsExpansion = BoundingBoxHelpers::GetAngularBoundsExpansion(motionState.Velocity.Angular.Length(), dt, maximumRadius, maximumAngularExpansion);
            float speculativeMargin = motionState.Velocity.Linear.Length() * dt + angularBoundsExpansion;
            speculativeMargin = MathFMax(collidable.MinimumSpeculativeMargin, MathFMin(collidable.MaximumSpeculativeMargin, speculativeMargin));
            collidable.SpeculativeMargin = speculativeMargin;
            float maximumAllowedExpansion = collidable.AllowExpansionBeyondSpeculativeMargin ? 3.402823466e+38f : speculativeMargin;
            Vector3 minExpansion, maxExpansion;
            BoundingBoxHelpers::GetBoundsExpansion(motionState.Velocity.Linear, dt, angularBoundsExpansion, minExpansion, maxExpansion);
            Vector3 broadcastMaximumBoundsExpansion = Broadcast3(maximumAllowedExpansion);
            minExpansion = Max(-broadcastMaximumBoundsExpansion, minExpansion);
            maxExpansion = Min(broadcastMaximumBoundsExpansion, maxExpansion);
            world.boundsMin[bodyIndex] = motionState.Pose.Position + (min + minExpansion);
            world.boundsMax[bodyIndex] = motionState.Pose.Position + (max + maxExpansion);
        }
    }

    void AddCompoundChild(int bodyIndex, int shapeType, int shapeIndex, const RigidPose& pose, const BodyVelocity& velocity) {  // :330-333
        Add(shapeType, shapeIndex, pose, velocity, BoundsContinuation::CreateCompoundChildContinuation(bodyIndex));
    }
    // Compound.AddChildBoundsToBatcher (Compound.cs:198-221; BigCompound.cs:128-131 forwards to it)
    void AddChildBoundsToBatcher(const std::vector<CompoundChild>& children, const RigidPose& pose, const BodyVelocity& velocity, int bodyIndex) {
        BodyVelocity childVelocity;
        childVelocity.Angular = velocity.Angular;
        for (size_t i = 0; i < children.size(); ++i) {
            const CompoundChild& child = children[i];
            RigidPose childPose;
            QuaternionEx::ConcatenateWithoutOverlap(child.LocalOrientation, pose.Orientation, childPose.Orientation);  // Compound.GetRotatedChildPose (:153-157)
            QuaternionEx::Transform(child.LocalPosition, pose.Orientation, childPose.Position);
            Vector3 angularContributionToChildLinear = Cross(velocity.Angular, childPose.Position);
            float contributionLengthSquared = angularContributionToChildLinear.LengthSquared();
            float localPoseRadiusSquared = childPose.Position.LengthSquared();
            if (contributionLengthSquared > localPoseRadiusSquared) {
                angularContributionToChildLinear = angularContributionToChildLinear * (float)(sqrt((double)localPoseRadiusSquared) / sqrt((double)contributionLengthSquared));
            }
            childVelocity.Linear = velocity.Linear + angularContributionToChildLinear;
            childPose.Position = childPose.Position + pose.Position;
            AddCompoundChild(bodyIndex, child.ShapeType, child.ShapeIndex, childPose, childVelocity);
        }
    }
    void ExecuteCompoundBatch(int typeId) {  // :268-287
        BoundingBoxBatch& batch = batches[typeId];
        Vector3 minValue = Broadcast3(3.402823466e+38f);
        Vector3 maxValue = Broadcast3(-3.402823466e+38f);
        for (int i = 0; i < batch.Count; ++i) {
            int bodyIndex = batch.Continuations[i].BodyIndex();
            const MotionState& motionState = batch.MotionStates[i];
            Collidable& collidable = world.collidables[bodyIndex];
            collidable.SpeculativeMargin = 0;
            world.boundsMin[bodyIndex] = minValue;
            world.boundsMax[bodyIndex] = maxValue;
            AddChildBoundsToBatcher(world.shapes.compounds[batch.ShapeIndices[i]].Children, motionState.Pose, motionState.Velocity, bodyIndex);
        }
    }

    void ComputeBounds(int typeIndex) {  // shapes[typeIndex].ComputeBounds(ref batcher): each ShapeBatch subclass calls its Execute*Batch (ShapeBatch.cs)
        switch (typeIndex) {
            case SphereId: ExecuteConvexBatch<Sphere, SphereWide>(typeIndex, world.shapes.spheres); break;
            case CapsuleId: ExecuteConvexBatch<Capsule, CapsuleWide>(typeIndex, world.shapes.capsules); break;
            case BoxId: ExecuteConvexBatch<Box, BoxWide>(typeIndex, world.shapes.boxes); break;
            case TriangleId: ExecuteConvexBatch<Triangle, TriangleWide>(typeIndex, world.shapes.triangles); break;
            case CylinderId: ExecuteConvexBatch<Cylinder, CylinderWide>(typeIndex, world.shapes.cylinders); break;
            case ConvexHullId: ExecuteConvexBatch<ConvexHull, ConvexHullWide>(typeIndex, world.shapes.hulls); break;
            case CompoundId: case BigCompoundId: ExecuteCompoundBatch(typeIndex); break;
            case MeshId: ExecuteHomogeneousCompoundBatch(typeIndex); break;
        }
    }
    void Add(int typeIndex, int shapeIndex, const RigidPose& pose, const BodyVelocity& velocity, BoundsContinuation continuation) {  // :290-313
        BoundingBoxBatch& batchSlot = batches[typeIndex];
        if (!batchSlot.Allocated) {
            batchSlot.Allocate(CollidablesPerFlush);
            if (typeIndex < minimumBatchIndex) minimumBatchIndex = typeIndex;
            if (typeIndex > maximumBatchIndex) maximumBatchIndex = typeIndex;
        }
        batchSlot.Add(shapeIndex, pose, velocity, continuation);
        if (batchSlot.Count == CollidablesPerFlush) {
            ComputeBounds(typeIndex);
            batchSlot.Count = 0;
        }
    }
    void Add(int bodyIndex, const RigidPose& pose, const BodyVelocity& velocity, const Collidable& collidable) {  // :316-328
        if (collidable.ShapeType >= 0) Add(collidable.ShapeType, collidable.ShapeIndex, pose, velocity, BoundsContinuation::CreateContinuation(bodyIndex));
    }
    void Flush() {  // :336-372: compounds first (reverse order), so that the children they add are flushed by the convex batches after them
        for (int i = maximumBatchIndex; i >= minimumBatchIndex; --i) {
            BoundingBoxBatch& batch = batches[i];
            if (batch.Count > 0) ComputeBounds(i);
            batch.Count = 0;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------- PoseIntegrator.cs:286-370
static inline void UpdateSleepCandidacy(float velocityHeuristic, BodyActivity& activity) {  // :287-305
    if (velocityHeuristic > activity.SleepThreshold) {
        activity.TimestepsUnderThresholdCount = 0;
        activity.SleepCandidate = false;
    } else {
        if (activity.TimestepsUnderThresholdCount < 255) {  // byte.MaxValue
            ++activity.TimestepsUnderThresholdCount;
            if (activity.TimestepsUnderThresholdCount >= activity.MinimumTimestepsUnderThreshold) activity.SleepCandidate = true;
        }
    }
}

static inline void PredictBoundingBoxes(Bodies& bodies, PoseIntegratorCallbacks& Callbacks, World& world, int startBundleIndex, int endBundleIndex, float dt,
                                        BoundingBoxBatcher& boundingBoxBatcher, int workerIndex) {  // :307-370
    VI laneIndexOffsets = VI{0, 1, 2, 3, 4, 5, 6, 7};
    VF dtWide = vf(dt);
    int bodyCount = bodies.count;
    for (int bundleIndex = startBundleIndex; bundleIndex < endBundleIndex; ++bundleIndex) {
        int bundleStartBodyIndex = bundleIndex * W;
        int countInBundle = bodyCount - bundleStartBodyIndex;
        if (countInBundle > W) countInBundle = W;
        VI laneIndices = vi(bundleStartBodyIndex) + laneIndexOffsets;
        VI maskForCountInBundle = (VI)(laneIndexOffsets < vi(countInBundle));  // BundleIndexing.CreateMaskForCountInBundle
        Vector3Wide position;
        QuaternionWide orientation;
        BodyVelocityWide velocity;
        BodyInertiaWide inertia;
        // (the reference gathers lanes past the last body out of the buffer's slack; they are masked out of everything below, so they are read as empty lanes here)
        bodies.GatherState(BitwiseOr(laneIndices, OnesComplement(maskForCountInBundle)), false, position, orientation, velocity, inertia);
        VI integrationMask;
        if (Callbacks.IntegrateVelocityForKinematics) {
            integrationMask = maskForCountInBundle;
        } else {
            integrationMask = AndNot(maskForCountInBundle, Bodies::IsKinematic(inertia));
        }
        laneIndices = BitwiseOr(OnesComplement(integrationMask), laneIndices);
        VF sleepEnergy = (velocity.Linear.X * velocity.Linear.X + velocity.Linear.Y * velocity.Linear.Y + velocity.Linear.Z * velocity.Linear.Z) +
                         (velocity.Angular.X * velocity.Angular.X + velocity.Angular.Y * velocity.Angular.Y + velocity.Angular.Z * velocity.Angular.Z);  // LengthSquared() + LengthSquared()
        if (LessThanAny(integrationMask, vi(0)))
            Callbacks.IntegrateVelocity(laneIndices, position, orientation, inertia, integrationMask, workerIndex, dtWide, velocity);  // nothing masks the result here
        for (int i = 0; i < countInBundle; ++i) {
            int bodyIndex = i + bundleStartBodyIndex;
            UpdateSleepCandidacy(sleepEnergy[i], world.activities[bodyIndex]);
            RigidPose bodyPose;
            bodyPose.Position = {position.X[i], position.Y[i], position.Z[i]};
            bodyPose.Orientation = {orientation.X[i], orientation.Y[i], orientation.Z[i], orientation.W[i]};
            BodyVelocity bodyVelocity;
            bodyVelocity.Linear = {velocity.Linear.X[i], velocity.Linear.Y[i], velocity.Linear.Z[i]};
            bodyVelocity.Angular = {velocity.Angular.X[i], velocity.Angular.Y[i], velocity.Angular.Z[i]};
            boundingBoxBatcher.Add(bodyIndex, bodyPose, bodyVelocity, world.collidables[bodyIndex]);
        }
    }
}

}  // namespace bounds
}  // namespace wide
