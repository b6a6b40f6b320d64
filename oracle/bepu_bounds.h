// TEST INFRASTRUCTURE — CPU restatement of PoseIntegrator.PredictBoundingBoxes for one body (parity unpinned, see bepu_oracle.cpp); derived from the
// device text with tools/port_constraints_to_oracle.py, pinned by the behavioural tests in tests/test_bounds.py.
#pragma once

#include <cmath>

#include "bepu_math.h"

namespace bo {

// ======================================================================================
// PoseIntegrator.PredictBoundingBoxes for one body (PoseIntegrator.cs:287-370) with BoundingBoxBatcher.ExecuteConvexBatch (BoundingBoxBatcher.cs:142-223)
// for the five primitive convex shapes and BoundingBoxHelpers (BoundingBoxHelpers.cs:12-61). One lane = one body.
// ======================================================================================
enum ShapeType { kShapeNone = -1, kShapeSphere = 0, kShapeCapsule = 1, kShapeBox = 2, kShapeTriangle = 3, kShapeCylinder = 4, kShapeConvexHull = 5 };  // Sphere.Id ... ConvexHull.Id
// Convex hull point sets (ConvexHull.Points without the bundle padding, which repeats real points): hull h owns points [begin[h], begin[h + 1]).
struct HullTable { const float* points; const int* begin; int count; };

static inline V3 transformUnitY(Q r) {  // QuaternionWide.cs:389-405
    float x2 = r.x + r.x, y2 = r.y + r.y, z2 = r.z + r.z;
    float xx2 = r.x * x2, xy2 = r.x * y2, yz2 = r.y * z2, zz2 = r.z * z2, wx2 = r.w * x2, wz2 = r.w * z2;
    return {xy2 - wz2, 1.0f - xx2 - zz2, yz2 + wx2};
}

// TShapeWide.GetBounds: local bounds around the body's position, the largest distance of any point from it, and how far a rotation can move a point outward.
static inline bool shapeBounds(int type, const float* s, Q orientation, const HullTable& hulls, float& maximumRadius, float& maximumAngularExpansion, V3& mn, V3& mx) {
    switch (type) {
        case kShapeSphere: {  // Sphere.cs:149-160
            maximumRadius = 0.0f; maximumAngularExpansion = 0.0f;
            mx = {s[0], s[0], s[0]};
            mn = {-s[0], -s[0], -s[0]};
            return true;
        }
        case kShapeCapsule: {  // Capsule.cs:226-239  {Radius, HalfLength}
            V3 segmentOffset = transformUnitY(orientation);
            segmentOffset = scale(segmentOffset, s[1]);
            segmentOffset = {vabs(segmentOffset.x), vabs(segmentOffset.y), vabs(segmentOffset.z)};
            mx = {segmentOffset.x + s[0], segmentOffset.y + s[0], segmentOffset.z + s[0]};
            mn = {-mx.x, -mx.y, -mx.z};
            maximumRadius = s[1] + s[0];
            maximumAngularExpansion = s[1];
            return true;
        }
        case kShapeBox: {  // Box.cs:211-222  {HalfWidth, HalfHeight, HalfLength}
            M3 basis = createFromQuaternion(orientation);
            mx.x = vabs(s[0] * basis.X.x) + vabs(s[1] * basis.Y.x) + vabs(s[2] * basis.Z.x);
            mx.y = vabs(s[0] * basis.X.y) + vabs(s[1] * basis.Y.y) + vabs(s[2] * basis.Z.y);
            mx.z = vabs(s[0] * basis.X.z) + vabs(s[1] * basis.Y.z) + vabs(s[2] * basis.Z.z);
            mn = {-mx.x, -mx.y, -mx.z};
            maximumRadius = sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
            maximumAngularExpansion = maximumRadius - vmin(s[2], vmin(s[1], s[2]));  // as written in the reference: HalfLength appears twice, HalfWidth not at all (:221)
            return true;
        }
        case kShapeTriangle: {  // Triangle.cs:203-221  {A, B, C}
            M3 basis = createFromQuaternion(orientation);
            V3 a = {s[0], s[1], s[2]}, b = {s[3], s[4], s[5]}, c = {s[6], s[7], s[8]};
            V3 worldA = transform(a, basis), worldB = transform(b, basis), worldC = transform(c, basis);
            mn = {vmin(worldA.x, vmin(worldB.x, worldC.x)), vmin(worldA.y, vmin(worldB.y, worldC.y)), vmin(worldA.z, vmin(worldB.z, worldC.z))};
            mx = {vmax(worldA.x, vmax(worldB.x, worldC.x)), vmax(worldA.y, vmax(worldB.y, worldC.y)), vmax(worldA.z, vmax(worldB.z, worldC.z))};
            maximumRadius = sqrtf(vmax(lengthSquared(a), vmax(lengthSquared(b), lengthSquared(c))));
            maximumAngularExpansion = maximumRadius;
            return true;
        }
        case kShapeCylinder: {  // Cylinder.cs:222-235  {Radius, HalfLength}
            V3 y = transformUnitY(orientation);
            V3 squared = {1.0f - y.x * y.x, 1.0f - y.y * y.y, 1.0f - y.z * y.z};
            mx.x = vabs(s[1] * y.x) + sqrtf(vmax(0.0f, squared.x)) * s[0];
            mx.y = vabs(s[1] * y.y) + sqrtf(vmax(0.0f, squared.y)) * s[0];
            mx.z = vabs(s[1] * y.z) + sqrtf(vmax(0.0f, squared.z)) * s[0];
            mn = {-mx.x, -mx.y, -mx.z};
            maximumRadius = sqrtf(s[1] * s[1] + s[0] * s[0]);
            maximumAngularExpansion = maximumRadius - vmin(s[1], s[0]);
            return true;
        }
        case kShapeConvexHull: {  // ConvexHull.cs:319-364: every point rotated, componentwise min / max; the farthest point bounds both expansions. s[0] = hull index.
            if (!hulls.points) return false;
            const int hull = (int)s[0];
            M3 basis = createFromQuaternion(orientation);
            mn = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
            mx = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
            float maximumRadiusSquared = 0.0f;
            for (int j = hulls.begin[hull]; j < hulls.begin[hull + 1]; ++j) {
                const V3 local = {hulls.points[3 * (size_t)j], hulls.points[3 * (size_t)j + 1], hulls.points[3 * (size_t)j + 2]};
                const V3 p = transform(local, basis);  // Matrix3x3Wide.TransformWithoutOverlap
                maximumRadiusSquared = vmax(lengthSquared(local), maximumRadiusSquared);
                mn = {vmin(mn.x, p.x), vmin(mn.y, p.y), vmin(mn.z, p.z)};
                mx = {vmax(mx.x, p.x), vmax(mx.y, p.y), vmax(mx.z, p.z)};
            }
            maximumRadius = sqrtf(maximumRadiusSquared);
            maximumAngularExpansion = maximumRadius;
            return true;
        }
        default: return false;
    }
}

static inline float angularBoundsExpansion(float angularSpeed, float dt, float maximumRadius, float maximumAngularExpansion) {  // BoundingBoxHelpers.cs:12-47
    float a = vmin(angularSpeed * dt, 3.14159274f / 3.0f);
    float a2 = a * a;
    float a4 = a2 * a2;
    float a6 = a4 * a2;
    float cosAngleMinusOne = a2 * (-1.0f / 2.0f) + a4 * (1.0f / 24.0f) - a6 * (1.0f / 720.0f);
    return vmin(maximumAngularExpansion, sqrtf(-2.0f * maximumRadius * maximumRadius * cosAngleMinusOne));
}

struct CollidableIn {  // mirrors bepuhip_collidable (include/bepuhip.h), 16 words
    int shape_type; float shape[9];
    float minimum_speculative_margin, maximum_speculative_margin; int allow_expansion_beyond_speculative_margin;
    float sleep_threshold; int minimum_timesteps_under_threshold; int activity;  // activity: bits 0-7 TimestepsUnderThresholdCount, bit 8 SleepCandidate
};
struct PredictedBounds { float min[3]; float speculative_margin; float max[3]; int activity; };

// UpdateSleepCandidacy, PoseIntegrator.cs:287-305 (the count is a byte in the reference and stops at 255).
static inline int updateSleepCandidacy(float velocityHeuristic, float sleepThreshold, int minimumTimestepsUnderThreshold, int activity) {
    int count = activity & 0xFF;
    bool candidate = (activity & 0x100) != 0;
    if (velocityHeuristic > sleepThreshold) {
        count = 0;
        candidate = false;
    } else if (count < 255) {
        ++count;
        if (count >= minimumTimestepsUnderThreshold) candidate = true;
    }
    return count | (candidate ? 0x100 : 0);
}

// `velocity` is the body's velocity after the integration callback ran on it for the full dt (only used for the prediction, never stored: :331-333);
// `sleepEnergy` was taken from the stored velocity (:329).
static inline void predictBounds(V3 position, Q orientation, const BodyVel& velocity, float sleepEnergy, float dt, const CollidableIn& c, const HullTable& hulls, PredictedBounds& out) {
    out.activity = updateSleepCandidacy(sleepEnergy, c.sleep_threshold, c.minimum_timesteps_under_threshold, c.activity);
    float maximumRadius, maximumAngularExpansion; V3 mn, mx;
    if (!shapeBounds(c.shape_type, c.shape, orientation, hulls, maximumRadius, maximumAngularExpansion, mn, mx)) {  // Shape.Exists == false: nothing to bound (BoundingBoxBatcher.cs:326)
        out.min[0] = out.min[1] = out.min[2] = 0.0f; out.max[0] = out.max[1] = out.max[2] = 0.0f; out.speculative_margin = 0.0f;
        return;
    }
    // BoundingBoxBatcher.cs:174-191
    float angularExpansion = angularBoundsExpansion(length(velocity.ang), dt, maximumRadius, maximumAngularExpansion);
    float speculativeMargin = length(velocity.lin) * dt + angularExpansion;
    speculativeMargin = vmax(c.minimum_speculative_margin, vmin(c.maximum_speculative_margin, speculativeMargin));
    float maximumBoundsExpansion = sel(c.allow_expansion_beyond_speculative_margin != 0, 3.402823466e+38f, speculativeMargin);
    // BoundingBoxHelpers.GetBoundsExpansion :51-60
    V3 linearDisplacement = scale(velocity.lin, dt);
    V3 minExpansion = {vmin(0.0f, linearDisplacement.x), vmin(0.0f, linearDisplacement.y), vmin(0.0f, linearDisplacement.z)};
    V3 maxExpansion = {vmax(0.0f, linearDisplacement.x), vmax(0.0f, linearDisplacement.y), vmax(0.0f, linearDisplacement.z)};
    minExpansion = {minExpansion.x - angularExpansion, minExpansion.y - angularExpansion, minExpansion.z - angularExpansion};
    maxExpansion = {maxExpansion.x + angularExpansion, maxExpansion.y + angularExpansion, maxExpansion.z + angularExpansion};
    minExpansion = {vmax(-maximumBoundsExpansion, minExpansion.x), vmax(-maximumBoundsExpansion, minExpansion.y), vmax(-maximumBoundsExpansion, minExpansion.z)};
    maxExpansion = {vmin(maximumBoundsExpansion, maxExpansion.x), vmin(maximumBoundsExpansion, maxExpansion.y), vmin(maximumBoundsExpansion, maxExpansion.z)};
    V3 lo = add(position, add(mn, minExpansion));
    V3 hi = add(position, add(mx, maxExpansion));
    out.min[0] = lo.x; out.min[1] = lo.y; out.min[2] = lo.z;
    out.max[0] = hi.x; out.max[1] = hi.y; out.max[2] = hi.z;
    out.speculative_margin = speculativeMargin;
}


// ======================================================================================
// Compounds (Compound.Id 6, BigCompound.Id 7) and meshes (Mesh.Id 8). Written from BoundingBoxBatcher.cs / Compound.cs / Mesh.cs on their own: the device
// text (bepu_device_bounds.h) was written separately and is shaped differently (one lane walks the children through a shared convex helper; here every child
// goes through predictBounds() as the collidable the batcher makes of it).
// ======================================================================================
enum { kShapeCompound = 6, kShapeBigCompound = 7, kShapeMesh = 8 };
struct CompoundChildIn { int shape_type; float shape[9]; float local_position[3]; float local_orientation[4]; };  // mirrors bepuhip_compound_child (CompoundChild: Compound.cs:13-40)
struct CompoundTable { const CompoundChildIn* children; const int* begin; int count; };                          // compound k owns children [begin[k], begin[k + 1])
struct MeshTable { const float* triangles; const int* begin; const float* scales; int count; };                  // mesh m: triangles [begin[m], begin[m + 1]) of 9 floats (A, B, C), scales[3m..3m+2]
struct ShapeTables { HullTable hulls; CompoundTable compounds; MeshTable meshes; };

static inline float mathfMax(float a, float b) {  // System.MathF.Max: NaN wins, +0 over -0
    if (a != a) return a;
    if (b != b) return b;
    if (a == b) return std::signbit(a) ? b : a;
    return a > b ? a : b;
}
static inline float mathfMin(float a, float b) {  // System.MathF.Min: NaN wins, -0 under +0
    if (a != a) return a;
    if (b != b) return b;
    if (a == b) return std::signbit(a) ? a : b;
    return a < b ? a : b;
}

// BoundingBoxBatcher.ExecuteCompoundBatch (BoundingBoxBatcher.cs:268-287): margin 0, box (MaxValue, -MaxValue), then every child is handed to the batcher by
// Compound.AddChildBoundsToBatcher (Compound.cs:198-221) with the pose and the velocity the compound's motion gives it, and comes back through
// ExecuteConvexBatch with continuation.CompoundChild set (:203-209): the body's margin is the largest child margin, its box the union of the child boxes.
// Both merges are order-free for finite numbers, so the flush order (16 children per type) does not matter.
static inline void predictCompoundBounds(V3 position, Q orientation, const BodyVel& velocity, float dt, const CollidableIn& body, const ShapeTables& tables, PredictedBounds& out) {
    const int compound = (int)body.shape[0];
    float speculativeMargin = 0.0f;                                                  // :279
    V3 mn = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};                 // :281
    V3 mx = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};              // :282
    for (int k = tables.compounds.begin[compound]; k < tables.compounds.begin[compound + 1]; ++k) {
        const CompoundChildIn& child = tables.compounds.children[k];
        const V3 localPosition = {child.local_position[0], child.local_position[1], child.local_position[2]};
        const Q localOrientation = {child.local_orientation[0], child.local_orientation[1], child.local_orientation[2], child.local_orientation[3]};
        // Compound.GetRotatedChildPose (Compound.cs:153-157)
        const Q childOrientation = concatenate(localOrientation, orientation);
        V3 childPosition = transform(localPosition, orientation);
        // Compound.cs:210-218: the child's linear velocity is the parent's plus angular x offset, never longer than the offset itself; the quotient is taken in double (:215)
        V3 angularContributionToChildLinear = cross(velocity.ang, childPosition);
        const float contributionLengthSquared = lengthSquared(angularContributionToChildLinear);
        const float localPoseRadiusSquared = lengthSquared(childPosition);
        if (contributionLengthSquared > localPoseRadiusSquared)
            angularContributionToChildLinear = scale(angularContributionToChildLinear, (float)(std::sqrt((double)localPoseRadiusSquared) / std::sqrt((double)contributionLengthSquared)));
        BodyVel childVelocity;
        childVelocity.ang = velocity.ang;
        childVelocity.lin = add(velocity.lin, angularContributionToChildLinear);
        childPosition = add(childPosition, position);
        // ExecuteConvexBatch reads the margins and the continuity of the BODY's collidable (:164-168) and the shape of the child
        CollidableIn asConvex = body;
        asConvex.shape_type = child.shape_type;
        for (int f = 0; f < 9; ++f) asConvex.shape[f] = child.shape[f];
        PredictedBounds childBounds;
        predictBounds(childPosition, childOrientation, childVelocity, 0.0f, dt, asConvex, tables.hulls, childBounds);
        speculativeMargin = mathfMax(speculativeMargin, childBounds.speculative_margin);  // :205
        mn = {vmin(mn.x, childBounds.min[0]), vmin(mn.y, childBounds.min[1]), vmin(mn.z, childBounds.min[2])};  // BoundingBox.CreateMerged (BoundingBox.cs:173-177)
        mx = {vmax(mx.x, childBounds.max[0]), vmax(mx.y, childBounds.max[1]), vmax(mx.z, childBounds.max[2])};
    }
    out.min[0] = mn.x; out.min[1] = mn.y; out.min[2] = mn.z;
    out.max[0] = mx.x; out.max[1] = mx.y; out.max[2] = mx.z;
    out.speculative_margin = speculativeMargin;
}

// BoundingBoxBatcher.ExecuteHomogeneousCompoundBatch (:225-266) with Mesh.ComputeBounds (Mesh.cs:232-255): the scalar (Vector3) path of the reference.
static inline void predictMeshBounds(V3 position, Q orientation, const BodyVel& velocity, float dt, const CollidableIn& body, const ShapeTables& tables, PredictedBounds& out) {
    const int mesh = (int)body.shape[0];
    const V3 meshScale = {tables.meshes.scales[3 * mesh], tables.meshes.scales[3 * mesh + 1], tables.meshes.scales[3 * mesh + 2]};
    const M3 r = createFromQuaternion(orientation);
    V3 mn = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    V3 mx = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int t = tables.meshes.begin[mesh]; t < tables.meshes.begin[mesh + 1]; ++t) {
        const float* tri = tables.meshes.triangles + 9 * (size_t)t;
        const V3 a = transform(V3{meshScale.x * tri[0], meshScale.y * tri[1], meshScale.z * tri[2]}, r);
        const V3 b = transform(V3{meshScale.x * tri[3], meshScale.y * tri[4], meshScale.z * tri[5]}, r);
        const V3 c = transform(V3{meshScale.x * tri[6], meshScale.y * tri[7], meshScale.z * tri[8]}, r);
        const V3 min0 = {vmin(a.x, b.x), vmin(a.y, b.y), vmin(a.z, b.z)};
        const V3 min1 = {vmin(c.x, mn.x), vmin(c.y, mn.y), vmin(c.z, mn.z)};
        const V3 max0 = {vmax(a.x, b.x), vmax(a.y, b.y), vmax(a.z, b.z)};
        const V3 max1 = {vmax(c.x, mx.x), vmax(c.y, mx.y), vmax(c.z, mx.z)};
        mn = {vmin(min0.x, min1.x), vmin(min0.y, min1.y), vmin(min0.z, min1.z)};
        mx = {vmax(max0.x, max1.x), vmax(max0.y, max1.y), vmax(max0.z, max1.z)};
    }
    // :238-245: a plain upper bound of the angular expansion, from the box alone
    const V3 absMin = {vabs(mn.x), vabs(mn.y), vabs(mn.z)};
    const V3 absMax = {vabs(mx.x), vabs(mx.y), vabs(mx.z)};
    const float maximumRadius = length(V3{vmax(absMin.x, absMax.x), vmax(absMin.y, absMax.y), vmax(absMin.z, absMax.z)});
    const V3 minimumComponents = {vmin(absMin.x, absMax.x), vmin(absMin.y, absMax.y), vmin(absMin.z, absMax.z)};
    const float minimumRadius = vmin(minimumComponents.x, vmin(minimumComponents.y, minimumComponents.z));  // MathHelper.Min: a < b ? a : b
    const float maximumAngularExpansion = maximumRadius - minimumRadius;
    // :250-259 (BoundingBoxHelpers.cs:129-148, the scalar overloads: float sqrt through Math.Sqrt(double) rounds to the same float)
    const float angularExpansion = angularBoundsExpansion(length(velocity.ang), dt, maximumRadius, maximumAngularExpansion);
    float speculativeMargin = length(velocity.lin) * dt + angularExpansion;
    speculativeMargin = mathfMax(body.minimum_speculative_margin, mathfMin(body.maximum_speculative_margin, speculativeMargin));
    const float maximumAllowedExpansion = body.allow_expansion_beyond_speculative_margin != 0 ? 3.402823466e+38f : speculativeMargin;
    const V3 linearDisplacement = scale(velocity.lin, dt);
    V3 minExpansion = {vmin(0.0f, linearDisplacement.x) - angularExpansion, vmin(0.0f, linearDisplacement.y) - angularExpansion, vmin(0.0f, linearDisplacement.z) - angularExpansion};
    V3 maxExpansion = {vmax(0.0f, linearDisplacement.x) + angularExpansion, vmax(0.0f, linearDisplacement.y) + angularExpansion, vmax(0.0f, linearDisplacement.z) + angularExpansion};
    minExpansion = {vmax(-maximumAllowedExpansion, minExpansion.x), vmax(-maximumAllowedExpansion, minExpansion.y), vmax(-maximumAllowedExpansion, minExpansion.z)};
    maxExpansion = {vmin(maximumAllowedExpansion, maxExpansion.x), vmin(maximumAllowedExpansion, maxExpansion.y), vmin(maximumAllowedExpansion, maxExpansion.z)};
    const V3 lo = add(position, add(mn, minExpansion));  // :263-265
    const V3 hi = add(position, add(mx, maxExpansion));
    out.min[0] = lo.x; out.min[1] = lo.y; out.min[2] = lo.z;
    out.max[0] = hi.x; out.max[1] = hi.y; out.max[2] = hi.z;
    out.speculative_margin = speculativeMargin;
}

// Shapes[typeIndex].ComputeBounds(ref batcher): convex batches, compound batches and the homogeneous compound batch (mesh) each run their own Execute*Batch.
static inline void predictBoundsOfAnyShape(V3 position, Q orientation, const BodyVel& velocity, float sleepEnergy, float dt, const CollidableIn& c, const ShapeTables& tables,
                                           PredictedBounds& out) {
    if (c.shape_type == kShapeCompound || c.shape_type == kShapeBigCompound) {
        out.activity = updateSleepCandidacy(sleepEnergy, c.sleep_threshold, c.minimum_timesteps_under_threshold, c.activity);
        predictCompoundBounds(position, orientation, velocity, dt, c, tables, out);
    } else if (c.shape_type == kShapeMesh) {
        out.activity = updateSleepCandidacy(sleepEnergy, c.sleep_threshold, c.minimum_timesteps_under_threshold, c.activity);
        predictMeshBounds(position, orientation, velocity, dt, c, tables, out);
    } else {
        predictBounds(position, orientation, velocity, sleepEnergy, dt, c, tables.hulls, out);
    }
}

}  // namespace bo
