// Device-side bounding-box prediction and sleep candidacy: the per-body work of PoseIntegrator.PredictBoundingBoxes, the stage
// that runs right before collision detection and streams the same body array the solver leaves in HBM.
#pragma once

#include "bepu_device_math.h"

namespace bd {

// ======================================================================================
// PoseIntegrator.PredictBoundingBoxes for one body (PoseIntegrator.cs:287-370) with BoundingBoxBatcher.ExecuteConvexBatch (BoundingBoxBatcher.cs:142-223)
// for the five primitive convex shapes and BoundingBoxHelpers (BoundingBoxHelpers.cs:12-61). One lane = one body.
// ======================================================================================
enum ShapeType { kShapeNone = -1, kShapeSphere = 0, kShapeCapsule = 1, kShapeBox = 2, kShapeTriangle = 3, kShapeCylinder = 4, kShapeConvexHull = 5 };  // Sphere.Id ... ConvexHull.Id
// Convex hull point sets (ConvexHull.Points without the bundle padding, which repeats real points): hull h owns points [begin[h], begin[h + 1]).
struct HullTable { const float* points; const int* begin; int count; };

BD_FN V3 transformUnitY(Q r) {  // QuaternionWide.cs:389-405
    float x2 = r.x + r.x, y2 = r.y + r.y, z2 = r.z + r.z;
    float xx2 = r.x * x2, xy2 = r.x * y2, yz2 = r.y * z2, zz2 = r.z * z2, wx2 = r.w * x2, wz2 = r.w * z2;
    return {xy2 - wz2, 1.0f - xx2 - zz2, yz2 + wx2};
}

// TShapeWide.GetBounds: local bounds around the body's position, the largest distance of any point from it, and how far a rotation can move a point outward.
BD_FN bool shapeBounds(int type, const float* s, Q orientation, const HullTable& hulls, float& maximumRadius, float& maximumAngularExpansion, V3& mn, V3& mx) {
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

BD_FN float angularBoundsExpansion(float angularSpeed, float dt, float maximumRadius, float maximumAngularExpansion) {  // BoundingBoxHelpers.cs:12-47
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
BD_FN int updateSleepCandidacy(float velocityHeuristic, float sleepThreshold, int minimumTimestepsUnderThreshold, int activity) {
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
BD_FN void predictBounds(V3 position, Q orientation, const BodyVel& velocity, float sleepEnergy, float dt, const CollidableIn& c, const HullTable& hulls, PredictedBounds& out) {
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
// Compounds (Compound.Id 6, BigCompound.Id 7) and meshes (Mesh.Id 8): one lane still owns one body and walks its children / triangles.
// ======================================================================================
enum { kShapeCompound = 6, kShapeBigCompound = 7, kShapeMesh = 8 };
struct CompoundChildIn { int shape_type; float shape[9]; float local_position[3]; float local_orientation[4]; };  // mirrors bepuhip_compound_child
struct ShapeTables {
    HullTable hulls;
    const CompoundChildIn* children; const int* child_begin; int compound_count;               // compound k: children [child_begin[k], child_begin[k + 1])
    const float* triangles; const int* triangle_begin; const float* mesh_scales; int mesh_count;  // mesh m: 9-float triangles [triangle_begin[m], triangle_begin[m + 1]), scale xyz
};

struct Box3 { V3 lo, hi; };
BD_FN Box3 emptyBox() { return {{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f}}; }
BD_FN V3 min3(V3 a, V3 b) { return {vmin(a.x, b.x), vmin(a.y, b.y), vmin(a.z, b.z)}; }  // Vector3.Min / Vector3.Max, first operand kept on ties like minps
BD_FN V3 max3(V3 a, V3 b) { return {vmax(a.x, b.x), vmax(a.y, b.y), vmax(a.z, b.z)}; }
BD_FN float largerMargin(float a, float b) {  // MathF.Max
    if (a != a) return a;
    if (b != b) return b;
    if (a == b) return (__float_as_uint(a) >> 31) ? b : a;
    return a > b ? a : b;
}
BD_FN float smallerMargin(float a, float b) {  // MathF.Min
    if (a != a) return a;
    if (b != b) return b;
    if (a == b) return (__float_as_uint(a) >> 31) ? a : b;
    return a < b ? a : b;
}

// The velocity expansion every Execute*Batch ends with, given the local box and the two radii of the shape. `wideClamp` = the Vector<float> form of the margin
// clamp (convex batches, BoundingBoxBatcher.cs:186), otherwise the MathF form of the homogeneous compound batch (:253).
BD_FN float expandByVelocity(const BodyVel& velocity, float dt, float maximumRadius, float maximumAngularExpansion, float minimumMargin, float maximumMargin, int allowBeyondMargin,
                             bool wideClamp, V3 position, Box3& box) {
    const float angular = angularBoundsExpansion(length(velocity.ang), dt, maximumRadius, maximumAngularExpansion);
    float margin = length(velocity.lin) * dt + angular;
    margin = wideClamp ? vmax(minimumMargin, vmin(maximumMargin, margin)) : largerMargin(minimumMargin, smallerMargin(maximumMargin, margin));
    const float limit = allowBeyondMargin != 0 ? 3.402823466e+38f : margin;
    const V3 swept = scale(velocity.lin, dt);
    const V3 grow_lo = {vmax(-limit, vmin(0.0f, swept.x) - angular), vmax(-limit, vmin(0.0f, swept.y) - angular), vmax(-limit, vmin(0.0f, swept.z) - angular)};
    const V3 grow_hi = {vmin(limit, vmax(0.0f, swept.x) + angular), vmin(limit, vmax(0.0f, swept.y) + angular), vmin(limit, vmax(0.0f, swept.z) + angular)};
    box.lo = add(position, add(box.lo, grow_lo));
    box.hi = add(position, add(box.hi, grow_hi));
    return margin;
}

// Compound.AddChildBoundsToBatcher (Compound.cs:198-221) feeding ExecuteConvexBatch with CompoundChild continuations (BoundingBoxBatcher.cs:142-223, merge :203-209),
// after ExecuteCompoundBatch reset the body's box and margin (:268-287).
BD_FN void compoundBounds(V3 position, Q orientation, const BodyVel& velocity, float dt, const CollidableIn& c, const ShapeTables& tables, PredictedBounds& out) {
    const int compound = (int)c.shape[0];
    Box3 whole = emptyBox();
    float margin = 0.0f;
    const int last = tables.child_begin[compound + 1];
    for (int k = tables.child_begin[compound]; k < last; ++k) {
        const CompoundChildIn* child = tables.children + k;
        const Q local_q = {child->local_orientation[0], child->local_orientation[1], child->local_orientation[2], child->local_orientation[3]};
        const V3 offset = transform(V3{child->local_position[0], child->local_position[1], child->local_position[2]}, orientation);  // GetRotatedChildPose, Compound.cs:153-157
        const Q child_q = concatenate(local_q, orientation);
        V3 swing = cross(velocity.ang, offset);  // the child's share of the angular motion, capped at the length of its offset (:210-216; the ratio is formed in double there)
        const float swing2 = lengthSquared(swing), offset2 = lengthSquared(offset);
        if (swing2 > offset2) swing = scale(swing, (float)(sqrt((double)offset2) / sqrt((double)swing2)));
        const BodyVel child_v = {add(velocity.lin, swing), velocity.ang};
        float shape[9];
        for (int f = 0; f < 9; ++f) shape[f] = child->shape[f];
        float maximumRadius, maximumAngularExpansion;
        Box3 box;
        shapeBounds(child->shape_type, shape, child_q, tables.hulls, maximumRadius, maximumAngularExpansion, box.lo, box.hi);  // children are convex (checked at the boundary)
        const float child_margin = expandByVelocity(child_v, dt, maximumRadius, maximumAngularExpansion, c.minimum_speculative_margin, c.maximum_speculative_margin,
                                                    c.allow_expansion_beyond_speculative_margin, true, add(offset, position), box);
        margin = largerMargin(margin, child_margin);
        whole.lo = min3(whole.lo, box.lo);  // BoundingBox.CreateMerged, BoundingBox.cs:173-177
        whole.hi = max3(whole.hi, box.hi);
    }
    out.min[0] = whole.lo.x; out.min[1] = whole.lo.y; out.min[2] = whole.lo.z;
    out.max[0] = whole.hi.x; out.max[1] = whole.hi.y; out.max[2] = whole.hi.z;
    out.speculative_margin = margin;
}

// ExecuteHomogeneousCompoundBatch (BoundingBoxBatcher.cs:225-266) over Mesh.ComputeBounds (Mesh.cs:232-255).
BD_FN void meshBounds(V3 position, Q orientation, const BodyVel& velocity, float dt, const CollidableIn& c, const ShapeTables& tables, PredictedBounds& out) {
    const int mesh = (int)c.shape[0];
    const float sx = tables.mesh_scales[3 * mesh], sy = tables.mesh_scales[3 * mesh + 1], sz = tables.mesh_scales[3 * mesh + 2];
    const M3 basis = createFromQuaternion(orientation);
    Box3 box = emptyBox();
    const int last = tables.triangle_begin[mesh + 1];
    for (int t = tables.triangle_begin[mesh]; t < last; ++t) {
        const float* v = tables.triangles + 9 * (size_t)t;
        const V3 a = transform(V3{sx * v[0], sy * v[1], sz * v[2]}, basis), b = transform(V3{sx * v[3], sy * v[4], sz * v[5]}, basis), cc = transform(V3{sx * v[6], sy * v[7], sz * v[8]}, basis);
        box.lo = min3(min3(a, b), min3(cc, box.lo));  // the reference's pairing: Min(Min(a, b), Min(c, min)) (:248-253)
        box.hi = max3(max3(a, b), max3(cc, box.hi));
    }
    const V3 abs_lo = {vabs(box.lo.x), vabs(box.lo.y), vabs(box.lo.z)}, abs_hi = {vabs(box.hi.x), vabs(box.hi.y), vabs(box.hi.z)};
    const float maximumRadius = length(max3(abs_lo, abs_hi));
    const V3 inner = min3(abs_lo, abs_hi);
    const float maximumAngularExpansion = maximumRadius - vmin(inner.x, vmin(inner.y, inner.z));
    const float margin = expandByVelocity(velocity, dt, maximumRadius, maximumAngularExpansion, c.minimum_speculative_margin, c.maximum_speculative_margin,
                                          c.allow_expansion_beyond_speculative_margin, false, position, box);
    out.min[0] = box.lo.x; out.min[1] = box.lo.y; out.min[2] = box.lo.z;
    out.max[0] = box.hi.x; out.max[1] = box.hi.y; out.max[2] = box.hi.z;
    out.speculative_margin = margin;
}

// ---- a whole wave on one body: compounds, meshes and large hulls (predict_heavy_bounds_kernel) ----
// The lanes share the children / triangles / points of the body and merge what they found. The serial text above fixes which of two EQUAL candidates a
// minimum or maximum keeps (it matters for the sign of a zero only): vmin(acc, p) keeps the later one, the mesh pairing keeps the earlier triangle and c over b
// over a. A candidate therefore carries a key, larger = kept on ties, and the merge is the same in a lane's loop and across lanes, so the result does not depend
// on how the work was dealt out.
struct Keyed { float v; int k; };
BD_FN void keepSmaller(Keyed& acc, float v, int k) { if (v < acc.v || (v == acc.v && k > acc.k)) { acc.v = v; acc.k = k; } }
BD_FN void keepLarger(Keyed& acc, float v, int k) { if (v > acc.v || (v == acc.v && k > acc.k)) { acc.v = v; acc.k = k; } }
struct KeyedBox {
    Keyed lo[3], hi[3];
    __device__ __forceinline__ void reset(int seedKey) {
        for (int a = 0; a < 3; ++a) { lo[a] = {3.402823466e+38f, seedKey}; hi[a] = {-3.402823466e+38f, seedKey}; }
    }
    __device__ __forceinline__ void take(V3 lower, V3 upper, int key) {
        keepSmaller(lo[0], lower.x, key); keepSmaller(lo[1], lower.y, key); keepSmaller(lo[2], lower.z, key);
        keepLarger(hi[0], upper.x, key); keepLarger(hi[1], upper.y, key); keepLarger(hi[2], upper.z, key);
    }
};
#if defined(__HIPCC__)
__device__ __forceinline__ void mergeAcrossWave(KeyedBox& box) {  // butterfly over the 64 lanes; every lane ends with the wave's result
    for (int step = 32; step > 0; step >>= 1)
        for (int a = 0; a < 3; ++a) {
            const float lv = __shfl_xor(box.lo[a].v, step), hv = __shfl_xor(box.hi[a].v, step);
            const int lk = __shfl_xor(box.lo[a].k, step), hk = __shfl_xor(box.hi[a].k, step);
            keepSmaller(box.lo[a], lv, lk);
            keepLarger(box.hi[a], hv, hk);
        }
}
__device__ __forceinline__ float largestAcrossWave(float v, bool marginRule) {
    for (int step = 32; step > 0; step >>= 1) {
        const float other = __shfl_xor(v, step);
        v = marginRule ? largerMargin(v, other) : vmax(v, other);
    }
    return v;
}
// One wave, one body of shape type ConvexHull (many points), Compound, BigCompound or Mesh; `lane` = 0..63. The result is written by every lane into `out` (callers keep lane 0's).
__device__ __forceinline__ void heavyBounds(int lane, V3 position, Q orientation, const BodyVel& velocity, float dt, const CollidableIn& c, const ShapeTables& tables, PredictedBounds& out) {
    KeyedBox found;
    float margin = 0.0f;
    Box3 box;
    const int entry = (int)c.shape[0];
    if (c.shape_type == kShapeConvexHull) {  // shapeBounds' hull case with the points dealt to the lanes: vmin(acc, p) keeps the later point
        found.reset(-1);
        const M3 basis = createFromQuaternion(orientation);
        float farthest2 = 0.0f;
        for (int j = tables.hulls.begin[entry] + lane; j < tables.hulls.begin[entry + 1]; j += 64) {
            const V3 local = {tables.hulls.points[3 * (size_t)j], tables.hulls.points[3 * (size_t)j + 1], tables.hulls.points[3 * (size_t)j + 2]};
            const V3 p = transform(local, basis);
            farthest2 = vmax(lengthSquared(local), farthest2);
            found.take(p, p, j);
        }
        mergeAcrossWave(found);
        const float maximumRadius = sqrtf(largestAcrossWave(farthest2, false));
        box = {{found.lo[0].v, found.lo[1].v, found.lo[2].v}, {found.hi[0].v, found.hi[1].v, found.hi[2].v}};
        margin = expandByVelocity(velocity, dt, maximumRadius, maximumRadius, c.minimum_speculative_margin, c.maximum_speculative_margin, c.allow_expansion_beyond_speculative_margin, true,
                                  position, box);
    } else if (c.shape_type == kShapeMesh) {  // meshBounds with the triangles dealt to the lanes: the earlier triangle is kept, within one c over b over a
        found.reset(0x7fffffff);
        const float sx = tables.mesh_scales[3 * entry], sy = tables.mesh_scales[3 * entry + 1], sz = tables.mesh_scales[3 * entry + 2];
        const M3 basis = createFromQuaternion(orientation);
        const int first = tables.triangle_begin[entry];
        for (int t = first + lane; t < tables.triangle_begin[entry + 1]; t += 64) {
            const float* v = tables.triangles + 9 * (size_t)t;
            const int key = -3 * (t - first);
            const V3 a = transform(V3{sx * v[0], sy * v[1], sz * v[2]}, basis), b = transform(V3{sx * v[3], sy * v[4], sz * v[5]}, basis), cc = transform(V3{sx * v[6], sy * v[7], sz * v[8]}, basis);
            found.take(a, a, key);
            found.take(b, b, key + 1);
            found.take(cc, cc, key + 2);
        }
        mergeAcrossWave(found);
        box = {{found.lo[0].v, found.lo[1].v, found.lo[2].v}, {found.hi[0].v, found.hi[1].v, found.hi[2].v}};
        const V3 abs_lo = {vabs(box.lo.x), vabs(box.lo.y), vabs(box.lo.z)}, abs_hi = {vabs(box.hi.x), vabs(box.hi.y), vabs(box.hi.z)};
        const float maximumRadius = length(max3(abs_lo, abs_hi));
        const V3 inner = min3(abs_lo, abs_hi);
        margin = expandByVelocity(velocity, dt, maximumRadius, maximumRadius - vmin(inner.x, vmin(inner.y, inner.z)), c.minimum_speculative_margin, c.maximum_speculative_margin,
                                  c.allow_expansion_beyond_speculative_margin, false, position, box);
    } else {  // compoundBounds with the children dealt to the lanes: vmin(acc, child) keeps the later child
        found.reset(-1);
        for (int k = tables.child_begin[entry] + lane; k < tables.child_begin[entry + 1]; k += 64) {
            const CompoundChildIn* child = tables.children + k;
            const Q local_q = {child->local_orientation[0], child->local_orientation[1], child->local_orientation[2], child->local_orientation[3]};
            const V3 offset = transform(V3{child->local_position[0], child->local_position[1], child->local_position[2]}, orientation);
            const Q child_q = concatenate(local_q, orientation);
            V3 swing = cross(velocity.ang, offset);
            const float swing2 = lengthSquared(swing), offset2 = lengthSquared(offset);
            if (swing2 > offset2) swing = scale(swing, (float)(sqrt((double)offset2) / sqrt((double)swing2)));
            const BodyVel child_v = {add(velocity.lin, swing), velocity.ang};
            float shape[9];
            for (int f = 0; f < 9; ++f) shape[f] = child->shape[f];
            float maximumRadius, maximumAngularExpansion;
            Box3 child_box;
            shapeBounds(child->shape_type, shape, child_q, tables.hulls, maximumRadius, maximumAngularExpansion, child_box.lo, child_box.hi);
            const float child_margin = expandByVelocity(child_v, dt, maximumRadius, maximumAngularExpansion, c.minimum_speculative_margin, c.maximum_speculative_margin,
                                                        c.allow_expansion_beyond_speculative_margin, true, add(offset, position), child_box);
            margin = largerMargin(margin, child_margin);
            found.take(child_box.lo, child_box.hi, k);
        }
        mergeAcrossWave(found);
        margin = largestAcrossWave(margin, true);
        box = {{found.lo[0].v, found.lo[1].v, found.lo[2].v}, {found.hi[0].v, found.hi[1].v, found.hi[2].v}};
    }
    out.min[0] = box.lo.x; out.min[1] = box.lo.y; out.min[2] = box.lo.z;
    out.max[0] = box.hi.x; out.max[1] = box.hi.y; out.max[2] = box.hi.z;
    out.speculative_margin = margin;
}
#endif
// Which bodies get a wave of their own: hulls, compounds and meshes with more points / children / triangles than a lane should walk alone.
BD_FN bool isHeavyShape(const CollidableIn& c, const ShapeTables& tables, int threshold) {
    if (c.shape_type < kShapeConvexHull) return false;
    const int entry = (int)c.shape[0];
    const int* begin = c.shape_type == kShapeConvexHull ? tables.hulls.begin : c.shape_type == kShapeMesh ? tables.triangle_begin : tables.child_begin;
    return begin[entry + 1] - begin[entry] > threshold;
}

BD_FN void predictBoundsOfAnyShape(V3 position, Q orientation, const BodyVel& velocity, float sleepEnergy, float dt, const CollidableIn& c, const ShapeTables& tables, PredictedBounds& out) {
    if (c.shape_type < kShapeCompound) return predictBounds(position, orientation, velocity, sleepEnergy, dt, c, tables.hulls, out);
    out.activity = updateSleepCandidacy(sleepEnergy, c.sleep_threshold, c.minimum_timesteps_under_threshold, c.activity);
    if (c.shape_type == kShapeMesh) meshBounds(position, orientation, velocity, dt, c, tables, out);
    else compoundBounds(position, orientation, velocity, dt, c, tables, out);
}

}  // namespace bd
