// Host-side mirror (C++) of the reference's managed surface for the solver hot path — see bepu_host.h for the file:line map.
#include "bepu_host.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "bepuhip.h"

namespace bepu {

MotorSettings::MotorSettings(float maximumForce, float softness)  // MotorSettings.cs:19,44-48
    : MaximumForce(maximumForce), Damping(softness <= 0 ? 3.402823466e+38f : 1.0f / softness) {}

bool GetTypeInfo(int typeId, TypeInfo& info) {
    // sizeof(TPrestepData)/sizeof(Vector<float>), sizeof(TAccumulatedImpulse)/sizeof(Vector<float>) (TypeProcessor.cs:247);
    // ids: ContactConvexTypes.cs (one-body N -> N-1, two-body N -> 3+N), BatchTypeId constants of the joint types.
    switch (typeId) {
        case 0: info = {1, 11, 4, true}; return true;
        case 1: info = {1, 15, 5, true}; return true;
        case 2: info = {1, 19, 6, true}; return true;
        case 3: info = {1, 23, 7, true}; return true;
        case 4: info = {2, 14, 4, true}; return true;
        case 5: info = {2, 18, 5, true}; return true;
        case 6: info = {2, 22, 6, true}; return true;
        case 7: info = {2, 26, 7, true}; return true;
        case 22: info = {2, 8, 3, false}; return true;   // BallSocket
        case 23: info = {2, 8, 2, false}; return true;   // AngularHinge
        case 25: info = {2, 9, 1, false}; return true;   // SwingLimit
        case 26: info = {2, 14, 1, false}; return true;  // TwistServo
        case 27: info = {2, 12, 1, false}; return true;  // TwistLimit
        case 30: info = {2, 5, 3, false}; return true;   // AngularMotor
        case 46: info = {2, 14, 4, false}; return true;  // SwivelHinge
        case 24: info = {2, 8, 1, false}; return true;   // AngularSwivelHinge
        case 28: info = {2, 9, 1, false}; return true;   // TwistMotor
        case 29: info = {2, 9, 3, false}; return true;   // AngularServo
        case 31: info = {2, 9, 6, false}; return true;   // Weld (Weld.cs:70-81, :224)
        case 33: info = {2, 12, 1, false}; return true;  // DistanceServo
        case 34: info = {2, 10, 1, false}; return true;  // DistanceLimit
        case 41: info = {2, 6, 1, false}; return true;   // AngularAxisMotor
        case 42: info = {1, 9, 3, false}; return true;   // OneBodyAngularServo
        case 43: info = {1, 5, 3, false}; return true;   // OneBodyAngularMotor
        case 44: info = {1, 11, 3, false}; return true;  // OneBodyLinearServo
        case 45: info = {1, 8, 3, false}; return true;   // OneBodyLinearMotor
        case 52: info = {2, 8, 3, false}; return true;   // BallSocketMotor
        case 53: info = {2, 11, 3, false}; return true;  // BallSocketServo
        case 47: info = {2, 14, 5, false}; return true;  // Hinge
        case 8: info = {1, 18, 6, true}; return true;    // Contact2NonconvexOneBody (ContactNonconvexTypes.cs:161-167, :192)
        case 9: info = {1, 25, 9, true}; return true;    // Contact3NonconvexOneBody
        case 10: info = {1, 32, 12, true}; return true;  // Contact4NonconvexOneBody
        case 15: info = {2, 21, 6, true}; return true;   // Contact2Nonconvex (ContactNonconvexTypes.cs:58-66, :109)
        case 16: info = {2, 28, 9, true}; return true;   // Contact3Nonconvex
        case 17: info = {2, 35, 12, true}; return true;  // Contact4Nonconvex
        case 37: info = {2, 14, 2, false}; return true;  // PointOnLineServo
        case 38: info = {2, 15, 1, false}; return true;  // LinearAxisServo
        case 39: info = {2, 12, 1, false}; return true;  // LinearAxisMotor
        case 40: info = {2, 13, 1, false}; return true;  // LinearAxisLimit
        case 54: info = {2, 6, 1, false}; return true;   // AngularAxisGearMotor
        case 32: info = {4, 3, 1, false}; return true;   // VolumeConstraint (four bodies, FourBodyTypeProcessor.cs)
        case 35: info = {2, 3, 1, false}; return true;   // CenterDistanceConstraint
        case 36: info = {3, 3, 1, false}; return true;   // AreaConstraint (three bodies, ThreeBodyTypeProcessor.cs)
        case 55: info = {2, 4, 1, false}; return true;   // CenterDistanceLimit
    }
    return false;
}

// ---- Bodies ----
int32_t Bodies::Add(const BodyDescription& d) {  // BodySet.cs:83-134
    BodyDynamics b;
    std::memset(&b, 0, sizeof(b));
    b.f[0] = d.Pose.Orientation.X; b.f[1] = d.Pose.Orientation.Y; b.f[2] = d.Pose.Orientation.Z; b.f[3] = d.Pose.Orientation.W;
    b.f[4] = d.Pose.Position.X; b.f[5] = d.Pose.Position.Y; b.f[6] = d.Pose.Position.Z;
    b.f[8] = d.Velocity.Linear.X; b.f[9] = d.Velocity.Linear.Y; b.f[10] = d.Velocity.Linear.Z;
    b.f[12] = d.Velocity.Angular.X; b.f[13] = d.Velocity.Angular.Y; b.f[14] = d.Velocity.Angular.Z;
    const Symmetric3x3& t = d.LocalInertia.InverseInertiaTensor;
    b.f[16] = t.XX; b.f[17] = t.YX; b.f[18] = t.YY; b.f[19] = t.ZX; b.f[20] = t.ZY; b.f[21] = t.ZZ; b.f[22] = d.LocalInertia.InverseMass;
    // World inertia slot is zeroed: valid for kinematics forever, refreshed by the solver for dynamics (BodySet.cs:131-134).
    ++TopologyVersion;
    int32_t handle = (int32_t)HandleToIndex.size();
    HandleToIndex.push_back((int32_t)DynamicsState.size());
    IndexToHandle.push_back(handle);
    DynamicsState.push_back(b);
    return handle;
}
bool Bodies::IsKinematic(int index) const {  // Bodies.cs:326-349
    const float* f = DynamicsState[index].f;
    for (int i = 16; i < 23; ++i)
        if (f[i] != 0) return false;
    return true;
}

// ---- TypeBatch / ConstraintBatch ----
int TypeBatch::Allocate(int constraintHandle, const int32_t* encodedBodyIndices) {  // TypeProcessor.cs:314-334
    const int W = kBundleWidth;
    int index = ConstraintCount++;
    if (ConstraintCount > (int)IndexToHandle.size()) IndexToHandle.resize(ConstraintCount);
    IndexToHandle[index] = constraintHandle;
    int bundles = BundleCount();
    if ((size_t)bundles * Info.bodies * W > BodyReferences.size()) {
        size_t nb = std::max<size_t>((size_t)bundles * 2, 4);
        BodyReferences.resize(nb * Info.bodies * W, -1);  // trailing lanes of the last bundle hold -1 (TypeProcessor.cs:287-298)
        PrestepData.resize(nb * Info.prestepFloats * W, 0.0f);
        AccumulatedImpulses.resize(nb * Info.impulseFloats * W, 0.0f);
    }
    int bundle = index / W, lane = index % W;
    for (int k = 0; k < Info.bodies; ++k) BodyReferences[(size_t)bundle * Info.bodies * W + (size_t)k * W + lane] = encodedBodyIndices[k];
    for (int f = 0; f < Info.impulseFloats; ++f) AccumulatedImpulses[(size_t)bundle * Info.impulseFloats * W + (size_t)f * W + lane] = 0.0f;  // a reused lane starts from rest (:327)
    return index;
}
TypeBatch& ConstraintBatch::GetOrCreateTypeBatch(int typeId) {  // ConstraintBatch.cs:60-90
    auto it = TypeIndexToTypeBatchIndex.find(typeId);
    if (it != TypeIndexToTypeBatchIndex.end()) return TypeBatches[it->second];
    TypeIndexToTypeBatchIndex[typeId] = (int)TypeBatches.size();
    TypeBatches.emplace_back();
    TypeBatch& tb = TypeBatches.back();
    tb.TypeId = typeId;
    GetTypeInfo(typeId, tb.Info);
    return tb;
}

// ---- SolveDescription ----
SolveDescription::SolveDescription(int velocityIterationCount, int substepCount, int fallbackBatchThreshold)
    : VelocityIterationCount(velocityIterationCount), SubstepCount(substepCount), FallbackBatchThreshold(fallbackBatchThreshold) {
    // SolveDescription.cs:42-47: ArgumentException
    if (substepCount < 1) throw std::invalid_argument("Substep count must be positive.");
    if (velocityIterationCount < 1) throw std::invalid_argument("Velocity iteration count must be positive.");
    if (fallbackBatchThreshold < 1) throw std::invalid_argument("Fallback batch threshold must be positive.");
}
std::vector<int32_t> SolveDescription::ResolveIterations() const {  // Solver_Solve.cs:743-751
    std::vector<int32_t> out(SubstepCount);
    for (int s = 0; s < SubstepCount; ++s) {
        int n = VelocityIterationCount;
        if (VelocityIterationScheduler) {
            int scheduled = VelocityIterationScheduler(s);
            if (scheduled >= 1) n = scheduled;
        }
        out[s] = n;
    }
    return out;
}

// ---- Solver ----
int Solver::Add(const int32_t* bodyHandles, int bodyCount, int typeId, const float* prestepLane) {  // Solver.cs:1182-1199
    TypeInfo info;
    if (!GetTypeInfo(typeId, info)) throw std::invalid_argument("unknown constraint type id");
    if (bodyCount != info.bodies) throw std::invalid_argument("body count does not match constraint type");
    ++TopologyVersion;
    int32_t encoded[4], blocking[4];
    int blockingCount = 0;
    for (int i = 0; i < bodyCount; ++i) {  // GetBlockingBodyHandles, Solver.cs:1058-1078: kinematics never block
        int index = bodies.HandleToIndex[bodyHandles[i]];
        if (bodies.IsKinematic(index)) {
            encoded[i] = index | kKinematicMask;
            if ((size_t)bodyHandles[i] >= kinematicConstrained.size()) kinematicConstrained.resize(bodyHandles[i] + 1, 0);
            if (kinematicConstrained[bodyHandles[i]]++ == 0) ConstrainedKinematicHandles.push_back(bodyHandles[i]);  // Solver.cs:1025
        } else {
            encoded[i] = index;
            blocking[blockingCount++] = bodyHandles[i];
        }
    }
    for (int b = 0; b <= (int)Batches.size(); ++b) {
        if (b == (int)Batches.size()) {  // AllocateNewConstraintBatch, Solver.cs:1080-1091
            if (b >= kFallbackBatchThreshold) throw std::runtime_error("sequential fallback batch is not supported by this mirror");
            Batches.emplace_back();
            batchReferencedHandles.emplace_back();
        } else {
            bool fits = true;  // IndexSet.CanFit, IndexSet.cs:70-80
            for (int i = 0; i < blockingCount; ++i)
                if (batchReferencedHandles[b].Contains(blocking[i])) { fits = false; break; }
            if (!fits) continue;
        }
        int handle = (int)HandleToConstraint.size();
        TypeBatch& tb = Batches[b].GetOrCreateTypeBatch(typeId);
        int index = tb.Allocate(handle, encoded);
        for (int i = 0; i < blockingCount; ++i) batchReferencedHandles[b].Set(blocking[i]);
        // ApplyDescription: write the lane (GetOffsetInstance/GetFirst, BepuUtilities/GatherScatter.cs)
        const int W = kBundleWidth;
        float* lane = tb.PrestepData.data() + (size_t)(index / W) * info.prestepFloats * W + (index % W);
        for (int f = 0; f < info.prestepFloats; ++f) lane[(size_t)f * W] = prestepLane[f];
        HandleToConstraint.push_back({b, typeId, index});
        ++liveConstraints;
        StructuralChange change{true, b, typeId, index, {-1, -1, -1, -1}, std::vector<float>(prestepLane, prestepLane + info.prestepFloats)};
        for (int i = 0; i < bodyCount; ++i) change.encoded[i] = encoded[i];
        StructuralLog.push_back(std::move(change));
        return handle;
    }
    return -1;
}

void Solver::Remove(int constraintHandle) {
    if (constraintHandle < 0 || constraintHandle >= (int)HandleToConstraint.size() || HandleToConstraint[constraintHandle].BatchIndex < 0)
        throw std::invalid_argument("Can only remove elements that are actually in the batch!");  // TypeProcessor.cs:636
    const ConstraintLocation loc = HandleToConstraint[constraintHandle];
    ConstraintBatch& batch = Batches[loc.BatchIndex];
    TypeBatch& tb = batch.TypeBatches[batch.TypeIndexToTypeBatchIndex.at(loc.TypeId)];
    const int W = kBundleWidth, nb = tb.Info.bodies, pf = tb.Info.prestepFloats, imf = tb.Info.impulseFloats;
    const int index = loc.IndexInTypeBatch, last = tb.ConstraintCount - 1;
    auto ref = [&](int i, int k) -> int32_t& { return tb.BodyReferences[(size_t)(i / W) * nb * W + (size_t)k * W + (i % W)]; };
    for (int k = 0; k < nb; ++k)  // ConstraintBatch.RemoveBodyHandlesFromBatchForConstraint (ConstraintBatch.cs:196-214): dynamic bodies only
        if ((uint32_t)ref(index, k) < (uint32_t)kKinematicMask) batchReferencedHandles[loc.BatchIndex].Unset(bodies.IndexToHandle[ref(index, k)]);
    for (int k = 0; k < nb; ++k) {  // RemoveConstraintReferencesFromBodiesEnumerator (Solver.cs:1368-1377): a kinematic body's last constraint takes it out of the set (FastRemove)
        if ((uint32_t)ref(index, k) < (uint32_t)kKinematicMask) continue;
        const int32_t handle = bodies.IndexToHandle[ref(index, k) & (kKinematicMask - 1)];
        if (--kinematicConstrained[handle] == 0) {
            auto at = std::find(ConstrainedKinematicHandles.begin(), ConstrainedKinematicHandles.end(), handle);
            *at = ConstrainedKinematicHandles.back();
            ConstrainedKinematicHandles.pop_back();
        }
    }
    if (index < last) {  // TypeProcessor.Move
        for (int k = 0; k < nb; ++k) ref(index, k) = ref(last, k);
        for (int f = 0; f < pf; ++f) tb.PrestepData[(size_t)(index / W) * pf * W + (size_t)f * W + (index % W)] = tb.PrestepData[(size_t)(last / W) * pf * W + (size_t)f * W + (last % W)];
        for (int f = 0; f < imf; ++f)
            tb.AccumulatedImpulses[(size_t)(index / W) * imf * W + (size_t)f * W + (index % W)] = tb.AccumulatedImpulses[(size_t)(last / W) * imf * W + (size_t)f * W + (last % W)];
        tb.IndexToHandle[index] = tb.IndexToHandle[last];
        HandleToConstraint[tb.IndexToHandle[index]].IndexInTypeBatch = index;
    }
    for (int k = 0; k < nb; ++k) ref(last, k) = -1;  // the vacated lane reads as empty again (TypeProcessor.cs:287-298)
    tb.ConstraintCount = last;
    HandleToConstraint[constraintHandle].BatchIndex = -1;
    --liveConstraints;
    ++TopologyVersion;
    StructuralLog.push_back(StructuralChange{false, loc.BatchIndex, loc.TypeId, index, {-1, -1, -1, -1}, {}});
}

void Solver::ValidateBatches() const {
    for (size_t b = 0; b < Batches.size(); ++b) {
        std::vector<uint8_t> seen(bodies.Count(), 0);
        for (const TypeBatch& tb : Batches[b].TypeBatches) {
            const int W = kBundleWidth;
            for (int i = 0; i < tb.ConstraintCount; ++i)
                for (int k = 0; k < tb.Info.bodies; ++k) {
                    int32_t ref = tb.BodyReferences[(size_t)(i / W) * tb.Info.bodies * W + (size_t)k * W + (i % W)];
                    if ((uint32_t)ref < (uint32_t)kKinematicMask) {
                        if (seen[ref]) throw std::logic_error("dynamic body referenced twice in a non-fallback batch");
                        seen[ref] = 1;
                    }
                }
        }
    }
}

Solver::IntegrationResponsibilities Solver::PrepareConstraintIntegrationResponsibilities() const {  // Solver_Solve.cs:1072-1388
    IntegrationResponsibilities r;
    const int W = kBundleWidth;
    const int batchCount = (int)Batches.size();
    if (batchCount == 0) return r;
    size_t words = (bodies.HandleToIndex.size() + 64) / 64;  // (HighestPossiblyClaimedId + 64) / 64
    r.integrationFlags.resize(batchCount);
    r.coarseBatchIntegrationResponsibilities.resize(batchCount);
    IndexSet& merged = r.mergedConstrainedBodyHandles;
    merged.Flags.assign(words, 0);
    for (size_t w = 0; w < std::min(words, batchReferencedHandles[0].Flags.size()); ++w) merged.Flags[w] = batchReferencedHandles[0].Flags[w];
    IndexSet firstObserved;
    for (int b = 1; b < batchCount; ++b) {
        const IndexSet& batchHandles = batchReferencedHandles[b];
        firstObserved.Flags.assign(words, 0);
        size_t n = std::min(words, batchHandles.Flags.size());
        for (size_t w = 0; w < n; ++w) {  // :1198-1207
            uint64_t mergeBundle = merged.Flags[w], batchBundle = batchHandles.Flags[w];
            merged.Flags[w] = mergeBundle | batchBundle;
            firstObserved.Flags[w] = ~mergeBundle & batchBundle;
        }
        const ConstraintBatch& batch = Batches[b];
        r.integrationFlags[b].resize(batch.TypeBatches.size());
        r.coarseBatchIntegrationResponsibilities[b].assign(batch.TypeBatches.size(), 0);
        for (size_t t = 0; t < batch.TypeBatches.size(); ++t) {  // ComputeIntegrationResponsibilitiesForConstraintRegion, :951-1044
            const TypeBatch& tb = batch.TypeBatches[t];
            auto& flagsForTypeBatch = r.integrationFlags[b][t];
            flagsForTypeBatch.resize(tb.Info.bodies);
            size_t flagWords = ((size_t)tb.ConstraintCount + 63) / 64;
            for (auto& s : flagsForTypeBatch) s.Flags.assign(std::max<size_t>(flagWords, 1), 0);
            uint64_t mergedFlagBundles = 0;
            for (int i = 0; i < tb.ConstraintCount; ++i) {
                for (int k = 0; k < tb.Info.bodies; ++k) {
                    int bodyIndex = tb.BodyReferences[(size_t)(i / W) * tb.Info.bodies * W + (size_t)k * W + (i % W)] & kBodyReferenceMask;
                    int bodyHandle = bodies.IndexToHandle[bodyIndex];
                    if (firstObserved.Contains(bodyHandle)) {
                        flagsForTypeBatch[k].Flags[i >> 6] |= 1ull << (i & 63);
                        mergedFlagBundles |= 1;
                    }
                }
            }
            r.coarseBatchIntegrationResponsibilities[b][t] = mergedFlagBundles != 0;
        }
    }
    for (int32_t h : ConstrainedKinematicHandles) merged.Set(h);  // :1378-1381
    return r;
}

// ---- Simulation ----
void Simulation::Timestep(float dt) {  // Simulation.cs:316-326
    if (!(dt > 0)) throw std::invalid_argument("Timestep duration must be positive.");
    if (!timestepper) throw std::logic_error("no timestepper (this mirror has no CPU solver: attach a HipTimestepper)");
    timestepper->Timestep(*this, dt);
}

// ---- The structural changes of a frame, reconstructed from the type batches themselves (C++ twin of integration/csharp/HipTimestepper.cs, DiffTypeBatch) ----
// The reference changes a type batch in three ways — append (TypeProcessor.AllocateInTypeBatch, TypeProcessor.cs:314-334), swap-with-last removal (Remove :695-717) and
// single body references patched when a body moves in memory (UpdateForBodyMemoryMove :807) — and from several places (Solver.Add / Remove, the narrow phase's
// pending adds and ConstraintRemover, the sleeper, the awakener's bulk copies). A device mirror does not have to see any of them happen: constraint handles are
// stable (TypeBatch.IndexToHandle, TypeBatch.cs:16-19), so last frame's copy of a type batch's handles and references against this frame's tells which constraints left,
// which came and which references changed. What it does not tell is the ORDER of the removals, which decides where swap-with-last left the survivors — hence the
// fourth operation, swap: removals (in any order) + additions (in index order) + swaps (at most one per index that still disagrees) + reference patches reproduce
// exactly this frame's arrangement. Operations are emitted in the form bepuhip_apply_structural_ops takes.
void DiffTypeBatch(int batch, int typeId, int bodies, int prestepFloats, const int32_t* oldHandles, int oldCount, const int32_t* oldReferences /* [oldCount][bodies] */,
                   const int32_t* newHandles, int newCount, const int32_t* newReferencesAosoa, const float* newPrestepAosoa,
                   std::vector<bepuhip_structural_op>& ops, std::vector<uint32_t>& payload) {
    auto lane_ref = [&](int index, int k) { return newReferencesAosoa[(size_t)(index / kBundleWidth) * bodies * kBundleWidth + (size_t)k * kBundleWidth + index % kBundleWidth]; };
    auto lane_prestep = [&](int index, int f) { return newPrestepAosoa[(size_t)(index / kBundleWidth) * prestepFloats * kBundleWidth + (size_t)f * kBundleWidth + index % kBundleWidth]; };
    if (oldCount == newCount && (oldCount == 0 || std::memcmp(oldHandles, newHandles, (size_t)oldCount * 4) == 0)) {  // the common case: same constraints at the same indices
        for (int i = 0; i < newCount; ++i)
            for (int k = 0; k < bodies; ++k)
                if (oldReferences[(size_t)i * bodies + k] != lane_ref(i, k)) ops.push_back({2, batch, typeId, i, k, lane_ref(i, k), 0, 0});
        return;
    }
    std::unordered_map<int32_t, int32_t> newIndexOf, position, oldIndexOf;
    newIndexOf.reserve((size_t)newCount * 2); position.reserve((size_t)oldCount * 2); oldIndexOf.reserve((size_t)oldCount * 2);
    for (int i = 0; i < newCount; ++i) newIndexOf[newHandles[i]] = i;
    std::vector<int32_t> list(oldHandles, oldHandles + oldCount);  // the device's type batch as the operations so far leave it
    for (int i = 0; i < oldCount; ++i) { position[oldHandles[i]] = i; oldIndexOf[oldHandles[i]] = i; }
    for (int i = oldCount - 1; i >= 0; --i) {  // removals, highest old index first (any order is right; this one moves the fewest survivors)
        const int32_t handle = oldHandles[i];
        if (newIndexOf.count(handle)) continue;
        const int at = position[handle], last = (int)list.size() - 1;
        ops.push_back({1, batch, typeId, at, 0, 0, 0, 0});
        if (at != last) { list[at] = list[last]; position[list[at]] = at; }
        list.pop_back();
        position.erase(handle);
    }
    for (int i = 0; i < newCount; ++i) {  // additions, in the order of their final indices
        const int32_t handle = newHandles[i];
        if (oldIndexOf.count(handle)) continue;
        ops.push_back({0, batch, typeId, (int32_t)list.size(), 0, 0, (int32_t)payload.size(), 0});
        for (int k = 0; k < bodies; ++k) payload.push_back((uint32_t)lane_ref(i, k));
        for (int f = 0; f < prestepFloats; ++f) { const float v = lane_prestep(i, f); uint32_t w; std::memcpy(&w, &v, 4); payload.push_back(w); }
        position[handle] = (int32_t)list.size();
        list.push_back(handle);
    }
    for (int i = 0; i < newCount; ++i) {  // the same set by now: put every index right
        if (list[i] == newHandles[i]) continue;
        const int j = position[newHandles[i]];
        ops.push_back({3, batch, typeId, i, j, 0, 0, 0});
        std::swap(list[i], list[j]);
        position[list[i]] = i; position[list[j]] = j;
    }
    for (int i = 0; i < newCount; ++i) {  // survivors whose bodies moved in memory
        auto was = oldIndexOf.find(newHandles[i]);
        if (was == oldIndexOf.end()) continue;
        for (int k = 0; k < bodies; ++k)
            if (oldReferences[(size_t)was->second * bodies + k] != lane_ref(i, k)) ops.push_back({2, batch, typeId, i, k, lane_ref(i, k), 0, 0});
    }
}

// ---- HipTimestepper: DefaultTimestepper.Timestep (DefaultTimestepper.cs:28-43) with simulation.Solve replaced by the C ABI ----
struct HipApi {
    void* lib = nullptr;
#define DECL(name) decltype(&::name) name = nullptr;
    DECL(bepuhip_last_error) DECL(bepuhip_create) DECL(bepuhip_destroy) DECL(bepuhip_set_bodies) DECL(bepuhip_begin_constraints)
    DECL(bepuhip_set_type_batch) DECL(bepuhip_end_constraints) DECL(bepuhip_set_constrained_kinematics) DECL(bepuhip_solve)
    DECL(bepuhip_get_bodies) DECL(bepuhip_get_accumulated_impulses) DECL(bepuhip_get_prestep) DECL(bepuhip_add_constraint) DECL(bepuhip_remove_constraint)
    DECL(bepuhip_apply_structural_ops)
#undef DECL
    bool load(const char* path, std::string& err) {
        lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = dlerror(); return false; }
#define LOAD(name) name = (decltype(name))dlsym(lib, #name); if (!name) { err = std::string("missing symbol ") + #name; return false; }
        LOAD(bepuhip_last_error) LOAD(bepuhip_create) LOAD(bepuhip_destroy) LOAD(bepuhip_set_bodies) LOAD(bepuhip_begin_constraints)
        LOAD(bepuhip_set_type_batch) LOAD(bepuhip_end_constraints) LOAD(bepuhip_set_constrained_kinematics) LOAD(bepuhip_solve)
        LOAD(bepuhip_get_bodies) LOAD(bepuhip_get_accumulated_impulses) LOAD(bepuhip_get_prestep) LOAD(bepuhip_add_constraint) LOAD(bepuhip_remove_constraint)
        LOAD(bepuhip_apply_structural_ops)
#undef LOAD
        return true;
    }
};

class HipTimestepper : public ITimestepper {
public:
    HipApi api;
    bepuhip_ctx* ctx = nullptr;
    uint64_t uploadedSolverVersion = ~0ull, uploadedBodiesVersion = ~0ull;
    int fullUploads = 0, structuralReplays = 0, diffOperations = 0;
    size_t uploadedKinematics = 0;
    // How the constraint changes of a frame reach the device: 0 = the solver's structural log (what a listener inside the reference would record), 1 = the diff of the
    // type batches against last frame's copy (public API only: what integration/csharp/HipTimestepper.cs does by default).
    int mode = 0;
    struct Mirror { std::vector<int32_t> handles, references; int bodies = 0; };
    std::unordered_map<uint64_t, Mirror> mirrors;  // (batch index << 32 | type id) -> the type batch as the device has it
    void RememberTypeBatches(const Solver& solver) {
        mirrors.clear();
        for (size_t b = 0; b < solver.Batches.size(); ++b)
            for (const TypeBatch& tb : solver.Batches[b].TypeBatches) {
                Mirror& m = mirrors[((uint64_t)b << 32) | (uint32_t)tb.TypeId];
                m.bodies = tb.Info.bodies;
                m.handles.assign(tb.IndexToHandle.begin(), tb.IndexToHandle.begin() + tb.ConstraintCount);
                m.references.resize((size_t)tb.ConstraintCount * tb.Info.bodies);
                for (int i = 0; i < tb.ConstraintCount; ++i)
                    for (int k = 0; k < tb.Info.bodies; ++k)
                        m.references[(size_t)i * tb.Info.bodies + k] = tb.BodyReferences[(size_t)(i / kBundleWidth) * tb.Info.bodies * kBundleWidth + (size_t)k * kBundleWidth + i % kBundleWidth];
            }
    }
    // One bepuhip_apply_structural_ops call for everything that changed in the solver's type batches since RememberTypeBatches.
    void DiffAndApply(const Solver& solver) {
        std::vector<bepuhip_structural_op> ops;
        std::vector<uint32_t> payload;
        std::unordered_map<uint64_t, bool> seen;
        for (size_t b = 0; b < solver.Batches.size(); ++b)
            for (const TypeBatch& tb : solver.Batches[b].TypeBatches) {
                const uint64_t key = ((uint64_t)b << 32) | (uint32_t)tb.TypeId;
                seen[key] = true;
                auto found = mirrors.find(key);
                static const Mirror empty;
                const Mirror& was = found == mirrors.end() ? empty : found->second;
                DiffTypeBatch((int)b, tb.TypeId, tb.Info.bodies, tb.Info.prestepFloats, was.handles.data(), (int)was.handles.size(), was.references.data(), tb.IndexToHandle.data(),
                              tb.ConstraintCount, tb.BodyReferences.data(), tb.PrestepData.data(), ops, payload);
            }
        for (auto& kv : mirrors)  // a type batch that no longer exists (ConstraintBatch.RemoveTypeBatchIfEmpty): its constraints went
            if (!seen.count(kv.first))
                for (int i = (int)kv.second.handles.size() - 1; i >= 0; --i) ops.push_back({1, (int32_t)(kv.first >> 32), (int32_t)(uint32_t)kv.first, i, 0, 0, 0, 0});
        diffOperations += (int)ops.size();
        if (!ops.empty()) {
            int32_t failed = -1;
            if (payload.empty()) payload.push_back(0u);
            check(api.bepuhip_apply_structural_ops(ctx, ops.data(), (int32_t)ops.size(), payload.data(), (int32_t)payload.size(), &failed));
        }
    }
    HipTimestepper(const char* libraryPath, int device) {
        std::string err;
        if (!api.load(libraryPath, err)) throw std::runtime_error("cannot load libbepuhip: " + err);
        bepuhip_config cfg{device, kBundleWidth, BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS};  // a simulation adds and removes constraints between frames: plan room for them
        if (api.bepuhip_create(&cfg, &ctx) != BEPUHIP_OK) throw std::runtime_error(std::string("bepuhip_create: ") + api.bepuhip_last_error());
    }
    ~HipTimestepper() override { if (ctx) api.bepuhip_destroy(ctx); }
    void check(int32_t status) {
        if (status == BEPUHIP_OK) return;
        std::string msg = api.bepuhip_last_error();
        if (status == BEPUHIP_E_INVALID_ARGUMENT) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);  // the C# shim converts these into InvalidOperationException / falls back on UNSUPPORTED
    }
    void Timestep(Simulation& sim, float dt) override {
        // simulation.Sleep / PredictBoundingBoxes / CollisionDetection (DefaultTimestepper.cs:30-37) are out of scope: no-ops here.
        // ---- simulation.Solve(dt) replaced (DefaultTimestepper.cs:39) ----
        Solver& solver = sim.solver;
        // Topology is re-uploaded only when it changed: keyed on the version counters Solver.Add / Bodies.Add bump (a count comparison would miss a
        // remove + add, or a body move that renumbers references); bodies are host-authoritative every frame.
        check(api.bepuhip_set_bodies(ctx, sim.bodies.DynamicsState.data(), sim.bodies.Count()));
        // Constraint changes since the last frame: replayed through the structural calls when the log covers exactly what happened since the upload (and is short
        // next to a re-upload); the library keeps them on the island layout where it can (include/bepuhip.h).
        if (mode == 1 && uploadedBodiesVersion == sim.bodies.TopologyVersion && uploadedSolverVersion != ~0ull) {
            if (uploadedSolverVersion != solver.TopologyVersion) {
                DiffAndApply(solver);
                RememberTypeBatches(solver);
                std::vector<int32_t> kin;
                for (int32_t h : solver.ConstrainedKinematicHandles) kin.push_back(sim.bodies.HandleToIndex[h]);
                check(api.bepuhip_set_constrained_kinematics(ctx, kin.data(), (int)kin.size()));
                uploadedKinematics = kin.size();
                uploadedSolverVersion = solver.TopologyVersion;
                ++structuralReplays;
            }
        } else
        if (uploadedBodiesVersion == sim.bodies.TopologyVersion && uploadedSolverVersion != solver.TopologyVersion && solver.StructuralLogBase == uploadedSolverVersion &&
            solver.StructuralLogBase + solver.StructuralLog.size() == solver.TopologyVersion && (int)solver.StructuralLog.size() * 8 < std::max(64, solver.ConstraintCount())) {
            for (const Solver::StructuralChange& change : solver.StructuralLog) {
                if (change.add) {
                    int32_t index = -1;
                    check(api.bepuhip_add_constraint(ctx, change.batch, change.typeId, change.encoded, change.prestep.data(), &index));
                    if (index != change.index) throw std::logic_error("device and host disagree about the index of an added constraint");
                } else {
                    check(api.bepuhip_remove_constraint(ctx, change.batch, change.typeId, change.index));
                }
            }
            {  // additions bring kinematic bodies into Solver.ConstrainedKinematicHandles, removals take them out (the count alone does not tell: one of each leaves it unchanged)
                std::vector<int32_t> kin;
                for (int32_t h : solver.ConstrainedKinematicHandles) kin.push_back(sim.bodies.HandleToIndex[h]);
                check(api.bepuhip_set_constrained_kinematics(ctx, kin.data(), (int)kin.size()));
                uploadedKinematics = kin.size();
            }
            uploadedSolverVersion = solver.TopologyVersion;
            ++structuralReplays;
        }
        solver.ConsumeStructuralLog();
        if (uploadedSolverVersion != solver.TopologyVersion || uploadedBodiesVersion != sim.bodies.TopologyVersion) {
            ++fullUploads;
            check(api.bepuhip_begin_constraints(ctx, (int)solver.Batches.size(), sim.solveDescription.FallbackBatchThreshold));
            for (size_t b = 0; b < solver.Batches.size(); ++b)
                for (const TypeBatch& tb : solver.Batches[b].TypeBatches)
                    check(api.bepuhip_set_type_batch(ctx, (int)b, tb.TypeId, tb.ConstraintCount, tb.BodyReferences.data(), tb.PrestepData.data(), tb.AccumulatedImpulses.data()));
            check(api.bepuhip_end_constraints(ctx));
            std::vector<int32_t> kin;
            for (int32_t h : solver.ConstrainedKinematicHandles) kin.push_back(sim.bodies.HandleToIndex[h]);
            check(api.bepuhip_set_constrained_kinematics(ctx, kin.data(), (int)kin.size()));
            uploadedKinematics = kin.size();
            uploadedSolverVersion = solver.TopologyVersion;
            uploadedBodiesVersion = sim.bodies.TopologyVersion;
            if (mode == 1) RememberTypeBatches(solver);
        }
        std::vector<int32_t> iterations = sim.solveDescription.ResolveIterations();
        bepuhip_integrator in{};
        in.gravity[0] = sim.callbacks.Gravity.X; in.gravity[1] = sim.callbacks.Gravity.Y; in.gravity[2] = sim.callbacks.Gravity.Z;
        in.linear_damping = sim.callbacks.LinearDamping; in.angular_damping = sim.callbacks.AngularDamping;
        in.angular_integration_mode = 0;
        in.allow_substeps_for_unconstrained = sim.callbacks.AllowSubstepsForUnconstrainedBodies;
        in.integrate_velocity_for_kinematics = sim.callbacks.IntegrateVelocityForKinematics;
        check(api.bepuhip_solve(ctx, dt, sim.solveDescription.SubstepCount, iterations.data(), &in));
        check(api.bepuhip_get_bodies(ctx, sim.bodies.DynamicsState.data(), sim.bodies.Count()));
        for (size_t b = 0; b < solver.Batches.size(); ++b)
            for (TypeBatch& tb : solver.Batches[b].TypeBatches) {
                if (tb.ConstraintCount == 0) continue;
                check(api.bepuhip_get_accumulated_impulses(ctx, (int)b, tb.TypeId, tb.AccumulatedImpulses.data()));
                if (tb.Info.incremental) check(api.bepuhip_get_prestep(ctx, (int)b, tb.TypeId, tb.PrestepData.data()));
            }
        // simulation.IncrementallyOptimizeDataStructures (DefaultTimestepper.cs:42): out of scope.
    }
};

}  // namespace bepu

// ------------------------------------------------------------------------------------------------
// C exports for Python (ctypes): scene construction, export of the reference-layout buffers, HipTimestepper.
// ------------------------------------------------------------------------------------------------
using namespace bepu;

namespace bepu { Simulation* BuildScene(const char* name, int64_t a, int64_t b, int64_t c, uint32_t seed); }

static thread_local std::string g_err;

extern "C" {

const char* bepuhost_last_error() { return g_err.c_str(); }

void* bepuhost_simulation_create(const float* gravity, float linearDamping, float angularDamping, int velocityIterations, int substeps) {
    try {
        PoseIntegratorCallbacks cb;
        cb.Gravity = {gravity[0], gravity[1], gravity[2]};
        cb.LinearDamping = linearDamping; cb.AngularDamping = angularDamping;
        return new Simulation(cb, SolveDescription(velocityIterations, substeps));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void* bepuhost_scene_create(const char* name, int64_t a, int64_t b, int64_t c, uint32_t seed) {
    try { return BuildScene(name, a, b, c, seed); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void bepuhost_simulation_destroy(void* s) {
    Simulation* sim = (Simulation*)s;
    if (sim) { delete sim->timestepper; delete sim; }
}
int32_t bepuhost_add_body(void* s, const float* pose7, const float* velocity6, const float* inertia7) {
    BodyDescription d;
    d.Pose.Position = {pose7[0], pose7[1], pose7[2]};
    d.Pose.Orientation = {pose7[3], pose7[4], pose7[5], pose7[6]};
    d.Velocity.Linear = {velocity6[0], velocity6[1], velocity6[2]};
    d.Velocity.Angular = {velocity6[3], velocity6[4], velocity6[5]};
    d.LocalInertia.InverseInertiaTensor = {inertia7[0], inertia7[1], inertia7[2], inertia7[3], inertia7[4], inertia7[5]};
    d.LocalInertia.InverseMass = inertia7[6];
    return ((Simulation*)s)->bodies.Add(d);
}
int32_t bepuhost_add_constraint(void* s, int typeId, const int32_t* bodyHandles, int bodyCount, const float* prestepLane) {
    try { return ((Simulation*)s)->solver.Add(bodyHandles, bodyCount, typeId, prestepLane); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int32_t bepuhost_remove_constraint(void* s, int32_t constraintHandle) {
    try { ((Simulation*)s)->solver.Remove(constraintHandle); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// Per type batch: the handles of its constraints in index order (TypeBatch.IndexToHandle, TypeBatch.cs:19).
const int32_t* bepuhost_type_batch_handles(void* s, int b, int t) { return ((Simulation*)s)->solver.Batches[b].TypeBatches[t].IndexToHandle.data(); }
int32_t bepuhost_validate(void* s) {
    try { ((Simulation*)s)->solver.ValidateBatches(); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int32_t bepuhost_body_count(void* s) { return ((Simulation*)s)->bodies.Count(); }
float* bepuhost_bodies_ptr(void* s) { return ((Simulation*)s)->bodies.DynamicsState.data()->f; }
const int32_t* bepuhost_index_to_handle(void* s) { return ((Simulation*)s)->bodies.IndexToHandle.data(); }
const int32_t* bepuhost_handle_to_index(void* s) { return ((Simulation*)s)->bodies.HandleToIndex.data(); }
int32_t bepuhost_handle_capacity(void* s) { return (int32_t)((Simulation*)s)->bodies.HandleToIndex.size(); }
int32_t bepuhost_constraint_count(void* s) { return ((Simulation*)s)->solver.ConstraintCount(); }
int32_t bepuhost_batch_count(void* s) { return (int32_t)((Simulation*)s)->solver.Batches.size(); }
int32_t bepuhost_type_batch_count(void* s, int b) { return (int32_t)((Simulation*)s)->solver.Batches[b].TypeBatches.size(); }
struct bepuhost_type_batch_view { int32_t type_id, count, bodies, prestep_floats, impulse_floats, bundle_count; int32_t* refs; float* prestep; float* accumulated; };
void bepuhost_type_batch(void* s, int b, int t, bepuhost_type_batch_view* out) {
    TypeBatch& tb = ((Simulation*)s)->solver.Batches[b].TypeBatches[t];
    *out = {tb.TypeId, tb.ConstraintCount, tb.Info.bodies, tb.Info.prestepFloats, tb.Info.impulseFloats, tb.BundleCount(), tb.BodyReferences.data(), tb.PrestepData.data(), tb.AccumulatedImpulses.data()};
}
int32_t bepuhost_kinematic_count(void* s) { return (int32_t)((Simulation*)s)->solver.ConstrainedKinematicHandles.size(); }
const int32_t* bepuhost_kinematic_handles(void* s) { return ((Simulation*)s)->solver.ConstrainedKinematicHandles.data(); }
void bepuhost_solve_description(void* s, int32_t* velocityIterations, int32_t* substeps) {
    *velocityIterations = ((Simulation*)s)->solveDescription.VelocityIterationCount;
    *substeps = ((Simulation*)s)->solveDescription.SubstepCount;
}

// Prepass export for parity tests against the oracle: same packing as oracle_prepare_flags.
int32_t bepuhost_prepare_flags(void* s, uint64_t* outMerged, int64_t mergedCapacity, uint64_t* outFlags, int64_t flagsCapacity, uint8_t* outCoarse) {
    Simulation* sim = (Simulation*)s;
    auto r = sim->solver.PrepareConstraintIntegrationResponsibilities();
    for (int64_t w = 0; w < mergedCapacity; ++w) outMerged[w] = (size_t)w < r.mergedConstrainedBodyHandles.Flags.size() ? r.mergedConstrainedBodyHandles.Flags[w] : 0;
    int64_t o = 0;
    size_t flat = 0;
    for (size_t b = 0; b < sim->solver.Batches.size(); ++b) {
        for (size_t t = 0; t < sim->solver.Batches[b].TypeBatches.size(); ++t, ++flat) {
            outCoarse[flat] = b == 0 ? 0 : r.coarseBatchIntegrationResponsibilities[b][t];
            if (b == 0) continue;
            int64_t words = (sim->solver.Batches[b].TypeBatches[t].ConstraintCount + 63) / 64;
            for (auto& slot : r.integrationFlags[b][t]) {
                if (o + words > flagsCapacity) return -1;
                for (int64_t w = 0; w < words; ++w) outFlags[o + w] = slot.Flags[w];
                o += words;
            }
        }
    }
    return 0;
}

// Attach a HipTimestepper (ITimestepper) bound to libbepuhip at `libraryPath` and run Simulation.Timestep(dt).
int32_t bepuhost_attach_hip_timestepper(void* s, const char* libraryPath, int device) {
    try {
        Simulation* sim = (Simulation*)s;
        delete sim->timestepper;
        sim->timestepper = nullptr;
        sim->timestepper = new HipTimestepper(libraryPath, device);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// 0: the attached HipTimestepper replays the solver's structural log (a listener's view); 1: it diffs the type batches against last frame's copy (public API only).
int32_t bepuhost_timestepper_mode(void* s, int mode) {
    HipTimestepper* t = dynamic_cast<HipTimestepper*>(((Simulation*)s)->timestepper);
    if (!t) { g_err = "no HipTimestepper attached"; return -1; }
    t->mode = mode;
    return 0;
}
// The diff on its own, for tests that keep their own mirror of a type batch: writes at most `opCapacity` operations (8 int32 each, bepuhip_structural_op) and
// `payloadCapacity` payload words; returns the number of operations, *payloadWords the words used, or -1 when a capacity is too small.
int32_t bepuhost_diff_type_batch(int batch, int typeId, int bodies, int prestepFloats, const int32_t* oldHandles, int oldCount, const int32_t* oldReferences,
                                 const int32_t* newHandles, int newCount, const int32_t* newReferencesAosoa, const float* newPrestepAosoa,
                                 int32_t* opsOut, int opCapacity, uint32_t* payloadOut, int payloadCapacity, int32_t* payloadWords) {
    std::vector<bepuhip_structural_op> ops;
    std::vector<uint32_t> payload;
    DiffTypeBatch(batch, typeId, bodies, prestepFloats, oldHandles, oldCount, oldReferences, newHandles, newCount, newReferencesAosoa, newPrestepAosoa, ops, payload);
    if ((int)ops.size() > opCapacity || (int)payload.size() > payloadCapacity) return -1;
    if (!ops.empty()) std::memcpy(opsOut, ops.data(), ops.size() * sizeof(bepuhip_structural_op));
    if (!payload.empty()) std::memcpy(payloadOut, payload.data(), payload.size() * 4);
    *payloadWords = (int32_t)payload.size();
    return (int32_t)ops.size();
}
// How often the attached HipTimestepper re-uploaded the topology / replayed a structural log instead.
void bepuhost_timestepper_stats(void* s, int32_t* fullUploads, int32_t* structuralReplays);
int32_t bepuhost_timestep(void* s, float dt) {
    try { ((Simulation*)s)->Timestep(dt); return 0; }
    catch (const std::invalid_argument& e) { g_err = e.what(); return -1; }
    catch (const std::exception& e) { g_err = e.what(); return -2; }
}

void bepuhost_timestepper_stats(void* s, int32_t* fullUploads, int32_t* structuralReplays) {
    HipTimestepper* t = dynamic_cast<HipTimestepper*>(((Simulation*)s)->timestepper);
    *fullUploads = t ? t->fullUploads : 0;
    *structuralReplays = t ? t->structuralReplays : 0;
}

}  // extern "C"
