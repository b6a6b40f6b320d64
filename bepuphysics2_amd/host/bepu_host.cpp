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
        int handle;
        if (!HandlePool.empty()) { handle = HandlePool.back(); HandlePool.pop_back(); }  // IdPool.Take: the most recently returned id first
        else { handle = (int)HandleToConstraint.size(); HandleToConstraint.push_back({-1, 0, 0}); }
        TypeBatch& tb = Batches[b].GetOrCreateTypeBatch(typeId);
        int index = tb.Allocate(handle, encoded);
        for (int i = 0; i < blockingCount; ++i) batchReferencedHandles[b].Set(blocking[i]);
        // ApplyDescription: write the lane (GetOffsetInstance/GetFirst, BepuUtilities/GatherScatter.cs)
        const int W = kBundleWidth;
        float* lane = tb.PrestepData.data() + (size_t)(index / W) * info.prestepFloats * W + (index % W);
        for (int f = 0; f < info.prestepFloats; ++f) lane[(size_t)f * W] = prestepLane[f];
        HandleToConstraint[handle] = {b, typeId, index};
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
    HandlePool.push_back(constraintHandle);
    --liveConstraints;
    ++TopologyVersion;
    StructuralLog.push_back(StructuralChange{false, loc.BatchIndex, loc.TypeId, index, {-1, -1, -1, -1}, {}});
}

static TypeBatch& LocateTypeBatch(Solver& solver, int constraintHandle, int& index) {
    if (constraintHandle < 0 || constraintHandle >= (int)solver.HandleToConstraint.size() || solver.HandleToConstraint[constraintHandle].BatchIndex < 0)
        throw std::invalid_argument("the constraint handle does not name a live constraint");
    const ConstraintLocation loc = solver.HandleToConstraint[constraintHandle];
    ConstraintBatch& batch = solver.Batches[loc.BatchIndex];
    index = loc.IndexInTypeBatch;
    return batch.TypeBatches[batch.TypeIndexToTypeBatchIndex.at(loc.TypeId)];
}
void Solver::ApplyDescription(int constraintHandle, const float* prestepLane) {  // Solver.cs:1162-1185 -> TDescription.ApplyDescription (GetOffsetInstance / GetFirst writes)
    int index;
    TypeBatch& tb = LocateTypeBatch(*this, constraintHandle, index);
    const int W = kBundleWidth, pf = tb.Info.prestepFloats;
    for (int f = 0; f < pf; ++f) tb.PrestepData[(size_t)(index / W) * pf * W + (size_t)f * W + (index % W)] = prestepLane[f];
}
void Solver::SetAccumulatedImpulses(int constraintHandle, const float* impulseLane) {
    int index;
    TypeBatch& tb = LocateTypeBatch(*this, constraintHandle, index);
    const int W = kBundleWidth, imf = tb.Info.impulseFloats;
    for (int f = 0; f < imf; ++f) tb.AccumulatedImpulses[(size_t)(index / W) * imf * W + (size_t)f * W + (index % W)] = impulseLane[f];
}
void Solver::GetAccumulatedImpulses(int constraintHandle, float* impulseLane) const {
    int index;
    const TypeBatch& tb = LocateTypeBatch(const_cast<Solver&>(*this), constraintHandle, index);
    const int W = kBundleWidth, imf = tb.Info.impulseFloats;
    for (int f = 0; f < imf; ++f) impulseLane[f] = tb.AccumulatedImpulses[(size_t)(index / W) * imf * W + (size_t)f * W + (index % W)];
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
// Identity: a constraint is the same constraint as last frame's when its handle AND the handles of its bodies are the same. The constraint handle alone is not enough:
// Solver.HandlePool hands a freed handle out again last-in-first-out (IdPool.Take), so Solver.Remove(h) followed by Solver.Add(...) in one frame returns h for a different
// constraint — possibly at the same index of the same type batch. `oldBodyHandles` / `newBodyHandles` ([count][bodies], Bodies.ActiveSet.IndexToHandle of every
// reference) tell the two apart; a reused handle becomes a removal plus an addition. A body that only moved in memory keeps its handle: that is the reference patch.
// Removals go to `removals`, everything else to `ops`: the caller sends EVERY type batch's removals before any addition (a constraint that replaces another one on the same
// bodies in the same batch — a hinge swapped for a weld — would otherwise arrive while its bodies still look taken: the island layout checks the batch invariant).
// `survivorOldIndex` (optional, newCount entries): for every constraint of the new arrangement the index it had last frame, or -1 when it is new — what the caller needs to
// carry its copy of the device's prestep data and impulses along.
void DiffTypeBatch(int batch, int typeId, int bodies, int prestepFloats, const int32_t* oldHandles, int oldCount, const int32_t* oldReferences /* [oldCount][bodies] */,
                   const int32_t* oldBodyHandles /* [oldCount][bodies] or null */, const int32_t* newHandles, int newCount, const int32_t* newReferencesAosoa,
                   const int32_t* newBodyHandles /* [newCount][bodies] or null */, const float* newPrestepAosoa, std::vector<bepuhip_structural_op>& removals,
                   std::vector<bepuhip_structural_op>& ops, std::vector<uint32_t>& payload, std::vector<int32_t>* survivorOldIndex) {
    auto lane_ref = [&](int index, int k) { return newReferencesAosoa[(size_t)(index / kBundleWidth) * bodies * kBundleWidth + (size_t)k * kBundleWidth + index % kBundleWidth]; };
    auto lane_prestep = [&](int index, int f) { return newPrestepAosoa[(size_t)(index / kBundleWidth) * prestepFloats * kBundleWidth + (size_t)f * kBundleWidth + index % kBundleWidth]; };
    const bool identities = oldBodyHandles && newBodyHandles;
    if (survivorOldIndex) survivorOldIndex->assign((size_t)newCount, -1);
    if (oldCount == newCount && (oldCount == 0 || std::memcmp(oldHandles, newHandles, (size_t)oldCount * 4) == 0) &&
        (!identities || oldCount == 0 || std::memcmp(oldBodyHandles, newBodyHandles, (size_t)oldCount * bodies * 4) == 0)) {  // the common case: same constraints at the same indices
        for (int i = 0; i < newCount; ++i) {
            if (survivorOldIndex) (*survivorOldIndex)[i] = i;
            for (int k = 0; k < bodies; ++k)
                if (oldReferences[(size_t)i * bodies + k] != lane_ref(i, k)) ops.push_back({2, batch, typeId, i, k, lane_ref(i, k), 0, 0});
        }
        return;
    }
    std::unordered_map<int64_t, int32_t> newIndexOf, position, oldIndexOf;
    newIndexOf.reserve((size_t)newCount * 2); position.reserve((size_t)oldCount * 2); oldIndexOf.reserve((size_t)oldCount * 2);
    for (int i = 0; i < newCount; ++i) newIndexOf[newHandles[i]] = i;
    // keys: the handle — except for an old constraint whose handle names a DIFFERENT constraint now (other bodies): it gets a key no new constraint has
    std::vector<int64_t> list((size_t)oldCount);  // the device's type batch as the operations so far leave it
    for (int j = 0; j < oldCount; ++j) {
        int64_t key = oldHandles[j];
        auto now = newIndexOf.find(key);
        if (identities && now != newIndexOf.end() && std::memcmp(oldBodyHandles + (size_t)j * bodies, newBodyHandles + (size_t)now->second * bodies, (size_t)bodies * 4) != 0) key = -1 - key;
        list[j] = key; position[key] = j; oldIndexOf[key] = j;
    }
    const std::vector<int64_t> oldKeys = list;
    for (int i = oldCount - 1; i >= 0; --i) {  // removals, highest old index first (any order is right; this one moves the fewest survivors)
        const int64_t key = oldKeys[i];
        if (key >= 0 && newIndexOf.count(key)) continue;
        const int at = position[key], last = (int)list.size() - 1;
        removals.push_back({1, batch, typeId, at, 0, 0, 0, 0});
        if (at != last) { list[at] = list[last]; position[list[at]] = at; }
        list.pop_back();
        position.erase(key);
    }
    for (int i = 0; i < newCount; ++i) {  // additions, in the order of their final indices
        const int64_t key = newHandles[i];
        if (oldIndexOf.count(key)) continue;
        ops.push_back({0, batch, typeId, (int32_t)list.size(), 0, 0, (int32_t)payload.size(), 0});
        for (int k = 0; k < bodies; ++k) payload.push_back((uint32_t)lane_ref(i, k));
        for (int f = 0; f < prestepFloats; ++f) { const float v = lane_prestep(i, f); uint32_t w; std::memcpy(&w, &v, 4); payload.push_back(w); }
        position[key] = (int32_t)list.size();
        list.push_back(key);
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
        if (survivorOldIndex) (*survivorOldIndex)[i] = was->second;
        for (int k = 0; k < bodies; ++k)
            if (oldReferences[(size_t)was->second * bodies + k] != lane_ref(i, k)) ops.push_back({2, batch, typeId, i, k, lane_ref(i, k), 0, 0});
    }
}

// ---- HipTimestepper: DefaultTimestepper.Timestep (DefaultTimestepper.cs:28-43) with simulation.Solve replaced by the C ABI ----
struct HipApi {
    void* lib = nullptr;
#define BEPU_API(X) X(bepuhip_last_error) X(bepuhip_create) X(bepuhip_destroy) X(bepuhip_set_bodies) X(bepuhip_begin_constraints) X(bepuhip_set_type_batch) X(bepuhip_end_constraints) \
    X(bepuhip_set_constrained_kinematics) X(bepuhip_solve) X(bepuhip_get_bodies) X(bepuhip_get_accumulated_impulses) X(bepuhip_get_prestep) X(bepuhip_add_constraint) \
    X(bepuhip_remove_constraint) X(bepuhip_apply_structural_ops) X(bepuhip_transfer_rows_async) X(bepuhip_solve_async) X(bepuhip_sync) X(bepuhip_get_poses_and_velocities_async) \
    X(bepuhip_register_host_memory) X(bepuhip_unregister_host_memory) X(bepuhip_get_schedule) X(bepuhip_replan)
#define DECL(name) decltype(&::name) name = nullptr;
    BEPU_API(DECL)
#undef DECL
    bool load(const char* path, std::string& err) {
        lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = dlerror(); return false; }
#define LOAD(name) name = (decltype(name))dlsym(lib, #name); if (!name) { err = std::string("missing symbol ") + #name; return false; }
        BEPU_API(LOAD)
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
    int64_t refreshedBundles = 0;  // bundles of non-contact prestep data / impulses the resident frame found changed on the host and sent
    size_t uploadedKinematics = 0;
    // How the constraint changes of a frame reach the device:
    //   0  the solver's structural log (what a listener inside the reference would record);
    //   1  the diff of the type batches against last frame's copy (public API only), everything read back synchronously every frame (round 4);
    //   2  the RESIDENT frame integration/csharp/HipTimestepper.cs runs (round 5): the diff in two phases, the device's prestep data and impulses shadowed on the host so
    //      that what the host rewrote in place since the last frame — Solver.ApplyDescription on motors and servos (Tank.cs:100,139, SimpleCar.cs:25), impulses the awakener
    //      restored, a handle the pool handed out again — is found by comparison and sent by bundle range; poses, velocities and ALL accumulated impulses back
    //      asynchronously behind the solve, one sync per frame.
    int mode = 0;
    bool readBackContactDepths = false;  // mode 2: also fetch the contact type batches' prestep data (the depths substeps > 0 advanced): a narrow phase rewrites them anyway; tests compare them
    struct Mirror {
        std::vector<int32_t> handles, references, bodyHandles;
        std::vector<float> prestep, impulses;  // mode 2: what the device holds, in the type batch's own AOSOA layout (non-contact type batches)
        int bodies = 0; bool contact = false;
    };
    std::unordered_map<uint64_t, Mirror> mirrors;  // (batch index << 32 | type id) -> the type batch as the device has it
    std::vector<void*> registered;
    void Register(void* memory, size_t bytes) {  // BufferPool blocks are pinned for the simulation's life; this mirror's std::vectors move when they grow: re-registered when they do
        if (!memory || bytes == 0) return;
        if (api.bepuhip_register_host_memory(ctx, memory, (int64_t)bytes) == BEPUHIP_OK) registered.push_back(memory);
    }
    void UnregisterAll() { for (void* p : registered) api.bepuhip_unregister_host_memory(ctx, p); registered.clear(); }
    static void Remember(Mirror& m, const TypeBatch& tb, const Bodies& bodies) {
        m.bodies = tb.Info.bodies; m.contact = tb.Info.incremental;
        m.handles.assign(tb.IndexToHandle.begin(), tb.IndexToHandle.begin() + tb.ConstraintCount);
        m.references.resize((size_t)tb.ConstraintCount * tb.Info.bodies);
        m.bodyHandles.resize(m.references.size());
        for (int i = 0; i < tb.ConstraintCount; ++i)
            for (int k = 0; k < tb.Info.bodies; ++k) {
                const int32_t ref = tb.BodyReferences[(size_t)(i / kBundleWidth) * tb.Info.bodies * kBundleWidth + (size_t)k * kBundleWidth + i % kBundleWidth];
                m.references[(size_t)i * tb.Info.bodies + k] = ref;
                m.bodyHandles[(size_t)i * tb.Info.bodies + k] = bodies.IndexToHandle[ref & kBodyReferenceMask];
            }
    }
    void RememberTypeBatches(const Solver& solver, const Bodies& bodies, bool shadows) {
        mirrors.clear();
        for (size_t b = 0; b < solver.Batches.size(); ++b)
            for (const TypeBatch& tb : solver.Batches[b].TypeBatches) {
                Mirror& m = mirrors[((uint64_t)b << 32) | (uint32_t)tb.TypeId];
                Remember(m, tb, bodies);
                if (shadows && !m.contact) {
                    m.prestep.assign(tb.PrestepData.begin(), tb.PrestepData.begin() + (size_t)tb.BundleCount() * tb.Info.prestepFloats * kBundleWidth);
                    m.impulses.assign(tb.AccumulatedImpulses.begin(), tb.AccumulatedImpulses.begin() + (size_t)tb.BundleCount() * tb.Info.impulseFloats * kBundleWidth);
                }
            }
    }
    // Everything that changed in the solver's type batches since RememberTypeBatches, in ONE bepuhip_apply_structural_ops call: every type batch's removals first (the ones
    // of type batches that no longer exist included), then additions, swaps and reference patches. With `shadows` the mirrors' copies of the device's prestep data and
    // impulses follow the constraints to their new indices (a new constraint: the prestep lane it was added with, zero impulses — what the device now holds).
    std::vector<std::pair<int, int>> changedTypeBatches;  // (batch, type id) whose arrangement the last DiffAndApply changed
    void DiffAndApply(const Solver& solver, const Bodies& bodies, bool shadows) {
        changedTypeBatches.clear();
        std::vector<bepuhip_structural_op> removals, ops;
        std::vector<uint32_t> payload;
        std::unordered_map<uint64_t, bool> seen;
        std::vector<int32_t> newBodyHandles, survivor;
        for (size_t b = 0; b < solver.Batches.size(); ++b)
            for (const TypeBatch& tb : solver.Batches[b].TypeBatches) {
                const uint64_t key = ((uint64_t)b << 32) | (uint32_t)tb.TypeId;
                seen[key] = true;
                Mirror& was = mirrors[key];
                const int nb = tb.Info.bodies, W = kBundleWidth;
                newBodyHandles.resize((size_t)tb.ConstraintCount * nb);
                for (int i = 0; i < tb.ConstraintCount; ++i)
                    for (int k = 0; k < nb; ++k)
                        newBodyHandles[(size_t)i * nb + k] = bodies.IndexToHandle[tb.BodyReferences[(size_t)(i / W) * nb * W + (size_t)k * W + i % W] & kBodyReferenceMask];
                const size_t before = removals.size() + ops.size();
                DiffTypeBatch((int)b, tb.TypeId, nb, tb.Info.prestepFloats, was.handles.data(), (int)was.handles.size(), was.references.data(), was.bodyHandles.data(), tb.IndexToHandle.data(),
                              tb.ConstraintCount, tb.BodyReferences.data(), newBodyHandles.data(), tb.PrestepData.data(), removals, ops, payload, &survivor);
                if (removals.size() + ops.size() != before || was.handles.size() != (size_t)tb.ConstraintCount) changedTypeBatches.push_back({(int)b, tb.TypeId});
                if (shadows && !tb.Info.incremental && (removals.size() + ops.size() != before || was.handles.size() != (size_t)tb.ConstraintCount)) {
                    const int pf = tb.Info.prestepFloats, imf = tb.Info.impulseFloats;
                    std::vector<float> prestep((size_t)tb.BundleCount() * pf * W, 0.0f), impulses((size_t)tb.BundleCount() * imf * W, 0.0f);
                    for (int i = 0; i < tb.ConstraintCount; ++i) {
                        const int old = survivor[i];
                        for (int f = 0; f < pf; ++f)
                            prestep[(size_t)(i / W) * pf * W + (size_t)f * W + i % W] = old >= 0 ? was.prestep[(size_t)(old / W) * pf * W + (size_t)f * W + old % W]
                                                                                                 : tb.PrestepData[(size_t)(i / W) * pf * W + (size_t)f * W + i % W];
                        if (old >= 0) for (int f = 0; f < imf; ++f) impulses[(size_t)(i / W) * imf * W + (size_t)f * W + i % W] = was.impulses[(size_t)(old / W) * imf * W + (size_t)f * W + old % W];
                    }
                    was.prestep.swap(prestep); was.impulses.swap(impulses);
                }
                Remember(was, tb, bodies);
            }
        for (auto it = mirrors.begin(); it != mirrors.end();) {  // a type batch that no longer exists (ConstraintBatch.RemoveTypeBatchIfEmpty): its constraints went
            if (seen.count(it->first)) { ++it; continue; }
            for (int i = (int)it->second.handles.size() - 1; i >= 0; --i) removals.push_back({1, (int32_t)(it->first >> 32), (int32_t)(uint32_t)it->first, i, 0, 0, 0, 0});
            it = mirrors.erase(it);
        }
        diffOperations += (int)(removals.size() + ops.size());
        removals.insert(removals.end(), ops.begin(), ops.end());
        if (!removals.empty()) {
            int32_t failed = -1;
            if (payload.empty()) payload.push_back(0u);
            check(api.bepuhip_apply_structural_ops(ctx, removals.data(), (int32_t)removals.size(), payload.data(), (int32_t)payload.size(), &failed));
        }
    }
    HipTimestepper(const char* libraryPath, int device) {
        std::string err;
        if (!api.load(libraryPath, err)) throw std::runtime_error("cannot load libbepuhip: " + err);
        bepuhip_config cfg{device, kBundleWidth, BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS};  // a simulation adds and removes constraints between frames: plan room for them
        if (api.bepuhip_create(&cfg, &ctx) != BEPUHIP_OK) throw std::runtime_error(std::string("bepuhip_create: ") + api.bepuhip_last_error());
    }
    ~HipTimestepper() override { if (ctx) { UnregisterAll(); api.bepuhip_destroy(ctx); } }
    void check(int32_t status) {
        if (status == BEPUHIP_OK) return;
        std::string msg = api.bepuhip_last_error();
        if (status == BEPUHIP_E_INVALID_ARGUMENT) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);  // the C# shim converts these into InvalidOperationException / falls back on UNSUPPORTED
    }
    void SendKinematics(Simulation& sim) {
        std::vector<int32_t> kin;
        for (int32_t h : sim.solver.ConstrainedKinematicHandles) kin.push_back(sim.bodies.HandleToIndex[h]);
        check(api.bepuhip_set_constrained_kinematics(ctx, kin.data(), (int)kin.size()));
        uploadedKinematics = kin.size();
    }
    void Upload(Simulation& sim) {
        Solver& solver = sim.solver;
        ++fullUploads;
        check(api.bepuhip_begin_constraints(ctx, (int)solver.Batches.size(), sim.solveDescription.FallbackBatchThreshold));
        for (size_t b = 0; b < solver.Batches.size(); ++b)
            for (const TypeBatch& tb : solver.Batches[b].TypeBatches)
                check(api.bepuhip_set_type_batch(ctx, (int)b, tb.TypeId, tb.ConstraintCount, tb.BodyReferences.data(), tb.PrestepData.data(), tb.AccumulatedImpulses.data()));
        check(api.bepuhip_end_constraints(ctx));
        SendKinematics(sim);
        uploadedSolverVersion = solver.TopologyVersion;
        uploadedBodiesVersion = sim.bodies.TopologyVersion;
        if (mode >= 1) RememberTypeBatches(solver, sim.bodies, mode == 2);
    }
    bepuhip_integrator Integrator(const Simulation& sim) const {
        bepuhip_integrator in{};
        in.gravity[0] = sim.callbacks.Gravity.X; in.gravity[1] = sim.callbacks.Gravity.Y; in.gravity[2] = sim.callbacks.Gravity.Z;
        in.linear_damping = sim.callbacks.LinearDamping; in.angular_damping = sim.callbacks.AngularDamping;
        in.angular_integration_mode = 0;
        in.allow_substeps_for_unconstrained = sim.callbacks.AllowSubstepsForUnconstrainedBodies;
        in.integrate_velocity_for_kinematics = sim.callbacks.IntegrateVelocityForKinematics;
        return in;
    }
    // ---- mode 2: the resident frame ----
    // Bundles of a non-contact type batch whose host copy differs from what the device holds (the shadow), as ranges; the shadow takes the host's values.
    static void ChangedRanges(const float* host, float* shadow, int bundles, int floatsPerBundle, std::vector<std::pair<int, int>>& ranges) {
        ranges.clear();
        for (int b = 0; b < bundles; ++b) {
            const size_t at = (size_t)b * floatsPerBundle;
            if (std::memcmp(host + at, shadow + at, (size_t)floatsPerBundle * 4) == 0) continue;
            std::memcpy(shadow + at, host + at, (size_t)floatsPerBundle * 4);
            if (!ranges.empty() && ranges.back().first + ranges.back().second == b) ++ranges.back().second; else ranges.push_back({b, 1});
        }
    }
    void ResidentFrame(Simulation& sim, float dt) {
        Solver& solver = sim.solver;
        if (uploadedSolverVersion == ~0ull || uploadedBodiesVersion != sim.bodies.TopologyVersion) {
            check(api.bepuhip_set_bodies(ctx, sim.bodies.DynamicsState.data(), sim.bodies.Count()));
            Upload(sim);
        } else {
            DiffAndApply(solver, sim.bodies, true);  // every frame, like the C# shim: the reference has no version counter to ask
            if (uploadedSolverVersion != solver.TopologyVersion) { ++structuralReplays; uploadedSolverVersion = solver.TopologyVersion; }
            check(api.bepuhip_set_bodies(ctx, sim.bodies.DynamicsState.data(), sim.bodies.Count()));  // the host's array is authoritative (ResendBodiesEveryFrame)
            SendKinematics(sim);
            int32_t schedule = 0;
            check(api.bepuhip_get_schedule(ctx, &schedule));
            if (schedule == 0 && ++framesOffPlan >= replanInterval) { check(api.bepuhip_replan(ctx)); framesOffPlan = 0; }
        }
        solver.ConsumeStructuralLog();
        // what the host rewrote in place since the last frame
        std::vector<bepuhip_row_transfer> in, out;
        std::vector<std::pair<int, int>> ranges;
        for (size_t b = 0; b < solver.Batches.size(); ++b)
            for (TypeBatch& tb : solver.Batches[b].TypeBatches) {
                if (tb.ConstraintCount == 0) continue;
                const int pfb = tb.Info.prestepFloats * kBundleWidth, ifb = tb.Info.impulseFloats * kBundleWidth;
                if (tb.Info.incremental) {  // contacts: the narrow phase rewrites all of them (NarrowPhaseConstraintUpdate.cs:147-207)
                    in.push_back({BEPUHIP_ROWS_UPDATE_PRESTEP, (int32_t)b, tb.TypeId, 0, -1, 0, tb.PrestepData.data()});
                    in.push_back({BEPUHIP_ROWS_UPDATE_IMPULSES, (int32_t)b, tb.TypeId, 0, -1, 0, tb.AccumulatedImpulses.data()});
                    if (readBackContactDepths) out.push_back({BEPUHIP_ROWS_GET_PRESTEP, (int32_t)b, tb.TypeId, 0, -1, 0, tb.PrestepData.data()});
                } else {
                    Mirror& m = mirrors[((uint64_t)b << 32) | (uint32_t)tb.TypeId];
                    ChangedRanges(tb.PrestepData.data(), m.prestep.data(), tb.BundleCount(), pfb, ranges);
                    for (auto& r : ranges) { in.push_back({BEPUHIP_ROWS_UPDATE_PRESTEP, (int32_t)b, tb.TypeId, r.first, r.second, 0, tb.PrestepData.data() + (size_t)r.first * pfb}); refreshedBundles += r.second; }
                    ChangedRanges(tb.AccumulatedImpulses.data(), m.impulses.data(), tb.BundleCount(), ifb, ranges);
                    for (auto& r : ranges) { in.push_back({BEPUHIP_ROWS_UPDATE_IMPULSES, (int32_t)b, tb.TypeId, r.first, r.second, 0, tb.AccumulatedImpulses.data() + (size_t)r.first * ifb}); refreshedBundles += r.second; }
                }
                out.push_back({BEPUHIP_ROWS_GET_IMPULSES, (int32_t)b, tb.TypeId, 0, -1, 0, tb.AccumulatedImpulses.data()});
            }
        if (!in.empty()) check(api.bepuhip_transfer_rows_async(ctx, in.data(), (int32_t)in.size()));
        std::vector<int32_t> iterations = sim.solveDescription.ResolveIterations();
        bepuhip_integrator integ = Integrator(sim);
        check(api.bepuhip_solve_async(ctx, dt, sim.solveDescription.SubstepCount, iterations.data(), &integ));
        check(api.bepuhip_get_poses_and_velocities_async(ctx, sim.bodies.DynamicsState.data(), sim.bodies.Count()));
        if (!out.empty()) check(api.bepuhip_transfer_rows_async(ctx, out.data(), (int32_t)out.size()));
        check(api.bepuhip_sync(ctx));
        for (size_t b = 0; b < solver.Batches.size(); ++b)  // the device's impulses are the host's again: the shadows follow
            for (TypeBatch& tb : solver.Batches[b].TypeBatches) {
                if (tb.ConstraintCount == 0 || tb.Info.incremental) continue;
                Mirror& m = mirrors[((uint64_t)b << 32) | (uint32_t)tb.TypeId];
                std::memcpy(m.impulses.data(), tb.AccumulatedImpulses.data(), m.impulses.size() * 4);
            }
    }
    int framesOffPlan = 0, replanInterval = 30;
    void Timestep(Simulation& sim, float dt) override {
        // simulation.Sleep / PredictBoundingBoxes / CollisionDetection (DefaultTimestepper.cs:30-37) are out of scope: no-ops here.
        // ---- simulation.Solve(dt) replaced (DefaultTimestepper.cs:39) ----
        if (mode == 2) return ResidentFrame(sim, dt);
        Solver& solver = sim.solver;
        // Topology is re-uploaded only when it changed: keyed on the version counters Solver.Add / Bodies.Add bump (a count comparison would miss a
        // remove + add, or a body move that renumbers references); bodies are host-authoritative every frame.
        check(api.bepuhip_set_bodies(ctx, sim.bodies.DynamicsState.data(), sim.bodies.Count()));
        // Constraint changes since the last frame: replayed through the structural calls when the log covers exactly what happened since the upload (and is short
        // next to a re-upload); the library keeps them on the island layout where it can (include/bepuhip.h).
        if (mode == 1 && uploadedBodiesVersion == sim.bodies.TopologyVersion && uploadedSolverVersion != ~0ull) {
            if (uploadedSolverVersion != solver.TopologyVersion) {
                DiffAndApply(solver, sim.bodies, false);
                // A constraint that kept its handle AND its bodies is the same constraint to the diff — also when it was removed and added again in this frame (the pool
                // hands the handle back): the device then still holds the old one's impulses where the host's start from zero (TypeProcessor.cs:327). This mode reads
                // everything back every frame, so the host's buffers are the truth: the type batches whose arrangement changed are sent whole.
                std::vector<bepuhip_row_transfer> rows;
                for (auto& key : changedTypeBatches) {
                    const ConstraintBatch& batch = solver.Batches[key.first];
                    const TypeBatch& tb = batch.TypeBatches[batch.TypeIndexToTypeBatchIndex.at(key.second)];
                    if (tb.ConstraintCount == 0) continue;
                    rows.push_back({BEPUHIP_ROWS_UPDATE_PRESTEP, key.first, key.second, 0, -1, 0, const_cast<float*>(tb.PrestepData.data())});
                    rows.push_back({BEPUHIP_ROWS_UPDATE_IMPULSES, key.first, key.second, 0, -1, 0, const_cast<float*>(tb.AccumulatedImpulses.data())});
                }
                if (!rows.empty()) { check(api.bepuhip_transfer_rows_async(ctx, rows.data(), (int32_t)rows.size())); check(api.bepuhip_sync(ctx)); }
                SendKinematics(sim);
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
            SendKinematics(sim);  // additions bring kinematic bodies into Solver.ConstrainedKinematicHandles, removals take them out (the count alone does not tell: one of each leaves it unchanged)
            uploadedSolverVersion = solver.TopologyVersion;
            ++structuralReplays;
        }
        solver.ConsumeStructuralLog();
        if (uploadedSolverVersion != solver.TopologyVersion || uploadedBodiesVersion != sim.bodies.TopologyVersion) Upload(sim);
        std::vector<int32_t> iterations = sim.solveDescription.ResolveIterations();
        bepuhip_integrator in = Integrator(sim);
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
int32_t bepuhost_apply_description(void* s, int32_t constraintHandle, const float* prestepLane) {
    try { ((Simulation*)s)->solver.ApplyDescription(constraintHandle, prestepLane); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int32_t bepuhost_set_accumulated_impulses(void* s, int32_t constraintHandle, const float* impulseLane) {
    try { ((Simulation*)s)->solver.SetAccumulatedImpulses(constraintHandle, impulseLane); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// (batch index, type id, index in type batch) of a live constraint; -1 when the handle names none.
int32_t bepuhost_constraint_location(void* s, int32_t constraintHandle, int32_t* out3) {
    const Solver& solver = ((Simulation*)s)->solver;
    if (constraintHandle < 0 || constraintHandle >= (int)solver.HandleToConstraint.size() || solver.HandleToConstraint[constraintHandle].BatchIndex < 0) return -1;
    const ConstraintLocation& loc = solver.HandleToConstraint[constraintHandle];
    out3[0] = loc.BatchIndex; out3[1] = loc.TypeId; out3[2] = loc.IndexInTypeBatch;
    return 0;
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
    std::vector<bepuhip_structural_op> ops, rest;
    std::vector<uint32_t> payload;
    DiffTypeBatch(batch, typeId, bodies, prestepFloats, oldHandles, oldCount, oldReferences, nullptr, newHandles, newCount, newReferencesAosoa, nullptr, newPrestepAosoa, ops, rest, payload, nullptr);
    ops.insert(ops.end(), rest.begin(), rest.end());  // one type batch on its own: its removals, then the rest
    if ((int)ops.size() > opCapacity || (int)payload.size() > payloadCapacity) return -1;
    if (!ops.empty()) std::memcpy(opsOut, ops.data(), ops.size() * sizeof(bepuhip_structural_op));
    if (!payload.empty()) std::memcpy(payloadOut, payload.data(), payload.size() * 4);
    *payloadWords = (int32_t)payload.size();
    return (int32_t)ops.size();
}
// ... with the constraints' identities (constraint handle + the handles of its bodies, see DiffTypeBatch): a handle the pool handed out again for another constraint is a
// removal plus an addition. survivorOut (optional, newCount entries): last frame's index of every constraint of the new arrangement, -1 for a new one. The removals come first.
int32_t bepuhost_diff_type_batch_identities(int batch, int typeId, int bodies, int prestepFloats, const int32_t* oldHandles, int oldCount, const int32_t* oldReferences,
                                            const int32_t* oldBodyHandles, const int32_t* newHandles, int newCount, const int32_t* newReferencesAosoa, const int32_t* newBodyHandles,
                                            const float* newPrestepAosoa, int32_t* opsOut, int opCapacity, uint32_t* payloadOut, int payloadCapacity, int32_t* payloadWords,
                                            int32_t* survivorOut) {
    std::vector<bepuhip_structural_op> ops, rest;
    std::vector<uint32_t> payload;
    std::vector<int32_t> survivor;
    DiffTypeBatch(batch, typeId, bodies, prestepFloats, oldHandles, oldCount, oldReferences, oldBodyHandles, newHandles, newCount, newReferencesAosoa, newBodyHandles, newPrestepAosoa, ops,
                  rest, payload, &survivor);
    ops.insert(ops.end(), rest.begin(), rest.end());
    if ((int)ops.size() > opCapacity || (int)payload.size() > payloadCapacity) return -1;
    if (!ops.empty()) std::memcpy(opsOut, ops.data(), ops.size() * sizeof(bepuhip_structural_op));
    if (!payload.empty()) std::memcpy(payloadOut, payload.data(), payload.size() * 4);
    if (survivorOut && newCount > 0) std::memcpy(survivorOut, survivor.data(), (size_t)newCount * 4);
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
// The resident frame's counters: structural operations the diffs emitted, bundles of joint prestep data / impulses found changed on the host and sent, the schedule the
// context is on (bepuhip_get_schedule).
int32_t bepuhost_resident_stats(void* s, int64_t* out3) {
    HipTimestepper* t = dynamic_cast<HipTimestepper*>(((Simulation*)s)->timestepper);
    if (!t) { g_err = "no HipTimestepper attached"; return -1; }
    int32_t schedule = -1;
    t->api.bepuhip_get_schedule(t->ctx, &schedule);
    out3[0] = t->diffOperations; out3[1] = t->refreshedBundles; out3[2] = schedule;
    return 0;
}
int32_t bepuhost_timestepper_replan_interval(void* s, int frames) {
    HipTimestepper* t = dynamic_cast<HipTimestepper*>(((Simulation*)s)->timestepper);
    if (!t) { g_err = "no HipTimestepper attached"; return -1; }
    t->replanInterval = frames;
    return 0;
}
int32_t bepuhost_timestepper_read_back_contact_depths(void* s, int on) {
    HipTimestepper* t = dynamic_cast<HipTimestepper*>(((Simulation*)s)->timestepper);
    if (!t) { g_err = "no HipTimestepper attached"; return -1; }
    t->readBackContactDepths = on != 0;
    return 0;
}

}  // extern "C"
