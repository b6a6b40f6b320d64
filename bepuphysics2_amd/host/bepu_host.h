// Host-side mirror (C++) of the reference's managed surface for the solver hot path.
// The reference is C# and there is no .NET toolchain in this image, so the host side above the C ABI is C++
// mirroring the reference's names, argument meaning and error behaviour:
//   Bodies / BodyDescription        BepuPhysics/Bodies.cs, BodySet.cs:41,83-134, BodyProperties.cs:11-338
//   Solver.Add + batch colouring    BepuPhysics/Solver.cs:1058-1199  (greedy first-fit, kinematics never block)
//   TypeBatch / ConstraintBatch     BepuPhysics/Constraints/TypeBatch.cs:10-36, ConstraintBatch.cs:14-49
//   constraint descriptions         BepuPhysics/Constraints/*.cs ApplyDescription (AOSOA lane writes, BundleIndexing.cs:50-60)
//   SolveDescription                BepuPhysics/SolveDescription.cs:16-136
//   PrepareConstraintIntegrationResponsibilities   BepuPhysics/Solver_Solve.cs:951-1044,1072-1388
//   Simulation / ITimestepper       BepuPhysics/Simulation.cs:106-326, ITimestepper.cs:60-79, DefaultTimestepper.cs:28-43
// HipTimestepper replaces only `simulation.Solve(dt, dispatcher)` with the C ABI in include/bepuhip.h.
#pragma once
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace bepu {

constexpr int kBundleWidth = 8;  // Vector<float>.Count on the reference's AVX2 hosts; the C ABI takes it as a parameter.
constexpr int kKinematicMask = 1 << 30;           // Bodies_GatherScatter.cs:107-118
constexpr int kBodyReferenceMask = 0x3FFFFFFF;
constexpr int kFallbackBatchThreshold = 64;       // SolveDescription.cs:38

struct Vector3 { float X = 0, Y = 0, Z = 0; };
struct Quaternion { float X = 0, Y = 0, Z = 0, W = 1; };
struct Symmetric3x3 { float XX = 0, YX = 0, YY = 0, ZX = 0, ZY = 0, ZZ = 0; };
struct RigidPose { Vector3 Position; Quaternion Orientation; };
struct BodyVelocity { Vector3 Linear, Angular; };
struct BodyInertia { Symmetric3x3 InverseInertiaTensor; float InverseMass = 0; };

struct SpringSettings {  // Constraints/SpringSettings.cs:60-80
    float AngularFrequency = 0, TwiceDampingRatio = 0;
    SpringSettings() = default;
    SpringSettings(float frequency, float dampingRatio) : AngularFrequency(frequency * 6.283185307179586477f), TwiceDampingRatio(dampingRatio * 2) {}
};
struct ServoSettings { float MaximumSpeed, BaseSpeed, MaximumForce; };  // Constraints/ServoSettings.cs:60-66
struct MotorSettings {  // Constraints/MotorSettings.cs:15-48 (Damping = 1/softness, softness<=0 -> float.MaxValue)
    float MaximumForce, Damping;
    MotorSettings(float maximumForce, float softness);
};

struct BodyDescription {
    RigidPose Pose;
    BodyVelocity Velocity;
    BodyInertia LocalInertia;  // all zero => kinematic (Bodies.cs:326-349)
    static BodyDescription CreateDynamic(const RigidPose& pose, const BodyInertia& inertia) { return {pose, {}, inertia}; }
    static BodyDescription CreateKinematic(const RigidPose& pose, const BodyVelocity& velocity) { return {pose, velocity, {}}; }
};

// BodyDynamics: 32 floats, 128 bytes (BodyProperties.cs:318-338).
struct BodyDynamics { float f[32]; };

class Bodies {
public:
    std::vector<BodyDynamics> DynamicsState;  // ActiveSet.DynamicsState (BodySet.cs:41)
    std::vector<int32_t> IndexToHandle;
    std::vector<int32_t> HandleToIndex;
    int32_t Add(const BodyDescription& description);
    uint64_t TopologyVersion = 0;  // bumped by every change of the body set (Add; Remove / swap-with-last moves when they exist): what a device mirror keys its uploads on
    int Count() const { return (int)DynamicsState.size(); }
    bool IsKinematic(int index) const;
};

struct TypeInfo { int bodies, prestepFloats, impulseFloats; bool incremental; };
bool GetTypeInfo(int typeId, TypeInfo& info);

struct TypeBatch {  // Constraints/TypeBatch.cs:10-19
    int TypeId = 0;
    int ConstraintCount = 0;
    TypeInfo Info{};
    std::vector<int32_t> BodyReferences;       // AOSOA
    std::vector<float> PrestepData;            // AOSOA
    std::vector<float> AccumulatedImpulses;    // AOSOA
    std::vector<int32_t> IndexToHandle;
    int BundleCount() const { return (ConstraintCount + kBundleWidth - 1) / kBundleWidth; }
    int Allocate(int constraintHandle, const int32_t* encodedBodyIndices);
};
struct ConstraintBatch {  // ConstraintBatch.cs:14-49
    std::vector<TypeBatch> TypeBatches;
    std::unordered_map<int, int> TypeIndexToTypeBatchIndex;
    TypeBatch& GetOrCreateTypeBatch(int typeId);
};
struct IndexSet {  // BepuUtilities/Collections/IndexSet.cs:12-120
    std::vector<uint64_t> Flags;
    bool Contains(int i) const { return (size_t)(i >> 6) < Flags.size() && ((Flags[i >> 6] >> (i & 63)) & 1ull); }
    void Set(int i) { if ((size_t)(i >> 6) >= Flags.size()) Flags.resize((i >> 6) + 1, 0); Flags[i >> 6] |= 1ull << (i & 63); }
    void Unset(int i) { if ((size_t)(i >> 6) < Flags.size()) Flags[i >> 6] &= ~(1ull << (i & 63)); }
};
struct ConstraintLocation { int BatchIndex, TypeId, IndexInTypeBatch; };

struct SolveDescription {  // SolveDescription.cs:16-136
    int VelocityIterationCount = 1;
    int SubstepCount = 1;
    int FallbackBatchThreshold = kFallbackBatchThreshold;
    std::function<int(int)> VelocityIterationScheduler;  // returns <1 => use VelocityIterationCount (:33)
    SolveDescription(int velocityIterationCount, int substepCount, int fallbackBatchThreshold = kFallbackBatchThreshold);
    std::vector<int32_t> ResolveIterations() const;  // Solver_Solve.cs:743-751
};

struct PoseIntegratorCallbacks {  // Demos/DemoCallbacks.cs:20-109 (the only callback shape the device path supports)
    Vector3 Gravity{0, -10, 0};
    float LinearDamping = 0.03f, AngularDamping = 0.03f;
    bool AllowSubstepsForUnconstrainedBodies = false;
    bool IntegrateVelocityForKinematics = false;
};

class Solver {
public:
    explicit Solver(Bodies& bodies) : bodies(bodies) {}
    std::vector<ConstraintBatch> Batches;             // ActiveSet.Batches
    std::vector<IndexSet> batchReferencedHandles;     // Solver.cs:33
    std::vector<int32_t> ConstrainedKinematicHandles; // Solver.cs:68
    std::vector<ConstraintLocation> HandleToConstraint;
    int ConstraintCount() const { return liveConstraints; }
    uint64_t TopologyVersion = 0;  // bumped by every Add / Remove: a remove + add leaves the count unchanged but not the layout
    // Solver.Add(bodyHandles, description): prestepLane holds the description's fields in prestep order (what ApplyDescription writes).
    int Add(const int32_t* bodyHandles, int bodyCount, int typeId, const float* prestepLane);
    // Solver.ApplyDescription(handle, description) (Solver.cs:1162-1185): the description's fields written over the constraint's prestep lane, in place — what
    // Demos/Demos/Tanks/Tank.cs:100,139 and Cars/SimpleCar.cs:25 do every frame to their motors and servos. Nothing else changes; nobody is told.
    void ApplyDescription(int constraintHandle, const float* prestepLane);
    // The constraint's accumulated impulses, read and written in place (Solver.GetAccumulatedImpulses / the awakener's bulk copies that restore them, IslandAwakener.cs:388-400).
    void SetAccumulatedImpulses(int constraintHandle, const float* impulseLane);
    void GetAccumulatedImpulses(int constraintHandle, float* impulseLane) const;
    std::vector<int32_t> HandlePool;  // freed constraint handles, reused last-in-first-out like the reference's IdPool.Take (Solver.HandlePool): a Remove followed by an Add returns the same handle
    // Solver.Remove(handle) (Solver.cs:1528-1560 -> ConstraintBatch.Remove -> TypeProcessor.Remove, TypeProcessor.cs:634-731): the last constraint of the type batch moves
    // into the freed index (TypeProcessor.Move :578-592), handle -> location of the moved constraint is fixed up, the batch forgets the removed constraint's dynamic bodies.
    void Remove(int constraintHandle);
    // What changed since a device mirror last looked, in order, in the terms of include/bepuhip.h's structural updates: a mirror whose upload was taken at
    // StructuralLogBase replays the log instead of re-uploading (HipTimestepper) and then calls ConsumeStructuralLog.
    struct StructuralChange { bool add; int batch, typeId, index; int32_t encoded[4]; std::vector<float> prestep; };
    std::vector<StructuralChange> StructuralLog;
    uint64_t StructuralLogBase = 0;  // TopologyVersion the log starts from
    void ConsumeStructuralLog() { StructuralLog.clear(); StructuralLogBase = TopologyVersion; }
    // Integration-responsibility prepass (Solver_Solve.cs:1072-1388).
    struct IntegrationResponsibilities {
        IndexSet mergedConstrainedBodyHandles;
        // [batch][typeBatch][slot] -> IndexSet over constraint indices; batch 0 empty.
        std::vector<std::vector<std::vector<IndexSet>>> integrationFlags;
        std::vector<std::vector<uint8_t>> coarseBatchIntegrationResponsibilities;
    };
    IntegrationResponsibilities PrepareConstraintIntegrationResponsibilities() const;
    void ValidateBatches() const;  // "no dynamic body twice in a non-fallback batch" (Solver.cs:1046-1051)
private:
    Bodies& bodies;
    std::vector<int32_t> kinematicConstrained;  // per body handle: constraints that reference the (kinematic) body (Bodies' constraint lists, as far as this set needs them)
    int liveConstraints = 0;
};

class Simulation;
struct ITimestepper {  // ITimestepper.cs:60-79
    virtual ~ITimestepper() = default;
    virtual void Timestep(Simulation& simulation, float dt) = 0;
};

class Simulation {  // Simulation.cs:106-326 (collision detection, sleeping etc. are out of scope and are no-ops here)
public:
    Bodies bodies;
    Solver solver;
    SolveDescription solveDescription;
    PoseIntegratorCallbacks callbacks;
    ITimestepper* timestepper = nullptr;
    Simulation(const PoseIntegratorCallbacks& callbacks, const SolveDescription& solveDescription, ITimestepper* timestepper = nullptr)
        : solver(bodies), solveDescription(solveDescription), callbacks(callbacks), timestepper(timestepper) {}
    void Timestep(float dt);  // throws std::invalid_argument for dt <= 0 (Simulation.cs:318-319)
};

}  // namespace bepu
