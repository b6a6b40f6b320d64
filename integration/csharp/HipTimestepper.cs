// Reference-side binding of libbepuhip.so (include/bepuhip.h): drop this file into an application that references BepuPhysics and create the simulation with
// `new HipTimestepper<TCallbacks>()`. It uses public API only, on the UNPATCHED reference: the structural changes of a frame (narrow-phase adds and removes, sleeping,
// awakening, user calls, bodies moving in memory) are reconstructed by diffing every type batch's constraint handles (TypeBatch.IndexToHandle, TypeBatch.cs:16) and body
// references against last frame's copy, and sent to the device in one bepuhip_apply_structural_ops call; what the host rewrote IN PLACE since the last frame — Solver.ApplyDescription
// on motors and servos (Demos/Demos/Tanks/Tank.cs:100,139; Cars/SimpleCar.cs:25), impulses the awakener restored, a constraint handle the pool handed out again — is found by
// comparing the type batches' buffers with a shadow of what the device holds and sent by bundle range (bepuhip_transfer_rows_async); poses, velocities and ALL accumulated
// impulses come back behind the solve. INTEGRATION.md carries the same text (a test keeps the two
// identical, and the DllImport block is generated from the header by tools/gen_csharp_imports.py); the diff has a C++ twin that IS compiled and tested here
// (bepuphysics2_amd/host/bepu_host.cpp DiffTypeBatch; tests/test_structural_diff.py, tests/test_gpu_structural.py) — this text itself is not (no .NET SDK in this environment).
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using BepuPhysics;
using BepuPhysics.CollisionDetection;
using BepuPhysics.Constraints;
using BepuUtilities;
using BepuUtilities.Memory;

[StructLayout(LayoutKind.Sequential)]
struct BepuHipConfig { public int DeviceOrdinal, BundleWidth, Flags; }

[StructLayout(LayoutKind.Sequential)]
unsafe struct BepuHipIntegrator
{
    public fixed float Gravity[3];
    public float LinearDamping, AngularDamping;
    public int AngularIntegrationMode, AllowSubstepsForUnconstrained, IntegrateVelocityForKinematics;
}

[StructLayout(LayoutKind.Sequential)]
unsafe struct BepuHipVelocityModel { public int Model; public fixed float Center[3]; public float Gravity; }   // bepuhip_velocity_model: 0 uniform, 1 per body, 2 radial

[StructLayout(LayoutKind.Sequential)]
struct BepuHipStructuralOp { public int Kind, BatchIndex, TypeId, Index, Slot, Reference, PayloadOffset, Reserved; }   // bepuhip_structural_op: 0 add, 1 remove, 2 update reference, 3 swap

[StructLayout(LayoutKind.Sequential)]
unsafe struct BepuHipRowTransfer { public int Kind, BatchIndex, TypeId, FirstBundle, BundleCount, Reserved; public void* Bundles; }   // bepuhip_row_transfer: 0 / 1 update prestep / impulses, 2 / 3 get

[StructLayout(LayoutKind.Sequential)]
unsafe struct BepuHipCollidable   // 64 bytes, bepuhip_collidable
{
    public int ShapeType; public fixed float Shape[9];
    public float MinimumSpeculativeMargin, MaximumSpeculativeMargin; public int AllowExpansionBeyondSpeculativeMargin;
    public float SleepThreshold; public int MinimumTimestepsUnderThreshold, Activity;
}
[StructLayout(LayoutKind.Sequential)]
unsafe struct BepuHipPredictedBounds { public fixed float Min[3]; public float SpeculativeMargin; public fixed float Max[3]; public int Activity; }   // 32 bytes
[StructLayout(LayoutKind.Sequential)]
unsafe struct BepuHipCompoundChild { public int ShapeType; public fixed float Shape[9]; public fixed float LocalPosition[3]; public fixed float LocalOrientation[4]; }   // 68 bytes

static unsafe class BepuHip
{
    const string Lib = "bepuhip"; // libbepuhip.so
    public const int FlagNoGraph = 1, FlagNoClusters = 2, FlagReserveUpdateSlots = 8, FlagExclusiveDevice = 16;
    // ---- generated from include/bepuhip.h (tools/gen_csharp_imports.py): every entry point of the header ----
    [DllImport(Lib)] public static extern IntPtr bepuhip_last_error();
    [DllImport(Lib)] public static extern int bepuhip_create(BepuHipConfig* config, IntPtr* outCtx);
    [DllImport(Lib)] public static extern int bepuhip_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_set_velocity_model(IntPtr ctx, BepuHipVelocityModel* model, float* perBodyGravity, int bodyCount);
    [DllImport(Lib)] public static extern int bepuhip_set_bodies(IntPtr ctx, void* bodyDynamicsAos, int count);
    [DllImport(Lib)] public static extern int bepuhip_begin_constraints(IntPtr ctx, int batchCount, int fallbackBatchThreshold);
    [DllImport(Lib)] public static extern int bepuhip_set_type_batch(IntPtr ctx, int batchIndex, int typeId, int constraintCount, int* bodyReferencesAosoa, float* prestepAosoa, float* accumulatedImpulsesAosoa);
    [DllImport(Lib)] public static extern int bepuhip_end_constraints(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_set_constrained_kinematics(IntPtr ctx, int* bodyIndices, int count);
    [DllImport(Lib)] public static extern int bepuhip_solve(IntPtr ctx, float dt, int substepCount, int* velocityIterations, BepuHipIntegrator* integrator);
    [DllImport(Lib)] public static extern int bepuhip_solve_with_substep_events(IntPtr ctx, float dt, int substepCount, int* velocityIterations, BepuHipIntegrator* integrator, delegate* unmanaged[Cdecl]<void*, int, void> started, delegate* unmanaged[Cdecl]<void*, int, void> ended, void* user);
    [DllImport(Lib)] public static extern int bepuhip_set_boundary_bodies(IntPtr ctx, int* bodyIndices, int count);
    [DllImport(Lib)] public static extern int bepuhip_boundary_deltas(IntPtr ctx, float* deltasOut, int outIsDevicePointer);
    [DllImport(Lib)] public static extern int bepuhip_boundary_apply(IntPtr ctx, float* summedDeltas, int inIsDevicePointer);
    [DllImport(Lib)] public static extern int bepuhip_solve_exchanged(IntPtr ctx, float dt, int substepCount, int* velocityIterations, BepuHipIntegrator* integrator, delegate* unmanaged[Cdecl]<void*, int, int, int> fn, void* user);
    [DllImport(Lib)] public static extern int bepuhip_colour_constraints(int device, int* refs, int count, int bodyCount, int order, int fallbackBatchThreshold, int* coloursOut, int* batchCountOut, int* roundsOut);
    [DllImport(Lib)] public static extern int bepuhip_set_exchange_mode(IntPtr ctx, int mode);
    [DllImport(Lib)] public static extern int bepuhip_set_boundary_layout(IntPtr ctx, int* denseRows, int denseRowCount, float* holders);
    [DllImport(Lib)] public static extern int bepuhip_comm_unique_id(void* idOut);
    [DllImport(Lib)] public static extern int bepuhip_comm_init(IntPtr ctx, void* id, int rank, int world);
    [DllImport(Lib)] public static extern int bepuhip_comm_adopt(IntPtr ctx, void* ncclComm, int world);
    [DllImport(Lib)] public static extern int bepuhip_solve_lattice(IntPtr ctx, float dt, int substepCount, int* velocityIterations, BepuHipIntegrator* integrator);
    [DllImport(Lib)] public static extern int bepuhip_set_device_group(IntPtr ctx, int world, int rank);
    [DllImport(Lib)] public static extern int bepuhip_get_shared_records(IntPtr ctx, void** recordsOut, long* bytesOut);
    [DllImport(Lib)] public static extern int bepuhip_set_peer_records(IntPtr ctx, int peer, void* records);
    [DllImport(Lib)] public static extern int bepuhip_export_shared_records(IntPtr ctx, void* ipcHandleOut);
    [DllImport(Lib)] public static extern int bepuhip_import_peer_records(IntPtr ctx, int peer, void* ipcHandle);
    [DllImport(Lib)] public static extern int bepuhip_get_owned_bodies(IntPtr ctx, byte* maskOut, int count);
    [DllImport(Lib)] public static extern int bepuhip_get_owned_constraints(IntPtr ctx, int batchIndex, int typeId, byte* maskOut);
    [DllImport(Lib)] public static extern int bepuhip_sync_owned_bodies(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_get_bodies(IntPtr ctx, void* bodyDynamicsAosOut, int count);
    [DllImport(Lib)] public static extern int bepuhip_register_host_memory(IntPtr ctx, void* memory, long bytes);
    [DllImport(Lib)] public static extern int bepuhip_unregister_host_memory(IntPtr ctx, void* memory);
    [DllImport(Lib)] public static extern int bepuhip_get_poses_and_velocities(IntPtr ctx, void* bodyDynamicsAosOut, int count);
    [DllImport(Lib)] public static extern int bepuhip_get_poses_and_velocities_async(IntPtr ctx, void* bodyDynamicsAosOut, int count);
    [DllImport(Lib)] public static extern int bepuhip_get_accumulated_impulses(IntPtr ctx, int batchIndex, int typeId, float* accumulatedImpulsesAosoaOut);
    [DllImport(Lib)] public static extern int bepuhip_get_prestep(IntPtr ctx, int batchIndex, int typeId, float* prestepAosoaOut);
    [DllImport(Lib)] public static extern int bepuhip_update_bodies(IntPtr ctx, void* bodyDynamicsAos, int first, int count);
    [DllImport(Lib)] public static extern int bepuhip_update_prestep(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, float* prestepBundles);
    [DllImport(Lib)] public static extern int bepuhip_update_prestep_async(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, float* prestepBundles);
    [DllImport(Lib)] public static extern int bepuhip_update_accumulated_impulses_async(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, float* impulseBundles);
    [DllImport(Lib)] public static extern int bepuhip_update_accumulated_impulses(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, float* impulseBundles);
    [DllImport(Lib)] public static extern int bepuhip_get_bodies_range(IntPtr ctx, void* bodyDynamicsAosOut, int first, int count);
    [DllImport(Lib)] public static extern int bepuhip_get_prestep_range(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, float* prestepBundlesOut);
    [DllImport(Lib)] public static extern int bepuhip_get_accumulated_impulses_range(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, float* impulseBundlesOut);
    [DllImport(Lib)] public static extern int bepuhip_transfer_rows_async(IntPtr ctx, BepuHipRowTransfer* items, int count);
    [DllImport(Lib)] public static extern int bepuhip_add_constraint(IntPtr ctx, int batchIndex, int typeId, int* encodedBodyReferences, float* prestepLane, int* indexOut);
    [DllImport(Lib)] public static extern int bepuhip_remove_constraint(IntPtr ctx, int batchIndex, int typeId, int index);
    [DllImport(Lib)] public static extern int bepuhip_add_constraint_at(IntPtr ctx, int batchIndex, int typeId, int index, int* encodedBodyReferences, float* prestepLane);
    [DllImport(Lib)] public static extern int bepuhip_update_body_reference(IntPtr ctx, int batchIndex, int typeId, int index, int bodyIndexInConstraint, int encodedBodyReference);
    [DllImport(Lib)] public static extern int bepuhip_swap_constraints(IntPtr ctx, int batchIndex, int typeId, int indexA, int indexB);
    [DllImport(Lib)] public static extern int bepuhip_apply_structural_ops(IntPtr ctx, BepuHipStructuralOp* ops, int count, uint* payload, int payloadWords, int* failedOpOut);
    [DllImport(Lib)] public static extern int bepuhip_get_constraint_count(IntPtr ctx, int batchIndex, int typeId, int* countOut);
    [DllImport(Lib)] public static extern int bepuhip_get_schedule(IntPtr ctx, int* scheduleOut);
    [DllImport(Lib)] public static extern int bepuhip_get_kernel_family(IntPtr ctx, int* familyOut);
    [DllImport(Lib)] public static extern int bepuhip_specialise_units(IntPtr ctx, int wait, int* stateOut);
    [DllImport(Lib)] public static extern int bepuhip_prebuild_unit(ulong typeMask, int threadsBudget, int splitPlan, byte* pathOut, int pathCapacity);
    [DllImport(Lib)] public static extern int bepuhip_replan(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_replan_begin(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_replan_poll(IntPtr ctx, int* stateOut);
    [DllImport(Lib)] public static extern int bepuhip_replan_commit(IntPtr ctx, int wait, int* committedOut);
    [DllImport(Lib)] public static extern int bepuhip_replan_cancel(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_set_convex_hulls(IntPtr ctx, float* points, int* pointBegin, int hullCount);
    [DllImport(Lib)] public static extern int bepuhip_set_compounds(IntPtr ctx, BepuHipCompoundChild* children, int* childBegin, int compoundCount);
    [DllImport(Lib)] public static extern int bepuhip_set_meshes(IntPtr ctx, float* triangles, int* triangleBegin, float* scales, int meshCount);
    [DllImport(Lib)] public static extern int bepuhip_set_collidables(IntPtr ctx, BepuHipCollidable* collidables, int count);
    [DllImport(Lib)] public static extern int bepuhip_predict_bounding_boxes(IntPtr ctx, float dt, BepuHipIntegrator* integrator, BepuHipCollidable* collidables, int count, BepuHipPredictedBounds* boundsOut);
    [DllImport(Lib)] public static extern int bepuhip_get_constrained_flags(IntPtr ctx, byte* flagsOut, int count);
    [DllImport(Lib)] public static extern int bepuhip_set_solve_timing(IntPtr ctx, int enabled);
    [DllImport(Lib)] public static extern int bepuhip_last_solve_ms(IntPtr ctx, float* msOut);
    [DllImport(Lib)] public static extern int bepuhip_set_profiling(IntPtr ctx, int enabled);
    [DllImport(Lib)] public static extern int bepuhip_get_profile(IntPtr ctx, int family, float* msOut, int* launchesOut);
    [DllImport(Lib)] public static extern int bepuhip_set_cluster_trace(IntPtr ctx, int enabled);
    [DllImport(Lib)] public static extern int bepuhip_get_cluster_trace(IntPtr ctx, ulong* wordsOut, long capacityWords, int* itemsOut);
    [DllImport(Lib)] public static extern int bepuhip_get_cluster_cycles(IntPtr ctx, ulong* cyclesOut, int capacity, int* countOut);
    [DllImport(Lib)] public static extern int bepuhip_get_row_policy(IntPtr ctx, int* policyOut);
    [DllImport(Lib)] public static extern int bepuhip_debug_status(IntPtr ctx, uint* words16Out);
    [DllImport(Lib)] public static extern int bepuhip_last_constraint_iterations(IntPtr ctx, long* iterationsOut);
    [DllImport(Lib)] public static extern int bepuhip_get_stream(IntPtr ctx, void** streamOut);
    [DllImport(Lib)] public static extern int bepuhip_solve_async(IntPtr ctx, float dt, int substepCount, int* velocityIterations, BepuHipIntegrator* integrator);
    [DllImport(Lib)] public static extern int bepuhip_sync(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_reset_state(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_type_info(int typeId, int* bodiesPerConstraint, int* prestepFloats, int* impulseFloats);
    // ---- end of the generated block ----
}

/// What the device needs to know about a velocity callback: IPoseIntegratorCallbacks.IntegrateVelocity is arbitrary code and cannot cross a C ABI, so the callback
/// struct states which of the library's models it implements (bepuhip_velocity_model) by carrying one of these interfaces — its fields become the getters:
///   IHipVelocityModel        "velocity += gravity * dt, then exponential damping": Demos/DemoCallbacks.cs:100-109
///   IHipPerBodyGravityModel  "velocity.Linear.Y += BodyGravities[handle] * dt":    Demos/Demos/PerBodyGravityDemo.cs:57-88
///   IHipRadialGravityModel   "velocity.Linear -= dt * Gravity * offset / max(1, |offset|^3)": Demos/Demos/PlanetDemo.cs:36-47
/// A callback struct with none of them keeps simulation.Solve (the shim falls back), as does a host that subscribes to Solver.SubstepStarted / SubstepEnded.
public interface IHipVelocityModel { System.Numerics.Vector3 Gravity { get; } float LinearDamping { get; } float AngularDamping { get; } }
public interface IHipPerBodyGravityModel { CollidableProperty<float> BodyGravities { get; } }
public interface IHipRadialGravityModel { System.Numerics.Vector3 PlanetCenter { get; } float Gravity { get; } }

// [UnmanagedCallersOnly] methods may not live in a generic type: the native side calls these two, which hand the event to the timestepper behind the GCHandle.
interface IHipSubstepSink { void RaiseStarted(int substep); void RaiseEnded(int substep); }
static unsafe class HipSubstepTrampoline
{
    [UnmanagedCallersOnly(CallConvs = new[] { typeof(System.Runtime.CompilerServices.CallConvCdecl) })]
    public static void Started(void* user, int substep) => ((IHipSubstepSink)GCHandle.FromIntPtr((IntPtr)user).Target).RaiseStarted(substep);
    [UnmanagedCallersOnly(CallConvs = new[] { typeof(System.Runtime.CompilerServices.CallConvCdecl) })]
    public static void Ended(void* user, int substep) => ((IHipSubstepSink)GCHandle.FromIntPtr((IntPtr)user).Target).RaiseEnded(substep);
}

public unsafe class HipTimestepper<TCallbacks> : ITimestepper, IHipSubstepSink, IDisposable where TCallbacks : struct, IPoseIntegratorCallbacks
{
    IntPtr ctx;
    bool resident;                                    // the device holds the scene as of the last solve
    // What the device holds of every type batch, keyed by (batch index, type id): the constraint handles by index (TypeBatch.IndexToHandle is public, TypeBatch.cs:16-19;
    // handles are stable for a constraint's life, Solver.HandlePool) and the encoded body references by index. The diff against this frame's type batches IS the frame's
    // structural change: whoever made it — NarrowPhasePendingConstraintAdds, ConstraintRemover, Solver.Add / Remove, the sleeper, IslandAwakener's bulk copies
    // (IslandAwakener.cs:388-400, which no per-constraint hook sees), Bodies.RemoveAt's reference patches (Solver.UpdateForBodyMemoryMove, Solver.cs:1475).
    // A constraint's identity is its handle AND the handles of its bodies: Solver.HandlePool hands a freed handle out again last-in-first-out (IdPool.Take), so Solver.Remove(h)
    // followed by Solver.Add(...) in one frame returns h for a different constraint, possibly at the same index of the same type batch. BodyHandles tells the two apart.
    // Prestep / Impulses (non-contact type batches): what the DEVICE holds, in the type batch's own AOSOA layout — compared with the host's buffers every frame.
    sealed class Mirror
    {
        public int[] Handles = Array.Empty<int>(), References = Array.Empty<int>(), BodyHandles = Array.Empty<int>();
        public float[] Prestep = Array.Empty<float>(), Impulses = Array.Empty<float>();
        public int Count, Bodies, PrestepFloats, ImpulseFloats; public bool Seen, Contact;
    }
    readonly Dictionary<long, Mirror> mirrors = new Dictionary<long, Mirror>();
    readonly List<BepuHipStructuralOp> ops = new List<BepuHipStructuralOp>(), removals = new List<BepuHipStructuralOp>();
    readonly List<BepuHipRowTransfer> rowsIn = new List<BepuHipRowTransfer>(), rowsOut = new List<BepuHipRowTransfer>();
    /// Every frame the prestep data and accumulated impulses of every NON-contact type batch are compared, bundle by bundle, with what the device holds, and the bundles that
    /// differ are sent: Solver.ApplyDescription (Solver.cs:1162-1185) writes into TypeBatch.PrestepData in place and raises no event. About 1 ms per 40 MB of joint data on a
    /// few threads. A host that tells the shim what it rewrote (MarkDirty) may turn the comparison off.
    public bool CompareJointData = true;
    /// Accumulated impulses of the joints come back every frame (behind the solve, with the contacts' impulses): Solver.GetDescription / GetAccumulatedImpulses, the sleeper's
    /// copies (IslandSleeper.cs:174-260) and the awakener's restores (IslandAwakener.cs:388-400) then see this frame's values. 7 MB for 870,000 joints. A host that never sleeps
    /// islands and never reads joint impulses may turn it off; FetchJointImpulses() brings them on demand.
    public bool ReadBackJointImpulses = true;
    readonly HashSet<int> dirtyConstraints = new HashSet<int>();
    /// Tells the shim that the constraint's description (or its accumulated impulses) was rewritten since the last frame — only needed with CompareJointData off.
    public void MarkDirty(ConstraintHandle handle) => dirtyConstraints.Add(handle.Value);
    readonly List<uint> payload = new List<uint>();
    public int ReplanInterval = 30;                                // frames between two bepuhip_replan calls at most
    /// The re-plan's host planner (20-36 ms for a million constraints) on a thread of the library's own (bepuhip_replan_begin / _commit) instead of inside Timestep.
    public bool ReplanInBackground = true;
    /// The island kernel compiled for exactly the constraint types the simulation uses (bepuhip_specialise_units: in the background, cached on disk; the prebuilt families
    /// run until it is loaded). Worth 1-4 % of the solve, most for scenes that mix the sixteen common types with a few of the other 28.
    public bool SpecialiseKernels = true;
    bool replanInFlight;
    /// Every body's state is sent before every solve (bepuhip_set_bodies: one DMA from the registered DynamicsState buffer, 0.6 ms for 240,000 bodies): velocities the user
    /// set, ApplyImpulse, teleports, inertia changes and kinematic bodies driven by writing their velocity all reach the device. A host that never writes body state
    /// between frames may turn it off; bodies are then sent when the body count changed or a body moved in memory.
    public bool ResendBodiesEveryFrame = true;
    int residentBodyCount;
    int sentModel; System.Numerics.Vector3 sentCenter; float sentGravity; float[] gravities = Array.Empty<float>();   // the velocity model the context holds
    int framesSinceReplan = 30;
    readonly Dictionary<IntPtr, long> registered = new Dictionary<IntPtr, long>();
    public int ReplayLimit = 262144;                  // more operations than this in one frame are not cheaper than an upload
    public event TimestepperStageHandler BeforeCollisionDetection, CollisionsDetected, ConstraintsSolved; // ITimestepper.cs:62-74 (subset)
    /// Solver.SubstepStarted / SubstepEnded (Solver.cs:125-146) cannot be raised from outside the Solver (OnSubstepStarted is protected) and their subscribers cannot be
    /// seen from outside either: a host that needs the events subscribes HERE instead. With a subscriber the frame runs bepuhip_solve_with_substep_events — the
    /// launch-per-batch kernels substep by substep, handlers in between with the device idle (they may use the update_* calls through Context).
    public event Solver.SubstepEvent SubstepStarted, SubstepEnded;
    public IntPtr Context => ctx;
    void IHipSubstepSink.RaiseStarted(int substep) => SubstepStarted?.Invoke(substep);
    void IHipSubstepSink.RaiseEnded(int substep) => SubstepEnded?.Invoke(substep);

    public HipTimestepper(int device = 0, bool deviceIsExclusive = false)
    {
        var cfg = new BepuHipConfig { DeviceOrdinal = device, BundleWidth = System.Numerics.Vector<float>.Count,
                                      Flags = BepuHip.FlagReserveUpdateSlots | (deviceIsExclusive ? BepuHip.FlagExclusiveDevice : 0) };
        IntPtr c; Check(BepuHip.bepuhip_create(&cfg, &c)); ctx = c;
    }
    static void Check(int status)
    {
        if (status == 0) return;
        var message = Marshal.PtrToStringUTF8(BepuHip.bepuhip_last_error());
        if (status == -1) throw new ArgumentException(message);          // same exceptions the reference throws (Simulation.cs:318-319)
        if (status == -2) throw new NotSupportedException(message);      // caller falls back to simulation.Solve
        throw new InvalidOperationException(message);
    }

    public void Timestep(Simulation simulation, float dt, IThreadDispatcher threadDispatcher = null)
    {
        simulation.Sleep(threadDispatcher);                               // DefaultTimestepper.cs:30
        simulation.PredictBoundingBoxes(dt, threadDispatcher);            // :33
        BeforeCollisionDetection?.Invoke(dt, threadDispatcher);
        simulation.CollisionDetection(dt, threadDispatcher);              // :36
        CollisionsDetected?.Invoke(dt, threadDispatcher);
        try { SolveOnDevice(simulation, dt); }                            // replaces simulation.Solve(dt, threadDispatcher) (:39); NotSupportedException is thrown before anything is enqueued
        catch (NotSupportedException) { resident = false; simulation.Solve(dt, threadDispatcher); }
        ConstraintsSolved?.Invoke(dt, threadDispatcher);
        simulation.IncrementallyOptimizeDataStructures(threadDispatcher); // :42
    }

    // Registered (pinned) host memory is read and written by asynchronous DMA. BufferPool hands out sub-allocations of large blocks (BufferPool.cs:42,83) and neighbouring
    // buffers share pages, so ranges are registered once per address and the library refuses a range that an existing registration covers only in part
    // (bepuhip_register_host_memory): such a buffer stays unregistered and its copies are staged by the runtime instead — slower, never wrong.
    void Register(void* memory, long bytes)
    {
        if (memory == null || bytes <= 0) return;
        if (registered.TryGetValue((IntPtr)memory, out var known) && (known >= bytes || known < 0)) return;
        if (known > 0) Check(BepuHip.bepuhip_unregister_host_memory(ctx, memory));   // the buffer was resized in place
        registered[(IntPtr)memory] = BepuHip.bepuhip_register_host_memory(ctx, memory, bytes) == 0 ? bytes : -1;
    }

    static long Key(int batchIndex, int typeId) => ((long)batchIndex << 32) | (uint)typeId;
    // Floats per constraint of a type's prestep data (TypeProcessor.GetBundleTypeSizes is internal; the library knows every type id: bepuhip_type_info)
    readonly Dictionary<int, int> prestepFloats = new Dictionary<int, int>();
    int PrestepFloats(int typeId)
    {
        if (prestepFloats.TryGetValue(typeId, out var known)) return known;
        int bodiesPerConstraint, floats, impulses;
        Check(BepuHip.bepuhip_type_info(typeId, &bodiesPerConstraint, &floats, &impulses));
        return prestepFloats[typeId] = floats;
    }

    int ImpulseFloats(int typeId)
    {
        int bodiesPerConstraint, floats, impulses;
        Check(BepuHip.bepuhip_type_info(typeId, &bodiesPerConstraint, &floats, &impulses));
        return impulses;
    }
    static int Lane(int index, int field, int fields, int width) => (index / width) * fields * width + field * width + index % width;   // AOSOA: bundle, field, lane (TypeProcessor.cs:139-148)

    // Last frame's copy of a type batch: handles, references, the body handles behind the references; with `shadows` also the device's prestep data and impulses.
    void Remember(Simulation simulation, int batchIndex, ref TypeBatch tb, int bodiesPerConstraint, bool shadows)
    {
        if (!mirrors.TryGetValue(Key(batchIndex, tb.TypeId), out var m)) mirrors[Key(batchIndex, tb.TypeId)] = m = new Mirror();
        m.Bodies = bodiesPerConstraint; m.Count = tb.ConstraintCount; m.Seen = true; m.Contact = NarrowPhase.IsContactConstraintType(tb.TypeId);
        m.PrestepFloats = PrestepFloats(tb.TypeId); m.ImpulseFloats = ImpulseFloats(tb.TypeId);
        if (m.Handles.Length < tb.ConstraintCount)
        {
            m.Handles = new int[Math.Max(tb.ConstraintCount, m.Handles.Length * 2)];
            m.References = new int[m.Handles.Length * bodiesPerConstraint]; m.BodyHandles = new int[m.Handles.Length * bodiesPerConstraint];
        }
        var width = System.Numerics.Vector<int>.Count;
        var references = (int*)tb.BodyReferences.Memory;
        ref var activeBodies = ref simulation.Bodies.ActiveSet;
        for (int i = 0; i < tb.ConstraintCount; ++i)
        {
            m.Handles[i] = tb.IndexToHandle[i].Value;
            for (int k = 0; k < bodiesPerConstraint; ++k)
            {
                var reference = references[Lane(i, k, bodiesPerConstraint, width)];
                m.References[i * bodiesPerConstraint + k] = reference;
                m.BodyHandles[i * bodiesPerConstraint + k] = activeBodies.IndexToHandle[reference & Bodies.BodyReferenceMask].Value;
            }
        }
        if (shadows && !m.Contact)
        {   // the device holds exactly the host's bundles right now (an upload)
            int prestepFloats = tb.BundleCount * m.PrestepFloats * width, impulseFloats = tb.BundleCount * m.ImpulseFloats * width;
            if (m.Prestep.Length < prestepFloats) m.Prestep = new float[prestepFloats * 2];
            if (m.Impulses.Length < impulseFloats) m.Impulses = new float[impulseFloats * 2];
            new Span<float>(tb.PrestepData.Memory, prestepFloats).CopyTo(m.Prestep);
            new Span<float>(tb.AccumulatedImpulses.Memory, impulseFloats).CopyTo(m.Impulses);
        }
    }

    void Upload(Simulation simulation)
    {
        var bodies = simulation.Bodies; var solver = simulation.Solver;
        ref var activeBodies = ref bodies.ActiveSet; ref var activeSet = ref solver.ActiveSet;
        // Bodies.ActiveSet.DynamicsState: 128-byte BodyDynamics (BodySet.cs:41)
        Register(activeBodies.DynamicsState.Memory, (long)activeBodies.DynamicsState.Length * sizeof(BodyDynamics));
        Check(BepuHip.bepuhip_set_bodies(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
        replanInFlight = false;   // (an upload drops a re-plan in the making: it described the constraints this upload replaces)
        Check(BepuHip.bepuhip_begin_constraints(ctx, activeSet.Batches.Count, solver.FallbackBatchThreshold));
        mirrors.Clear();
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];   // TypeBatch.cs:10-19; the library reads prestep data and impulses until end_constraints returns
                Register(tb.PrestepData.Memory, tb.PrestepData.Length); Register(tb.AccumulatedImpulses.Memory, tb.AccumulatedImpulses.Length);
                Check(BepuHip.bepuhip_set_type_batch(ctx, b, tb.TypeId, tb.ConstraintCount, (int*)tb.BodyReferences.Memory, (float*)tb.PrestepData.Memory, (float*)tb.AccumulatedImpulses.Memory));
                Remember(simulation, b, ref tb, solver.TypeProcessors[tb.TypeId].BodiesPerConstraint, true);
            }
        }
        Check(BepuHip.bepuhip_end_constraints(ctx));
        if (SpecialiseKernels) { int unitState; Check(BepuHip.bepuhip_specialise_units(ctx, 0, &unitState)); }   // asks once: later plans of the context ask by themselves
        resident = true;
    }

    // One type batch: the operations that turn what the device holds (the mirror) into what the host holds now. The reference changes a type batch by append
    // (TypeProcessor.AllocateInTypeBatch, TypeProcessor.cs:314-334), swap-with-last removal (Remove :695-717) and reference patches (UpdateForBodyMemoryMove :807); the
    // identities (constraint handle + body handles) say which constraints left and which came, but not the ORDER of the removals, which decides where swap-with-last left
    // the survivors — hence the swaps: removals (any order) + additions (in index order) + swaps (at most one per index that still disagrees) + reference patches
    // reproduce this frame's arrangement. Removals go to `removals`: DiffAndApply sends EVERY type batch's removals before any addition — a constraint that replaces
    // another one on the same bodies in the same batch (a hinge swapped for a weld) would otherwise arrive while its bodies still look taken (the island layout checks the
    // batch invariant). survivor[i] = last frame's index of the constraint now at index i, -1 for a new one. (C++ twin, compiled and tested: bepu_host.cpp DiffTypeBatch;
    // tests/test_structural_diff.py.)
    readonly Dictionary<long, int> newIndexOf = new Dictionary<long, int>(), position = new Dictionary<long, int>(), oldIndexOf = new Dictionary<long, int>();
    readonly List<long> list = new List<long>(), oldKeys = new List<long>();
    int[] survivor = Array.Empty<int>(), nowBodyHandles = Array.Empty<int>();
    bool DiffTypeBatch(Simulation simulation, int batchIndex, ref TypeBatch tb, int bodiesPerConstraint, int prestepFloats, Mirror was)
    {
        var width = System.Numerics.Vector<int>.Count;
        var references = (int*)tb.BodyReferences.Memory; var prestep = (uint*)tb.PrestepData.Memory;
        ref var activeBodies = ref simulation.Bodies.ActiveSet;
        int newCount = tb.ConstraintCount, oldCount = was.Count;
        if (survivor.Length < newCount) survivor = new int[newCount * 2];
        if (nowBodyHandles.Length < newCount * bodiesPerConstraint) nowBodyHandles = new int[newCount * bodiesPerConstraint * 2];
        for (int i = 0; i < newCount; ++i)
            for (int k = 0; k < bodiesPerConstraint; ++k)
                nowBodyHandles[i * bodiesPerConstraint + k] = activeBodies.IndexToHandle[references[Lane(i, k, bodiesPerConstraint, width)] & Bodies.BodyReferenceMask].Value;
        bool same = newCount == oldCount;
        for (int i = 0; same && i < newCount; ++i)
        {
            same = was.Handles[i] == tb.IndexToHandle[i].Value;
            for (int k = 0; same && k < bodiesPerConstraint; ++k) same = was.BodyHandles[i * bodiesPerConstraint + k] == nowBodyHandles[i * bodiesPerConstraint + k];
        }
        int before = ops.Count + removals.Count;
        if (same)
        {   // the common case: the same constraints at the same indices; only references can have changed (bodies that moved in memory)
            for (int i = 0; i < newCount; ++i)
            {
                survivor[i] = i;
                for (int k = 0; k < bodiesPerConstraint; ++k)
                    if (was.References[i * bodiesPerConstraint + k] != references[Lane(i, k, bodiesPerConstraint, width)])
                        ops.Add(new BepuHipStructuralOp { Kind = 2, BatchIndex = batchIndex, TypeId = tb.TypeId, Index = i, Slot = k, Reference = references[Lane(i, k, bodiesPerConstraint, width)] });
            }
            return ops.Count + removals.Count != before;
        }
        newIndexOf.Clear(); position.Clear(); oldIndexOf.Clear(); list.Clear(); oldKeys.Clear();
        for (int i = 0; i < newCount; ++i) { newIndexOf[tb.IndexToHandle[i].Value] = i; survivor[i] = -1; }
        for (int j = 0; j < oldCount; ++j)
        {   // keys: the handle — except for an old constraint whose handle names a DIFFERENT constraint now (other bodies): it gets a key no new constraint has
            long key = was.Handles[j];
            if (newIndexOf.TryGetValue(key, out var now))
                for (int k = 0; k < bodiesPerConstraint; ++k)
                    if (was.BodyHandles[j * bodiesPerConstraint + k] != nowBodyHandles[now * bodiesPerConstraint + k]) { key = -1 - key; break; }
            list.Add(key); oldKeys.Add(key); position[key] = j; oldIndexOf[key] = j;
        }
        for (int i = oldCount - 1; i >= 0; --i)
        {   // removals, highest old index first
            long key = oldKeys[i];
            if (key >= 0 && newIndexOf.ContainsKey(key)) continue;
            int at = position[key], last = list.Count - 1;
            removals.Add(new BepuHipStructuralOp { Kind = 1, BatchIndex = batchIndex, TypeId = tb.TypeId, Index = at });
            if (at != last) { list[at] = list[last]; position[list[at]] = at; }
            list.RemoveAt(last); position.Remove(key);
        }
        for (int i = 0; i < newCount; ++i)
        {   // additions, in the order of their final indices: references, then the prestep lane, as raw words
            long key = tb.IndexToHandle[i].Value;
            if (oldIndexOf.ContainsKey(key)) continue;
            ops.Add(new BepuHipStructuralOp { Kind = 0, BatchIndex = batchIndex, TypeId = tb.TypeId, Index = list.Count, PayloadOffset = payload.Count });
            for (int k = 0; k < bodiesPerConstraint; ++k) payload.Add((uint)references[Lane(i, k, bodiesPerConstraint, width)]);
            for (int f = 0; f < prestepFloats; ++f) payload.Add(prestep[Lane(i, f, prestepFloats, width)]);
            position[key] = list.Count; list.Add(key);
        }
        for (int i = 0; i < newCount; ++i)
        {   // the same set by now: put every index right
            long key = tb.IndexToHandle[i].Value;
            if (list[i] == key) continue;
            int j = position[key];
            ops.Add(new BepuHipStructuralOp { Kind = 3, BatchIndex = batchIndex, TypeId = tb.TypeId, Index = i, Slot = j });
            long moved = list[i]; list[i] = list[j]; list[j] = moved;
            position[list[i]] = i; position[list[j]] = j;
        }
        for (int i = 0; i < newCount; ++i)
        {   // survivors: where they were, and whether their bodies moved in memory
            if (!oldIndexOf.TryGetValue(tb.IndexToHandle[i].Value, out var old)) continue;
            survivor[i] = old;
            for (int k = 0; k < bodiesPerConstraint; ++k)
                if (was.References[old * bodiesPerConstraint + k] != references[Lane(i, k, bodiesPerConstraint, width)])
                    ops.Add(new BepuHipStructuralOp { Kind = 2, BatchIndex = batchIndex, TypeId = tb.TypeId, Index = i, Slot = k, Reference = references[Lane(i, k, bodiesPerConstraint, width)] });
        }
        return true;
    }

    // The shadow of a type batch whose arrangement changed: every constraint's copy of the device's prestep data and impulses follows it to its new index; a new
    // constraint holds, on the device, the prestep lane it was added with and zero impulses (TypeProcessor.cs:327) — the comparison of the frame then finds and sends
    // whatever the host holds beyond that (the awakener restores impulses with its bulk copies, IslandAwakener.cs:388-400).
    void CarryShadow(ref TypeBatch tb, Mirror was)
    {
        var width = System.Numerics.Vector<int>.Count;
        int pf = was.PrestepFloats, imf = was.ImpulseFloats, newCount = tb.ConstraintCount;
        var prestep = new float[Math.Max(1, tb.BundleCount) * pf * width * 2]; var impulses = new float[Math.Max(1, tb.BundleCount) * imf * width * 2];
        var hostPrestep = (float*)tb.PrestepData.Memory;
        for (int i = 0; i < newCount; ++i)
        {
            int old = survivor[i];
            for (int f = 0; f < pf; ++f) prestep[Lane(i, f, pf, width)] = old >= 0 ? was.Prestep[Lane(old, f, pf, width)] : hostPrestep[Lane(i, f, pf, width)];
            if (old >= 0) for (int f = 0; f < imf; ++f) impulses[Lane(i, f, imf, width)] = was.Impulses[Lane(old, f, imf, width)];
        }
        was.Prestep = prestep; was.Impulses = impulses;
    }

    // Everything that changed in the solver's type batches since the last frame, in ONE call. Returns false when an upload is the better answer.
    bool DiffAndApply(Simulation simulation, out bool referencesChanged)
    {
        var solver = simulation.Solver; ref var activeSet = ref solver.ActiveSet;
        ops.Clear(); removals.Clear(); payload.Clear(); referencesChanged = false;
        foreach (var m in mirrors.Values) m.Seen = false;
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];
                var processor = solver.TypeProcessors[tb.TypeId];
                if (!mirrors.TryGetValue(Key(b, tb.TypeId), out var was))
                {
                    mirrors[Key(b, tb.TypeId)] = was = new Mirror { PrestepFloats = PrestepFloats(tb.TypeId), ImpulseFloats = ImpulseFloats(tb.TypeId), Contact = NarrowPhase.IsContactConstraintType(tb.TypeId) };
                }
                bool changed = DiffTypeBatch(simulation, b, ref tb, processor.BodiesPerConstraint, was.PrestepFloats, was);
                if (changed && !was.Contact) CarryShadow(ref tb, was);
                Remember(simulation, b, ref tb, processor.BodiesPerConstraint, false);   // the device holds this arrangement once the operations below are applied
            }
        }
        var gone = new List<long>();
        foreach (var kv in mirrors)   // a type batch that no longer exists (ConstraintBatch.RemoveTypeBatchIfEmpty): its constraints went
            if (!kv.Value.Seen)
            {
                for (int i = kv.Value.Count - 1; i >= 0; --i) removals.Add(new BepuHipStructuralOp { Kind = 1, BatchIndex = (int)(kv.Key >> 32), TypeId = (int)(uint)kv.Key, Index = i });
                gone.Add(kv.Key);
            }
        foreach (var key in gone) mirrors.Remove(key);
        if (ops.Count + removals.Count > ReplayLimit) return false;   // (the caller uploads, which rebuilds every mirror)
        foreach (var op in ops) referencesChanged |= op.Kind == 2;
        if (ops.Count + removals.Count > 0)
        {
            removals.AddRange(ops);   // phase one: every removal of every type batch; phase two: additions, swaps, reference patches
            if (payload.Count == 0) payload.Add(0);
            int failed;
            fixed (BepuHipStructuralOp* table = System.Runtime.InteropServices.CollectionsMarshal.AsSpan(removals)) fixed (uint* words = System.Runtime.InteropServices.CollectionsMarshal.AsSpan(payload))
                Check(BepuHip.bepuhip_apply_structural_ops(ctx, table, removals.Count, words, payload.Count, &failed));
        }
        // Updates the plan could not absorb (a new type batch, exhausted reserves) leave the context on the launch-per-batch schedule: a fresh plan costs tens of
        // milliseconds once, the slow schedule costs every frame from then on. Not more often than every ReplanInterval frames.
        int schedule;
        Check(BepuHip.bepuhip_get_schedule(ctx, &schedule));
        if (replanInFlight)
        {   // the worker plans beside the frames; when it is done the commit costs this thread the tables' upload and the replay of the operations since, not the planning
            int committed;
            Check(BepuHip.bepuhip_replan_commit(ctx, 0, &committed));
            if (committed != 0) { replanInFlight = false; framesSinceReplan = 0; }
        }
        else if (schedule == 0 && framesSinceReplan >= ReplanInterval)
        {
            if (ReplanInBackground) { Check(BepuHip.bepuhip_replan_begin(ctx)); replanInFlight = true; }
            else { Check(BepuHip.bepuhip_replan(ctx)); framesSinceReplan = 0; }
        }
        else ++framesSinceReplan;
        return true;
    }

    // What the host rewrote IN PLACE since the last solve, and what the solve will change, as two bepuhip_transfer_rows_async tables (rowsIn before the solve, rowsOut behind it):
    //  * contact type batches whole — the narrow phase rewrites prestep data and redistributed impulses of every persisting pair (NarrowPhaseConstraintUpdate.cs:147-207);
    //  * every other type batch: the bundles whose prestep data or impulses differ from the shadow (Solver.ApplyDescription on a motor or servo, Tank.cs:100,139,
    //    SimpleCar.cs:25; impulses the awakener restored; a handle the pool handed out again for a new joint) — compared on the thread pool, one type batch per task;
    //  * back: accumulated impulses of the contacts always, of the joints with ReadBackJointImpulses.
    // The buffers are not touched again before the sync at the end of SolveOnDevice.
    sealed class Range { public int Batch, TypeId, First, Count; public bool Prestep; }
    void CollectRowTraffic(Simulation simulation)
    {
        ref var activeSet = ref simulation.Solver.ActiveSet;
        rowsIn.Clear(); rowsOut.Clear();
        var width = System.Numerics.Vector<int>.Count;
        var work = new List<(int batch, int typeBatch)>();
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];
                if (tb.ConstraintCount == 0) continue;
                Register(tb.PrestepData.Memory, tb.PrestepData.Length); Register(tb.AccumulatedImpulses.Memory, tb.AccumulatedImpulses.Length);
                bool contact = NarrowPhase.IsContactConstraintType(tb.TypeId);
                if (contact)
                {
                    rowsIn.Add(new BepuHipRowTransfer { Kind = 0, BatchIndex = b, TypeId = tb.TypeId, FirstBundle = 0, BundleCount = -1, Bundles = tb.PrestepData.Memory });
                    rowsIn.Add(new BepuHipRowTransfer { Kind = 1, BatchIndex = b, TypeId = tb.TypeId, FirstBundle = 0, BundleCount = -1, Bundles = tb.AccumulatedImpulses.Memory });
                }
                else work.Add((b, t));
                if (contact || ReadBackJointImpulses)
                    rowsOut.Add(new BepuHipRowTransfer { Kind = 3, BatchIndex = b, TypeId = tb.TypeId, FirstBundle = 0, BundleCount = -1, Bundles = tb.AccumulatedImpulses.Memory });
            }
        }
        var found = new List<Range>[work.Count];
        var solver = simulation.Solver;
        System.Threading.Tasks.Parallel.For(0, work.Count, w =>
        {
            ref var tb = ref solver.ActiveSet.Batches[work[w].batch].TypeBatches[work[w].typeBatch];
            var m = mirrors[Key(work[w].batch, tb.TypeId)];
            var ranges = found[w] = new List<Range>();
            for (int pass = 0; pass < 2; ++pass)
            {
                int floatsPerBundle = (pass == 0 ? m.PrestepFloats : m.ImpulseFloats) * width;
                var host = new Span<float>(pass == 0 ? tb.PrestepData.Memory : tb.AccumulatedImpulses.Memory, tb.BundleCount * floatsPerBundle);
                var shadow = new Span<float>(pass == 0 ? m.Prestep : m.Impulses, 0, tb.BundleCount * floatsPerBundle);
                for (int bundle = 0; bundle < tb.BundleCount; ++bundle)
                {
                    bool dirty = false;
                    if (CompareJointData) dirty = !System.Runtime.InteropServices.MemoryMarshal.AsBytes(host.Slice(bundle * floatsPerBundle, floatsPerBundle)).SequenceEqual(System.Runtime.InteropServices.MemoryMarshal.AsBytes(shadow.Slice(bundle * floatsPerBundle, floatsPerBundle)));
                    else for (int lane = 0; lane < width && bundle * width + lane < tb.ConstraintCount; ++lane) dirty |= dirtyConstraints.Contains(tb.IndexToHandle[bundle * width + lane].Value);
                    if (!dirty) continue;
                    host.Slice(bundle * floatsPerBundle, floatsPerBundle).CopyTo(shadow.Slice(bundle * floatsPerBundle, floatsPerBundle));
                    if (ranges.Count > 0 && ranges[^1].Prestep == (pass == 0) && ranges[^1].First + ranges[^1].Count == bundle) ++ranges[^1].Count;
                    else ranges.Add(new Range { Batch = work[w].batch, TypeId = tb.TypeId, First = bundle, Count = 1, Prestep = pass == 0 });
                }
            }
        });
        for (int w = 0; w < work.Count; ++w)
        {
            ref var tb = ref activeSet.Batches[work[w].batch].TypeBatches[work[w].typeBatch];
            var m = mirrors[Key(work[w].batch, tb.TypeId)];
            foreach (var r in found[w])
                rowsIn.Add(new BepuHipRowTransfer { Kind = r.Prestep ? 0 : 1, BatchIndex = r.Batch, TypeId = r.TypeId, FirstBundle = r.First, BundleCount = r.Count,
                    Bundles = r.Prestep ? (float*)tb.PrestepData.Memory + (long)r.First * m.PrestepFloats * width : (float*)tb.AccumulatedImpulses.Memory + (long)r.First * m.ImpulseFloats * width });
        }
        dirtyConstraints.Clear();
    }
    void Transfer(List<BepuHipRowTransfer> rows)
    {
        if (rows.Count == 0) return;
        fixed (BepuHipRowTransfer* table = System.Runtime.InteropServices.CollectionsMarshal.AsSpan(rows)) Check(BepuHip.bepuhip_transfer_rows_async(ctx, table, rows.Count));
    }
    /// With ReadBackJointImpulses off: the joints' accumulated impulses now (before Solver.GetDescription on a joint, before the sleeper looks at an island).
    public void FetchJointImpulses(Simulation simulation)
    {
        bool keep = ReadBackJointImpulses; ReadBackJointImpulses = true;
        rowsOut.Clear();
        ref var activeSet = ref simulation.Solver.ActiveSet;
        for (int b = 0; b < activeSet.Batches.Count; ++b)
            for (int t = 0; t < activeSet.Batches[b].TypeBatches.Count; ++t)
            {
                ref var tb = ref activeSet.Batches[b].TypeBatches[t];
                if (tb.ConstraintCount > 0 && !NarrowPhase.IsContactConstraintType(tb.TypeId))
                    rowsOut.Add(new BepuHipRowTransfer { Kind = 3, BatchIndex = b, TypeId = tb.TypeId, FirstBundle = 0, BundleCount = -1, Bundles = tb.AccumulatedImpulses.Memory });
            }
        Transfer(rowsOut); Check(BepuHip.bepuhip_sync(ctx)); AdoptImpulses(simulation);
        ReadBackJointImpulses = keep;
    }
    // The joints' impulses the device just wrote into the host's buffers are what the device holds: the shadows follow.
    void AdoptImpulses(Simulation simulation)
    {
        if (!ReadBackJointImpulses) return;
        var width = System.Numerics.Vector<int>.Count;
        ref var activeSet = ref simulation.Solver.ActiveSet;
        for (int b = 0; b < activeSet.Batches.Count; ++b)
            for (int t = 0; t < activeSet.Batches[b].TypeBatches.Count; ++t)
            {
                ref var tb = ref activeSet.Batches[b].TypeBatches[t];
                if (tb.ConstraintCount == 0 || NarrowPhase.IsContactConstraintType(tb.TypeId)) continue;
                var m = mirrors[Key(b, tb.TypeId)];
                new Span<float>(tb.AccumulatedImpulses.Memory, tb.BundleCount * m.ImpulseFloats * width).CopyTo(m.Impulses);
            }
    }

    void SolveOnDevice(Simulation simulation, float dt)
    {
        var bodies = simulation.Bodies; var solver = simulation.Solver;
        ref var activeBodies = ref bodies.ActiveSet; ref var activeSet = ref solver.ActiveSet;
        bool referencesChanged = false;
        if (!resident || !DiffAndApply(simulation, out referencesChanged)) Upload(simulation);
        else
        {
            // Bodies: the host's array is authoritative — user writes (velocities, ApplyImpulse, teleports, inertia), Bodies.Add / RemoveAt (a count that changed, a body
            // that moved into a freed index: the reference patches above). Sent whole before the solve unless the host vouches that it never writes body state.
            if (ResendBodiesEveryFrame || activeBodies.Count != residentBodyCount || referencesChanged)
            {
                Register(activeBodies.DynamicsState.Memory, (long)activeBodies.DynamicsState.Length * sizeof(BodyDynamics));
                Check(BepuHip.bepuhip_set_bodies(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
            }
        }
        CollectRowTraffic(simulation);
        Transfer(rowsIn);
        residentBodyCount = activeBodies.Count;
        // Solver.ConstrainedKinematicHandles changes with the constraints (Solver.cs:1025, :1374): indices only, sent every frame
        var kinematics = stackalloc int[Math.Max(1, solver.ConstrainedKinematicHandles.Count)];
        for (int i = 0; i < solver.ConstrainedKinematicHandles.Count; ++i)
            kinematics[i] = bodies.HandleToLocation[solver.ConstrainedKinematicHandles[i]].Index;   // PoseIntegrator.cs:467
        Check(BepuHip.bepuhip_set_constrained_kinematics(ctx, kinematics, solver.ConstrainedKinematicHandles.Count));

        var iterations = stackalloc int[solver.SubstepCount];
        for (int s = 0; s < solver.SubstepCount; ++s) iterations[s] = GetVelocityIterationCountForSubstepIndex(solver, s); // Solver_Solve.cs:743-751
        var callbacks = ((PoseIntegrator<TCallbacks>)simulation.PoseIntegrator).Callbacks;  // the three IPoseIntegratorCallbacks properties (PoseIntegrator.cs:42-94) + the model interface
        var integ = new BepuHipIntegrator { AngularIntegrationMode = (int)callbacks.AngularIntegrationMode,
            AllowSubstepsForUnconstrained = callbacks.AllowSubstepsForUnconstrainedBodies ? 1 : 0,
            IntegrateVelocityForKinematics = callbacks.IntegrateVelocityForKinematics ? 1 : 0 };
        var model = new BepuHipVelocityModel();
        object boxed = callbacks;
        if (boxed is IHipVelocityModel uniform)
        {
            integ.LinearDamping = uniform.LinearDamping; integ.AngularDamping = uniform.AngularDamping;
            integ.Gravity[0] = uniform.Gravity.X; integ.Gravity[1] = uniform.Gravity.Y; integ.Gravity[2] = uniform.Gravity.Z;
            if (sentModel != 0) { Check(BepuHip.bepuhip_set_velocity_model(ctx, &model, null, 0)); sentModel = 0; }
        }
        else if (boxed is IHipPerBodyGravityModel perBody)
        {   // the demo looks a body's value up by handle inside the callback; the device wants it by index: gathered here, once per frame (bodies move in memory)
            model.Model = 1;
            if (gravities.Length < activeBodies.Count) gravities = new float[Math.Max(activeBodies.Count, gravities.Length * 2)];
            for (int i = 0; i < activeBodies.Count; ++i) gravities[i] = perBody.BodyGravities[activeBodies.IndexToHandle[i]];
            fixed (float* table = gravities) Check(BepuHip.bepuhip_set_velocity_model(ctx, &model, table, activeBodies.Count));
            sentModel = 1;
        }
        else if (boxed is IHipRadialGravityModel radial)
        {
            model.Model = 2; model.Center[0] = radial.PlanetCenter.X; model.Center[1] = radial.PlanetCenter.Y; model.Center[2] = radial.PlanetCenter.Z; model.Gravity = radial.Gravity;
            if (sentModel != 2 || sentCenter != radial.PlanetCenter || sentGravity != radial.Gravity)
            { Check(BepuHip.bepuhip_set_velocity_model(ctx, &model, null, 0)); sentModel = 2; sentCenter = radial.PlanetCenter; sentGravity = radial.Gravity; }
        }
        else throw new NotSupportedException("this IPoseIntegratorCallbacks states no velocity model the device knows (IHipVelocityModel, IHipPerBodyGravityModel, IHipRadialGravityModel)");
        if (SubstepStarted != null || SubstepEnded != null)
        {
            var self = GCHandle.Alloc(this);
            try { Check(BepuHip.bepuhip_solve_with_substep_events(ctx, dt, solver.SubstepCount, iterations, &integ, &HipSubstepTrampoline.Started, &HipSubstepTrampoline.Ended, (void*)GCHandle.ToIntPtr(self))); }
            finally { self.Free(); }
        }
        else Check(BepuHip.bepuhip_solve_async(ctx, dt, solver.SubstepCount, iterations, &integ));

        // Back to the host, behind the solve on the same stream: poses and velocities (the MotionState half of BodyDynamics: what collision detection and the user read)
        // and the accumulated impulses (contacts: the narrow phase redistributes them over next frame's manifolds; joints: Solver.GetDescription, the sleeper, the
        // awakener) — one kernel each that writes straight into the registered host buffers. Contact depths stay on the device (the narrow phase rewrites them);
        // bepuhip_get_prestep fetches them for a host that wants to look.
        Check(BepuHip.bepuhip_get_poses_and_velocities_async(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
        Transfer(rowsOut);
        Check(BepuHip.bepuhip_sync(ctx));
        AdoptImpulses(simulation);
    }
    static int GetVelocityIterationCountForSubstepIndex(Solver solver, int substepIndex)
    {
        if (solver.VelocityIterationScheduler == null) return solver.VelocityIterationCount;
        var scheduled = solver.VelocityIterationScheduler(substepIndex);
        return scheduled < 1 ? solver.VelocityIterationCount : scheduled;
    }
    public void Dispose() { if (ctx != IntPtr.Zero) { BepuHip.bepuhip_destroy(ctx); ctx = IntPtr.Zero; } }
}
