// Reference-side binding of libbepuhip.so (include/bepuhip.h): drop this file into an application that references BepuPhysics and create the simulation with
// `new HipTimestepper<TCallbacks>()`. It uses public API only, plus — for frames whose constraint set changed — three listener calls a maintainer adds to the
// reference (IHipStructureListener below; without them the shim re-uploads whenever the topology changed). INTEGRATION.md carries the same text (a test keeps the two
// identical, and the DllImport block is generated from the header by tools/gen_csharp_imports.py); not compiled in this repository's environment (no .NET SDK here).
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
    [DllImport(Lib)] public static extern int bepuhip_set_bodies(IntPtr ctx, void* bodyDynamicsAos, int count);
    [DllImport(Lib)] public static extern int bepuhip_begin_constraints(IntPtr ctx, int batchCount, int fallbackBatchThreshold);
    [DllImport(Lib)] public static extern int bepuhip_set_type_batch(IntPtr ctx, int batchIndex, int typeId, int constraintCount, int* bodyReferencesAosoa, float* prestepAosoa, float* accumulatedImpulsesAosoa);
    [DllImport(Lib)] public static extern int bepuhip_end_constraints(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_set_constrained_kinematics(IntPtr ctx, int* bodyIndices, int count);
    [DllImport(Lib)] public static extern int bepuhip_solve(IntPtr ctx, float dt, int substepCount, int* velocityIterations, BepuHipIntegrator* integrator);
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
    [DllImport(Lib)] public static extern int bepuhip_add_constraint(IntPtr ctx, int batchIndex, int typeId, int* encodedBodyReferences, float* prestepLane, int* indexOut);
    [DllImport(Lib)] public static extern int bepuhip_remove_constraint(IntPtr ctx, int batchIndex, int typeId, int index);
    [DllImport(Lib)] public static extern int bepuhip_update_body_reference(IntPtr ctx, int batchIndex, int typeId, int index, int bodyIndexInConstraint, int encodedBodyReference);
    [DllImport(Lib)] public static extern int bepuhip_get_constraint_count(IntPtr ctx, int batchIndex, int typeId, int* countOut);
    [DllImport(Lib)] public static extern int bepuhip_get_schedule(IntPtr ctx, int* scheduleOut);
    [DllImport(Lib)] public static extern int bepuhip_replan(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_set_convex_hulls(IntPtr ctx, float* points, int* pointBegin, int hullCount);
    [DllImport(Lib)] public static extern int bepuhip_set_compounds(IntPtr ctx, BepuHipCompoundChild* children, int* childBegin, int compoundCount);
    [DllImport(Lib)] public static extern int bepuhip_set_meshes(IntPtr ctx, float* triangles, int* triangleBegin, float* scales, int meshCount);
    [DllImport(Lib)] public static extern int bepuhip_set_collidables(IntPtr ctx, BepuHipCollidable* collidables, int count);
    [DllImport(Lib)] public static extern int bepuhip_predict_bounding_boxes(IntPtr ctx, float dt, BepuHipIntegrator* integrator, BepuHipCollidable* collidables, int count, BepuHipPredictedBounds* boundsOut);
    [DllImport(Lib)] public static extern int bepuhip_get_constrained_flags(IntPtr ctx, byte* flagsOut, int count);
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

/// What the device needs to know about a velocity callback: IPoseIntegratorCallbacks.IntegrateVelocity is arbitrary code and cannot cross a C ABI, so the
/// callback struct states the model it implements. Any IPoseIntegratorCallbacks whose IntegrateVelocity is "velocity += gravity * dt, then exponential
/// damping" (Demos/DemoCallbacks.cs:100-109 is one) implements this by returning its fields; anything else leaves the interface off and the shim uses simulation.Solve.
public interface IHipVelocityModel { System.Numerics.Vector3 Gravity { get; } float LinearDamping { get; } float AngularDamping { get; } }

/// The three places at which the reference changes a type batch between solves, as calls. A maintainer adds them to the reference (a `public IHipStructureListener
/// StructureListener;` on Solver, invoked behind the mutation with the indices the mutation used):
///   TypeProcessor.AllocateInTypeBatch (Constraints/TypeProcessor.cs:314-334, reached from Solver.Add)            -> ConstraintAdded
///   TypeProcessor.Remove              (:695-717; the last constraint moves into the hole, Move :578-592)           -> ConstraintRemoved
///   TypeProcessor.UpdateForBodyMemoryMove (:807, from Solver.UpdateForBodyMemoryMove, Solver.cs:1475)             -> BodyReferenceChanged
/// A new ConstraintBatch or TypeBatch (Solver.cs:1182-1199, ConstraintBatch.cs:66-80) is not an update of a type batch: -> TopologyReset (the next frame uploads).
public interface IHipStructureListener
{
    unsafe void ConstraintAdded(int batchIndex, int typeId, int indexInTypeBatch, int bodiesPerConstraint, int* encodedBodyReferences, int prestepFloats, float* prestepLane);
    void ConstraintRemoved(int batchIndex, int typeId, int indexInTypeBatch);
    void BodyReferenceChanged(int batchIndex, int typeId, int indexInTypeBatch, int bodyIndexInConstraint, int encodedBodyReference);
    void TopologyReset();
}

public unsafe class HipTimestepper<TCallbacks> : ITimestepper, IHipStructureListener, IDisposable where TCallbacks : struct, IPoseIntegratorCallbacks, IHipVelocityModel
{
    IntPtr ctx;
    bool resident;                                    // the device holds the scene as of the last solve
    struct StructuralOp { public int Kind, Batch, TypeId, Index, Slot, Reference; public int[] References; public float[] Prestep; }
    readonly List<StructuralOp> log = new List<StructuralOp>();   // what the listener saw since the last solve, in order
    public int ReplanInterval = 30;                                // frames between two bepuhip_replan calls at most
    int residentBodyCount;                                         // Bodies.ActiveSet.Count the device's body array was last sent with
    int framesSinceReplan = 30;
    readonly Dictionary<IntPtr, long> registered = new Dictionary<IntPtr, long>();
    public int ReplayLimit = 65536;                   // a longer log is not cheaper than an upload
    public event TimestepperStageHandler BeforeCollisionDetection, CollisionsDetected, ConstraintsSolved; // ITimestepper.cs:62-74 (subset)

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

    // ---- IHipStructureListener: the narrow phase's add / remove stream, recorded with the indices the reference used ----
    public void ConstraintAdded(int batchIndex, int typeId, int indexInTypeBatch, int bodiesPerConstraint, int* encodedBodyReferences, int prestepFloats, float* prestepLane)
    {
        var op = new StructuralOp { Kind = 0, Batch = batchIndex, TypeId = typeId, Index = indexInTypeBatch, References = new int[bodiesPerConstraint], Prestep = new float[prestepFloats] };
        for (int i = 0; i < bodiesPerConstraint; ++i) op.References[i] = encodedBodyReferences[i];
        for (int i = 0; i < prestepFloats; ++i) op.Prestep[i] = prestepLane[i];
        log.Add(op);
    }
    public void ConstraintRemoved(int batchIndex, int typeId, int indexInTypeBatch) { log.Add(new StructuralOp { Kind = 1, Batch = batchIndex, TypeId = typeId, Index = indexInTypeBatch }); }
    public void BodyReferenceChanged(int batchIndex, int typeId, int indexInTypeBatch, int bodyIndexInConstraint, int encodedBodyReference)
    { log.Add(new StructuralOp { Kind = 2, Batch = batchIndex, TypeId = typeId, Index = indexInTypeBatch, Slot = bodyIndexInConstraint, Reference = encodedBodyReference }); }
    public void TopologyReset() { resident = false; log.Clear(); }

    public void Timestep(Simulation simulation, float dt, IThreadDispatcher threadDispatcher = null)
    {
        simulation.Sleep(threadDispatcher);                               // DefaultTimestepper.cs:30
        simulation.PredictBoundingBoxes(dt, threadDispatcher);            // :33
        BeforeCollisionDetection?.Invoke(dt, threadDispatcher);
        simulation.CollisionDetection(dt, threadDispatcher);              // :36
        CollisionsDetected?.Invoke(dt, threadDispatcher);
        try { SolveOnDevice(simulation, dt); }                            // replaces simulation.Solve(dt, threadDispatcher) (:39)
        catch (NotSupportedException) { resident = false; simulation.Solve(dt, threadDispatcher); }
        ConstraintsSolved?.Invoke(dt, threadDispatcher);
        simulation.IncrementallyOptimizeDataStructures(threadDispatcher); // :42
    }

    // BufferPool blocks are pinned unmanaged memory (BufferPool.cs:42,83): registered once per address, copies from / to them are asynchronous DMA from then on.
    void Register(void* memory, long bytes)
    {
        if (memory == null || bytes <= 0) return;
        if (registered.TryGetValue((IntPtr)memory, out var known) && known >= bytes) return;
        if (known > 0) Check(BepuHip.bepuhip_unregister_host_memory(ctx, memory));   // the buffer was resized in place
        Check(BepuHip.bepuhip_register_host_memory(ctx, memory, bytes));
        registered[(IntPtr)memory] = bytes;
    }

    void Upload(Simulation simulation)
    {
        var bodies = simulation.Bodies; var solver = simulation.Solver;
        ref var activeBodies = ref bodies.ActiveSet; ref var activeSet = ref solver.ActiveSet;
        // Bodies.ActiveSet.DynamicsState: 128-byte BodyDynamics (BodySet.cs:41)
        Register(activeBodies.DynamicsState.Memory, (long)activeBodies.DynamicsState.Length * sizeof(BodyDynamics));
        Check(BepuHip.bepuhip_set_bodies(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
        Check(BepuHip.bepuhip_begin_constraints(ctx, activeSet.Batches.Count, solver.FallbackBatchThreshold));
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];   // TypeBatch.cs:10-19; the library reads prestep data and impulses until end_constraints returns
                Register(tb.PrestepData.Memory, tb.PrestepData.Length); Register(tb.AccumulatedImpulses.Memory, tb.AccumulatedImpulses.Length);
                Check(BepuHip.bepuhip_set_type_batch(ctx, b, tb.TypeId, tb.ConstraintCount, (int*)tb.BodyReferences.Memory, (float*)tb.PrestepData.Memory, (float*)tb.AccumulatedImpulses.Memory));
            }
        }
        Check(BepuHip.bepuhip_end_constraints(ctx));
        resident = true;
        log.Clear();
    }

    // The constraint set changed by what the log holds: the same mutations, with the same indices, on the rows in HBM (applied by the library at the start of the solve).
    void Replay()
    {
        foreach (var op in log)
        {
            if (op.Kind == 0)
            {
                int index;
                fixed (int* references = op.References) fixed (float* prestep = op.Prestep)
                    Check(BepuHip.bepuhip_add_constraint(ctx, op.Batch, op.TypeId, references, prestep, &index));
                if (index != op.Index) throw new InvalidOperationException("the device's type batch is out of step with the host's");
            }
            else if (op.Kind == 1) Check(BepuHip.bepuhip_remove_constraint(ctx, op.Batch, op.TypeId, op.Index));
            else Check(BepuHip.bepuhip_update_body_reference(ctx, op.Batch, op.TypeId, op.Index, op.Slot, op.Reference));
        }
        log.Clear();
        // Updates the plan could not absorb (a new type batch, a body's first or last constraint, exhausted reserves) leave the context on the launch-per-batch
        // schedule: a fresh plan costs tens of milliseconds once, the slow schedule costs every frame from then on. Not more often than every ReplanInterval frames.
        int schedule;
        Check(BepuHip.bepuhip_get_schedule(ctx, &schedule));
        if (schedule == 0 && framesSinceReplan >= ReplanInterval) { Check(BepuHip.bepuhip_replan(ctx)); framesSinceReplan = 0; }
        else ++framesSinceReplan;
    }

    // What the narrow phase rewrote in place for persisting pairs since the last solve (NarrowPhaseConstraintUpdate.cs:147-207): prestep data and redistributed impulses
    // of the contact type batches. Enqueued; the buffers are not touched again before the sync at the end of SolveOnDevice.
    void RefreshContacts(Simulation simulation)
    {
        ref var activeSet = ref simulation.Solver.ActiveSet;
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];
                if (!NarrowPhase.IsContactConstraintType(tb.TypeId) || tb.ConstraintCount == 0) continue;
                Register(tb.PrestepData.Memory, tb.PrestepData.Length); Register(tb.AccumulatedImpulses.Memory, tb.AccumulatedImpulses.Length);
                Check(BepuHip.bepuhip_update_prestep_async(ctx, b, tb.TypeId, 0, tb.BundleCount, (float*)tb.PrestepData.Memory));
                Check(BepuHip.bepuhip_update_accumulated_impulses_async(ctx, b, tb.TypeId, 0, tb.BundleCount, (float*)tb.AccumulatedImpulses.Memory));
            }
        }
    }

    void SolveOnDevice(Simulation simulation, float dt)
    {
        var bodies = simulation.Bodies; var solver = simulation.Solver;
        ref var activeBodies = ref bodies.ActiveSet; ref var activeSet = ref solver.ActiveSet;
        if (solver.StructureListener != this) { solver.StructureListener = this; resident = false; }   // (the maintainer's field, see IHipStructureListener)
        if (!resident || log.Count > ReplayLimit) Upload(simulation);
        else
        {
            // Bodies.Add / RemoveAt since the last frame (a count that changed, or the reference patches of a body that moved into a freed slot): the host's array is
            // current for every body (last frame's read-back) and authoritative for the new ones — sent whole, before the constraints that reference it
            bool bodiesChanged = activeBodies.Count != residentBodyCount;
            foreach (var op in log) bodiesChanged |= op.Kind == 2;
            Replay();
            if (bodiesChanged)
            {
                Register(activeBodies.DynamicsState.Memory, (long)activeBodies.DynamicsState.Length * sizeof(BodyDynamics));
                Check(BepuHip.bepuhip_set_bodies(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
            }
            RefreshContacts(simulation);
        }
        residentBodyCount = activeBodies.Count;
        // Solver.ConstrainedKinematicHandles changes with the constraints (Solver.cs:1025, :1374): indices only, sent every frame
        var kinematics = stackalloc int[Math.Max(1, solver.ConstrainedKinematicHandles.Count)];
        for (int i = 0; i < solver.ConstrainedKinematicHandles.Count; ++i)
            kinematics[i] = bodies.HandleToLocation[solver.ConstrainedKinematicHandles[i]].Index;   // PoseIntegrator.cs:467
        Check(BepuHip.bepuhip_set_constrained_kinematics(ctx, kinematics, solver.ConstrainedKinematicHandles.Count));

        var iterations = stackalloc int[solver.SubstepCount];
        for (int s = 0; s < solver.SubstepCount; ++s) iterations[s] = GetVelocityIterationCountForSubstepIndex(solver, s); // Solver_Solve.cs:743-751
        var callbacks = ((PoseIntegrator<TCallbacks>)simulation.PoseIntegrator).Callbacks;  // the three IPoseIntegratorCallbacks properties (PoseIntegrator.cs:42-94) + IHipVelocityModel
        var integ = new BepuHipIntegrator { LinearDamping = callbacks.LinearDamping, AngularDamping = callbacks.AngularDamping,
            AngularIntegrationMode = (int)callbacks.AngularIntegrationMode,
            AllowSubstepsForUnconstrained = callbacks.AllowSubstepsForUnconstrainedBodies ? 1 : 0,
            IntegrateVelocityForKinematics = callbacks.IntegrateVelocityForKinematics ? 1 : 0 };
        integ.Gravity[0] = callbacks.Gravity.X; integ.Gravity[1] = callbacks.Gravity.Y; integ.Gravity[2] = callbacks.Gravity.Z;
        Check(BepuHip.bepuhip_solve_async(ctx, dt, solver.SubstepCount, iterations, &integ));

        // Back to the host, behind the solve on the same stream: poses and velocities (the MotionState half of BodyDynamics: what collision detection and the user read),
        // and the accumulated impulses of the contact type batches (the narrow phase redistributes them over next frame's manifolds). Joint impulses and contact
        // depths stay on the device; bepuhip_get_accumulated_impulses / bepuhip_get_prestep fetch them for a host that wants to look.
        Check(BepuHip.bepuhip_get_poses_and_velocities_async(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
        Check(BepuHip.bepuhip_sync(ctx));
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];
                if (NarrowPhase.IsContactConstraintType(tb.TypeId) && tb.ConstraintCount > 0)
                    Check(BepuHip.bepuhip_get_accumulated_impulses(ctx, b, tb.TypeId, (float*)tb.AccumulatedImpulses.Memory));
            }
        }
    }
    static int GetVelocityIterationCountForSubstepIndex(Solver solver, int substepIndex)
    {
        if (solver.VelocityIterationScheduler == null) return solver.VelocityIterationCount;
        var scheduled = solver.VelocityIterationScheduler(substepIndex);
        return scheduled < 1 ? solver.VelocityIterationCount : scheduled;
    }
    public void Dispose() { if (ctx != IntPtr.Zero) { BepuHip.bepuhip_destroy(ctx); ctx = IntPtr.Zero; } }
}
