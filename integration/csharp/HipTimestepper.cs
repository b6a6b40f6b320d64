// Reference-side binding of libbepuhip.so: drop this file into an application that references BepuPhysics (it only uses public API) and create the
// simulation with `new HipTimestepper()`. Same text as INTEGRATION.md; not compiled in this repository's environment (no .NET SDK here).
using System;
using System.Runtime.InteropServices;
using BepuPhysics;
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

static unsafe class BepuHip
{
    const string Lib = "bepuhip"; // libbepuhip.so
    [DllImport(Lib)] public static extern IntPtr bepuhip_last_error();
    [DllImport(Lib)] public static extern int bepuhip_create(BepuHipConfig* config, IntPtr* ctx);
    [DllImport(Lib)] public static extern int bepuhip_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_set_bodies(IntPtr ctx, void* bodyDynamics, int count);
    [DllImport(Lib)] public static extern int bepuhip_begin_constraints(IntPtr ctx, int batchCount, int fallbackBatchThreshold);
    [DllImport(Lib)] public static extern int bepuhip_set_type_batch(IntPtr ctx, int batchIndex, int typeId, int constraintCount,
                                                                       void* bodyReferences, void* prestepData, void* accumulatedImpulses);
    [DllImport(Lib)] public static extern int bepuhip_end_constraints(IntPtr ctx);
    [DllImport(Lib)] public static extern int bepuhip_set_constrained_kinematics(IntPtr ctx, int* bodyIndices, int count);
    [DllImport(Lib)] public static extern int bepuhip_solve(IntPtr ctx, float dt, int substepCount, int* velocityIterations, BepuHipIntegrator* integrator);
    [DllImport(Lib)] public static extern int bepuhip_get_bodies(IntPtr ctx, void* bodyDynamics, int count);
    [DllImport(Lib)] public static extern int bepuhip_get_accumulated_impulses(IntPtr ctx, int batchIndex, int typeId, void* accumulatedImpulses);
    [DllImport(Lib)] public static extern int bepuhip_get_prestep(IntPtr ctx, int batchIndex, int typeId, void* prestepData);
    // PredictBoundingBoxes + sleep candidacy for every shape type (include/bepuhip.h: bepuhip_collidable = 64 bytes, bepuhip_predicted_bounds = 32 bytes, bepuhip_compound_child = 68 bytes);
    // hull points, compound children and mesh triangles live in device tables uploaded once, collidables name their entry in shape[0]
    [DllImport(Lib)] public static extern int bepuhip_set_convex_hulls(IntPtr ctx, float* points, int* pointBegin, int hullCount);
    [DllImport(Lib)] public static extern int bepuhip_set_compounds(IntPtr ctx, void* children, int* childBegin, int compoundCount);
    [DllImport(Lib)] public static extern int bepuhip_set_meshes(IntPtr ctx, float* triangles, int* triangleBegin, float* scales, int meshCount);
    [DllImport(Lib)] public static extern int bepuhip_set_collidables(IntPtr ctx, void* collidables, int count);
    [DllImport(Lib)] public static extern int bepuhip_predict_bounding_boxes(IntPtr ctx, float dt, BepuHipIntegrator* integrator, void* collidablesOrNull, int count, void* boundsOut);
    // ranged in-place updates / read-backs for frames whose topology did not change (INTEGRATION.md)
    [DllImport(Lib)] public static extern int bepuhip_update_bodies(IntPtr ctx, void* bodyDynamics, int first, int count);
    [DllImport(Lib)] public static extern int bepuhip_update_prestep(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, void* prestepBundles);
    [DllImport(Lib)] public static extern int bepuhip_update_accumulated_impulses(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, void* impulseBundles);
    [DllImport(Lib)] public static extern int bepuhip_get_bodies_range(IntPtr ctx, void* bodyDynamics, int first, int count);
    [DllImport(Lib)] public static extern int bepuhip_get_prestep_range(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, void* prestepBundles);
    [DllImport(Lib)] public static extern int bepuhip_get_accumulated_impulses_range(IntPtr ctx, int batchIndex, int typeId, int firstBundle, int bundleCount, void* impulseBundles);
}

public unsafe class HipTimestepper : ITimestepper, IDisposable
{
    IntPtr ctx;
    public event TimestepperStageHandler BeforeCollisionDetection, CollisionsDetected, ConstraintsSolved; // ITimestepper.cs:62-74 (subset)
    public HipTimestepper(int device = 0)
    {
        var cfg = new BepuHipConfig { DeviceOrdinal = device, BundleWidth = System.Numerics.Vector<float>.Count };
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
        try { SolveOnDevice(simulation, dt); }                            // replaces simulation.Solve(dt, threadDispatcher) (:39)
        catch (NotSupportedException) { simulation.Solve(dt, threadDispatcher); }
        ConstraintsSolved?.Invoke(dt, threadDispatcher);
        simulation.IncrementallyOptimizeDataStructures(threadDispatcher); // :42
    }

    void SolveOnDevice(Simulation simulation, float dt)
    {
        var bodies = simulation.Bodies; var solver = simulation.Solver;
        ref var activeBodies = ref bodies.ActiveSet; ref var activeSet = ref solver.ActiveSet;
        // Bodies.ActiveSet.DynamicsState: 128-byte BodyDynamics, pinned BufferPool memory (BodySet.cs:41, BufferPool.cs:42,83)
        Check(BepuHip.bepuhip_set_bodies(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
        // Type batches (TypeBatch.cs:10-19). A production shim uploads only changed ranges; v1 re-sends them when topology changed.
        Check(BepuHip.bepuhip_begin_constraints(ctx, activeSet.Batches.Count, solver.FallbackBatchThreshold));
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];
                Check(BepuHip.bepuhip_set_type_batch(ctx, b, tb.TypeId, tb.ConstraintCount, tb.BodyReferences.Memory, tb.PrestepData.Memory, tb.AccumulatedImpulses.Memory));
            }
        }
        Check(BepuHip.bepuhip_end_constraints(ctx));
        var kinematics = stackalloc int[Math.Max(1, solver.ConstrainedKinematicHandles.Count)];
        for (int i = 0; i < solver.ConstrainedKinematicHandles.Count; ++i)
            kinematics[i] = bodies.HandleToLocation[solver.ConstrainedKinematicHandles[i]].Index;   // PoseIntegrator.cs:467
        Check(BepuHip.bepuhip_set_constrained_kinematics(ctx, kinematics, solver.ConstrainedKinematicHandles.Count));

        var iterations = stackalloc int[solver.SubstepCount];
        for (int s = 0; s < solver.SubstepCount; ++s) iterations[s] = GetVelocityIterationCountForSubstepIndex(solver, s); // Solver_Solve.cs:743-751
        var callbacks = ((PoseIntegrator<DemoPoseIntegratorCallbacks>)simulation.PoseIntegrator).Callbacks;  // only this callback shape crosses the ABI
        var integ = new BepuHipIntegrator { LinearDamping = callbacks.LinearDamping, AngularDamping = callbacks.AngularDamping,
            AngularIntegrationMode = (int)callbacks.AngularIntegrationMode,
            AllowSubstepsForUnconstrained = callbacks.AllowSubstepsForUnconstrainedBodies ? 1 : 0,
            IntegrateVelocityForKinematics = callbacks.IntegrateVelocityForKinematics ? 1 : 0 };
        integ.Gravity[0] = callbacks.Gravity.X; integ.Gravity[1] = callbacks.Gravity.Y; integ.Gravity[2] = callbacks.Gravity.Z;
        Check(BepuHip.bepuhip_solve(ctx, dt, solver.SubstepCount, iterations, &integ));

        // After return the host buffers hold what Simulation.Solve would have left there.
        Check(BepuHip.bepuhip_get_bodies(ctx, activeBodies.DynamicsState.Memory, activeBodies.Count));
        for (int b = 0; b < activeSet.Batches.Count; ++b)
        {
            ref var batch = ref activeSet.Batches[b];
            for (int t = 0; t < batch.TypeBatches.Count; ++t)
            {
                ref var tb = ref batch.TypeBatches[t];
                Check(BepuHip.bepuhip_get_accumulated_impulses(ctx, b, tb.TypeId, tb.AccumulatedImpulses.Memory));   // warm start for the next frame
                if (solver.TypeProcessors[tb.TypeId].RequiresIncrementalSubstepUpdates)
                    Check(BepuHip.bepuhip_get_prestep(ctx, b, tb.TypeId, tb.PrestepData.Memory));                    // contact depths (PenetrationLimit.cs:42)
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
