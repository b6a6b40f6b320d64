// ReferenceDumper — runs a scene exported by this repository (oracle/pin/pin_format.py) through the REAL bepuphysics2 `Simulation.Solve` and writes
// what the library leaves in its own buffers, so that oracle/pin/compare_with_reference.py can diff the CPU oracle against the reference bit for bit.
// Part of the oracle's pinning kit (test infrastructure). NOT BUILT OR RUN IN THIS REPOSITORY'S ENVIRONMENT: there is no .NET SDK here, which is why
// the oracle is labelled "parity unpinned". Build: see oracle/pin/README.md.
//
// How the scene is rebuilt without per-type description code: constraints are added through Solver.Add with a default description of the right type
// (found by reflection over IConstraintDescription implementers' static ConstraintTypeId), in (batch, type batch, index) order — the greedy batch
// selection of Solver.Add (Solver.cs:1058-1140) then reproduces the exported batches and indices exactly — and the prestep data and accumulated
// impulses of each constraint are then written straight into the type batch's AOSOA buffers (TypeBatch.cs:13-16, BundleIndexing.cs:50-60).
// Bodies carry no shapes, so nothing but the solver touches them; only Simulation.Solve is called (Simulation.cs:278-290), never Timestep.
using System;
using System.Collections.Generic;
using System.IO;
using System.Numerics;
using System.Reflection;
using BepuPhysics;
using BepuPhysics.Collidables;
using BepuPhysics.CollisionDetection;
using BepuPhysics.Constraints;
using BepuUtilities;
using BepuUtilities.Memory;

struct NoContacts : INarrowPhaseCallbacks
{
    public void Initialize(Simulation simulation) { }
    public bool AllowContactGeneration(int workerIndex, CollidableReference a, CollidableReference b, ref float speculativeMargin) => false;
    public bool AllowContactGeneration(int workerIndex, CollidablePair pair, int childIndexA, int childIndexB) => false;
    public bool ConfigureContactManifold<TManifold>(int workerIndex, CollidablePair pair, ref TManifold manifold, out PairMaterialProperties pairMaterial)
        where TManifold : unmanaged, IContactManifold<TManifold> { pairMaterial = default; return false; }
    public bool ConfigureContactManifold(int workerIndex, CollidablePair pair, int childIndexA, int childIndexB, ref ConvexContactManifold manifold) => false;
    public void Dispose() { }
}

// Same arithmetic as the demo callbacks every bepuphysics2 sample uses (gravity and exponential damping, precomputed per dt) — the only callback shape
// that crosses this repository's C ABI (include/bepuhip.h: bepuhip_integrator).
struct GravityAndDamping : IPoseIntegratorCallbacks
{
    public Vector3 Gravity; public float LinearDamping, AngularDamping;
    public AngularIntegrationMode Mode; public bool SubstepUnconstrained, KinematicVelocities;
    Vector3Wide gravityDt; Vector<float> linearScale, angularScale;
    public readonly AngularIntegrationMode AngularIntegrationMode => Mode;
    public readonly bool AllowSubstepsForUnconstrainedBodies => SubstepUnconstrained;
    public readonly bool IntegrateVelocityForKinematics => KinematicVelocities;
    public void Initialize(Simulation simulation) { }
    public void PrepareForIntegration(float dt)
    {
        linearScale = new Vector<float>(MathF.Pow(MathHelper.Clamp(1 - LinearDamping, 0, 1), dt));
        angularScale = new Vector<float>(MathF.Pow(MathHelper.Clamp(1 - AngularDamping, 0, 1), dt));
        gravityDt = Vector3Wide.Broadcast(Gravity * dt);
    }
    public void IntegrateVelocity(Vector<int> bodyIndices, Vector3Wide position, QuaternionWide orientation, BodyInertiaWide localInertia,
        Vector<int> integrationMask, int workerIndex, Vector<float> dt, ref BodyVelocityWide velocity)
    {
        velocity.Linear = (velocity.Linear + gravityDt) * linearScale;
        velocity.Angular = velocity.Angular * angularScale;
    }
}

static unsafe class Program
{
    sealed class TypeBatchRecord { public int TypeId, Count, Bodies, PrestepFloats, ImpulseFloats; public int[] Refs; public float[] Prestep, Impulses; public ConstraintHandle[] Handles; }

    static Dictionary<int, Type> DescriptionTypesById()
    {
        var map = new Dictionary<int, Type>();
        foreach (var t in typeof(Simulation).Assembly.GetTypes())
        {
            if (!t.IsValueType || t.IsGenericTypeDefinition) continue;
            foreach (var i in t.GetInterfaces())
            {
                if (!i.IsGenericType || i.GetGenericTypeDefinition() != typeof(IConstraintDescription<>) || i.GetGenericArguments()[0] != t) continue;
                var id = t.GetProperty("ConstraintTypeId", BindingFlags.Public | BindingFlags.Static);
                if (id != null) map[(int)id.GetValue(null)] = t;
            }
        }
        return map;
    }

    static ConstraintHandle AddDefault(Solver solver, Type description, BodyHandle[] bodies)
    {
        // Solver.Add<T>(BodyHandle, in T) / Add<T>(BodyHandle, BodyHandle, in T): Solver.cs:1238-1266.
        foreach (var m in typeof(Solver).GetMethods())
        {
            if (m.Name != "Add" || !m.IsGenericMethodDefinition) continue;
            var ps = m.GetParameters();
            if (ps.Length != bodies.Length + 1 || ps[0].ParameterType != typeof(BodyHandle)) continue;
            var args = new object[ps.Length];
            for (int i = 0; i < bodies.Length; ++i) args[i] = bodies[i];
            args[bodies.Length] = Activator.CreateInstance(description);
            return (ConstraintHandle)m.MakeGenericMethod(description).Invoke(solver, args);
        }
        throw new InvalidOperationException("no Solver.Add overload for " + bodies.Length + " bodies");
    }

    static int Main(string[] args)
    {
        if (args.Length != 2) { Console.Error.WriteLine("usage: ReferenceDumper <scene.bin> <result.bin>"); return 2; }
        using var input = new BinaryReader(File.OpenRead(args[0]));
        if (new string(input.ReadChars(8)) != "BEPUPIN1") throw new InvalidDataException("not a BEPUPIN1 scene");
        int bodyCount = input.ReadInt32();
        var bodyData = new float[bodyCount * 32];
        for (int i = 0; i < bodyData.Length; ++i) bodyData[i] = input.ReadSingle();
        var callbacks = new GravityAndDamping { Gravity = new Vector3(input.ReadSingle(), input.ReadSingle(), input.ReadSingle()), LinearDamping = input.ReadSingle(), AngularDamping = input.ReadSingle() };
        callbacks.Mode = (AngularIntegrationMode)input.ReadInt32(); callbacks.SubstepUnconstrained = input.ReadInt32() != 0; callbacks.KinematicVelocities = input.ReadInt32() != 0;
        int velocityIterations = input.ReadInt32(), substeps = input.ReadInt32(), scheduledCount = input.ReadInt32();
        var scheduled = new int[scheduledCount];
        for (int i = 0; i < scheduledCount; ++i) scheduled[i] = input.ReadInt32();
        float dt = input.ReadSingle(); int frames = input.ReadInt32();
        var solveDescription = scheduledCount > 0 ? new SolveDescription(substeps, s => scheduled[s], velocityIterations) : new SolveDescription(velocityIterations, substeps);

        var pool = new BufferPool();
        var simulation = Simulation.Create(pool, new NoContacts(), callbacks, solveDescription);
        var neverSleeps = new BodyActivityDescription { SleepThreshold = -1, MinimumTimestepCountUnderThreshold = 32 };
        var handles = new BodyHandle[bodyCount];
        for (int i = 0; i < bodyCount; ++i)
        {
            var b = new ReadOnlySpan<float>(bodyData, i * 32, 32);
            var pose = new RigidPose(new Vector3(b[4], b[5], b[6]), new Quaternion(b[0], b[1], b[2], b[3]));
            var velocity = new BodyVelocity(new Vector3(b[8], b[9], b[10]), new Vector3(b[12], b[13], b[14]));
            var inertia = new BodyInertia { InverseInertiaTensor = new Symmetric3x3 { XX = b[16], YX = b[17], YY = b[18], ZX = b[19], ZY = b[20], ZZ = b[21] }, InverseMass = b[22] };
            handles[i] = simulation.Bodies.Add(new BodyDescription { Pose = pose, Velocity = velocity, LocalInertia = inertia, Activity = neverSleeps, Collidable = default });
            if (simulation.Bodies.HandleToLocation[handles[i].Value].Index != i) throw new InvalidOperationException("body index != add order");
        }

        var descriptionTypes = DescriptionTypesById();
        var records = new List<TypeBatchRecord>();
        int W = Vector<float>.Count;
        int batchCount = input.ReadInt32();
        for (int batch = 0; batch < batchCount; ++batch)
        {
            int typeBatchCount = input.ReadInt32();
            for (int t = 0; t < typeBatchCount; ++t)
            {
                var r = new TypeBatchRecord { TypeId = input.ReadInt32(), Count = input.ReadInt32(), Bodies = input.ReadInt32(), PrestepFloats = input.ReadInt32(), ImpulseFloats = input.ReadInt32() };
                r.Refs = new int[r.Count * r.Bodies]; r.Prestep = new float[r.Count * r.PrestepFloats]; r.Impulses = new float[r.Count * r.ImpulseFloats]; r.Handles = new ConstraintHandle[r.Count];
                for (int c = 0; c < r.Count; ++c)
                {
                    for (int k = 0; k < r.Bodies; ++k) r.Refs[c * r.Bodies + k] = input.ReadInt32();
                    for (int f = 0; f < r.PrestepFloats; ++f) r.Prestep[c * r.PrestepFloats + f] = input.ReadSingle();
                    for (int f = 0; f < r.ImpulseFloats; ++f) r.Impulses[c * r.ImpulseFloats + f] = input.ReadSingle();
                }
                if (!descriptionTypes.TryGetValue(r.TypeId, out var description)) throw new NotSupportedException("no constraint description with type id " + r.TypeId);
                for (int c = 0; c < r.Count; ++c)
                {
                    var bodies = new BodyHandle[r.Bodies];
                    for (int k = 0; k < r.Bodies; ++k) bodies[k] = handles[r.Refs[c * r.Bodies + k] & 0x3FFFFFFF];   // Bodies_GatherScatter.cs:107-118: low 30 bits = index
                    r.Handles[c] = AddDefault(simulation.Solver, description, bodies);
                    ref var location = ref simulation.Solver.HandleToConstraint[r.Handles[c].Value];
                    if (location.BatchIndex != batch || location.IndexInTypeBatch != c)
                        throw new InvalidOperationException($"constraint landed in batch {location.BatchIndex} index {location.IndexInTypeBatch}, scene says batch {batch} index {c}");
                    ref var typeBatch = ref simulation.Solver.ActiveSet.Batches[location.BatchIndex].GetTypeBatch(location.TypeId);
                    BundleIndexing.GetBundleIndices(location.IndexInTypeBatch, out var bundle, out var inner);
                    var prestep = (float*)typeBatch.PrestepData.Memory + (long)bundle * r.PrestepFloats * W + inner;
                    for (int f = 0; f < r.PrestepFloats; ++f) prestep[f * W] = r.Prestep[c * r.PrestepFloats + f];
                    var impulses = (float*)typeBatch.AccumulatedImpulses.Memory + (long)bundle * r.ImpulseFloats * W + inner;
                    for (int f = 0; f < r.ImpulseFloats; ++f) impulses[f * W] = r.Impulses[c * r.ImpulseFloats + f];
                }
                records.Add(r);
            }
        }

        for (int frame = 0; frame < frames; ++frame) simulation.Solve(dt, null);   // single-threaded: results do not depend on the dispatcher (DeterminismTest.cs)

        using var output = new BinaryWriter(File.Create(args[1]));
        output.Write("BEPUOUT1".ToCharArray());
        output.Write(bodyCount);
        var state = (float*)simulation.Bodies.ActiveSet.DynamicsState.Memory;      // 128-byte BodyDynamics, BodyProperties.cs:318-338
        for (int i = 0; i < bodyCount * 32; ++i) output.Write(state[i]);
        foreach (var r in records)
            for (int c = 0; c < r.Count; ++c)
            {
                ref var location = ref simulation.Solver.HandleToConstraint[r.Handles[c].Value];
                ref var typeBatch = ref simulation.Solver.ActiveSet.Batches[location.BatchIndex].GetTypeBatch(location.TypeId);
                BundleIndexing.GetBundleIndices(location.IndexInTypeBatch, out var bundle, out var inner);
                var impulses = (float*)typeBatch.AccumulatedImpulses.Memory + (long)bundle * r.ImpulseFloats * W + inner;
                for (int f = 0; f < r.ImpulseFloats; ++f) output.Write(impulses[f * W]);
                var prestep = (float*)typeBatch.PrestepData.Memory + (long)bundle * r.PrestepFloats * W + inner;
                for (int f = 0; f < r.PrestepFloats; ++f) output.Write(prestep[f * W]);
            }
        Console.WriteLine($"{bodyCount} bodies, {records.Count} type batches, {frames} frame(s) of Simulation.Solve(dt={dt}) -> {args[1]}");
        simulation.Dispose(); pool.Clear();
        return 0;
    }
}
