// Shared device-side definitions of libbepuhip's kernels: descriptors handed to the kernels, the body record in registers, its global-memory
// gather/scatter by access filter (Bodies_GatherScatter.cs:267-753), stage ids. Included by every translation unit of the library (everything here has internal linkage).
#pragma once
#include <cstddef>

namespace {


constexpr unsigned kDynamicLimit = 1u << 30;  // Bodies_GatherScatter.cs:107-118
constexpr int kRefMask = 0x3FFFFFFF;
constexpr int kBlock = 64;  // one wavefront per workgroup: a batch rarely fills the chip, so spread waves over as many CUs as possible

struct StepParams {
    float dt, inv_dt;
    float gx, gy, gz;  // gravity * dt
    float lin_damp, ang_damp;
    int angular_mode;  // AngularIntegrationMode (PoseIntegrator.cs:20-38): 0 Nonconserving, 1 ConserveMomentum, 2 ConserveMomentumWithGyroscopicTorque
    // IPoseIntegratorCallbacks.IntegrateVelocity as data (include/bepuhip.h, bepuhip_velocity_model): which of the models velocity_callback evaluates, and their parameters
    int velocity_model;         // 0 uniform gravity + damping (the fields above), 1 per-body gravity, 2 radial gravity
    float callback_dt;          // the dt IntegrateVelocity is called with
    float cx, cy, cz, radial;   // radial: the planet's centre and gravityDt = dt * Gravity (PlanetDemo.cs:36-40)
    const float* body_gravity;  // per-body: the body's gravity by body index (PerBodyGravityDemo.cs:57-88 gathers it by handle)
};

struct DevTypeBatch {
    int type_id, count, stride, block_begin;
    int* refs;
    float* prestep;
    float* accum;
    const int* indices;  // null: lanes [0, count) of the rows; else the `count` row indices this launch processes (one dependency level of the sequential fallback batch)
};

// ---- cluster path descriptors (see cluster_kernel) ----
constexpr int kMaxPreds = 6;
constexpr int kFallbackBatchLimit = 64;
struct ClusterItem {  // <= 64 consecutive constraints of one type batch, all owned by one cluster; 64 bytes, staged in LDS
    int type_id, count, stride, start;                // start: index of the first constraint inside the (reordered) type batch
    unsigned lrefs_off, prestep_off, accum_off;       // word offsets into the constraint slab: lrefs[bodies][stride], prestep[pf][stride], accum[imf][stride]
    int batch_npred;                                  // bits 0-15 batch, 16-19 predecessor count, 20-23 cross-pass predecessor count, 24 / 25 overflow flags
    unsigned short pred[kMaxPreds];                   // cluster-relative indices of the items that last touched this item's dynamic bodies (same pass)
    unsigned short xpred[kMaxPreds];                  // for bodies this item touches FIRST in a pass: their last toucher (previous pass); may be the item itself
    int tb, shape;                                    // host bookkeeping: type batch, bodies | prestep floats << 8 | impulse floats << 16; bits 24-26 of shape: merged manifold groups (below)
};
// Merged manifold items (split plans, bepu_cluster_kernel.h run_cluster_fused): (shape >> kItemFuseShift) & 3 = the members that follow a group's leader in the item
// array, & kItemFuseMember = this item is a member (run by its leader's wave). Structural updates never touch these bits: an item's lane range is fixed at planning.
constexpr int kItemFuseShift = 24;
constexpr int kItemFuseMember = 4;
static_assert(sizeof(ClusterItem) == 64, "ClusterItem is staged in LDS as four 16-byte vectors");
static_assert(offsetof(ClusterItem, xpred) == offsetof(ClusterItem, pred) + kMaxPreds * sizeof(unsigned short), "wait_predecessors indexes pred[] and xpred[] as one array");
// LDS words behind the work items: one flag per item, the table batch -> first item (batch_count + 1 entries; the sequential fallback batch is one more batch), the claim counter.
constexpr int kClusterBatchTable = kFallbackBatchLimit + 2;
__host__ __device__ inline size_t cluster_sync_words(int max_items) { return (size_t)max_items + kClusterBatchTable + 2; }
struct ClusterDesc { int body_begin, slot_count, item_begin, item_count, batch_item_offset; };
// LDS of a cluster workgroup: [planes x ncap float4 body table][work items][sync words][SHARED: slot -> body table][one scratch row of 256 B: the destination of the
// LDS-DMA reads that only exist to pull code into L2 (touch_code_ahead)].
__host__ __device__ inline size_t cluster_lds_core_bytes(int planes, int ncap, int max_items, bool shared) {
    return (size_t)planes * ncap * 16 + (size_t)max_items * sizeof(ClusterItem) + (cluster_sync_words(max_items) + 3) / 4 * 16 + (shared ? ((size_t)ncap * 4 + 15) / 16 * 16 : 0);
}
constexpr size_t kLdsScratchRowBytes = 256;
__host__ __device__ inline size_t cluster_lds_bytes(int planes, int ncap, int max_items, bool shared = false) {
    return cluster_lds_core_bytes(planes, ncap, max_items, shared) + kLdsScratchRowBytes;
}
// Slot table entries (cluster_bodies): body index | flags; -1 = unused slot.
constexpr int kSlotKinematic = 1 << 30;   // private read-only copy of a kinematic body
constexpr int kSlotGhost = 1 << 29;       // SHARED plan: pose / inertia copy of a shared body whose home is another cluster (never integrated here)
constexpr int kSlotSharedHome = 1 << 28;  // SHARED plan: this cluster integrates the body, but its velocity lives in the global shared table during the sweeps
constexpr int kSlotBodyMask = (1 << 28) - 1;
// Split-island ("shared body") plans, DESIGN.md 3.4: an island too large for one workgroup's LDS is cut into clusters. A dynamic body referenced by a
// constraint that another cluster runs is SHARED: during the sweeps its velocity lives in `vel`, two records per body index (substep parity, see
// shared_record), each {linear, n} {angular, n} moved with agent-scope accesses. n counts the events of the step that have happened on the body — one per
// substep for the home cluster's integration, then one per constraint application in the reference's batch order (rank r of d per pass). An application
// waits for "its" n and leaves n + 1 with the velocity it wrote. `info[body]` = d.
// Event numbers of one step start at `base`: the host advances it by more than any body's events per step (substeps + 255 x passes + 2, kept even so that the
// record parity of a substep does not depend on it), so whatever the previous step left in the records reads as "not yet" and nothing has to be cleared between steps.
// One scene on several devices (round 5, bepuhip_set_device_group): every device plans the SAME clusters and runs a contiguous range of them; each keeps its own copy
// of the record table, which its clusters poll, and every record is written to ALL copies — `peers` more tables in the other devices' memory (`peer[]`), with
// system-scope stores: a hand-off that crosses devices is the same one-way push it is between two clusters of one device. peers == 0: one device, nothing changes.
constexpr int kMaxPeers = 7;
struct SharedTables { float4* vel; const unsigned* info; int poll_sleep; unsigned base; int peers; float4* const* peer; };  // poll_sleep: 64-clock naps between two polls of a record; peer: device array of `peers` table pointers
constexpr unsigned kLrefDead = 0x80008000u;  // whole-island plans: the packed local references of a free device slot (reserved at planning, or left by a removal): both halves
                                             // name the kinematic copy in slot 0 — the lane computes on whatever that holds and, like every kinematic reference, writes no body back;
                                             // what it writes into its own rows is overwritten when the slot is taken again (no extra test in the kernel)
constexpr unsigned kLrefShared = 0x4000u;  // bit 14 of a 16-bit local reference: velocity through the shared table (bit 15 = kinematic copy, bits 0-13 slot)
constexpr int kSweepPlanes = 6;       // LDS body table: one plane per 16-byte field of BodyDynamics the sweeps touch (orientation, position, linear, angular, world inertia x 2)
constexpr int kAllPlanes = 8;         // ... plus the local inertia (read once per substep by the integration phase) when the cluster leaves room for it; otherwise that stays in memory
constexpr int kClusterThreads = 1024;  // default threads per cluster workgroup
constexpr int kSplitClusterThreads = 512;  // split-island plans: the shared-body code needs the 256-VGPR budget to stay out of scratch (spills sit on every hand-off's critical path)
constexpr int kMaxClusterSubsteps = 64;  // substeps one island-kernel launch runs (SolveDescription.SubstepCount is unbounded, SolveDescription.cs:16-136; the demos use up to 8): the per-substep
                                         // iteration counts travel in the kernel arguments. Beyond it the launch-per-batch schedule takes the step. (16 until round 4.)
constexpr int kCodeTouchMaxSpans = 4;  // the most 8 KB spans of code a wave may read ahead of its PC (code touch): every cluster unit ends in that much padding (code_pad_kernel)
constexpr int kClusterTracePasses = 128;  // passes (warm starts + velocity iterations) the cluster trace buffer holds; later passes are not recorded
struct ClusterParams {
    int substeps, batch_count, integrate_velocity_for_kinematics;
    int planes;  // kSweepPlanes or kAllPlanes
    int code_touch;     // 8 KB spans of its own upcoming code a wave pulls into L2 at the start of every work item (0: off), see touch_code_ahead
    int split_integration;  // between two substeps the pose half of the integration runs beside the contact items' incremental update, the velocity half behind a barrier (cluster_kernel; 0: one piece, behind the update)
    int slot_table_in_lds;  // whole-island plans: the launch asked for LDS room for the slot -> body table behind the sync words (split plans always have it)
    unsigned jitter;    // schedule fuzzing seed (BEPUHIP_DEBUG_JITTER; 0 = off): pseudo-random naps around every item's wait and publish, see jitter_nap
    int iters[kMaxClusterSubsteps];
    int pass_stage, pass_substep;  // the one-sweep-per-launch units (kPass): kStageWarmStart or kStageSolve, and the substep the sweep belongs to
    int fallback_batch;            // index of the sequential fallback batch (its items may depend on items of their own batch), -1 if the scene has none
    // A step may be a CHAIN of launches (round 5): more than kMaxClusterSubsteps substeps, or one substep per launch when the host raises Solver.SubstepStarted / SubstepEnded
    // between them. A launch runs substeps [substep_base, substep_base + substeps) of the step: "substep 0" rules (velocity only, no incremental contact update, the
    // conserving modes' backwards half-step and re-transformations) apply to the STEP's first substep only, and only the step's last launch integrates the trailing pose.
    int substep_base, final_launch;
    StepParams sp;
};


// The per-body work that follows the substep loop (PoseIntegrator.cs:451-535 for constrained kinematics, :537-693 for every other body), handed to
// cluster_kernel so that a whole step is ONE launch: workgroups beyond the clusters integrate the bodies no cluster owns.
struct TailParams {
    const unsigned* flags; const int* kinlist; unsigned* staged;
    int body_count, kin_count, cluster_count, body_blocks;
    int block_offset;  // added to blockIdx.x: the tail workgroups of a split plan are a launch of their own (the clusters' launch is cooperative: exactly the clusters)
    float dt, substep_dt;
    int substep_count, allow_substeps_for_unconstrained, integrate_velocity_for_kinematics;
    int substep_base, launch_substeps, final_launch;  // a chained step (ClusterParams): the constrained kinematic bodies advance by this launch's substeps; IntegrateAfterSubstepping runs in the last launch only
    StepParams final_sp;  // PrepareForIntegration(dt or dt / substeps) of the final pass (PoseIntegrator.cs:707-726), not the substep's
};

struct DBody {
    V3 pos; Q ori; BodyVel vel; Inertia inertia;
    float linw, angw;  // padding lanes of the velocity float4s, preserved on store
};

template <int ACCESS>
__device__ __forceinline__ void load_body(const float4* __restrict__ bodies, int ref, DBody& b) {
    const float4* base = bodies + (size_t)(ref & kRefMask) * 8;
    if (ACCESS & kOri) { float4 q = base[0]; b.ori = {q.x, q.y, q.z, q.w}; } else b.ori = {0, 0, 0, 0};
    if (ACCESS & kPos) { float4 p = base[1]; b.pos = {p.x, p.y, p.z}; } else b.pos = {0, 0, 0};
    if (ACCESS & kLin) { float4 l = base[2]; b.vel.lin = {l.x, l.y, l.z}; b.linw = l.w; } else { b.vel.lin = {0, 0, 0}; b.linw = 0; }
    if (ACCESS & kAng) { float4 a = base[3]; b.vel.ang = {a.x, a.y, a.z}; b.angw = a.w; } else { b.vel.ang = {0, 0, 0}; b.angw = 0; }
    if (ACCESS & kInertia) {
        float4 i0 = base[6], i1 = base[7];
        b.inertia.t = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y};
        b.inertia.invMass = i1.z;
    } else { b.inertia.t = {0, 0, 0, 0, 0, 0}; b.inertia.invMass = 0; }
}
// ScatterVelocities: kinematic / empty references are never written (Bodies_GatherScatter.cs:675-682,717-724).
template <int ACCESS>
__device__ __forceinline__ void store_velocity(float4* bodies, int ref, const DBody& b) {
    if ((unsigned)ref >= kDynamicLimit) return;
    float4* base = bodies + (size_t)ref * 8;
    if (ACCESS & kLin) base[2] = make_float4(b.vel.lin.x, b.vel.lin.y, b.vel.lin.z, b.linw);
    if (ACCESS & kAng) base[3] = make_float4(b.vel.ang.x, b.vel.ang.y, b.vel.ang.z, b.angw);
}

enum { kStageWarmStart = 0, kStageSolve = 1, kStageIncremental = 2 };
// The constraint functions call their gate once, right before the first use of the bodies' velocities (everything before it depends on
// poses, inertias and prestep data only). The launch-per-batch kernels have the velocities in registers already.
struct NoGate {
    static constexpr bool kPin = false;
    __device__ __forceinline__ void operator()(BodyVel&, BodyVel&) const {}
    __device__ __forceinline__ void many(BodyVel*) const {}
};

}  // namespace
