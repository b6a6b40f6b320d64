// Host-side state of a bepuhip context: kernel variant table, error plumbing, type table, per-type-batch bookkeeping, the context itself.
#pragma once
#include <set>
#include <unordered_set>

// ------------------------------------------------------------------------------------------------
// cluster_kernel instantiations live in their own translation units (bepu_cluster_{hot,wide}_{1024,768,512}.hip): type set x register budget, each
// with a traced twin. A kernel compiled for N threads per workgroup gets 65536 / N VGPRs per lane (128 / 168 / 256 after the allocation granule).
const void* bepu_cluster_kernel_hot_1024(bool trace);
const void* bepu_cluster_kernel_hot_512(bool trace);
const void* bepu_cluster_kernel_wide_1024(bool trace);
const void* bepu_cluster_kernel_wide_512(bool trace);
const void* bepu_cluster_kernel_hot_1024n(bool trace);   // non-temporal row loads (whole-island plans, 1024 threads)
const void* bepu_cluster_kernel_wide_1024n(bool trace);
const void* bepu_cluster_kernel_hot_512sn(bool trace);   // non-temporal row loads, split-island plans (512 threads)
const void* bepu_cluster_kernel_wide_512sn(bool trace);
const void* bepu_cluster_kernel_hot_512s(bool trace);    // split-island plans (shared bodies); the 1024-thread split units (372 / 3,437 spilled VGPRs, 1.3 MB: experiments only) are gone since round 6
const void* bepu_cluster_kernel_wide_512s(bool trace);
const void* bepu_cluster_kernel_hot_768s(bool trace);    // split-island plans at 768 threads (twelve waves, 168 VGPRs): plans with many work items per cluster (round 5)
const void* bepu_cluster_kernel_hot_1024c(bool trace);   // the momentum-conserving angular modes compiled in: whole-island plans at 1024 threads ...
const void* bepu_cluster_kernel_wide_1024c(bool trace);
const void* bepu_cluster_kernel_hot_512sc(bool trace);   // ... split-island plans at 512
const void* bepu_cluster_kernel_wide_512sc(bool trace);
const void* bepu_cluster_kernel_hot_1024p(bool trace);   // one sweep per launch (exchanged solves): whole-island plans at 1024 threads ...
const void* bepu_cluster_kernel_wide_1024p(bool trace);
const void* bepu_cluster_kernel_hot_512sp(bool trace);   // ... split-island plans at 512
const void* bepu_cluster_kernel_wide_512sp(bool trace);
const void* bepu_cluster_kernel_contacts_512s(bool trace);  // round 6: the contacts family (type ids 0-7 only): split plans at eight waves ...
const void* bepu_cluster_kernel_contacts_768s(bool trace);  // ... at twelve
const void* bepu_cluster_kernel_contacts_1024(bool trace);  // ... whole-island plans
constexpr int kClusterThreadChoices[3] = {1024, 768, 512};
// The smallest register budget that still fits `threads`; BEPUHIP_CLUSTER_VARIANT (experiments) asks for a tighter one, e.g. the 1024-thread build's 128 VGPRs for
// 512-thread workgroups, so that two of them are resident per CU.
static int cluster_variant_threads(int threads) {
    static const int forced = [] { const char* v = getenv("BEPUHIP_CLUSTER_VARIANT"); return v && *v ? atoi(v) : 0; }();
    const int fit = threads > 512 ? 1024 : 512;  // (round 5: the 768-thread units of round 2 are gone but one — bepu_cluster_hot_768s, the split plans' twelve-wave unit, picked in cluster_kernel_variant)
    return (forced == 1024 || forced == 512) && forced >= fit ? forced : fit;
}
// The conserving units exist for the default workgroup sizes only; other sizes (BEPUHIP_CLUSTER_THREADS / BEPUHIP_SPLIT_THREADS) keep such solves on the launch-per-batch schedule.
static bool conserving_variant_exists(int threads, bool shared) { return cluster_variant_threads(threads) == (shared ? 512 : 1024); }
static const void* cluster_pass_kernel(bool wide, bool shared) {  // (for the default workgroup sizes, like the conserving units)
    return shared ? (wide ? bepu_cluster_kernel_wide_512sp(false) : bepu_cluster_kernel_hot_512sp(false)) : (wide ? bepu_cluster_kernel_wide_1024p(false) : bepu_cluster_kernel_hot_1024p(false));
}
// Type-set families (round 6): kFamilyContacts = nothing but convex contact manifolds (type ids 0-7), kFamilyHot = SURVEY 8(a)'s sixteen, kFamilyWide = all 44. A family
// that has no unit for a (threads, policy, mode) combination runs the next larger family's: same bits by construction (a unit differs from its superset only in the
// switch cases it leaves out), asserted by tests/test_gpu_type_families.py.
enum { kFamilyContacts = 0, kFamilyHot = 1, kFamilyWide = 2, kFamilySpecial = 3 };
static bool contacts_family_enabled() { const char* v = getenv("BEPUHIP_CONTACTS_FAMILY"); return v == nullptr || atoi(v) != 0; }  // (developer switch, read per launch: tools/ab_scene.py compares the families on one box)
static const void* contacts_kernel_variant(int threads, bool trace, bool shared) {
    if (shared && threads > 512 && threads <= 768) return bepu_cluster_kernel_contacts_768s(trace);
    if (shared && cluster_variant_threads(threads) == 512) return bepu_cluster_kernel_contacts_512s(trace);
    if (!shared && cluster_variant_threads(threads) == 1024) return bepu_cluster_kernel_contacts_1024(trace);
    return nullptr;
}
static const void* cluster_kernel_variant(int threads, bool trace, bool wide, bool shared = false, bool nt = false, bool conserving = false, bool contacts_only = false) {
    if (contacts_only && !wide && !nt && !conserving && contacts_family_enabled()) {
        if (const void* fn = contacts_kernel_variant(threads, trace, shared)) return fn;
    }
    if (conserving) return shared ? (wide ? bepu_cluster_kernel_wide_512sc(trace) : bepu_cluster_kernel_hot_512sc(trace)) : (wide ? bepu_cluster_kernel_wide_1024c(trace) : bepu_cluster_kernel_hot_1024c(trace));
    // (the non-temporal units carry no traced twin: a traced solve runs the plain-row unit of the same size — same results, the timeline of the default policy)
    if (nt && !trace && !shared && cluster_variant_threads(threads) == 1024) return wide ? bepu_cluster_kernel_wide_1024n(false) : bepu_cluster_kernel_hot_1024n(false);
    if (nt && !trace && shared && cluster_variant_threads(threads) == 512) return wide ? bepu_cluster_kernel_wide_512sn(false) : bepu_cluster_kernel_hot_512sn(false);
    if (shared && !wide && !nt && threads > 512 && threads <= 768) return bepu_cluster_kernel_hot_768s(trace);
    if (shared) return wide ? bepu_cluster_kernel_wide_512s(trace) : bepu_cluster_kernel_hot_512s(trace);  // (cluster_threads keeps a split plan's workgroups within what these are compiled for)
    switch (cluster_variant_threads(threads)) {
        case 1024: return wide ? bepu_cluster_kernel_wide_1024(trace) : bepu_cluster_kernel_hot_1024(trace);
        default: return wide ? bepu_cluster_kernel_wide_512(trace) : bepu_cluster_kernel_hot_512(trace);
    }
}

static thread_local std::string g_last_error;
static int32_t fail(int32_t code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(BEPUHIP_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

using C1O = Contact<1, false>; using C2O = Contact<2, false>; using C3O = Contact<3, false>; using C4O = Contact<4, false>;
using C1T = Contact<1, true>; using C2T = Contact<2, true>; using C3T = Contact<3, true>; using C4T = Contact<4, true>;
struct TypeInfoH { int bodies, prestep, impulse; bool incremental; };
static bool type_info(int id, TypeInfoH& t) {
#define TI(T) { t = {T::bodies, T::prestepFloats, T::impulseFloats, T::incremental}; return true; }
    switch (id) {
        case kContact1OneBody: TI(C1O) case kContact2OneBody: TI(C2O)
        case kContact3OneBody: TI(C3O) case kContact4OneBody: TI(C4O)
        case kContact1: TI(C1T) case kContact2: TI(C2T)
        case kContact3: TI(C3T) case kContact4: TI(C4T)
#define X(ID, T) case ID: TI(T)
        BD_JOINT_TYPES(X)
        BD_NONCONVEX_CONTACT_TYPES(X)
        BD_MANY_BODY_TYPES(X)
#undef X
    }
#undef TI
    return false;
}

static bool is_widened_type(int id) {
    switch (id) {
#define X(ID, T) case ID: return true;
        BD_WIDENED_JOINT_TYPES(X)
        BD_NONCONVEX_CONTACT_TYPES(X)
        BD_MANY_BODY_TYPES(X)
#undef X
    }
    return false;
}

struct HostTypeBatch {
    int batch, type_id, count, stride;
    TypeInfoH info;
    size_t refs_off, prestep_off, accum_off, lrefs_off;  // offsets (in 4-byte words) into the constraint slab
    std::vector<uint8_t> occupied;  // type batches of the sequential fallback batch: lane i of the caller's layout holds a constraint (empty lanes carry -1 references, TypeProcessor.cs:451-571)
    std::vector<int32_t> perm;      // cluster path: device index -> host index inside the type batch (empty = identity)
    std::vector<int32_t> inv;       // host index -> device index (lazily built)
    // (`inv` has `count` entries, never `perm.size()`: on an island layout with reserved slots perm is longer than the type batch, and a type batch that was EMPTY when a plan
    // was built got a table of `slots` zeros here — the next addition appended its slot behind them and index 0 pointed at device slot 0, a dead one: the constraint was
    // solved where it was, read back and updated from somewhere else. Found by tools/fuzz_structural.py 6802, scene 490, in round 6's last hour; tests/test_gpu_schedule_fuzz.py replays it.)
    int perm_inverse(int host_index) {
        if (inv.empty() && count > 0) { inv.assign((size_t)count, 0); for (size_t d = 0; d < perm.size(); ++d) if (perm[d] >= 0 && perm[d] < count) inv[perm[d]] = (int32_t)d; }
        return (size_t)host_index < inv.size() ? inv[host_index] : 0;
    }
    int32_t* d_device_index = nullptr;  // device copy of `inv` for the upload, ranged update and read-back kernels: a slice of the context's index pool when built at
    bool index_pooled = false;          // end_constraints (never freed on its own), else allocated on first use
    const float* raw_prestep = nullptr; // the caller's AOSOA bundles as they were copied to the device by set_type_batch (ctx.raw_chunks): end_constraints transposes
    const float* raw_accum = nullptr;   // them into the rows ON the device; the host never touches the values
    // bepuhip_replan: the values are rows of the previous slab (and of its snapshot) in the caller's order instead of bundles
    const uint32_t* old_prestep[2] = {nullptr, nullptr};
    const uint32_t* old_accum[2] = {nullptr, nullptr};
    int old_stride = 0;
    // Island layout, whole-island plans (bepu_soft_updates.h): the rows hold `slots` device slots — every cluster's constraints of this type batch in one segment
    // [seg_begin[cluster], seg_begin[cluster + 1]), live ones and free ones (perm[d] == -1: reserved at planning, or left by a removal; their local references are
    // kLrefDead). `dev_refs` mirrors the encoded body references per device slot so that a removal knows whose constraint counts it lowers.
    int slots = 0;
    std::vector<int32_t> seg_begin;
    std::vector<int32_t> dev_refs;
    std::vector<int32_t> plan_lrefs;   // split plans: 32-bit local references (slot | kLrefShared | kinematic << 30) and rank words per device slot, the inputs of
    std::vector<uint32_t> plan_ranks;  // the predecessor rule (bepu_cluster_plan.h), kept for the structural updates
    int device_extent() const { return slots > 0 ? slots : count; }
    // structural updates since the last flush (bepu_soft_updates.h): device slot -> its record in ctx.soft_records (-1: untouched; sized on first use), and the
    // caller's indices whose device slot changed (the device copy of `inv` is patched from `inv` itself at the flush)
    std::vector<int32_t> soft_record_of;
    std::vector<int32_t> soft_dirty_indices;
    std::vector<int32_t> lrefs_soa; // cluster path: local (LDS) body indices
    std::vector<int32_t> refs_soa;
    std::vector<float> prestep_soa, accum_soa;  // only with ctx.host_values (the offline plan harness): host staging until the plan has permuted them
};

constexpr size_t kMaxCachedGraphs = 8;
struct GraphKey {
    std::vector<int> iterations;
    float dt;
    bepuhip_integrator integ;
    bool operator<(const GraphKey& o) const {
        if (iterations != o.iterations) return iterations < o.iterations;
        if (dt != o.dt) return dt < o.dt;
        return memcmp(&integ, &o.integ, sizeof(integ)) < 0;
    }
};

struct ReplanJob;  // bepuhip.hip
struct SpecialUnit;  // bepu_unit_cache.h
struct bepuhip_ctx {
    int device = 0, W = 8, flags = 0;
    ReplanJob* replan_job = nullptr;  // bepuhip_replan_begin ... _commit: the planning thread, its shadow of the type batches, the log of structural calls since
    bool replan_replaying = false;    // ... the commit is feeding that log back through the public calls: they do not log themselves again
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    float4* d_bodies = nullptr;
    float4* d_bodies0 = nullptr;  // pristine snapshot for reset_state
    unsigned* d_flags = nullptr;
    int body_count = 0, body_capacity = 0;
    int* d_kin = nullptr;
    int kin_count = 0;
    std::vector<int32_t> kin_indices;
    // upload staging, kept across uploads
    bool host_values = false;            // offline plan harness only (no device): set_type_batch converts the values on the host as well
    struct RawChunk { char* ptr; size_t capacity, used; };
    std::vector<RawChunk> raw_chunks;    // device memory the caller's prestep / impulse bundles are copied into as they are (set_type_batch)
    int32_t* d_index_pool = nullptr;     // host index -> device slot tables of every permuted type batch (HostTypeBatch::d_device_index points into it)
    void* h_staging = nullptr;           // pinned host buffer for what the host does build (references, local references, index tables)
    size_t h_staging_bytes = 0;
    std::vector<void*> registered_host;  // bepuhip_register_host_memory
    std::vector<size_t> registered_bytes;  // ... and how far each registration reaches (ranges inside one are read and written by kernels directly: mapped_pointer)
    // constraints
    bool building = false, built = false;
    int batch_count = 0;                 // the caller's batches, a sequential fallback batch included
    int fallback_threshold = 64;         // SolveDescription.FallbackBatchThreshold: batch index == threshold is the sequential fallback batch (Solver.cs:1878-1884)
    bool has_fallback = false;
    int launch_count = 0;                // launches per pass: the synchronized batches, then one per dependency level of the fallback batch
    int* d_fallback_indices = nullptr;   // row indices of every level's launches, concatenated
    // structural updates queued since the last flush (bepuhip_add_constraint / remove_constraint / update_body_reference)
    struct PendingOp { int tb; StructuralOp op; };
    std::vector<PendingOp> pending_ops;
    std::vector<uint32_t> pending_payload;
    bool structure_dirty = false;        // counts changed: descriptors, constrained flags and cached graphs are stale until flush_structural
    bool requirk_stale = false;          // the conserving angular modes' substep-0 lists describe the uploaded topology only
    std::vector<HostTypeBatch> tbs;
    std::unordered_map<uint64_t, int> tb_lookup;  // (batch, type id) -> ordinal in `tbs`: a cache, every hit is checked against the type batch it names
    std::vector<int> batch_begin;        // descriptor index of each launch's first type batch (size launch_count+1)
    std::vector<int> batch_blocks;       // grid size per launch
    uint32_t* d_slab = nullptr;          // all refs/prestep/accum
    uint32_t* d_slab0 = nullptr;         // pristine snapshot
    float* d_stage = nullptr;            // staging for ranged updates / read-backs (caller's AOSOA bundles)
    size_t stage_floats = 0;
    char* h_desc_ring = nullptr;         // pinned: descriptor tables of bepuhip_transfer_rows_async calls that have not been synchronised yet (bump-allocated, reset by bepuhip_sync)
    size_t desc_ring_bytes = 0, desc_ring_used = 0;
    size_t slab_words = 0;
    size_t slab_alloc_words = 0;           // what d_slab / d_slab0 were allocated with (>= slab_words: a re-upload takes the previous upload's pair back when it fits)
    uint32_t* spare_slab[2] = {nullptr, nullptr};  // the pair free_constraints set aside for the next upload (two 87 MB hipMalloc / hipFree pairs per upload of the bench scene otherwise)
    size_t spare_slab_words = 0;
    DevTypeBatch* d_tbs = nullptr;       // per (batch) descriptors, solve/warm-start grids
    DevTypeBatch* d_inc_tbs = nullptr;   // incremental-update grid (contacts of all batches)
    int inc_tb_count = 0, inc_blocks = 0;
    int64_t total_constraints = 0;
    int referenced_bodies = 0;           // 1 + the largest body index any constraint references
    // cluster path
    bool clusters_enabled = false;
    // Launch policy of the island schedule's default workgroup sizes (plain or non-temporal accesses to the constraint rows, code touched ahead or not; settle_row_policy).
    // Which one is faster depends on the box: the first solves cycle through the candidates — their results are bit-identical — each timed with its own event pair.
    // Structural updates that keep the island schedule (bepu_soft_updates.h)
    bool soft_ok = false;                        // whole-island plan with its host mirrors in place
    std::vector<int32_t> body_cluster, body_lref, body_degree;  // per dynamic body: its cluster (-1: none), its rotated LDS slot, its constraint count
    std::vector<uint64_t> body_batches;                          // per dynamic body: bit b = a constraint of synchronized batch b references it (the batch invariant, Solver.cs:1046-1051)
    std::vector<std::unordered_map<int32_t, int32_t>> cluster_kin;  // per cluster: kinematic body -> rotated LDS slot of its private copy
    std::vector<ClusterItem> items_host;         // the plan's work items (the predecessor lists of a cluster that received a constraint are replaced by batch-level waits)
    std::vector<ClusterDesc> clusters_host;
    std::vector<uint8_t> cluster_degraded;
    // Final state of every device slot touched since the last flush: flat tables (round 5; hash maps keyed by (type batch, slot) with a vector per entry until then:
    // two node allocations per operation). A slot touched again overwrites its record; the payload words (whole-island plans: references, packed local references,
    // prestep lane; split plans: the prestep lane) live in one pool.
    struct SoftSlotRecord { int32_t tb, slot; uint32_t live, payload_at, payload_words; };
    std::vector<SoftSlotRecord> soft_records;
    std::vector<uint32_t> soft_payload;
    bool soft_index_dirty = false;               // some type batch has entries in soft_dirty_indices
    // The flush's transfer: one pinned staging buffer, one device buffer, both kept (round 5; a hipMalloc / hipFree pair and three pageable copies per flush until then)
    char* h_flush = nullptr; char* d_flush = nullptr;
    size_t h_flush_bytes = 0, d_flush_bytes = 0;
    std::vector<int32_t> soft_orphans;           // bodies whose constraint count reached zero since the last flush
    bool soft_items_dirty = false;
    int64_t soft_adds = 0, soft_removes = 0;     // since the upload (diagnostics)
    // ... and on a split-island plan (bepu_soft_updates.h, second half): which bodies are shared, every cluster's ghost / kinematic copies and slot table, the applications
    // of every body (built when the first update arrives), and what the flush has to re-rank and patch
    bool soft_split = false;
    std::vector<uint8_t> split_shared;
    std::vector<std::unordered_map<int32_t, int32_t>> cluster_extra;  // per cluster: body | kSlotGhost / kSlotKinematic -> rotated LDS slot
    std::vector<int32_t> cluster_natural;                             // per cluster: natural slot indices handed out so far
    std::vector<std::vector<int32_t>> cluster_free_slots;             // per cluster: LDS slots of ghost / kinematic copies nothing references any more
    std::vector<std::unordered_map<int32_t, int32_t>> cluster_extra_uses;  // per cluster: references to each ghost / kinematic copy (body | kind -> count)
    std::vector<int32_t> cluster_bodies_host;                         // mirror of d_cluster_bodies
    std::vector<size_t> split_visit;                                  // order in which the type batches become a cluster's items
    struct SplitApp { int32_t tb, slot, k; };
    std::vector<std::vector<SplitApp>> body_apps;
    std::vector<uint8_t> split_rerank_flag;      // shared bodies whose applications changed since the last flush: re-ranked there (flag by body, and the list of flagged bodies)
    std::vector<int32_t> split_rerank_list;
    // A word of the split plan's device tables that differs from its host mirror until the next flush writes it (the VALUE is read from the mirror then: a slot may be
    // freed and taken again between the change and the flush). table: 0 constraint slab (both copies; tb / slot / row name the word), 1 shared_info, 2 cluster_bodies.
    struct WordPatch { int table; size_t index; int32_t tb, slot, row; };
    std::vector<WordPatch> split_patches;
    double soft_call_ms = 0.0;  // BEPUHIP_PLAN_STATS >= 2: time inside the structural calls since the last flush
    long soft_calls = 0;
    bool graphs_cleared_by_structure = false;  // set by flush_structural, consumed by the next solve (which then launches eagerly instead of capturing)
    int row_policy = -1;              // -1: still measuring; 0 plain rows; 1 non-temporal rows; 2 plain rows + one span of code touched per item (BEPUHIP_ROW_POLICY pins it)
    int policy_samples = 0;           // solves launched while measuring
    int policy_threads = 0;           // workgroup size the samples ran with
    hipEvent_t policy_events[16][2] = {};
    // One scene on several devices (bepuhip_set_device_group): this context plans `group_world` x the clusters one device holds and runs the contiguous range
    // [cluster_first, cluster_first + cluster_local) of them; its record table is one of `group_world` copies, the addresses of the other copies are in d_peer_table.
    int group_world = 1, group_rank = 0;
    int cluster_first = 0, cluster_local = 0;
    std::vector<int32_t> group_body_cluster;  // body -> cluster of the plan (group_world > 1: which device owns a body at the end of a step)
    std::vector<void*> peer_records;          // the record tables of the other devices, by peer ordinal (rank order, this rank left out)
    std::vector<void*> peer_opened;           // ... those opened from an IPC handle by this context, by peer ordinal (closed with it, or when the peer's table is imported again)
    std::vector<uint8_t> peer_on_this_device; // by peer ordinal: the table lies in THIS device's memory (members of a group sharing one device: a test configuration, see group_queue_check)
    float4** d_peer_table = nullptr;
    float4* group_records = nullptr;          // the record table of a group member: allocated once (for group_records_bodies bodies), reused by every later plan that fits
    size_t group_records_bodies = 0;
    bool group_records_on_host = false;       // BEPUHIP_GROUP_FAKE_REMOTE=1: the table lives in fine-grained host memory (see free_group_records)
    uint32_t* d_owned_dense = nullptr;        // bepuhip_sync_owned_bodies: 16 words per body
    uint8_t* d_owned_mask = nullptr;
    int owned_mask_bodies = 0;
    bool clusters_shared = false;    // split-island plan: bodies shared between clusters go through the tables below
    bool split_twelve_waves = false; // ... whose clusters hand out enough work items per pass to be short of wave time: 768 threads per cluster instead of 512 (enqueue_island_launch)
    float4* d_shared_vel = nullptr;   // per body two records (substep parity) of {linear, event number} {angular, event number}
    unsigned* d_shared_info = nullptr;
    size_t shared_bodies = 0;         // table length (bodies)
    unsigned shared_epoch = 0;        // event numbers of the next step start here (SharedTables.base): the records are cleared once, not per step
    unsigned long long type_mask = 0;  // bit t: a type batch of constraint type id t has been uploaded or added (never cleared by removals: a superset of the types present)
    SpecialUnit* special_unit = nullptr;  // bepuhip_specialise_units: the cluster_kernel unit compiled for exactly `special_mask` / `special_budget` / `special_shared` (bepu_unit_cache.h)
    unsigned long long special_mask = 0;
    int special_budget = 0;
    bool special_shared = false;
    bool specialise_auto = false;     // BEPUHIP_SPECIALISE=1 (or bepuhip_specialise_units once): every plan asks for its unit; launches use it once it is loaded
    bool has_widened_types = false;  // any type outside SURVEY 8(a)'s sixteen: selects the wider cluster_kernel variant
    int last_kernel_family = -1;     // bepuhip_get_kernel_family: the family of the last island launch
    bool has_joint_types = false;    // any type that is not a convex contact manifold (type id > 7): without one, the contacts family's units run the scene (round 6)
    int cluster_count = 0, cluster_max_slots = 0, cluster_max_items = 0, cluster_total_items = 0, cluster_planes = 8;
    ClusterDesc first_cluster = {0, 0, 0, 0, 0};
    int* d_requirk = nullptr;            // conserving angular modes: per batch, the bodies momentum_requirk_kernel transforms in substep 0
    std::vector<int> requirk_begin;     // batch -> offset into d_requirk (batch_count + 1 entries)
    int* d_boundary = nullptr;          // boundary body indices (see bepuhip_set_boundary_bodies)
    float4* d_boundary_snapshot = nullptr;
    float* d_boundary_buf = nullptr;     // count * 6 floats staging for host-pointer exchanges
    int boundary_count = 0;
    int exchange_mode = 0;               // BEPUHIP_EXCHANGE_*
    int* d_boundary_rows = nullptr;      // on-stream exchange: row of every boundary body in the dense buffer (same row on every rank)
    float* d_boundary_dense = nullptr;   // dense_rows x 6 words, all-reduced in place
    float* d_boundary_holders = nullptr; // per dense row: ranks holding the body (mass-split shares), null = sums applied as they are
    int dense_rows = 0;
    void* comm = nullptr;                // ncclComm_t of the on-stream exchange (null: single rank, the exchange only re-bases)
    bool comm_owned = false;
    int comm_world = 1;
    unsigned long long* d_cycles = nullptr;  // per cluster: shader clocks of the last cluster_kernel launch
    unsigned* d_status = nullptr;  // cluster schedule watchdog words (see report_stall)
    bool in_substep_event = false;  // inside a handler of bepuhip_solve_with_substep_events
    bepuhip_velocity_model velocity_model = {BEPUHIP_VELOCITY_UNIFORM_GRAVITY, {0, 0, 0}, 0};  // bepuhip_set_velocity_model
    float* d_body_gravity = nullptr;  // per-body gravity model: one float per body index
    int body_gravity_capacity = 0, body_gravity_count = 0;
    unsigned long long* d_trace = nullptr;  // optional per-item timeline of cluster 0 (diagnostics)
    size_t trace_words = 0;
    ClusterDesc* d_clusters = nullptr;
    ClusterItem* d_items = nullptr;
    int* d_batch_item_begin = nullptr;
    int* d_cluster_bodies = nullptr;
    int* d_clustered_dynamic = nullptr;
    int clustered_dynamic_count = 0;
    CollidableIn* d_collidables = nullptr;  // device-resident collidable records (bepuhip_set_collidables)
    int collidable_count = 0;
    float* d_hull_points = nullptr;  // convex hulls (bepuhip_set_convex_hulls): xyz triplets, and hull -> first point (hull_count + 1 entries)
    int* d_hull_begin = nullptr;
    int hull_count = 0;
    CompoundChildIn* d_compound_children = nullptr;  // compounds (bepuhip_set_compounds): children of every compound one after the other, and compound -> first child
    int* d_compound_begin = nullptr;
    int compound_count = 0;
    int compound_hulls_needed = 0;  // 1 + the largest hull index any compound child names (checked against the hull table at every prediction)
    float* d_mesh_triangles = nullptr;  // meshes (bepuhip_set_meshes): 9 floats per triangle, mesh -> first triangle, xyz scale per mesh
    int* d_mesh_begin = nullptr;
    float* d_mesh_scales = nullptr;
    int mesh_count = 0;
    int resident_hulls_needed = 0, resident_compounds_needed = 0, resident_meshes_needed = 0;  // 1 + the largest table index the resident collidables name
    unsigned* d_staged = nullptr;   // island schedule: clusters that have staged their bodies in the current launch (see cluster_kernel's kinematic block)
    int* d_kinlist = nullptr;       // constrained kinematic body indices derived from the body references
    int kinlist_count = 0;
    // Structural updates on an island layout change which kinematic bodies are constrained (the list above is the plan's): how many constraints reference each of them
    // (counted when the first update arrives), the list's host mirror, and the bodies whose count has touched zero since the last flush.
    std::vector<int32_t> kinlist_host;
    std::unordered_map<int32_t, int32_t> kin_uses;
    std::vector<int32_t> kin_touched;
    bool kin_uses_ready = false;
    // Bodies that join or leave a plan with their first / last constraint (bepu_soft_updates.h): the list behind kFlagClustered (device copy with spare capacity) ...
    std::vector<int32_t> clustered_dynamic_host;
    std::unordered_map<int32_t, int32_t> clustered_position;  // ... body -> its position in it (built on first use)
    int clustered_dynamic_capacity = 0;
    bool kinlist_dirty = false;                                // ... and of the constrained kinematic bodies (a kinematic body moved)
    bool clustered_dirty = false;                              // the device copy of the list is behind its mirror
    bool free_slots_ready = false;                             // cluster_free_slots lists every unused LDS slot of every cluster (scanned from the slot tables on first use)
    std::unordered_map<int32_t, int32_t> body_moves;           // Bodies.RemoveAt moves seen since the last flush: old index -> new index
    std::vector<BitMark> requirk_marks;  // the conserving modes' bits currently set in the island layout's rows (build_requirk_lists)
    bool soft_flags_stale = false;
    // measurement
    float last_ms = 0;
    bool solve_timing = false, solve_timed = false, last_ms_valid = false;  // bepuhip_set_solve_timing: events around a solve only when asked for (solve_event)
    int64_t last_constraint_iterations = 0;
    bool profiling = false;
    float prof_ms[6] = {0, 0, 0, 0, 0, 0};
    int prof_launches[6] = {0, 0, 0, 0, 0, 0};
    std::map<GraphKey, hipGraphExec_t> graphs;  // captured launch sequences, one per (iteration schedule, dt, integrator); at most kMaxCachedGraphs
};

// Every copy and fill of the library runs on the context's own (non-blocking) stream — never on the legacy stream. A legacy-stream call (hipMemcpy, hipMemset) made on one
// host thread while ANOTHER thread of the process has a stream in capture fails with hipErrorStreamCaptureImplicit and invalidates that thread's capture, whatever the
// capture mode and although the streams are non-blocking (ROCm 7.2; found by tests/test_gpu_soak.py: two contexts on two threads, one capturing its launch-per-batch
// graph while the other uploads); the legacy stream is also not ordered with a non-blocking stream, so a fill issued on it is not ordered with the kernels that follow
// on the context's stream. copy_sync returns when the bytes have arrived (pageable or registered host memory alike).
static hipError_t copy_sync(bepuhip_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
}
static hipError_t fill_async(bepuhip_ctx* c, void* dst, int value, size_t bytes) { return bytes == 0 ? hipSuccess : hipMemsetAsync(dst, value, bytes, c->stream); }


static void clear_graphs(bepuhip_ctx* c) {
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
}

static void free_constraints(bepuhip_ctx* c) {
    clear_graphs(c);
    if (c->d_fallback_indices) hipFree(c->d_fallback_indices);
    c->d_fallback_indices = nullptr; c->has_fallback = false; c->launch_count = 0;
    if (c->d_slab && c->d_slab0 && !c->spare_slab[0] && c->slab_alloc_words > 0) {  // kept for the next upload (build_constraints)
        c->spare_slab[0] = c->d_slab; c->spare_slab[1] = c->d_slab0; c->spare_slab_words = c->slab_alloc_words;
    } else {
        if (c->d_slab) hipFree(c->d_slab);
        if (c->d_slab0) hipFree(c->d_slab0);
    }
    c->slab_alloc_words = 0;
    if (c->d_tbs) hipFree(c->d_tbs);
    if (c->d_inc_tbs) hipFree(c->d_inc_tbs);
    if (c->d_clusters) hipFree(c->d_clusters);
    if (c->d_items) hipFree(c->d_items);
    if (c->d_batch_item_begin) hipFree(c->d_batch_item_begin);
    if (c->d_cluster_bodies) hipFree(c->d_cluster_bodies);
    if (c->d_clustered_dynamic) hipFree(c->d_clustered_dynamic);
    if (c->d_kinlist) hipFree(c->d_kinlist);
    if (c->d_requirk) hipFree(c->d_requirk);
    c->d_requirk = nullptr; c->requirk_begin.clear(); c->requirk_marks.clear();
    if (c->d_trace) hipFree(c->d_trace);
    c->d_trace = nullptr; c->trace_words = 0;
    if (c->d_cycles) hipFree(c->d_cycles);
    c->d_cycles = nullptr;
    // (a device group's record table is mapped by the other members: it keeps its address across uploads and re-plans — group_records below — and dies with the context)
    if (c->d_shared_vel && c->d_shared_vel != c->group_records) hipFree(c->d_shared_vel);
    if (c->d_shared_info) hipFree(c->d_shared_info);
    c->d_shared_vel = nullptr; c->d_shared_info = nullptr; c->clusters_shared = false; c->shared_bodies = 0;
    c->d_clusters = nullptr; c->d_items = nullptr; c->d_batch_item_begin = nullptr; c->d_cluster_bodies = nullptr;
    c->d_clustered_dynamic = nullptr; c->d_kinlist = nullptr;
    c->clusters_enabled = false; c->cluster_count = 0; c->clustered_dynamic_count = 0; c->kinlist_count = 0;
    c->d_slab = c->d_slab0 = nullptr;
    c->d_tbs = c->d_inc_tbs = nullptr;
    for (auto& tb : c->tbs) if (tb.d_device_index && !tb.index_pooled) hipFree(tb.d_device_index);
    if (c->d_index_pool) hipFree(c->d_index_pool);
    c->d_index_pool = nullptr;
    c->tbs.clear();
    c->batch_count = 0; c->batch_begin.clear(); c->batch_blocks.clear();
    c->inc_blocks = 0; c->inc_tb_count = 0; c->total_constraints = 0; c->slab_words = 0; c->referenced_bodies = 0;
    c->built = false;
    c->pending_ops.clear(); c->pending_payload.clear(); c->structure_dirty = false; c->requirk_stale = false;
    c->soft_ok = false; c->soft_split = false; c->clustered_dynamic_host.clear(); c->clustered_position.clear(); c->clustered_dynamic_capacity = 0; c->free_slots_ready = false; c->body_moves.clear(); c->kinlist_host.clear(); c->kin_uses.clear(); c->kin_touched.clear(); c->kin_uses_ready = false; c->cluster_free_slots.clear(); c->cluster_extra_uses.clear(); c->body_apps.clear(); c->split_rerank_flag.clear(); c->split_rerank_list.clear(); c->split_patches.clear(); c->soft_records.clear(); c->soft_payload.clear(); c->soft_index_dirty = false; c->soft_orphans.clear(); c->soft_items_dirty = false; c->items_host.clear(); c->clusters_host.clear(); c->cluster_degraded.clear();
}
