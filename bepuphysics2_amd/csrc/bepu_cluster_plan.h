// Host-side planning of the island-per-workgroup schedule (once per topology upload).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <unistd.h>
#include <pthread.h>
#include <sched.h>
#include <memory>

#include "bepu_host_state.h"
#include <atomic>
#include <chrono>
#include <thread>
#include <unordered_map>

// ---- cluster planning (host, once per topology upload) ----
// Islands = connected components through dynamic bodies (kinematic references never connect: they are read-only to the solver).
// Whole islands are packed, in body-index order, into clusters of at most `cap` LDS-resident bodies; each type batch is
// reordered so that every cluster's constraints are contiguous (coalesced loads per <=64-lane work item), and every work item
// records which earlier items last touched its dynamic bodies (the only ordering the solve has to respect).
struct ClusterPlan {
    bool enabled = false;
    bool shared = false;                 // split islands: some dynamic bodies are referenced from more than one cluster (SharedTables)
    std::vector<uint32_t> shared_info;   // per body index: applications per pass (d), 0 for bodies that are not shared
    std::vector<ClusterDesc> clusters;
    std::vector<ClusterItem> items;
    std::vector<int> batch_item_begin, cluster_bodies, clustered_dynamic, kinlist;
    int max_slots = 0, max_items = 0;
    // whole-island plans: what structural updates need to stay on the island schedule (bepu_soft_updates.h)
    std::vector<int32_t> body_cluster, body_lref, body_degree;
    std::vector<std::unordered_map<int32_t, int32_t>> cluster_kin;
    // split plans: which bodies are shared, their constraint counts, every cluster's ghost / kinematic copies (body | kSlotGhost / kSlotKinematic -> rotated slot), the number of
    // natural slot indices each cluster has handed out, and the order in which the type batches were turned into items
    std::vector<uint8_t> split_shared;
    std::vector<int32_t> split_degree, cluster_natural;
    std::vector<std::unordered_map<int32_t, int32_t>> cluster_extra;
    std::vector<size_t> split_visit;
    int planes = kAllPlanes;  // LDS planes per body slot: all eight fields when they fit, else the six the sweeps touch (the local inertia is then read from memory)
};

static int env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : fallback;
}

constexpr size_t kLdsBudgetBytes = 160 * 1024;  // cluster_lds_bytes (bepu_kernels_common.h) counts everything a workgroup asks for, the scratch row included
// Slot rotation inside every group of 16 (see the LDS layout note above cluster_kernel).
static inline int rotated_slot(int i) { return (i & ~15) | ((i + (i >> 4)) & 15); }

static void plan_split_clusters(bepuhip_ctx* c, ClusterPlan& plan, int universe);
// Device slots a cluster's segment of a type batch gets for `live` constraints: with BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS an eighth more (at least two), so that the
// narrow phase's additions find room without a new plan.
static inline int segment_slots(int live, bool reserve) { return (reserve && live > 0) ? live + std::max(2, live / 8) : live; }
// ... and on a split-island plan: a quarter more (at least four), also in segments that hold nothing yet — additions across the cut land in whichever of the two home
// clusters has room, and a pile's or crowd's contacts come and go everywhere.
static inline int split_segment_slots(int live, bool reserve) { return reserve ? live + std::max(4, live / 4) : live; }
constexpr int32_t kPlanDeadLref = (int32_t)kDynamicLimit;  // 32-bit planning form of a free slot's local references: the kinematic copy in slot 0 (packs to kLrefDead)

// Host threads of the planner: BEPUHIP_PLAN_THREADS, default a quarter of the hardware threads between 8 and 16 (the phases are memory-bound; more only adds start-up cost).
// (round 6) The worker of a background re-plan (bepuhip_replan_begin) plans BESIDE the frames: it takes BEPUHIP_REPLAN_THREADS threads (default 4) from a pool of its own
// — sixteen planner threads next to the caller's own flush loops oversubscribe a sixteen-CPU quota and the frames in between pay for it (crowd: 18 ms frames).
static thread_local int tl_plan_thread_cap = 0;
static int plan_workers(size_t jobs) {
    const int hw = (int)std::thread::hardware_concurrency();
    const int fallback = std::max(1, std::min(16, std::max(8, hw / 4)));
    int wanted = env_int("BEPUHIP_PLAN_THREADS", fallback);
    if (tl_plan_thread_cap > 0) wanted = std::min(wanted, tl_plan_thread_cap);
    return std::max(1, std::min<int>({wanted, hw > 0 ? hw : 1, (int)std::max<size_t>(jobs, 1)}));
}
// Where the parked threads run. The planner's loops hand cache lines back and forth (the union-find's parents, the per-cluster lists): on a two-socket host with sixteen
// L3 domains the scheduler spreads sixteen fresh threads over all of them, and every hand-over is a trip across the fabric — the same loops run a quarter faster when
// the threads share a last-level cache (end_constraints of the bench scene under `taskset -c 0-7`: 12.4 ms against 14.5, profiles/r05_s27_upload_probe_before.txt). The
// threads are therefore kept on the logical CPUs that share an L3 with the CPU the first caller runs on, plus neighbouring L3 domains of the same package until there
// is a CPU per thread — never outside the caller's own affinity mask; nothing is pinned when the topology cannot be read or BEPUHIP_PLAN_PIN=0. The caller's thread is
// left alone.
static bool plan_read_cpu_list(const char* path, cpu_set_t& out) {
    CPU_ZERO(&out);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char text[4096];
    const bool got = fgets(text, sizeof(text), f) != nullptr;
    fclose(f);
    if (!got) return false;
    for (char* at = text; *at && *at != '\n';) {  // "0-7,128-135"
        char* end = nullptr;
        const long a = strtol(at, &end, 10);
        if (end == at) return false;
        long b = a;
        if (*end == '-') { at = end + 1; b = strtol(at, &end, 10); if (end == at) return false; }
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu) if (cpu >= 0) CPU_SET((int)cpu, &out);
        at = (*end == ',') ? end + 1 : end;
        if (*end != ',' && *end != '\n' && *end != 0) return false;
    }
    return CPU_COUNT(&out) > 0;
}
static bool plan_home_cpus(int threads, cpu_set_t& home) {
    CPU_ZERO(&home);
    if (env_int("BEPUHIP_PLAN_PIN", 1) == 0) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    const int here = sched_getcpu();
    if (here < 0 || !CPU_ISSET(here, &allowed)) return false;
    auto l3_of = [](int cpu, cpu_set_t& set) {
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        return plan_read_cpu_list(path, set);
    };
    auto package_of = [](int cpu) {
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu);
        int id = -1;
        if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &id) != 1) id = -1; fclose(f); }
        return id;
    };
    cpu_set_t domain;
    if (!l3_of(here, domain)) return false;
    CPU_AND(&home, &domain, &allowed);
    const int package = package_of(here);
    for (int step = 1; step < CPU_SETSIZE && CPU_COUNT(&home) < threads && package >= 0; ++step)
        for (int cpu : {here + step, here - step}) {
            if (cpu < 0 || cpu >= CPU_SETSIZE || !CPU_ISSET(cpu, &allowed) || CPU_ISSET(cpu, &home) || CPU_COUNT(&home) >= threads) continue;
            if (package_of(cpu) != package || !l3_of(cpu, domain)) continue;
            cpu_set_t more;
            CPU_AND(&more, &domain, &allowed);
            CPU_OR(&home, &home, &more);
        }
    return CPU_COUNT(&home) > 0;
}

// The host threads themselves: started once per process and parked on a condition variable between loops (a planner run is a dozen short parallel loops, a flush of
// structural updates two: starting sixteen threads for each costs more than most of the loops). One loop at a time; a second caller (another context planning on another
// thread) runs its loop on threads of its own.
struct PlanPool {
    std::mutex owner, m;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    const std::function<void(int)>* work = nullptr;
    uint64_t generation = 0;
    int wanted = 0, running = 0;
    pid_t pid = getpid();
    bool placed = false, pinned = false;
    cpu_set_t home;
    void serve(int worker) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* mine = nullptr;
            {
                std::unique_lock<std::mutex> lock(m);
                wake.wait(lock, [&] { return generation != seen; });
                seen = generation;
                if (worker < wanted) mine = work;
            }
            if (!mine) continue;
            (*mine)(worker);
            std::lock_guard<std::mutex> lock(m);
            if (--running == 0) done.notify_one();
        }
    }
    void run(int workers, const std::function<void(int)>& fn) {  // fn(0) here, fn(1 .. workers - 1) on the parked threads
        {
            std::lock_guard<std::mutex> lock(m);
            if (pid != getpid()) { pid = getpid(); threads.clear(); placed = false; }  // a forked child inherits the bookkeeping, not the threads
            if ((int)threads.size() + 1 < workers && !placed) { placed = true; pinned = plan_home_cpus(std::max(workers, plan_workers((size_t)1 << 20)), home); }
            while ((int)threads.size() + 1 < workers) {
                const int id = (int)threads.size() + 1;
                threads.emplace_back([this, id] { serve(id); });
                if (pinned) pthread_setaffinity_np(threads.back().native_handle(), sizeof(home), &home);  // (refused: the thread runs where the scheduler puts it)
                threads.back().detach();
            }
            work = &fn; wanted = workers; running = workers - 1; ++generation;
        }
        wake.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lock(m);
        done.wait(lock, [&] { return running == 0; });
        work = nullptr; wanted = 0;
    }
};
// Never destroyed: its threads outlive main's statics. A forked child gets a pool of its own: the parent's mutexes may have been held by threads that do not
// exist in the child, where locking them again would never return; the old object is abandoned, not freed (its condition variables may be mid-wait in the copy).
static PlanPool*& plan_pool_slot(int which = 0) { static PlanPool* pool[2] = {nullptr, nullptr}; return pool[which]; }  // [1]: the background re-plan workers' pool
static PlanPool& plan_pool() {
    static std::once_flag once;
    std::call_once(once, [] {
        plan_pool_slot(0) = new PlanPool(); plan_pool_slot(1) = new PlanPool();
        pthread_atfork(nullptr, nullptr, [] { plan_pool_slot(0) = new PlanPool(); plan_pool_slot(1) = new PlanPool(); });
    });
    return *plan_pool_slot(tl_plan_thread_cap > 0 ? 1 : 0);
}
template <class Fn>
static void plan_parallel_for_workers(size_t jobs, Fn&& fn) {  // fn(job, worker, workers) for every job, dynamically scheduled; results must not depend on the order
    const int workers = plan_workers(jobs);
    std::atomic<size_t> next{0};
    auto work = [&](int worker) { for (size_t j; (j = next.fetch_add(1)) < jobs;) fn(j, worker, workers); };
    if (workers == 1) { work(0); return; }
    PlanPool& pool = plan_pool();
    static const bool parked = env_int("BEPUHIP_PLAN_POOL", 1) != 0;
    if (parked && pool.owner.try_lock()) {
        const std::function<void(int)> shared_work = work;
        pool.run(workers, shared_work);
        pool.owner.unlock();
        return;
    }
    std::vector<std::thread> own;
    for (int w = 1; w < workers; ++w) own.emplace_back(work, w);
    work(0);
    for (auto& th : own) th.join();
}
template <class Fn>
static void plan_parallel_for(size_t jobs, Fn&& fn) { plan_parallel_for_workers(jobs, [&](size_t j, int, int) { fn(j); }); }
template <class Fn>
static void plan_parallel_ranges(size_t n, size_t grain, Fn&& fn) {  // fn(begin, end) over [0, n) in pieces of `grain`
    plan_parallel_for((n + grain - 1) / grain, [&](size_t j) { fn(j * grain, std::min(n, (j + 1) * grain)); });
}
// A piece of a type batch: the planner's loops over all constraints run over pieces of at most kPlanPiece constraints, so that the threads share the work evenly whatever
// the sizes of the type batches are (the headline scene: 32 type batches of 7,500 to 90,000 constraints on 16 threads).
constexpr int kPlanPiece = 16384;
struct PlanPiece { int32_t t, begin, end; };

// One work item of a sequential fallback type batch: from position `at` of `members` (constraints of one cluster, in the reference's order) as many consecutive constraints
// as share no dynamic body, at most 64. Returns the end position. Before the rows are permuted the references are read at the caller's index, afterwards at the device
// slot segment_begin + position (same constraints, same order).
static size_t fallback_item_end(const HostTypeBatch& tb, const std::vector<int32_t>& members, size_t at, bool rows_are_permuted = false, int segment_begin = 0) {
    int32_t seen[64 * 4];
    int nseen = 0;
    size_t end = at;
    for (; end < members.size() && end - at < 64; ++end) {
        const size_t row = rows_are_permuted ? (size_t)segment_begin + end : (size_t)members[end];
        bool repeats = false;
        for (int k = 0; k < tb.info.bodies && !repeats; ++k) {
            const int32_t r = tb.refs_soa[(size_t)k * tb.stride + row];
            if ((uint32_t)r >= kDynamicLimit) continue;
            for (int q = 0; q < nseen; ++q) repeats |= seen[q] == r;
        }
        if (repeats) break;
        for (int k = 0; k < tb.info.bodies; ++k) {
            const int32_t r = tb.refs_soa[(size_t)k * tb.stride + row];
            if ((uint32_t)r < kDynamicLimit) seen[nseen++] = r;
        }
    }
    return end > at ? end : at + 1;
}

static void plan_clusters(bepuhip_ctx* c, ClusterPlan& plan) {
    const bool plan_stats = env_int("BEPUHIP_PLAN_STATS", 0) >= 2;
    auto plan_t = std::chrono::steady_clock::now();
    auto plan_lap = [&](const char* what) {
        if (!plan_stats) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "bepuhip plan_clusters: %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - plan_t).count());
        plan_t = now;
    };
    int universe = c->referenced_bodies;  // end_constraints has just computed it
    if (universe <= 0)
        for (auto& tb : c->tbs)
            for (int32_t r : tb.refs_soa)
                if (r >= 0) universe = std::max(universe, (r & kRefMask) + 1);
    // One pass over the references on several threads: the kinematic list (Solver.ConstrainedKinematicHandles equivalent, always built; in the order the bodies first
    // appear when the type batches are scanned body slot by body slot — kept through the position of each body's first appearance), which bodies are dynamic, and the
    // islands as a lock-free union-find whose roots are the smallest body index of each component (so the result does not depend on the threads).
    // (arrays the threads fill: allocated raw and first touched in parallel — zero-filling a fresh vector of this size on one thread is page faults, a millisecond of them)
    std::unique_ptr<uint64_t[]> first_seen(new uint64_t[universe]);
    std::unique_ptr<uint8_t[]> is_dyn(new uint8_t[universe]);
    std::unique_ptr<int32_t[]> parent_store(new int32_t[universe]);
    int32_t* parent = parent_store.get();
    plan_parallel_ranges((size_t)universe, 32768, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) { first_seen[i] = UINT64_MAX; is_dyn[i] = 0; parent[i] = (int32_t)i; }
    });
    // The sequential fallback batch (Solver_Solve.cs:546-583: bundles one after the other, a body may repeat ACROSS bundles) runs the island schedule too since round 4:
    // its constraints become work items that are cut wherever a dynamic body would repeat, in the reference's order, and the predecessor lists order them like any
    // other items — a hub body's surplus constraints form a chain of one-constraint items (whole-island plans only; BEPUHIP_FALLBACK_CLUSTERS=0: launch-per-batch levels).
    const bool fallback_here = c->has_fallback;
    const int fallback_batch = c->fallback_threshold;
    const bool want_plan = !((c->flags & BEPUHIP_FLAG_NO_CLUSTERS) || env_int("BEPUHIP_NO_CLUSTERS", 0) || universe == 0 || c->total_constraints == 0 || c->batch_count > kFallbackBatchLimit + 1 ||
                             (fallback_here && (env_int("BEPUHIP_FALLBACK_CLUSTERS", 1) == 0 || (c->flags & BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS))));
    auto find_root = [&](int x) {
        for (;;) {
            const int p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
            if (p == x) return x;
            const int gp = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
            if (gp != p) { int expected = p; __atomic_compare_exchange_n(&parent[x], &expected, gp, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); }  // path halving
            x = p;
        }
    };
    auto unite = [&](int a, int b) {
        for (;;) {
            int ra = find_root(a), rb = find_root(b);
            if (ra == rb) return;
            if (ra < rb) std::swap(ra, rb);  // the larger root goes under the smaller one
            int expected = ra;
            if (__atomic_compare_exchange_n(&parent[ra], &expected, rb, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
        }
    };
    std::atomic<int> bodiless{0};
    std::vector<PlanPiece> pieces;  // (a fallback type batch stays in one piece: an empty lane takes its cluster from the lane before it)
    for (size_t t = 0; t < c->tbs.size(); ++t) {
        const int count = c->tbs[t].count, step = (fallback_here && c->tbs[t].batch == fallback_batch) ? std::max(count, 1) : kPlanPiece;
        for (int b = 0; b < count; b += step) pieces.push_back({(int32_t)t, b, std::min(count, b + step)});
    }
    plan_parallel_for(pieces.size(), [&](size_t piece) {
        const size_t t = (size_t)pieces[piece].t;
        const HostTypeBatch& tb = c->tbs[t];
        auto note_kinematic = [&](int32_t r, int k, int i) {  // earliest (type batch, body slot, index) at which the body appears
            const uint64_t key = ((uint64_t)t << 40) | ((uint64_t)k << 32) | (uint32_t)i;
            uint64_t seen = __atomic_load_n(&first_seen[r & kRefMask], __ATOMIC_RELAXED);
            while (key < seen && !__atomic_compare_exchange_n(&first_seen[r & kRefMask], &seen, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        };
        if (!want_plan) {
            for (int k = 0; k < tb.info.bodies; ++k)
                for (int i = pieces[piece].begin; i < pieces[piece].end; ++i) {
                    const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r >= kDynamicLimit && r >= 0) note_kinematic(r, k, i);
                }
            return;
        }
        // Type batches list their constraints in creation order, i.e. roughly by island: threads that walk different type batches at the same relative position would
        // hammer the same few parents at the same time (a ragdoll's sixteen parents share one cache line). The pieces are handed out type batch by type batch, which
        // puts the threads at different positions.
        for (int i = pieces[piece].begin; i < pieces[piece].end; ++i) {
            int first = -1;
            for (int k = 0; k < tb.info.bodies; ++k) {
                const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                if ((uint32_t)r >= kDynamicLimit) { if (r >= 0) note_kinematic(r, k, i); continue; }
                if (!__atomic_load_n(&is_dyn[r], __ATOMIC_RELAXED)) __atomic_store_n(&is_dyn[r], (uint8_t)1, __ATOMIC_RELAXED);  // read-mostly: an unconditional store makes the line bounce between the threads
                if (first < 0) first = r; else unite(first, r);
            }
            if (first < 0 && tb.refs_soa[i] != -1) bodiless.store(1, std::memory_order_relaxed);  // a constraint with no dynamic body (an empty lane of the fallback batch is not a constraint)
        }
    });
    {
        std::vector<std::pair<uint64_t, int32_t>> order;
        for (int i = 0; i < universe; ++i) if (first_seen[i] != UINT64_MAX) order.push_back({first_seen[i], i});
        std::sort(order.begin(), order.end());
        for (auto& kv : order) plan.kinlist.push_back(kv.second);
    }
    if (!want_plan || bodiless.load()) return;  // (a constraint with no dynamic body: leave everything to the global path)
    const bool reserve = (c->flags & BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS) != 0;
    plan_lap("universe, kinematic list, union-find");
    // Every body's root (the smallest body index of its component), looked up side by side without writing — the unions are over, the paths are short — and from here
    // on parent[i] IS the root; then the component sizes.
    {
        std::unique_ptr<int32_t[]> root_store(new int32_t[universe]);
        int32_t* const root = root_store.get();
        const int32_t* const up = parent;
        plan_parallel_ranges((size_t)universe, 32768, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) { int32_t x = (int32_t)i; while (up[x] != x) x = up[x]; root[i] = x; }
        });
        parent_store.swap(root_store);
    }
    parent = parent_store.get();
    std::vector<int32_t> comp_size(universe, 0);
    int64_t total_dyn = 0;
    int32_t largest = 0;
    for (int i = 0; i < universe; ++i) if (is_dyn[i]) { largest = std::max(largest, ++comp_size[parent[i]]); ++total_dyn; }
    int cap = env_int("BEPUHIP_CLUSTER_BODIES", 0);
    if (cap <= 0) {
        // default: one resident workgroup per CU (the kernel's LDS footprint admits one workgroup per CU), a single round
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const int per_round = std::max(1, cus * 31 / 32) * std::max(1, c->group_world);  // (a device group: every device gets a full round of clusters)
        int64_t target = (total_dyn + per_round - 1) / per_round;
        // More bodies than one round of workgroups holds (1,600 per workgroup): whole rounds of equal clusters. A cap of 1,600 left 345 clusters for 2 M constraints —
        // one full round and a second one that kept 95 of the 256 CUs busy: 0.44 ms/step where two full rounds take 0.35 (profiles/r04_s4_bench_fastbox.json, scale_sweep).
        if (target > 1600) { const int64_t rounds = (target + 1599) / 1600; target = (total_dyn + rounds * per_round - 1) / (rounds * per_round); }
        cap = (int)std::min<int64_t>(std::max<int64_t>(target, 64), 1600);
    }
    if (largest > cap) cap = largest;
    // (tests: BEPUHIP_FORCE_SPLIT=<bodies> cuts the scene's islands as if no workgroup held an island of that many bodies, so that small scenes reach the split-island kernels)
    if (const int force = env_int("BEPUHIP_FORCE_SPLIT", 0); force > 0 && largest >= force) { plan_split_clusters(c, plan, universe); return; }
    // ---- phase A: find a cap whose clusters fit the LDS budget (no mutation yet) ----
    std::vector<int32_t> cluster_of(universe, -1);  // every dynamic body's cluster: its component's
    std::vector<std::vector<int32_t>> cl_of_constraint(c->tbs.size());
    plan_parallel_for(c->tbs.size(), [&](size_t t) { cl_of_constraint[t].resize(c->tbs[t].count); });
    std::vector<std::vector<int32_t>> kin_lists;  // per cluster: the kinematic bodies its constraints reference (of the accepted attempt)
    int nclusters = 0;
    std::vector<int32_t> dyn_count;
    for (int attempt = 0;; ++attempt) {
        nclusters = 0;
        int cur = 0;
        dyn_count.clear();
        for (int i = 0; i < universe; ++i) {  // (a root is the smallest index of its component: it has its cluster before its bodies ask for it)
            if (!is_dyn[i]) continue;
            if (parent[i] == i) {  // roots, ascending: components are packed in that order
                if (nclusters == 0 || cur + comp_size[i] > cap) { ++nclusters; cur = 0; dyn_count.push_back(0); }
                cluster_of[i] = nclusters - 1;
                cur += comp_size[i];
            } else {
                cluster_of[i] = cluster_of[parent[i]];
            }
            ++dyn_count[cluster_of[i]];
        }
        std::vector<int32_t> item_count(nclusters, 0);
        std::vector<std::vector<int32_t>>& kin_seen = kin_lists;
        kin_seen.assign(nclusters, {});
        // per piece of a type batch, side by side: the cluster of every constraint, the constraints per cluster, and the (cluster, kinematic body) pairs in the order they
        // are met; merged in type-batch order afterwards, so that the kinematic copies get the slots a serial scan would give them
        std::vector<std::vector<int32_t>> per_cluster(c->tbs.size()), piece_clusters(pieces.size());
        std::vector<std::vector<std::pair<int32_t, int32_t>>> kin_met(pieces.size());
        plan_parallel_for(pieces.size(), [&](size_t piece) {
            const size_t t = (size_t)pieces[piece].t;
            const HostTypeBatch& tb = c->tbs[t];
            std::vector<int32_t>& counts = piece_clusters[piece];
            counts.assign(nclusters, 0);
            auto& met = kin_met[piece];
            for (int i = pieces[piece].begin; i < pieces[piece].end; ++i) {
                int cl = -1;
                for (int k = 0; k < tb.info.bodies && cl < 0; ++k) {
                    int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r < kDynamicLimit) cl = cluster_of[r];
                }
                if (cl < 0) cl = i > 0 ? cl_of_constraint[t][i - 1] : 0;  // an empty lane of the fallback batch: a dead device slot next to its neighbour
                cl_of_constraint[t][i] = cl;
                counts[cl]++;
                for (int k = 0; k < tb.info.bodies; ++k) {
                    int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if (r >= 0 && (uint32_t)r >= kDynamicLimit) {
                        const std::pair<int32_t, int32_t> pair{cl, r & kRefMask};
                        if (met.empty() || (met.back() != pair && std::find(met.begin(), met.end(), pair) == met.end())) met.push_back(pair);
                    }
                }
            }
        });
        for (size_t t = 0; t < c->tbs.size(); ++t) per_cluster[t].assign(nclusters, 0);
        for (size_t piece = 0; piece < pieces.size(); ++piece) {  // (in type-batch order, a type batch's pieces in index order)
            std::vector<int32_t>& sum = per_cluster[pieces[piece].t];
            for (int cl = 0; cl < nclusters; ++cl) sum[cl] += piece_clusters[piece][cl];
            for (auto& pair : kin_met[piece]) {
                auto& ks = kin_seen[pair.first];
                if (std::find(ks.begin(), ks.end(), pair.second) == ks.end()) ks.push_back(pair.second);
            }
        }
        for (size_t t = 0; t < c->tbs.size(); ++t) {
            if (fallback_here && c->tbs[t].batch == fallback_batch) {  // items end where a dynamic body would repeat: counted as they will be cut (fallback_item_end)
                const HostTypeBatch& tb = c->tbs[t];
                std::vector<std::vector<int32_t>> members(nclusters);
                for (int i = 0; i < tb.count; ++i) members[cl_of_constraint[t][i]].push_back(i);
                for (int cl = 0; cl < nclusters; ++cl)
                    for (size_t at = 0; at < members[cl].size(); ++item_count[cl]) at = fallback_item_end(tb, members[cl], at);
                continue;
            }
            for (int cl = 0; cl < nclusters; ++cl) item_count[cl] += (segment_slots(per_cluster[t][cl], reserve) + 63) / 64;
        }
        int max_slots = 0, max_items = 0;
        for (int cl = 0; cl < nclusters; ++cl) {
            max_slots = std::max(max_slots, (dyn_count[cl] + (int)kin_seen[cl].size() + 15) / 16 * 16);
            max_items = std::max(max_items, item_count[cl]);
        }
        if (max_items < 65536 && cluster_lds_bytes(plan.planes, max_slots, max_items) <= kLdsBudgetBytes) break;
        if (plan.planes == kAllPlanes) { plan.planes = kSweepPlanes; continue; }  // leave the local inertia in memory before giving up cluster size
        if (cap <= largest || attempt > 24) {  // an island (plus its work items) does not fit one workgroup: cut it (shared bodies), or leave it to the global path
            plan_split_clusters(c, plan, universe);
            return;
        }
        cap = std::max<int>(largest, cap * 7 / 8);
    }
    plan_lap("phase A: cluster sizes");
    // ---- phase B: local slots, reordered type batches, work items with predecessor lists ----
    std::vector<std::vector<int32_t>> cl_bodies(nclusters);  // natural local order: dynamics ascending, kinematics appended on first use
    std::vector<int32_t> local_of(universe, -1);
    for (int cl = 0; cl < nclusters; ++cl) cl_bodies[cl].reserve((size_t)dyn_count[cl] + kin_lists[cl].size());
    plan.clustered_dynamic.reserve(plan.clustered_dynamic.size() + (size_t)total_dyn);
    for (int i = 0; i < universe; ++i)
        if (is_dyn[i]) { int cl = cluster_of[i]; local_of[i] = (int)cl_bodies[cl].size(); cl_bodies[cl].push_back(i); plan.clustered_dynamic.push_back(i); }
    // Kinematic copies: slots behind the cluster's dynamic bodies, in the order phase A met them. Assigned before the rows are built so that building them
    // (the bulk of this function's time: every prestep / impulse row of every type batch is permuted) can run on several threads.
    std::vector<std::unordered_map<int32_t, int32_t>> cl_kin(nclusters);  // kinematic body -> natural local index
    for (int cl = 0; cl < nclusters; ++cl)
        for (int32_t body : kin_lists[cl]) { cl_kin[cl].emplace(body, (int)cl_bodies[cl].size()); cl_bodies[cl].push_back(body | (int)kDynamicLimit); }
    std::vector<std::vector<int32_t>> last_toucher(nclusters);  // by slot: cluster-relative index of the item that last touched the (dynamic) body
    for (int cl = 0; cl < nclusters; ++cl) last_toucher[cl].assign((cl_bodies[cl].size() + 15) / 16 * 16, -1);
    std::vector<std::vector<ClusterItem>> cl_items(nclusters);
    std::vector<std::vector<std::pair<int32_t, int32_t>>> first_touch(nclusters);  // (item, slot): the item is the slot's first toucher in a pass
    // Items are claimed in list order. Inside a batch any order is legal (a batch never references a body twice); the types that move the most data per
    // constraint go first so that their loads and velocity-independent work start as early as the claim sequence allows.
    std::vector<size_t> visit(c->tbs.size());
    for (size_t t = 0; t < visit.size(); ++t) visit[t] = t;
    if (env_int("BEPUHIP_CLUSTER_ORDER", 1) != 0)
        std::stable_sort(visit.begin(), visit.end(), [&](size_t a, size_t b) {
            const HostTypeBatch &x = c->tbs[a], &y = c->tbs[b];
            if (x.batch != y.batch) return x.batch < y.batch;
            if (fallback_here && x.batch == fallback_batch) return false;  // the sequential fallback batch: type batches in the reference's order (Solver_Solve.cs:546-583)
            return x.info.prestep + 2 * x.info.impulse > y.info.prestep + 2 * y.info.impulse;
        });
    // Rows of every type batch in cluster order (stable: inside a cluster the caller's order stays). Type batches are independent of each other here.
    auto permute_type_batch = [&](size_t t) {
        HostTypeBatch& tb = c->tbs[t];
        const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse;
        const std::vector<int32_t>& clc = cl_of_constraint[t];
        std::vector<int32_t> live(nclusters, 0);
        for (int i = 0; i < tb.count; ++i) ++live[clc[i]];
        tb.seg_begin.assign(nclusters + 1, 0);  // every cluster's constraints in one segment of device slots, free slots (if reserved) behind the live ones
        for (int cl = 0; cl < nclusters; ++cl) tb.seg_begin[cl + 1] = tb.seg_begin[cl] + segment_slots(live[cl], reserve);
        tb.slots = tb.seg_begin[nclusters];
        const int stride = std::max(tb.stride, (tb.slots + 63) / 64 * 64);
        std::vector<int32_t> next(tb.seg_begin.begin(), tb.seg_begin.end() - 1);
        tb.perm.assign(tb.slots, -1);
        for (int i = 0; i < tb.count; ++i) tb.perm[next[clc[i]]++] = i;  // counting sort by cluster: inside a cluster the caller's order stays
        tb.inv.assign(tb.count, 0);
        std::vector<int32_t> refs((size_t)nb * stride, -1), lrefs((size_t)nb * stride, -1);
        const bool host_values = c->host_values;  // the product leaves prestep data and impulses on the device (transposed there through `inv`); only the offline harness permutes them here
        std::vector<float> pre(host_values ? (size_t)pf * stride : 0, 0.0f), acc(host_values ? (size_t)imf * stride : 0, 0.0f);
        for (int k = 0; k < nb; ++k) {
            const int32_t* src = tb.refs_soa.data() + (size_t)k * tb.stride;
            int32_t* dst = refs.data() + (size_t)k * stride;
            int32_t* ldst = lrefs.data() + (size_t)k * stride;
            for (int d = 0; d < tb.slots; ++d) {
                const int h = tb.perm[d];
                if (h < 0) { ldst[d] = kPlanDeadLref; continue; }
                if (k == 0) tb.inv[h] = d;
                const int32_t r = src[h];
                dst[d] = r;
                if (r < 0) { ldst[d] = kPlanDeadLref; continue; }  // an empty lane of the fallback batch: computes on the kinematic copy in slot 0 and writes no body, like a free slot
                ldst[d] = ((uint32_t)r < kDynamicLimit) ? rotated_slot(local_of[r]) : (rotated_slot(cl_kin[clc[h]].find(r & kRefMask)->second) | (int)kDynamicLimit);
            }
        }
        for (int f = 0; f < pf && host_values; ++f) {
            const float* src = tb.prestep_soa.data() + (size_t)f * tb.stride;
            float* dst = pre.data() + (size_t)f * stride;
            for (int d = 0; d < tb.slots; ++d) if (tb.perm[d] >= 0) dst[d] = src[tb.perm[d]];
        }
        for (int f = 0; f < imf && host_values; ++f) {
            const float* src = tb.accum_soa.data() + (size_t)f * tb.stride;
            float* dst = acc.data() + (size_t)f * stride;
            for (int d = 0; d < tb.slots; ++d) if (tb.perm[d] >= 0) dst[d] = src[tb.perm[d]];
        }
        tb.stride = stride;
        tb.dev_refs = refs;
        tb.refs_soa.swap(refs); tb.prestep_soa.swap(pre); tb.accum_soa.swap(acc); tb.lrefs_soa.swap(lrefs);
    };
    plan_lap("slots, kinematic copies");
    plan_parallel_for(c->tbs.size(), permute_type_batch);
    plan_lap("permuted rows (threads)");
    // Work items and their predecessor lists: a cluster's items depend on that cluster's bodies only, so the clusters are built side by side; inside a cluster the type
    // batches are visited in the fixed order above, which is the order of its item list.
    plan_parallel_for((size_t)nclusters, [&](size_t cluster) {
        const int cl = (int)cluster;
        for (size_t t : visit) {
            HostTypeBatch& tb = c->tbs[t];
            const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse;
            const int d = tb.seg_begin[cl], e = tb.seg_begin[cl + 1];
            if (d == e) continue;
            const bool sequential = fallback_here && tb.batch == fallback_batch;
            std::vector<int32_t> segment;  // fallback type batches: the segment's constraints by the caller's index, to cut the items exactly as phase A counted them
            if (sequential) for (int j = d; j < e; ++j) segment.push_back(tb.perm[j]);
            for (int s0 = d, next = d; s0 < e; s0 = next) {
                next = sequential ? d + (int)fallback_item_end(tb, segment, (size_t)(s0 - d), /*rows_are_permuted=*/true, d) : std::min(e, s0 + 64);
                ClusterItem it;
                memset(&it, 0, sizeof(it));
                it.type_id = tb.type_id; it.count = next - s0; it.stride = tb.stride; it.start = s0;
                it.tb = (int)t; it.shape = nb | (pf << 8) | (imf << 16);
                const int self = (int)cl_items[cl].size();
                int npred = 0, overflow = 0;
                std::vector<int32_t>& lt = last_toucher[cl];
                for (int j = s0; j < s0 + it.count; ++j)
                    for (int k = 0; k < nb; ++k) {
                        const int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + j];
                        if ((uint32_t)lr >= kDynamicLimit || tb.perm[j] < 0) continue;  // kinematic copy, or a free slot
                        if ((size_t)lr >= lt.size()) lt.resize((size_t)lr + 16, -1);
                        const int pred = lt[lr];
                        if (pred < 0) { first_touch[cl].push_back({self, lr}); continue; }  // this item is the body's first toucher in a pass
                        if (pred == self) continue;
                        bool known = false;
                        for (int q = 0; q < npred; ++q) known |= it.pred[q] == pred;
                        if (known) continue;
                        if (npred < kMaxPreds) it.pred[npred++] = (unsigned short)pred; else overflow = 1;
                    }
                for (int j = s0; j < s0 + it.count; ++j)
                    for (int k = 0; k < nb; ++k) {
                        const int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + j];
                        if ((uint32_t)lr < kDynamicLimit && tb.perm[j] >= 0) lt[lr] = self;
                    }
                if (overflow) npred = 0;
                it.batch_npred = (tb.batch & 0xFFFF) | (npred << 16) | (overflow << 24);
                cl_items[cl].push_back(it);
            }
        }
    });
    plan_lap("work items, predecessor lists");
    // The kernel reads the local references as 16-bit halves, two body slots per word (slot < 32768; bit 15 = kinematic copy): half the bytes, and one
    // load instead of two for a two-body constraint. The 32-bit form above was only needed for the predecessor search.
    plan_parallel_for(c->tbs.size(), [&](size_t t) {
        HostTypeBatch& tb = c->tbs[t];
        const int nb = tb.info.bodies, rows = (nb + 1) / 2;
        std::vector<int32_t> packed((size_t)rows * tb.stride, 0);
        for (int k = 0; k < nb; ++k)
            for (int d = 0; d < tb.slots; ++d) {
                const int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + d];
                const uint32_t half = ((uint32_t)lr & 0x7FFFu) | (((uint32_t)lr >= kDynamicLimit) ? 0x8000u : 0u);  // a free slot packs to kLrefDead
                packed[(size_t)(k / 2) * tb.stride + d] |= (int32_t)(half << (16 * (k & 1)));
            }
        tb.lrefs_soa.swap(packed);
    });
    plan_lap("packed local references");
    // Cross-pass predecessors: the last toucher (end of a pass) of every body an item touches first.
    plan_parallel_for((size_t)nclusters, [&](size_t cluster) {
        const int cl = (int)cluster;
        for (auto& fs : first_touch[cl]) {
            ClusterItem& it = cl_items[cl][fs.first];
            const int last = last_toucher[cl][fs.second];
            int nx = (it.batch_npred >> 20) & 0xF;
            if ((it.batch_npred >> 25) & 1) continue;
            bool known = false;
            for (int q = 0; q < nx; ++q) known |= it.xpred[q] == last;
            if (known) continue;
            if (nx < kMaxPreds) { it.xpred[nx++] = (unsigned short)last; it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (nx << 20); }
            else it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (1 << 25);
        }
    });
    {
        size_t slots = 0, items = 0;
        for (int cl = 0; cl < nclusters; ++cl) { slots += (cl_bodies[cl].size() + 15) / 16 * 16; items += cl_items[cl].size(); }
        plan.cluster_bodies.reserve(plan.cluster_bodies.size() + slots);
        plan.items.reserve(plan.items.size() + items);
        plan.batch_item_begin.reserve(plan.batch_item_begin.size() + (size_t)nclusters * (c->batch_count + 1));
        plan.clusters.reserve(plan.clusters.size() + nclusters);
    }
    for (int cl = 0; cl < nclusters; ++cl) {
        ClusterDesc d;
        d.body_begin = (int)plan.cluster_bodies.size();
        d.slot_count = ((int)cl_bodies[cl].size() + 15) / 16 * 16;
        std::vector<int32_t> slots(d.slot_count, -1);
        for (size_t i = 0; i < cl_bodies[cl].size(); ++i) slots[rotated_slot((int)i)] = cl_bodies[cl][i];
        plan.cluster_bodies.insert(plan.cluster_bodies.end(), slots.begin(), slots.end());
        d.item_begin = (int)plan.items.size();
        d.item_count = (int)cl_items[cl].size();
        d.batch_item_offset = (int)plan.batch_item_begin.size();
        // items were appended in type-batch order == batch order
        int k = 0;
        for (int b = 0; b <= c->batch_count; ++b) {
            while (k < d.item_count && (cl_items[cl][k].batch_npred & 0xFFFF) < b) ++k;
            plan.batch_item_begin.push_back(d.item_begin + k);
        }
        plan.items.insert(plan.items.end(), cl_items[cl].begin(), cl_items[cl].end());
        plan.clusters.push_back(d);
        plan.max_slots = std::max(plan.max_slots, d.slot_count);
        plan.max_items = std::max(plan.max_items, d.item_count);
    }
    plan.enabled = nclusters > 0 && cluster_lds_bytes(plan.planes, plan.max_slots, plan.max_items) <= kLdsBudgetBytes;
    // what structural updates need in order to stay on this plan
    plan.body_cluster.swap(cluster_of);  // (-1 for every body that is not dynamic here)
    plan_parallel_ranges((size_t)universe, 32768, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) if (local_of[i] >= 0) local_of[i] = rotated_slot(local_of[i]); });
    plan.body_lref.swap(local_of);
    // (the bodies' constraint counts, which only structural updates need, are counted from dev_refs by the first of them: soft_ensure_degrees)
    plan.cluster_kin.resize(nclusters);
    for (int cl = 0; cl < nclusters; ++cl)
        for (auto& kv : cl_kin[cl]) plan.cluster_kin[cl].emplace(kv.first, rotated_slot(kv.second));
    plan_lap("cross-pass lists, descriptors, mirrors");
}


// ---- split-island plans ----
// An island that no workgroup's LDS can hold is cut into clusters of neighbouring bodies (breadth-first regions of the constraint graph). A constraint runs
// in the cluster of its first dynamic body; a dynamic body that a constraint of ANOTHER cluster references is shared: its velocity lives in a global table
// during the sweeps and every application on it waits for its event number in the body's record (kernel side: SharedRef, acquire_shared). Everything else — private
// bodies in LDS, work items, predecessor flags — is the island schedule's. Islands that fit are still packed whole. All clusters must be resident at once
// (they wait for each other), so the plan is refused (global path) when it needs more clusters than the device has CUs.
static void plan_split_clusters(bepuhip_ctx* c, ClusterPlan& plan, int universe) {
    if (env_int("BEPUHIP_NO_SPLIT", 0) || c->has_fallback) return;  // (a body's ranks are counted per batch; the sequential fallback batch keeps to whole islands)
    if (env_int("BEPUHIP_SPLIT_MANY_BODY", 1) == 0)
        for (auto& tb : c->tbs) if (tb.info.bodies > 2) return;  // (round 2: three- and four-body constraints kept to whole islands)
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const bool split_stats = env_int("BEPUHIP_PLAN_STATS", 0) >= 2;
    auto split_t0 = std::chrono::steady_clock::now();
    auto split_lap = [&](const char* what) {
        if (!split_stats) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "bepuhip plan_split_clusters: %-44s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - split_t0).count());
        split_t0 = now;
    };
    // adjacency of dynamic bodies (CSR), degree d per body. One thread: on the hosts this was measured on, an owner-computes build (every thread walks all references and
    // fills its own bodies' lists, so that the lists keep the serial order) and atomic counts LOSE to this loop — 11.4 against 8.2 ms for the crowd — the tables fit the
    // cache of one core and sixteen walks over all references do not.
    std::vector<uint8_t> is_dyn(universe, 0);
    std::vector<int32_t> deg(universe, 0);
    for (auto& tb : c->tbs)
        for (int k = 0; k < tb.info.bodies; ++k)
            for (int i = 0; i < tb.count; ++i) {
                const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                if ((uint32_t)r < kDynamicLimit) { is_dyn[r] = 1; ++deg[r]; }
            }
    // (a three- or four-body constraint links its bodies in a chain: body slot k with k + 1)
    auto for_each_edge = [&](auto&& fn) {
        for (auto& tb : c->tbs)
            for (int k = 0; k + 1 < tb.info.bodies; ++k)
                for (int i = 0; i < tb.count; ++i) {
                    const int32_t a = tb.refs_soa[(size_t)k * tb.stride + i], b = tb.refs_soa[(size_t)(k + 1) * tb.stride + i];
                    if ((uint32_t)a < kDynamicLimit && (uint32_t)b < kDynamicLimit) fn(a, b);
                }
    };
    std::vector<int64_t> adj_begin(universe + 1, 0);
    for_each_edge([&](int32_t a, int32_t b) { ++adj_begin[a + 1]; ++adj_begin[b + 1]; });
    for (int i = 0; i < universe; ++i) adj_begin[i + 1] += adj_begin[i];
    std::unique_ptr<int32_t[]> adj_store(new int32_t[std::max<int64_t>(adj_begin[universe], 1)]);  // (every entry is written below)
    int32_t* const adj = adj_store.get();
    {
        std::vector<int64_t> fill(adj_begin.begin(), adj_begin.end() - 1);
        for_each_edge([&](int32_t a, int32_t b) { adj[fill[a]++] = b; adj[fill[b]++] = a; });
    }
    int64_t total_dyn = 0;
    for (int i = 0; i < universe; ++i) total_dyn += is_dyn[i];
    if (total_dyn == 0) return;
    for (int i = 0; i < universe; ++i) if (deg[i] > 255) return;  // rank | degree travel as bytes
    // (a device group, bepuhip_set_device_group: the clusters of ALL devices are planned here, identically on every device; each runs a contiguous range of them and
    // only that range has to be resident on it)
    const int group = std::max(1, c->group_world);
    const int target_clusters = std::max(1, std::min(cus * 31 / 32 * group, env_int("BEPUHIP_SPLIT_CLUSTERS", cus * 31 / 32 * group)));
    int region = (int)((total_dyn + target_clusters - 1) / target_clusters);
    region = std::max(region, 32);
    // ---- regions: grown breadth-first around a seed until they hold `region` bodies (compact balls of the constraint graph: the fewer bodies on a
    // region's surface, the fewer are shared). The next seed is a body the last region's frontier already reached, so consecutive clusters are neighbours;
    // a component that ends early lets the next one continue the cluster (small islands are packed together, as in the whole-island plan). ----
    std::vector<int32_t> body_cluster(universe, -1);
    int nclusters = 0;
    {
        std::vector<int32_t> queue, carry;
        int cur = 0, scan = 0;
        size_t carry_at = 0;
        for (;;) {
            int seed = -1;
            while (carry_at < carry.size() && seed < 0) { const int v = carry[carry_at++]; if (body_cluster[v] < 0) seed = v; }
            if (seed < 0) {
                carry.clear(); carry_at = 0;
                while (scan < universe && (!is_dyn[scan] || body_cluster[scan] >= 0)) ++scan;
                if (scan == universe) break;
                seed = scan;
            }
            if (nclusters == 0 || cur >= region) { ++nclusters; cur = 0; }
            queue.clear();
            queue.push_back(seed);
            size_t q = 0;
            for (; q < queue.size() && cur < region; ++q) {
                const int u = queue[q];
                if (body_cluster[u] >= 0) continue;
                body_cluster[u] = nclusters - 1;
                ++cur;
                for (int64_t e = adj_begin[u]; e < adj_begin[u + 1]; ++e)
                    if (body_cluster[adj[e]] < 0) queue.push_back(adj[e]);
            }
            if (q < queue.size()) {  // region full: what the frontier had reached seeds the next regions
                if (carry_at > 0) { carry.erase(carry.begin(), carry.begin() + carry_at); carry_at = 0; }
                carry.insert(carry.end(), queue.begin() + q, queue.end());
            }
        }
    }
    if ((nclusters + group - 1) / group > cus) return;
    split_lap("adjacency, regions");
    // Smooth the regions' surfaces (BEPUHIP_SPLIT_REFINE = sweeps, default 2; 0 = off). A body moves to the neighbouring region that holds more of its constraint
    // partners than its own does (ties stay), as long as no region leaves [7/8, 9/8] of the target size: fewer crossing constraints for the same regions.
    if (const int sweeps = env_int("BEPUHIP_SPLIT_REFINE", 2)) {
        std::vector<int32_t> size(nclusters, 0);
        for (int v = 0; v < universe; ++v) if (body_cluster[v] >= 0) ++size[body_cluster[v]];
        const int lo = region * 7 / 8, hi = region * 9 / 8;
        std::vector<int32_t> seen_cluster, seen_count;
        for (int sweep = 0; sweep < sweeps; ++sweep) {
            int moved = 0;
            for (int v = 0; v < universe; ++v) {
                const int own = body_cluster[v];
                if (own < 0 || adj_begin[v] == adj_begin[v + 1]) continue;
                seen_cluster.clear(); seen_count.clear();
                int own_count = 0;
                for (int64_t e = adj_begin[v]; e < adj_begin[v + 1]; ++e) {
                    const int cl = body_cluster[adj[e]];
                    if (cl == own) { ++own_count; continue; }
                    size_t k = 0;
                    while (k < seen_cluster.size() && seen_cluster[k] != cl) ++k;
                    if (k == seen_cluster.size()) { seen_cluster.push_back(cl); seen_count.push_back(0); }
                    ++seen_count[k];
                }
                int best = -1, best_count = own_count;
                for (size_t k = 0; k < seen_cluster.size(); ++k)
                    if (seen_count[k] > best_count && size[seen_cluster[k]] < hi) { best = seen_cluster[k]; best_count = seen_count[k]; }
                if (best < 0 || size[own] <= lo) continue;
                body_cluster[v] = best; --size[own]; ++size[best]; ++moved;
            }
            if (moved == 0) break;
        }
    }
    split_lap("refine sweeps");
    // ---- constraints -> clusters, shared bodies, per-pass rank of every application on a shared body (type batches are in batch order) ----
    std::vector<std::vector<int32_t>> cl_of_constraint(c->tbs.size());
    std::vector<uint8_t> shared(universe, 0);
    // Which side runs a constraint that crosses the cut decides which of its two bodies becomes shared. The shared bodies are a greedy vertex cover of the cut
    // constraints — the body with the most crossing constraints first — and a crossing constraint runs on the side of its body that is NOT in the cover (either side
    // when both are): fewer distinct shared bodies for the same cut than letting the first dynamic body's cluster run it (BEPUHIP_SPLIT_COVER=0 restores that rule).
    // Measured in round 3 on an MI355X (profiles/r03_split_cut_ab.txt): pile 0.437 -> 0.421 ms/step, ragdoll crowd 0.554 -> 0.518 with two refine sweeps.
    std::vector<uint8_t> in_cover;
    if (env_int("BEPUHIP_SPLIT_COVER", 1)) {
        std::vector<int32_t> crossing(universe, 0);
        std::vector<PlanPiece> pieces;  // two-body type batches in pieces: the counts are sums, and the cover only asks WHETHER a body's crossing constraints are covered
        for (size_t t = 0; t < c->tbs.size(); ++t)
            if (c->tbs[t].info.bodies == 2)
                for (int b = 0; b < c->tbs[t].count; b += kPlanPiece) pieces.push_back({(int32_t)t, b, std::min(c->tbs[t].count, b + kPlanPiece)});
        auto for_each_crossing = [&](auto&& fn) {
            plan_parallel_for(pieces.size(), [&](size_t piece) {
                const HostTypeBatch& tb = c->tbs[pieces[piece].t];
                for (int i = pieces[piece].begin; i < pieces[piece].end; ++i) {
                    const int32_t a = tb.refs_soa[i], b = tb.refs_soa[(size_t)tb.stride + i];
                    if ((uint32_t)a < kDynamicLimit && (uint32_t)b < kDynamicLimit && body_cluster[a] != body_cluster[b]) fn(a, b);
                }
            });
        };
        for_each_crossing([&](int32_t a, int32_t b) { __atomic_fetch_add(&crossing[a], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&crossing[b], 1, __ATOMIC_RELAXED); });
        std::vector<int32_t> order;
        for (int v = 0; v < universe; ++v) if (crossing[v] > 0) order.push_back(v);
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return crossing[x] > crossing[y]; });
        // greedy cover in one sweep: a body enters the cover if one of its crossing constraints is still uncovered (its other body is not in the cover yet)
        std::vector<int32_t> across_begin(universe + 1, 0);  // the other bodies of every body's crossing constraints (CSR; a body's entries in no particular order)
        for (int v = 0; v < universe; ++v) across_begin[v + 1] = across_begin[v] + crossing[v];
        std::vector<int32_t> across(across_begin[universe]), fill(across_begin.begin(), across_begin.end() - 1);
        for_each_crossing([&](int32_t a, int32_t b) {
            across[__atomic_fetch_add(&fill[a], 1, __ATOMIC_RELAXED)] = b;
            across[__atomic_fetch_add(&fill[b], 1, __ATOMIC_RELAXED)] = a;
        });
        in_cover.assign(universe, 0);
        for (int32_t v : order) {
            bool uncovered = false;
            for (int32_t e = across_begin[v]; e < across_begin[v + 1]; ++e) uncovered |= !in_cover[across[e]];
            in_cover[v] = uncovered;
        }
    }
    std::atomic<int> bodiless{0};
    std::vector<PlanPiece> all_pieces;
    for (size_t t = 0; t < c->tbs.size(); ++t)
        for (int b = 0; b < c->tbs[t].count; b += kPlanPiece) all_pieces.push_back({(int32_t)t, b, std::min(c->tbs[t].count, b + kPlanPiece)});
    plan_parallel_for(c->tbs.size(), [&](size_t t) { cl_of_constraint[t].resize(c->tbs[t].count); });
    plan_parallel_for(all_pieces.size(), [&](size_t piece) {  // (several constraints may mark the same body shared: the same byte, the same value)
        const size_t t = (size_t)all_pieces[piece].t;
        HostTypeBatch& tb = c->tbs[t];
        for (int i = all_pieces[piece].begin; i < all_pieces[piece].end; ++i) {
            int cl = -1;
            for (int k = 0; k < tb.info.bodies && cl < 0; ++k) {
                const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                if ((uint32_t)r < kDynamicLimit) cl = body_cluster[r];
            }
            if (!in_cover.empty() && tb.info.bodies == 2) {
                const int32_t a = tb.refs_soa[i], b = tb.refs_soa[(size_t)tb.stride + i];
                if ((uint32_t)a < kDynamicLimit && (uint32_t)b < kDynamicLimit && body_cluster[a] != body_cluster[b] && in_cover[a] && !in_cover[b]) cl = body_cluster[b];
            }
            if (cl < 0) { bodiless.store(1, std::memory_order_relaxed); cl = 0; }
            cl_of_constraint[t][i] = cl;
            for (int k = 0; k < tb.info.bodies; ++k) {
                const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                if ((uint32_t)r < kDynamicLimit && body_cluster[r] != cl) __atomic_store_n(&shared[r], (uint8_t)1, __ATOMIC_RELAXED);
            }
        }
    });
    if (bodiless.load()) return;  // a constraint with no dynamic body: leave everything to the global path
    split_lap("vertex cover, constraints -> clusters");
    if (env_int("BEPUHIP_PLAN_STATS", 0) >= 2) {  // how many hand-offs of shared bodies stay inside one cluster (rank r and r + 1 of a pass run by the same cluster)
        std::vector<int32_t> last_cluster(universe, -1);
        long long applications = 0, local_pairs = 0, pairs = 0;
        for (size_t t = 0; t < c->tbs.size(); ++t) {
            const HostTypeBatch& tb = c->tbs[t];
            for (int k = 0; k < tb.info.bodies; ++k)
                for (int i = 0; i < tb.count; ++i) {
                    const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r >= kDynamicLimit || !shared[r]) continue;
                    ++applications;
                    if (last_cluster[r] >= 0) { ++pairs; local_pairs += last_cluster[r] == cl_of_constraint[t][i]; }
                    last_cluster[r] = cl_of_constraint[t][i];
                }
        }
        fprintf(stderr, "bepuhip split plan: %lld applications on shared bodies per pass, %lld consecutive pairs of which %lld (%.1f %%) inside one cluster\n", applications, pairs, local_pairs,
                100.0 * local_pairs / std::max(1LL, pairs));
    }
    // Rank words: rank | degree << 8, plus the two hand-off flags of bepu_cluster_kernel.h (kRankPredLocal / kRankSuccLocal): consecutive applications of a pass on a
    // shared body that the SAME cluster runs pass the velocity through that cluster's LDS slot of the body instead of the record in HBM (pile: 42 % of the hand-offs,
    // ragdoll crowd: 78 %). BEPUHIP_SPLIT_LOCAL_HANDOFF=0 turns them off.
    constexpr uint32_t kPlanRankPredLocal = 1u << 16, kPlanRankSuccLocal = 1u << 17;
    const int handoff_mode = env_int("BEPUHIP_SPLIT_LOCAL_HANDOFF", 1);  // debugging: 2 = only inside the body's home cluster, 3 = only inside clusters that hold a ghost copy, 4 = only contacts
    const bool local_handoff = handoff_mode != 0;
    split_lap("(statistics)");
    std::vector<std::vector<uint32_t>> srank(c->tbs.size());
    {
        std::vector<int32_t> next_rank(universe, 0), last_cluster(universe, -1);
        std::vector<uint32_t*> last_word(universe, nullptr);
        plan_parallel_for(c->tbs.size(), [&](size_t t) { srank[t].assign((size_t)c->tbs[t].info.bodies * c->tbs[t].stride, 0u); });
        // Batch order == type batch order, and inside a batch a body appears at most once: the batches one after the other, every batch's constraints side by side
        // (what a constraint reads and writes here belongs to its own bodies).
        std::vector<PlanPiece> pieces;
        for (size_t t0 = 0; t0 < c->tbs.size();) {
            size_t t1 = t0;
            pieces.clear();
            for (; t1 < c->tbs.size() && c->tbs[t1].batch == c->tbs[t0].batch; ++t1)
                for (int b = 0; b < c->tbs[t1].count; b += kPlanPiece) pieces.push_back({(int32_t)t1, b, std::min(c->tbs[t1].count, b + kPlanPiece)});
            t0 = t1;
            plan_parallel_for(pieces.size(), [&](size_t piece) {
            const size_t t = (size_t)pieces[piece].t;
            HostTypeBatch& tb = c->tbs[t];
            for (int k = 0; k < tb.info.bodies; ++k)
                for (int i = pieces[piece].begin; i < pieces[piece].end; ++i) {
                    const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r >= kDynamicLimit || !shared[r]) continue;
                    uint32_t& word = srank[t][(size_t)k * tb.stride + i];
                    word = (uint32_t)next_rank[r]++ | ((uint32_t)deg[r] << 8);
                    const int mode = handoff_mode;
                    const bool allowed = mode == 1 || (mode == 2 && cl_of_constraint[t][i] == body_cluster[r]) || (mode == 3 && cl_of_constraint[t][i] != body_cluster[r]) || (mode == 4 && tb.type_id < 8);
                    if (local_handoff && allowed && last_word[r] != nullptr && last_cluster[r] == cl_of_constraint[t][i]) { word |= kPlanRankPredLocal; *last_word[r] |= kPlanRankSuccLocal; }
                    last_word[r] = &word; last_cluster[r] = cl_of_constraint[t][i];
                }
            });
        }
    }
    split_lap("rank words");
    // ---- slots: home bodies (ascending), then ghosts and kinematic copies on first use ----
    std::vector<std::vector<int32_t>> cl_bodies(nclusters);
    std::vector<int32_t> local_of(universe, -1);
    {
        std::vector<int32_t> homes(nclusters, 0);
        for (int i = 0; i < universe; ++i) if (is_dyn[i]) ++homes[body_cluster[i]];
        for (int cl = 0; cl < nclusters; ++cl) cl_bodies[cl].reserve((size_t)homes[cl] + homes[cl] / 2 + 16);  // (room for the ghosts and kinematic copies that follow)
        plan.clustered_dynamic.reserve(plan.clustered_dynamic.size() + (size_t)total_dyn);
    }
    for (int i = 0; i < universe; ++i)
        if (is_dyn[i]) { const int cl = body_cluster[i]; local_of[i] = (int)cl_bodies[cl].size(); cl_bodies[cl].push_back(i | (shared[i] ? kSlotSharedHome : 0)); plan.clustered_dynamic.push_back(i); }
    std::vector<std::unordered_map<int32_t, int32_t>> cl_extra(nclusters);  // body | kind flag -> natural local index of ghosts and kinematic copies
    auto extra_local = [&](int cl, int tagged) {
        auto found = cl_extra[cl].find(tagged);
        if (found != cl_extra[cl].end()) return found->second;
        const int l = (int)cl_bodies[cl].size();
        cl_bodies[cl].push_back(tagged);
        cl_extra[cl].emplace(tagged, l);
        return l;
    };
    // pre-size: every cluster must fit the LDS budget with its ghosts, kinematic copies and items
    const bool reserve = (c->flags & BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS) != 0;
    int slot_reserve = 0;  // free LDS slots behind every cluster's bodies
    {
        // A copy gets its natural index when the cluster first meets it, type batches in order, constraints in order. The references that need a copy are listed per
        // type batch on the plan threads, strung together per cluster in that order, and every cluster then numbers its own (the hash tables are the expensive part).
        std::vector<int32_t> item_count(nclusters, 0);
        std::vector<std::vector<std::pair<int32_t, int32_t>>> wanted(all_pieces.size());  // (cluster, tagged body) in the order a serial scan meets them, piece by piece
        std::vector<std::vector<int32_t>> per_cluster(c->tbs.size()), piece_clusters(all_pieces.size());
        plan_parallel_for(all_pieces.size(), [&](size_t piece) {
            const size_t t = (size_t)all_pieces[piece].t;
            const HostTypeBatch& tb = c->tbs[t];
            std::vector<int32_t>& counts = piece_clusters[piece];
            counts.assign(nclusters, 0);
            for (int i = all_pieces[piece].begin; i < all_pieces[piece].end; ++i) {
                const int cl = cl_of_constraint[t][i];
                counts[cl]++;
                for (int k = 0; k < tb.info.bodies; ++k) {
                    const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                    if ((uint32_t)r >= kDynamicLimit) wanted[piece].push_back({cl, (r & kRefMask) | kSlotKinematic});
                    else if (body_cluster[r] != cl) wanted[piece].push_back({cl, r | kSlotGhost});
                }
            }
        });
        std::vector<std::vector<int32_t>> met(nclusters);
        for (size_t t = 0; t < c->tbs.size(); ++t) per_cluster[t].assign(nclusters, 0);
        for (size_t piece = 0; piece < all_pieces.size(); ++piece) {  // (type batches in order, a type batch's pieces in order)
            for (auto& pair : wanted[piece]) met[pair.first].push_back(pair.second);
            std::vector<int32_t>& sum = per_cluster[all_pieces[piece].t];
            for (int cl = 0; cl < nclusters; ++cl) sum[cl] += piece_clusters[piece][cl];
        }
        for (size_t t = 0; t < c->tbs.size(); ++t)
            for (int cl = 0; cl < nclusters; ++cl) item_count[cl] += (split_segment_slots(per_cluster[t][cl], reserve) + 63) / 64 + 1;
        plan_parallel_for((size_t)nclusters, [&](size_t cl) { for (int32_t tagged : met[cl]) extra_local((int)cl, tagged); });
        int max_slots = 0, max_items = 0;
        for (int cl = 0; cl < nclusters; ++cl) {
            max_slots = std::max(max_slots, ((int)cl_bodies[cl].size() + 15) / 16 * 16);
            max_items = std::max(max_items, item_count[cl]);
        }
        // Spare LDS slots for the ghost and kinematic copies structural updates may need (BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS): an eighth more (at least sixteen), if the workgroup's LDS has the room
        if (reserve) {
            const int wanted = std::min(0x3FF0, (max_slots + std::max(16, max_slots / 8) + 15) / 16 * 16);
            // (the two planes of local inertia give way to the reserve where both do not fit: they cost one read per body and substep, a plan that is lost costs the schedule)
            if (cluster_lds_bytes(kSweepPlanes, wanted, max_items, true) <= kLdsBudgetBytes) slot_reserve = wanted - max_slots;
        }
        max_slots += slot_reserve;
        plan.planes = cluster_lds_bytes(kAllPlanes, max_slots, max_items, true) <= kLdsBudgetBytes ? kAllPlanes : kSweepPlanes;
        if (max_slots >= 0x4000 || max_items >= 65536 || cluster_lds_bytes(plan.planes, max_slots, max_items, true) > kLdsBudgetBytes) {
            if (env_int("BEPUHIP_PLAN_STATS", 0))
                fprintf(stderr, "bepuhip split plan: declined, %d clusters (region %d) would need %d slots and %d items = %zu B of LDS per workgroup (budget %zu)\n", nclusters, region,
                        max_slots, max_items, cluster_lds_bytes(plan.planes, max_slots, max_items, true), kLdsBudgetBytes);
            return;
        }
    }
    split_lap("slots, ghost and kinematic copies, sizes");
    // (the pre-size pass above has met every (cluster, copy) pair: from here on the tables are only read, from several threads)
    auto extra_known = [&](int cl, int tagged) { return cl_extra[cl].find(tagged)->second; };
    auto slot_of = [&](int cl, int32_t r) {  // 32-bit local reference of body reference r as seen from cluster cl: slot | shared << 14 | kinematic << 30
        if ((uint32_t)r >= kDynamicLimit) return rotated_slot(extra_known(cl, (r & kRefMask) | kSlotKinematic)) | (int)kDynamicLimit;
        const int l = body_cluster[r] == cl ? local_of[r] : extra_known(cl, r | kSlotGhost);
        return rotated_slot(l) | (shared[r] ? (int)kLrefShared : 0);
    };
    std::vector<std::vector<int32_t>> last_toucher(nclusters);
    for (int cl = 0; cl < nclusters; ++cl) last_toucher[cl].assign((cl_bodies[cl].size() + 15) / 16 * 16 + slot_reserve + 16, -1);
    std::vector<std::vector<ClusterItem>> cl_items(nclusters);
    std::vector<std::vector<std::pair<int32_t, int32_t>>> first_touch(nclusters);
    std::vector<size_t> visit(c->tbs.size());
    for (size_t t = 0; t < visit.size(); ++t) visit[t] = t;
    std::stable_sort(visit.begin(), visit.end(), [&](size_t a, size_t b) {
        const HostTypeBatch &x = c->tbs[a], &y = c->tbs[b];
        if (x.batch != y.batch) return x.batch < y.batch;
        return x.info.prestep + 2 * x.info.impulse > y.info.prestep + 2 * y.info.impulse;
    });
    plan.split_visit = visit;
    // Every type batch's rows in their segmented order. Three steps so that one large type batch does not keep the other threads waiting: the order of every type
    // batch (a counting sort: cluster by cluster, private before shared, otherwise in the caller's order), the rows in chunks of slots, the swaps.
    struct RowJob { std::vector<int32_t> refs, lrefs; std::vector<uint32_t> ranks; std::vector<float> pre, acc; int stride = 0; std::vector<int32_t> private_live; };
    std::vector<RowJob> row_jobs(c->tbs.size());
    const bool host_values = c->host_values;
    plan_parallel_for(c->tbs.size(), [&](size_t t) {
        HostTypeBatch& tb = c->tbs[t];
        const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse;
        const std::vector<int32_t>& clc = cl_of_constraint[t];
        // inside a cluster: constraints that touch only private bodies first, the ones with shared bodies behind them. They share work items: a wave spends the
        // same time on an item whatever its lane count, and the waves' time is what a split cluster runs out of.
        std::vector<uint8_t> touches_shared(tb.count, 0);
        for (int k = 0; k < nb; ++k)
            for (int i = 0; i < tb.count; ++i) {
                const int32_t r = tb.refs_soa[(size_t)k * tb.stride + i];
                if ((uint32_t)r < kDynamicLimit && shared[r]) touches_shared[i] = 1;
            }
        // Every cluster's constraints of the type batch in one segment of device slots, live ones first (private before shared), free slots (if reserved) behind them —
        // the layout of the whole-island plans, so that structural updates find the same structures (bepu_soft_updates.h).
        std::vector<int32_t> live(nclusters, 0), private_live(nclusters, 0);
        for (int i = 0; i < tb.count; ++i) { ++live[clc[i]]; private_live[clc[i]] += !touches_shared[i]; }
        tb.seg_begin.assign(nclusters + 1, 0);
        for (int cl = 0; cl < nclusters; ++cl) tb.seg_begin[cl + 1] = tb.seg_begin[cl] + split_segment_slots(live[cl], reserve);
        tb.slots = tb.seg_begin[nclusters];
        RowJob& job = row_jobs[t];
        job.stride = std::max(tb.stride, (tb.slots + 63) / 64 * 64);
        tb.perm.assign(tb.slots, -1);
        {
            std::vector<int32_t> next_private(tb.seg_begin.begin(), tb.seg_begin.end() - 1), next_shared(nclusters);
            for (int cl = 0; cl < nclusters; ++cl) next_shared[cl] = tb.seg_begin[cl] + private_live[cl];
            for (int i = 0; i < tb.count; ++i) tb.perm[(touches_shared[i] ? next_shared : next_private)[clc[i]]++] = i;
        }
        tb.inv.assign(tb.count, 0);
        job.private_live = private_live;
        job.refs.assign((size_t)nb * job.stride, -1); job.lrefs.assign((size_t)nb * job.stride, kPlanDeadLref); job.ranks.assign((size_t)nb * job.stride, 0u);
        job.pre.assign(host_values ? (size_t)pf * job.stride : 0, 0.0f); job.acc.assign(host_values ? (size_t)imf * job.stride : 0, 0.0f);
    });
    split_lap("row order (threads)");
    constexpr int kRowChunk = 8192;
    std::vector<std::pair<int32_t, int32_t>> row_chunks;  // (type batch, first slot)
    for (size_t t = 0; t < c->tbs.size(); ++t) for (int d = 0; d < c->tbs[t].slots; d += kRowChunk) row_chunks.push_back({(int32_t)t, d});
    plan_parallel_for(row_chunks.size(), [&](size_t chunk) {
        const size_t t = (size_t)row_chunks[chunk].first;
        HostTypeBatch& tb = c->tbs[t];
        RowJob& job = row_jobs[t];
        const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse, stride = job.stride;
        const std::vector<int32_t>& clc = cl_of_constraint[t];
        for (int d = row_chunks[chunk].second; d < std::min(tb.slots, row_chunks[chunk].second + kRowChunk); ++d) {
            const int h = tb.perm[d];
            if (h < 0) continue;
            const int cl = clc[h];
            tb.inv[h] = d;
            for (int k = 0; k < nb; ++k) {
                const int32_t r = tb.refs_soa[(size_t)k * tb.stride + h];
                job.refs[(size_t)k * stride + d] = r;
                job.lrefs[(size_t)k * stride + d] = slot_of(cl, r);
                job.ranks[(size_t)k * stride + d] = srank[t][(size_t)k * tb.stride + h];
            }
            for (int f = 0; f < pf && host_values; ++f) job.pre[(size_t)f * stride + d] = tb.prestep_soa[(size_t)f * tb.stride + h];
            for (int f = 0; f < imf && host_values; ++f) job.acc[(size_t)f * stride + d] = tb.accum_soa[(size_t)f * tb.stride + h];
        }
    });
    split_lap("rows in chunks (threads)");
    plan_parallel_for(c->tbs.size(), [&](size_t t) {
        HostTypeBatch& tb = c->tbs[t];
        RowJob& job = row_jobs[t];
        tb.stride = job.stride;
        tb.dev_refs = job.refs;
        tb.refs_soa.swap(job.refs); tb.prestep_soa.swap(job.pre); tb.accum_soa.swap(job.acc); tb.lrefs_soa.swap(job.lrefs);
        srank[t].swap(job.ranks);
        tb.plan_lrefs = tb.lrefs_soa;  // 32-bit local references and rank words per device slot: what the predecessor rule reads (kept for the structural updates)
        tb.plan_ranks = srank[t];
    });
    split_lap("row swaps, mirrors (threads)");
    // Merged manifold items (round 5; bepu_cluster_kernel.h, run_cluster_fused): inside a batch, typed items of one convex manifold family (two-body Contact1..4,
    // or the one-body four) that do not fill a wave are grouped, lane counts summing to at most 64, at most four to a group — first fit over the items in
    // descending lane count. A group's items sit next to each other in the item array, leader first (shape bits 24-25: members behind it; bit 26: a member), at the
    // place of the group's first item; everything else keeps its claim order. BEPUHIP_FUSE_ITEMS=0 plans without groups (for A/Bs on one box).
    const bool fuse_items = env_int("BEPUHIP_FUSE_ITEMS", 1) != 0;
    const bool boundary_items = env_int("BEPUHIP_SPLIT_BOUNDARY_ITEMS", 0) != 0;
    struct ItemEntry { size_t t; int s0, count, fuse; };
    auto group_entries = [&](std::vector<ItemEntry>& entries) {
        std::vector<ItemEntry> out;
        out.reserve(entries.size());
        for (size_t b0 = 0; b0 < entries.size();) {
            size_t b1 = b0;
            while (b1 < entries.size() && c->tbs[entries[b1].t].batch == c->tbs[entries[b0].t].batch) ++b1;
            std::vector<int> group_of(b1 - b0, -1);  // entry -> group
            std::vector<std::vector<size_t>> groups;
            for (int family = 0; family < 2; ++family) {
                std::vector<size_t> partial;
                for (size_t e = b0; e < b1; ++e) {
                    const int type = c->tbs[entries[e].t].type_id;
                    if (type >= family * 4 && type < family * 4 + 4 && entries[e].count < 64) partial.push_back(e);
                }
                std::stable_sort(partial.begin(), partial.end(), [&](size_t x, size_t y) { return entries[x].count > entries[y].count; });
                std::vector<std::vector<size_t>> bins;
                std::vector<int> lanes;
                for (size_t e : partial) {
                    size_t at = 0;
                    while (at < bins.size() && (lanes[at] + entries[e].count > 64 || bins[at].size() >= 4)) ++at;
                    if (at == bins.size()) { bins.emplace_back(); lanes.push_back(0); }
                    bins[at].push_back(e); lanes[at] += entries[e].count;
                }
                for (auto& bin : bins) {
                    if (bin.size() < 2) continue;
                    std::sort(bin.begin(), bin.end());  // claim order inside the group: the heavier type leads
                    for (size_t e : bin) group_of[e - b0] = (int)groups.size();
                    groups.push_back(bin);
                }
            }
            for (size_t e = b0; e < b1; ++e) {
                const int g = group_of[e - b0];
                if (g < 0) { out.push_back(entries[e]); continue; }
                if (groups[g][0] != e) continue;  // emitted with its leader
                for (size_t m = 0; m < groups[g].size(); ++m) {
                    ItemEntry member = entries[groups[g][m]];
                    member.fuse = m == 0 ? (int)groups[g].size() - 1 : 4;
                    out.push_back(member);
                }
            }
            b0 = b1;
        }
        entries.swap(out);
    };
    plan_parallel_for((size_t)nclusters, [&](size_t cluster) {  // every cluster's work items with their predecessor lists, type batches in claim order
        const int cl = (int)cluster;
        std::vector<ItemEntry> entries;
        for (size_t t : visit) {
            const HostTypeBatch& tb = c->tbs[t];
            const int d = tb.seg_begin[cl], e = tb.seg_begin[cl + 1];
            // BEPUHIP_SPLIT_BOUNDARY_ITEMS=1 (an experiment of round 6): the segment's constraints on private bodies and the ones that touch a shared body in items of
            // their own — an item waits for the slowest of its lanes, and a lane behind a record that another cluster has yet to publish holds up the sixty-three
            // that only needed the LDS, and through them whatever comes next on THEIR bodies; more items per pass is what it costs.
            const int boundary = boundary_items ? std::min(e, d + row_jobs[t].private_live[cl]) : d;
            for (int s0 = d; s0 < boundary; s0 += 64) entries.push_back({t, s0, std::min(64, boundary - s0), 0});
            for (int s0 = boundary; s0 < e; s0 += 64) entries.push_back({t, s0, std::min(64, e - s0), 0});
        }
        if (fuse_items) group_entries(entries);
        for (const ItemEntry& entry : entries) {
            const size_t t = entry.t;
            HostTypeBatch& tb = c->tbs[t];
            const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse;
            {
                const int s0 = entry.s0;
                ClusterItem it;
                memset(&it, 0, sizeof(it));
                it.type_id = tb.type_id; it.count = entry.count; it.stride = tb.stride; it.start = s0;
                it.tb = (int)t; it.shape = nb | (pf << 8) | (imf << 16) | (entry.fuse << 24);
                const int self = (int)cl_items[cl].size();
                int npred = 0, overflow = 0;
                std::vector<int32_t>& lt = last_toucher[cl];
                // Shared bodies are ordered by the event numbers in their records, not by LDS flags — except where two consecutive applications run in this cluster
                // (the rank word's hand-off flags): there the later one waits for the earlier one's item like on a private body, under the body's slot without its flag bit.
                for (int j = s0; j < s0 + it.count; ++j)
                    for (int k = 0; k < nb; ++k) {
                        int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + j];
                        if ((uint32_t)lr >= kDynamicLimit || tb.perm[j] < 0) continue;  // kinematic copy, or a free slot
                        const bool is_shared = (lr & (int)kLrefShared) != 0;
                        if (is_shared && !(srank[t][(size_t)k * tb.stride + j] & kPlanRankPredLocal)) continue;
                        lr &= ~(int)kLrefShared;
                        const int pred = lt[lr];
                        if (pred < 0) { if (!is_shared) first_touch[cl].push_back({self, lr}); continue; }
                        if (pred == self) continue;
                        bool known = false;
                        for (int q = 0; q < npred; ++q) known |= it.pred[q] == pred;
                        if (known) continue;
                        if (npred < kMaxPreds) it.pred[npred++] = (unsigned short)pred; else overflow = 1;
                    }
                for (int j = s0; j < s0 + it.count; ++j)
                    for (int k = 0; k < nb; ++k) {
                        const int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + j];
                        if ((uint32_t)lr >= kDynamicLimit || tb.perm[j] < 0) continue;
                        if (!(lr & (int)kLrefShared)) lt[lr] = self;
                        else if (srank[t][(size_t)k * tb.stride + j] & kPlanRankSuccLocal) lt[lr & ~(int)kLrefShared] = self;
                    }
                if (overflow) npred = 0;
                it.batch_npred = (tb.batch & 0xFFFF) | (npred << 16) | (overflow << 24);
                cl_items[cl].push_back(it);
            }
        }
    });
    split_lap("work items, predecessor lists (threads)");
    // 16-bit local references (slot | shared << 14 | kinematic << 15), two per word, followed by the rank rows (one word per body slot)
    plan_parallel_for(c->tbs.size(), [&](size_t t) {
        HostTypeBatch& tb = c->tbs[t];
        const int nb = tb.info.bodies, rows = (nb + 1) / 2;
        std::vector<int32_t> packed((size_t)(rows + nb) * tb.stride, 0);
        for (int k = 0; k < nb; ++k)
            for (int d = 0; d < tb.slots; ++d) {
                const int32_t lr = tb.lrefs_soa[(size_t)k * tb.stride + d];
                const uint32_t half = ((uint32_t)lr & 0x7FFFu) | (((uint32_t)lr >= kDynamicLimit) ? 0x8000u : 0u);  // a free slot packs to kLrefDead
                packed[(size_t)(k / 2) * tb.stride + d] |= (int32_t)(half << (16 * (k & 1)));
                packed[(size_t)(rows + k) * tb.stride + d] = (int32_t)srank[t][(size_t)k * tb.stride + d];
            }
        if (nb == 1) for (int d = 0; d < tb.slots; ++d) if (tb.perm[d] < 0) packed[d] = (int32_t)kLrefDead;
        tb.lrefs_soa.swap(packed);
    });
    plan_parallel_for((size_t)nclusters, [&](size_t cluster) {
        const int cl = (int)cluster;
        for (auto& fs : first_touch[cl]) {
            ClusterItem& it = cl_items[cl][fs.first];
            const int last = last_toucher[cl][fs.second];
            int nx = (it.batch_npred >> 20) & 0xF;
            if ((it.batch_npred >> 25) & 1) continue;
            bool known = false;
            for (int q = 0; q < nx; ++q) known |= it.xpred[q] == last;
            if (known) continue;
            if (nx < kMaxPreds) { it.xpred[nx++] = (unsigned short)last; it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (nx << 20); }
            else it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (1 << 25);
        }
    });
    split_lap("packed local references, cross-pass lists");
    int64_t shared_count = 0, ghost_slots = 0;
    {
        size_t slots = 0, items = 0;
        for (int cl = 0; cl < nclusters; ++cl) { slots += (cl_bodies[cl].size() + 15) / 16 * 16 + (size_t)slot_reserve; items += cl_items[cl].size(); }
        plan.cluster_bodies.reserve(plan.cluster_bodies.size() + slots);
        plan.items.reserve(plan.items.size() + items);
        plan.batch_item_begin.reserve(plan.batch_item_begin.size() + (size_t)nclusters * (c->batch_count + 1));
        plan.clusters.reserve(plan.clusters.size() + nclusters);
    }
    for (int cl = 0; cl < nclusters; ++cl) {
        ClusterDesc d;
        d.body_begin = (int)plan.cluster_bodies.size();
        d.slot_count = ((int)cl_bodies[cl].size() + 15) / 16 * 16 + slot_reserve;  // the reserve: free slots (-1) structural updates turn into ghost / kinematic copies
        std::vector<int32_t> slots(d.slot_count, -1);
        for (size_t i = 0; i < cl_bodies[cl].size(); ++i) { slots[rotated_slot((int)i)] = cl_bodies[cl][i]; ghost_slots += (cl_bodies[cl][i] & kSlotGhost) != 0; }
        plan.cluster_bodies.insert(plan.cluster_bodies.end(), slots.begin(), slots.end());
        d.item_begin = (int)plan.items.size();
        d.item_count = (int)cl_items[cl].size();
        d.batch_item_offset = (int)plan.batch_item_begin.size();
        int k = 0;
        for (int b = 0; b <= c->batch_count; ++b) {
            while (k < d.item_count && (cl_items[cl][k].batch_npred & 0xFFFF) < b) ++k;
            plan.batch_item_begin.push_back(d.item_begin + k);
        }
        plan.items.insert(plan.items.end(), cl_items[cl].begin(), cl_items[cl].end());
        plan.clusters.push_back(d);
        plan.max_slots = std::max(plan.max_slots, d.slot_count);
        plan.max_items = std::max(plan.max_items, d.item_count);
    }
    plan.shared_info.assign(universe, 0u);
    for (int i = 0; i < universe; ++i) if (shared[i]) { plan.shared_info[i] = (uint32_t)deg[i]; ++shared_count; }
    plan.shared = true;
    // what structural updates need in order to stay on this plan (bepu_soft_updates.h, split part); the planner's own tables are handed over, not copied
    plan_parallel_ranges((size_t)universe, 32768, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) if (local_of[i] >= 0) local_of[i] = rotated_slot(local_of[i]); });
    plan.body_lref.swap(local_of);  // (-1 for every body that is not dynamic here)
    plan.body_cluster.swap(body_cluster);
    plan.split_shared.swap(shared);
    plan.split_degree.swap(deg);
    plan.cluster_natural.resize(nclusters);
    plan.cluster_extra.resize(nclusters);
    plan_parallel_for((size_t)nclusters, [&](size_t cl) {
        plan.cluster_natural[cl] = (int32_t)cl_bodies[cl].size();
        for (auto& kv : cl_extra[cl]) plan.cluster_extra[cl].emplace(kv.first, rotated_slot(kv.second));
    });
    split_lap("descriptors, mirrors");
    plan.planes = cluster_lds_bytes(kAllPlanes, plan.max_slots, plan.max_items, true) <= kLdsBudgetBytes ? kAllPlanes : kSweepPlanes;
    plan.enabled = nclusters > 0 && plan.max_slots < 0x4000 && cluster_lds_bytes(plan.planes, plan.max_slots, plan.max_items, true) <= kLdsBudgetBytes;
    if (env_int("BEPUHIP_PLAN_STATS", 0))
        fprintf(stderr, "bepuhip split plan: %d clusters (region %d), %lld dynamic bodies, %lld shared (%.1f %%), %lld ghost slots, max slots %d, max items %d, LDS %zu B, enabled %d\n", nclusters,
                region, (long long)total_dyn, (long long)shared_count, 100.0 * shared_count / std::max<int64_t>(total_dyn, 1), (long long)ghost_slots, plan.max_slots, plan.max_items,
                cluster_lds_bytes(plan.planes, plan.max_slots, plan.max_items, true), (int)plan.enabled);
}
