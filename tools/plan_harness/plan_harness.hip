// Offline harness of the host-side planner (a developer tool, CPU only, not part of the product): includes the library's translation unit, builds a context by hand
// (no HIP call is made: no device is needed), feeds it a scene dumped by dump_scene.py, runs the host phases of bepuhip_end_constraints — the AOSOA -> row conversion of
// set_type_batch and plan_clusters — and prints their timings, the plan's shape and a 64-bit digest of everything the plan hands to the device, so that a change of the
// planner's implementation can be shown to leave its output byte-identical without a GPU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../include -o plan_harness plan_harness.hip && ./plan_harness scene.bin
#include "../../bepuphysics2_amd/csrc/bepuhip.hip"

#include <cstdio>
#include <map>
#include <tuple>

// the cluster_kernel variants live in their own translation units; nothing is launched here
#define BEPU_STUB(name) const void* name(bool) { return nullptr; }
BEPU_STUB(bepu_cluster_kernel_hot_1024) BEPU_STUB(bepu_cluster_kernel_hot_512)
BEPU_STUB(bepu_cluster_kernel_wide_1024) BEPU_STUB(bepu_cluster_kernel_wide_512)
BEPU_STUB(bepu_cluster_kernel_hot_1024n) BEPU_STUB(bepu_cluster_kernel_wide_1024n) BEPU_STUB(bepu_cluster_kernel_hot_512sn) BEPU_STUB(bepu_cluster_kernel_wide_512sn)
BEPU_STUB(bepu_cluster_kernel_hot_1024s) BEPU_STUB(bepu_cluster_kernel_wide_1024s)
BEPU_STUB(bepu_cluster_kernel_hot_512s) BEPU_STUB(bepu_cluster_kernel_wide_512s) BEPU_STUB(bepu_cluster_kernel_hot_768s)
BEPU_STUB(bepu_cluster_kernel_hot_1024p) BEPU_STUB(bepu_cluster_kernel_wide_1024p) BEPU_STUB(bepu_cluster_kernel_hot_512sp) BEPU_STUB(bepu_cluster_kernel_wide_512sp)
BEPU_STUB(bepu_cluster_kernel_contacts_512s) BEPU_STUB(bepu_cluster_kernel_contacts_768s) BEPU_STUB(bepu_cluster_kernel_contacts_1024)
BEPU_STUB(bepu_cluster_kernel_hot_1024c) BEPU_STUB(bepu_cluster_kernel_wide_1024c) BEPU_STUB(bepu_cluster_kernel_hot_512sc) BEPU_STUB(bepu_cluster_kernel_wide_512sc)

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
template <class T> static uint64_t fnv_vec(uint64_t h, const std::vector<T>& v) { return v.empty() ? h : fnv(h, v.data(), v.size() * sizeof(T)); }

// A coarse timing model of one pass over a plan (PLAN_SIMULATE=1), to compare cuts offline: every cluster has WAVES waves that claim its items in order; a claimed item spends
// LOAD clocks on its row loads and velocity-independent work, then waits for its predecessors inside the cluster (LDS flags) and, per shared body, for the previous
// application on that body (rank order = batch order; HANDOFF clocks more when that ran in another cluster), then spends APPLY clocks and frees its wave. Every dependency
// points to a lower batch, so one sweep over the batches in order settles all times.
static void simulate(bepuhip_ctx* c, const ClusterPlan& plan) {
    auto knob = [](const char* name, double fallback) { const char* v = getenv(name); return v ? atof(v) : fallback; };
    const int waves = (int)knob("SIM_WAVES", 8);
    const double load = knob("SIM_LOAD", 6000), apply = knob("SIM_APPLY", 4000), handoff = knob("SIM_HANDOFF", 3300), passes = knob("SIM_PASSES", 12), ghz = knob("SIM_GHZ", 2.2);
    const size_t ncl = plan.clusters.size();
    std::vector<std::vector<double>> wave_free(ncl, std::vector<double>(waves, 0.0)), finish(ncl);
    std::vector<double> last_claim(ncl, 0.0);
    std::vector<size_t> cursor(ncl, 0);
    std::vector<double> body_finish(plan.shared_info.size(), 0.0);
    std::vector<int32_t> body_cluster(plan.shared_info.size(), -1);
    for (size_t cl = 0; cl < ncl; ++cl) finish[cl].assign(plan.clusters[cl].item_count, 0.0);
    double makespan = 0, waited = 0, waited_remote = 0;
    size_t items = 0;
    for (int batch = 0; batch < c->batch_count; ++batch)
        for (size_t cl = 0; cl < ncl; ++cl) {
            const ClusterDesc& cd = plan.clusters[cl];
            while (cursor[cl] < (size_t)cd.item_count && (plan.items[cd.item_begin + cursor[cl]].batch_npred & 0xFFFF) == batch) {
                const size_t k = cursor[cl]++;
                const ClusterItem& it = plan.items[cd.item_begin + k];
                const int fuse = (it.shape >> kItemFuseShift) & 7;
                if (fuse & kItemFuseMember) continue;  // run by its leader (merged manifold groups): no wave, no loads of its own
                const size_t group = 1 + (size_t)(fuse & 3);
                auto w = std::min_element(wave_free[cl].begin(), wave_free[cl].end());
                const double claim = std::max(*w, last_claim[cl]);
                last_claim[cl] = claim;
                double gate = claim + load, local_wait = gate;
                for (size_t m = 0; m < group; ++m) {
                    const ClusterItem& mi = plan.items[cd.item_begin + k + m];
                    const int npred = (mi.batch_npred >> 16) & 0xF;
                    if ((mi.batch_npred >> 24) & 1) { for (size_t q = 0; q < k; ++q) if ((plan.items[cd.item_begin + q].batch_npred & 0xFFFF) < batch) gate = std::max(gate, finish[cl][q]); }
                    else for (int q = 0; q < npred; ++q) gate = std::max(gate, finish[cl][mi.pred[q]]);
                }
                local_wait = gate;
                for (size_t m = 0; m < group; ++m) {
                    const ClusterItem& mi = plan.items[cd.item_begin + k + m];
                    const HostTypeBatch& tb = c->tbs[mi.tb];
                    for (int j = mi.start; j < mi.start + mi.count; ++j)
                        for (int b = 0; b < tb.info.bodies; ++b) {
                            const int32_t r = tb.refs_soa[(size_t)b * tb.stride + j];
                            if (r < 0 || (uint32_t)r >= kDynamicLimit || (size_t)r >= plan.shared_info.size() || plan.shared_info[r] == 0) continue;
                            if (body_cluster[r] >= 0) gate = std::max(gate, body_finish[r] + (body_cluster[r] != (int)cl ? handoff : 0.0));
                        }
                }
                waited += gate - (claim + load);
                waited_remote += gate - local_wait;
                const double done = gate + apply;
                for (size_t m = 0; m < group; ++m) {
                    const ClusterItem& mi = plan.items[cd.item_begin + k + m];
                    const HostTypeBatch& tb = c->tbs[mi.tb];
                    for (int j = mi.start; j < mi.start + mi.count; ++j)
                        for (int b = 0; b < tb.info.bodies; ++b) {
                            const int32_t r = tb.refs_soa[(size_t)b * tb.stride + j];
                            if (r < 0 || (uint32_t)r >= kDynamicLimit || (size_t)r >= plan.shared_info.size() || plan.shared_info[r] == 0) continue;
                            body_finish[r] = done; body_cluster[r] = (int)cl;
                        }
                    finish[cl][k + m] = done;
                }
                *w = done;
                makespan = std::max(makespan, done);
                ++items;
            }
        }
    printf("model: one pass %.0f clocks, x %.0f passes at %.1f GHz = %.3f ms | mean wait per item %.0f clocks, of which for another cluster %.0f\n", makespan, passes, ghz,
           makespan * passes / (ghz * 1e6), waited / items, waited_remote / items);
}

// PLAN_VALIDATE=1: invariants the device relies on, checked from the plan's OUTPUT alone (the rows, the packed local references, the rank rows, the slot tables, the items).
// Returns the number of violations (printed, first few).
static int validate(bepuhip_ctx* c, const ClusterPlan& plan) {
    int bad = 0;
    auto fail = [&](const char* what, long a, long b, long d) { if (bad++ < 12) fprintf(stderr, "plan violation: %s (%ld, %ld, %ld)\n", what, a, b, d); };
    const size_t universe = plan.shared ? plan.shared_info.size() : (size_t)c->referenced_bodies;
    // every live row of every type batch is covered by exactly one item, and that item belongs to the row's batch
    std::vector<std::vector<uint8_t>> covered(c->tbs.size());
    for (size_t t = 0; t < c->tbs.size(); ++t) covered[t].assign((size_t)std::max(c->tbs[t].slots, c->tbs[t].count), 0);
    struct Application { int batch, cluster, rank, degree; bool pred_local, succ_local; };
    std::vector<std::vector<Application>> applications(plan.shared ? universe : 0);
    for (size_t cl = 0; cl < plan.clusters.size(); ++cl) {
        const ClusterDesc& cd = plan.clusters[cl];
        const int32_t* slots = plan.cluster_bodies.data() + cd.body_begin;
        int previous_batch = -1;
        std::vector<int> last_batch_of_slot(cd.slot_count, -1);
        int members_due = 0, group_lanes = 0, group_family = -1, group_batch = -1;  // merged manifold groups (kItemFuseShift): a leader, then exactly its members
        for (int k = 0; k < cd.item_count; ++k) {
            const ClusterItem& it = plan.items[cd.item_begin + k];
            const HostTypeBatch& tb = c->tbs[it.tb];
            const int batch = it.batch_npred & 0xFFFF, nb = tb.info.bodies, rows = (nb + 1) / 2;
            {
                const int fuse = (it.shape >> kItemFuseShift) & 7;
                const bool member = (fuse & kItemFuseMember) != 0;
                if (member != (members_due > 0)) fail("group member without a leader, or a leader short of members", (long)cl, k, fuse);
                if (member) {
                    --members_due; group_lanes += it.count;
                    if (fuse & 3) fail("a member that leads", (long)cl, k, fuse);
                    if (it.type_id > 7 || it.type_id / 4 != group_family || batch != group_batch) fail("group member of another family or batch", (long)cl, k, it.type_id);
                    if (group_lanes > 64) fail("a group with more than 64 lanes", (long)cl, k, group_lanes);
                } else if (fuse & 3) {
                    if (!plan.shared) fail("a group outside a split plan", (long)cl, k, fuse);
                    if (it.type_id > 7) fail("a group led by a type that is no convex manifold", (long)cl, k, it.type_id);
                    members_due = fuse & 3; group_lanes = it.count; group_family = it.type_id / 4; group_batch = batch;
                }
                if (k + 1 == cd.item_count && members_due > 0) fail("a group runs past the cluster's items", (long)cl, k, members_due);
            }
            if (batch < previous_batch) fail("items out of batch order", (long)cl, k, batch);
            previous_batch = batch;
            if (batch != tb.batch || it.type_id != tb.type_id || it.stride != tb.stride || it.count < 1 || it.count > 64) fail("item header", (long)cl, k, it.tb);
            const int npred = (it.batch_npred >> 16) & 0xF, nxpred = (it.batch_npred >> 20) & 0xF;
            for (int q = 0; q < npred; ++q)
                if (it.pred[q] >= k || (plan.items[cd.item_begin + it.pred[q]].batch_npred & 0xFFFF) >= batch) fail("predecessor is not an earlier batch's item", (long)cl, k, it.pred[q]);
            for (int q = 0; q < nxpred; ++q) if (it.xpred[q] >= cd.item_count) fail("cross-pass predecessor out of range", (long)cl, k, it.xpred[q]);
            for (int j = it.start; j < it.start + it.count; ++j) {
                if ((size_t)j >= covered[it.tb].size()) { fail("item row out of range", (long)cl, k, j); continue; }
                if (covered[it.tb][j]++) fail("row covered twice", it.tb, j, 0);
                const bool live = tb.perm.empty() || (j < (int)tb.perm.size() && tb.perm[j] >= 0);
                for (int b = 0; b < nb; ++b) {
                    const uint32_t word = (uint32_t)tb.lrefs_soa[(size_t)(b / 2) * tb.stride + j];
                    const uint32_t half = (word >> (16 * (b & 1))) & 0xFFFFu;
                    const int32_t r = tb.refs_soa[(size_t)b * tb.stride + j];
                    if (!live) { if (half != 0x8000u) fail("free slot is not the dead reference", it.tb, j, (long)half); continue; }
                    const int slot = (int)(half & 0x3FFFu);
                    const bool kinematic = (half & 0x8000u) != 0, shared_ref = (half & kLrefShared) != 0;
                    if (slot >= cd.slot_count) { fail("local reference beyond the cluster's slots", (long)cl, k, slot); continue; }
                    const int32_t entry = slots[slot];
                    if (entry < 0 || (entry & kSlotBodyMask) != (r & kRefMask)) fail("slot table and global reference disagree", (long)cl, slot, r & kRefMask);
                    if (kinematic != ((uint32_t)r >= kDynamicLimit) || kinematic != ((entry & kSlotKinematic) != 0)) fail("kinematic marking", (long)cl, slot, r);
                    if (kinematic) continue;
                    if (last_batch_of_slot[slot] == batch) fail("a body twice in one batch of a cluster", (long)cl, slot, batch);
                    last_batch_of_slot[slot] = batch;
                    if (plan.shared) {
                        const bool is_shared = plan.shared_info[r] != 0;
                        if (shared_ref != is_shared) fail("shared bit of a local reference", (long)cl, r, (long)half);
                        if (((entry & kSlotGhost) != 0) != (is_shared && true && !(entry & kSlotSharedHome)) && is_shared) fail("ghost / home marking of a shared body's slot", (long)cl, r, entry);
                        if (!is_shared && (entry & (kSlotGhost | kSlotSharedHome))) fail("private body marked ghost or shared home", (long)cl, r, entry);
                        if (is_shared) {
                            const uint32_t rank_word = (uint32_t)tb.lrefs_soa[(size_t)(rows + b) * tb.stride + j];
                            applications[r].push_back({batch, (int)cl, (int)(rank_word & 0xFFu), (int)((rank_word >> 8) & 0xFFu), (rank_word & (1u << 16)) != 0, (rank_word & (1u << 17)) != 0});
                        }
                    }
                }
            }
        }
    }
    for (size_t t = 0; t < c->tbs.size(); ++t) {
        const HostTypeBatch& tb = c->tbs[t];
        for (int j = 0; j < (int)covered[t].size(); ++j) {
            const bool live = tb.perm.empty() ? j < tb.count : (j < (int)tb.perm.size() && tb.perm[j] >= 0);
            if (live && !covered[t][j]) fail("live row without an item", (long)t, j, 0);
        }
    }
    if (plan.shared) {
        std::vector<int> homes(universe, 0);
        for (size_t cl = 0; cl < plan.clusters.size(); ++cl) {
            const ClusterDesc& cd = plan.clusters[cl];
            for (int s = 0; s < cd.slot_count; ++s) {
                const int32_t entry = plan.cluster_bodies[cd.body_begin + s];
                if (entry >= 0 && !(entry & (kSlotKinematic | kSlotGhost))) ++homes[entry & kSlotBodyMask];
            }
        }
        for (size_t r = 0; r < universe; ++r) {
            auto& apps = applications[r];
            if (plan.shared_info[r] == 0) { if (!apps.empty()) fail("applications recorded for a private body", (long)r, 0, 0); continue; }
            if (homes[r] != 1) fail("a shared body needs exactly one home", (long)r, homes[r], 0);
            if (apps.size() != plan.shared_info[r]) fail("degree of a shared body", (long)r, (long)apps.size(), (long)plan.shared_info[r]);
            std::sort(apps.begin(), apps.end(), [](const Application& x, const Application& y) { return x.rank < y.rank; });
            for (size_t q = 0; q < apps.size(); ++q) {
                if (apps[q].rank != (int)q || apps[q].degree != (int)apps.size()) fail("rank / degree of an application", (long)r, apps[q].rank, apps[q].degree);
                if (q > 0 && apps[q].batch <= apps[q - 1].batch) fail("ranks are not in batch order", (long)r, apps[q - 1].batch, apps[q].batch);
                // the hand-off flags: set in pairs, only between consecutive applications of ONE cluster, never on the first / last application of a pass
                const bool pair = q > 0 && apps[q].pred_local;
                if (pair && (apps[q - 1].cluster != apps[q].cluster || !apps[q - 1].succ_local)) fail("a local hand-off whose two ends disagree", (long)r, (long)q, apps[q].cluster);
                if (apps[q].succ_local && (q + 1 >= apps.size() || !apps[q + 1].pred_local)) fail("a local hand-off nobody receives", (long)r, (long)q, apps[q].cluster);
                if (q == 0 && apps[q].pred_local) fail("the first application of a pass must poll the record", (long)r, 0, 0);
            }
        }
    }
    printf("validate: %d violation(s)\n", bad);
    return bad;
}

// PLAN_CHURN=frames: what the narrow phase does to a plan, without a device. Every frame a hundredth of every two-body type batch is removed (every hundredth constraint, another
// residue every frame) and as many constraints come back: most between the same bodies (a persisting pair's refresh), every PLAN_CHURN_NEW-th between the first
// body and a NEAR one — the nearest body index the batch does not hold yet (PLAN_CHURN_FAR=1: second bodies exchanged among the removed ones instead, pairs from all over
// the scene, which no narrow phase produces). The structural updates run on the plan's host mirrors (bepu_soft_updates.h); after every frame (1) a device image — the rows as uploaded,
// then only what flush_soft would write: whole slots and single words — must equal the mirrors, and (2) the mirrors must still be a valid plan (validate above).
static int churn(bepuhip_ctx* c, ClusterPlan& plan, int frames) {
    const int every = getenv("PLAN_CHURN_NEW") ? atoi(getenv("PLAN_CHURN_NEW")) : 5;
    const bool far = getenv("PLAN_CHURN_FAR") != nullptr;
    std::vector<int32_t> cluster_bodies_image = plan.cluster_bodies;
    std::vector<unsigned> shared_info = plan.shared_info;
    const bool shared_plan = plan.shared;
    size_t offset = 0;
    constexpr size_t kRefsBase = (size_t)1 << 40;  // (the references' words get addresses of their own: patches of body moves name them)
    for (auto& tb : c->tbs) { tb.lrefs_off = offset; tb.refs_off = kRefsBase + offset; offset += std::max(tb.lrefs_soa.size(), tb.refs_soa.size()); }  // the word patches name slab offsets
    const bool body_events = getenv("PLAN_CHURN_BODIES") != nullptr;
    long removals = 0, moves = 0, adoptions = 0;
    std::vector<std::vector<int32_t>> refs_image(c->tbs.size()), lrefs_image(c->tbs.size());
    for (size_t t = 0; t < c->tbs.size(); ++t) { refs_image[t] = c->tbs[t].refs_soa; lrefs_image[t] = c->tbs[t].lrefs_soa; }
    c->shared_bodies = plan.shared_info.size() + 1024;  // what build_constraints sets up on a device
    shared_info.resize(c->shared_bodies, 0u);
    c->clustered_dynamic_host = plan.clustered_dynamic;
    c->clustered_dynamic_capacity = (int)plan.clustered_dynamic.size() + 256;
    soft_setup(c, plan);
    if (!c->soft_ok) { printf("churn: the plan takes no structural updates\n"); return 0; }
    // (a whole-island plan keeps no host mirror of its local references: its image is built from what the slot writes carry and checked by the validator alone)
    long calls = 0;
    double calls_ms = 0.0, flush_ms = 0.0, library_ms = 0.0, resolve_ms = 0.0;  // library_ms: inside soft_remove / soft_add alone (the generator's own searches left out)
    auto timed = [&](auto&& call) { const auto a = std::chrono::steady_clock::now(); const bool ok = call(); library_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); return ok; };
    for (int frame = 0; frame < frames; ++frame) {
        const auto frame_begin = std::chrono::steady_clock::now();
        for (size_t t = 0; t < c->tbs.size(); ++t) {
            HostTypeBatch* tb = &c->tbs[t];
            if (tb->info.bodies != 2 || tb->count < 100) continue;
            const int n = tb->count / 100;
            const int stride = tb->count / n, first = (frame * 37) % stride;  // every stride-th constraint, another residue every frame: the churn is spread over the scene
            std::vector<std::array<int32_t, 2>> lanes;
            for (int i = 0; i < n; ++i) { const int d = tb->inv[first + i * stride]; lanes.push_back({tb->dev_refs[d], tb->dev_refs[(size_t)tb->stride + d]}); }
            const bool prefetching = getenv("PLAN_CHURN_NO_PREFETCH") == nullptr;  // what bepuhip_apply_structural_ops does with a table of operations
            for (int i = n - 1; i >= 0; --i, ++calls) if (!timed([&] {
                    if (prefetching) {
                        if (i >= 12) soft_prefetch_remove(c, tb, first + (i - 12) * stride, 0);
                        if (i >= 8) soft_prefetch_remove(c, tb, first + (i - 8) * stride, 1);
                        if (i >= 5) soft_prefetch_remove(c, tb, first + (i - 5) * stride, 2);
                        if (i >= 2) soft_prefetch_remove(c, tb, first + (i - 2) * stride, 3);
                    }
                    return soft_remove(c, tb, first + i * stride); })) { printf("churn: frame %d, removal refused\n", frame); return 0; }
            std::vector<int> fresh;
            for (int i = 0; i < n; i += every) fresh.push_back(i);
            std::vector<float> prestep(tb->info.prestep, 0.25f);
            for (int i = 0; i < n; ++i, ++calls) {
                int32_t refs[2] = {lanes[i][0], lanes[i][1]};
                if (i % every == 0 && far) refs[1] = lanes[fresh[(i / every + 1) % fresh.size()]][1];
                else if (i % every == 0 && (uint32_t)refs[0] < kDynamicLimit && (uint32_t)refs[1] < kDynamicLimit) {
                    for (int step = 1; step <= 16; ++step) {
                        const int32_t near = refs[0] + ((step & 1) ? (step + 1) / 2 : -(step / 2));
                        if (near < 0 || near >= (int32_t)c->body_cluster.size() || c->body_cluster[near] < 0 || near == refs[1] || (tb->batch < 64 && (c->body_batches[near] >> tb->batch) & 1)) continue;
                        if (!shared_plan && c->body_cluster[near] != c->body_cluster[refs[0]]) continue;  // (a whole-island plan keeps islands apart: a pair across two clusters leaves it)
                        bool taken = false;  // ... by a lane of this window that is still to come back
                        for (int q = i + 1; q < n && !taken; ++q) taken = lanes[q][0] == near || lanes[q][1] == near;
                        if (!taken) { refs[1] = near; break; }
                    }
                }
                bool violation = false;
                if (!timed([&] {
                        if (prefetching) {  // (the harness decides a lane's second body late: it prefetches for the pair as removed, which most additions are)
                            if (i + 8 < n) { const int32_t ahead[2] = {lanes[i + 8][0], lanes[i + 8][1]}; soft_prefetch_add(c, tb, ahead, 0); }
                            if (i + 4 < n) { const int32_t ahead[2] = {lanes[i + 4][0], lanes[i + 4][1]}; soft_prefetch_add(c, tb, ahead, 1); }
                        }
                        return soft_add(c, tb, refs, prestep.data(), &violation); })) {
                    for (int k = 0; k < 2; ++k) {
                        if ((uint32_t)refs[k] >= kDynamicLimit) continue;
                        const int home = c->body_cluster[refs[k]];
                        int live = 0, total = tb->seg_begin[home + 1] - tb->seg_begin[home];
                        for (int q = tb->seg_begin[home]; q < tb->seg_begin[home + 1]; ++q) live += tb->perm[q] >= 0;
                        fprintf(stderr, "churn: refused lane %d of %d (was %d %d): body %d of tb %zu (batch %d type %d count %d) home %d, its segment %d live of %d\n", i, n, lanes[i][0], lanes[i][1], refs[k], t, tb->batch, tb->type_id, tb->count, home, live, total);
                    } printf("churn: frame %d, addition refused (%s): the context would leave the plan here\n", frame, violation ? "batch invariant" : "no room"); return 0; }
            }
        }
        if (body_events) {
            // PLAN_CHURN_BODIES: Bodies.RemoveAt and Bodies.Add as the plan sees them. (1) a body loses all its constraints (it leaves the plan at the flush — or right away
            // when (2) needs its index); (2) the highest body of the plan takes its index: every reference to it is patched (TypeProcessor.UpdateForBodyMemoryMove);
            // (3) the index that became free gets a constraint with a body of the plan: it joins that body's cluster.
            const int universe = (int)c->body_cluster.size();
            auto references_of = [&](int32_t body, auto&& fn) {  // fn(type batch, device slot, body slot) for every live reference to `body`
                for (size_t t = 0; t < c->tbs.size(); ++t) {
                    HostTypeBatch& tb = c->tbs[t];
                    for (int k = 0; k < tb.info.bodies; ++k)
                        for (int d = 0; d < tb.slots; ++d)
                            if (tb.perm[d] >= 0 && tb.dev_refs[(size_t)k * tb.stride + d] >= 0 && (tb.dev_refs[(size_t)k * tb.stride + d] & kRefMask) == body && (uint32_t)tb.dev_refs[(size_t)k * tb.stride + d] < kDynamicLimit)
                                if (!fn(&tb, d, k)) return;
                }
            };
            int victim = -1;
            for (int probe = 0; probe < universe && victim < 0; ++probe) {
                const int v = (frame * 131 + 7 + probe * 17) % universe;
                if (c->body_cluster[v] < 0) continue;
                int degree = 0; bool two_body_only = true;
                references_of(v, [&](HostTypeBatch* tb, int, int) { ++degree; two_body_only &= tb->info.bodies <= 2; return true; });
                if (degree > 0 && degree <= 8 && two_body_only) victim = v;
            }
            if (victim >= 0) {
                for (;;) {  // every removal renumbers its type batch: find the next reference afresh
                    HostTypeBatch* found = nullptr; int index = -1;
                    references_of(victim, [&](HostTypeBatch* tb, int d, int) { found = tb; index = tb->perm[d]; return false; });
                    if (!found) break;
                    if (!soft_remove(c, found, index)) { printf("churn: frame %d, removal refused\n", frame); return 0; }
                    ++calls; ++removals;
                }
                int last = -1;
                for (int b = universe - 1; b >= 0 && last < 0; --b) if (b != victim && c->body_cluster[b] >= 0) last = b;
                bool moved = last > victim;
                if (moved) {
                    std::vector<std::array<int64_t, 3>> patches;
                    references_of(last, [&](HostTypeBatch* tb, int d, int k) { patches.push_back({(int64_t)(tb - c->tbs.data()), tb->perm[d], k}); return true; });
                    for (auto& p : patches) {
                        if (!soft_update_reference(c, &c->tbs[p[0]], (int)p[1], (int)p[2], victim)) { printf("churn: frame %d, body move refused: the context would leave the plan here\n", frame); return 0; }
                        ++calls;
                    }
                    ++moves;
                    // (3) `last` is a free index now: a constraint with a body of the plan brings it back in
                    for (size_t t = 0; t < c->tbs.size() && moved; ++t) {
                        HostTypeBatch* tb = &c->tbs[t];
                        if (tb->info.bodies != 2 || tb->count == 0 || tb->batch >= 64) continue;
                        for (int partner = victim; partner < universe; ++partner) {
                            if (partner == last || c->body_cluster[partner] < 0 || ((c->body_batches[partner] >> tb->batch) & 1)) continue;
                            if (!shared_plan) {  // a free device slot in the partner's cluster, or the addition (rightly) leaves the plan
                                bool room = false;
                                for (int q = tb->seg_begin[c->body_cluster[partner]]; q < tb->seg_begin[c->body_cluster[partner] + 1]; ++q) room |= tb->perm[q] < 0;
                                if (!room) continue;
                            }
                            int32_t refs[2] = {last, partner};
                            std::vector<float> prestep(tb->info.prestep, 0.5f);
                            bool violation = false;
                            if (soft_add(c, tb, refs, prestep.data(), &violation)) { ++calls; ++adoptions; moved = false; }
                            break;
                        }
                    }
                }
            }
        }
        if (!soft_bodies_still_constrained(c)) { printf("churn: frame %d, a body lost its last constraint\n", frame); return 0; }
        const auto f0 = std::chrono::steady_clock::now();
        calls_ms += std::chrono::duration<double, std::milli>(f0 - frame_begin).count();
        flush_soft_host(c);
        const auto f1 = std::chrono::steady_clock::now();
        std::vector<ResolvedWord> resolved_for_timing = split_resolve_patches(c);  // (the listing half of the flush, timed with it)
        flush_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count();
        resolve_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f1).count();
        if (frame < 3) fprintf(stderr, "churn: frame %d, the host half of the flush %.3f ms\n", frame, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count());
        // what flush_soft would send, applied to the image
        for (const auto& record : c->soft_records) {
            const HostTypeBatch& tb = c->tbs[record.tb];
            const int d = record.slot, nb = tb.info.bodies, rows = (nb + 1) / 2;
            const uint32_t* words = c->soft_payload.data() + record.payload_at;
            if (!shared_plan) {  // payload of a live slot: references, the packed local references, the prestep lane
                for (int k = 0; k < nb; ++k) refs_image[record.tb][(size_t)k * tb.stride + d] = record.live ? (int32_t)words[k] : -1;
                lrefs_image[record.tb][d] = record.live ? (int32_t)words[nb] : (int32_t)kLrefDead;
                continue;
            }
            for (int k = 0; k < nb; ++k) refs_image[record.tb][(size_t)k * tb.stride + d] = record.live ? tb.dev_refs[(size_t)k * tb.stride + d] : -1;
            for (int r = 0; r < rows; ++r) lrefs_image[record.tb][(size_t)r * tb.stride + d] = record.live ? (int32_t)split_packed_lrefs(tb, d, r) : (int32_t)kLrefDead;
            if (record.live) for (int k = 0; k < nb; ++k) lrefs_image[record.tb][(size_t)(rows + k) * tb.stride + d] = (int32_t)tb.plan_ranks[(size_t)k * tb.stride + d];
        }
        for (auto& word : split_resolve_patches(c)) {
            if (word.table == 1) { shared_info[word.index] = word.value; continue; }
            if (word.table == 2) { cluster_bodies_image[word.index] = (int32_t)word.value; continue; }
            if (word.index >= kRefsBase) {  // a body reference
                size_t t = 0;
                while (t + 1 < c->tbs.size() && c->tbs[t + 1].refs_off <= word.index) ++t;
                refs_image[t][word.index - c->tbs[t].refs_off] = (int32_t)word.value;
                continue;
            }
            size_t t = 0;
            while (t + 1 < c->tbs.size() && c->tbs[t + 1].lrefs_off <= word.index) ++t;
            lrefs_image[t][word.index - c->tbs[t].lrefs_off] = (int32_t)word.value;
        }
        c->split_patches.clear(); soft_clear_notes(c); c->soft_items_dirty = false;
        // (1) image == mirrors
        int differences = 0;
        if (cluster_bodies_image != c->cluster_bodies_host) { ++differences; fprintf(stderr, "churn: frame %d, the slot table's image differs from its mirror\n", frame); }
        for (size_t t = 0; t < c->tbs.size(); ++t) {
            const HostTypeBatch& tb = c->tbs[t];
            const int nb = tb.info.bodies, rows = (nb + 1) / 2;
            for (int d = 0; d < tb.slots; ++d) {
                const bool live = tb.perm[d] >= 0;
                for (int k = 0; k < nb; ++k) {
                    if (refs_image[t][(size_t)k * tb.stride + d] != tb.dev_refs[(size_t)k * tb.stride + d] && differences++ < 8) fprintf(stderr, "churn: frame %d, reference image != mirror (tb %zu slot %d body %d)\n", frame, t, d, k);
                    if (live && shared_plan && (uint32_t)lrefs_image[t][(size_t)(rows + k) * tb.stride + d] != tb.plan_ranks[(size_t)k * tb.stride + d] && differences++ < 8)
                        fprintf(stderr, "churn: frame %d, rank word image %x != mirror %x (tb %zu slot %d body %d)\n", frame, lrefs_image[t][(size_t)(rows + k) * tb.stride + d], tb.plan_ranks[(size_t)k * tb.stride + d], t, d, k);
                }
                for (int r = 0; r < rows && shared_plan; ++r)
                    if ((((uint32_t)lrefs_image[t][(size_t)r * tb.stride + d] ^ split_packed_lrefs(tb, d, r)) & ((nb & 1) && r == rows - 1 ? 0xFFFFu : 0xFFFFFFFFu)) != 0 && differences++ < 8)
                        fprintf(stderr, "churn: frame %d, local reference image %x != mirror %x (tb %zu slot %d row %d, live %d)\n", frame, lrefs_image[t][(size_t)r * tb.stride + d], split_packed_lrefs(tb, d, r), t, d, r, (int)live);
            }
        }
        if (differences) { printf("churn: frame %d, %d difference(s) between the device image and the mirrors\n", frame, differences); return 4; }
        // (2) the image is a valid plan
        ClusterPlan now;
        now.enabled = true; now.shared = shared_plan; now.items = c->items_host; now.clusters = c->clusters_host; now.cluster_bodies = cluster_bodies_image; now.shared_info = shared_info;
        for (size_t t = 0; t < c->tbs.size(); ++t) { c->tbs[t].refs_soa = refs_image[t]; c->tbs[t].lrefs_soa = lrefs_image[t]; }
        if (validate(c, now) != 0) { printf("churn: frame %d leaves an invalid plan\n", frame); return 3; }
    }
    size_t live_clusters = c->clusters_host.size();
    printf("churn: host time per frame: the structural calls (with this harness's generator) %.2f ms, of which inside soft_remove / soft_add %.2f ms; ranks + predecessor lists + word list %.2f ms (the word list alone %.2f)\n", calls_ms / frames, library_ms / frames, flush_ms / frames, resolve_ms / frames);
    if (body_events) printf("churn: bodies: %ld constraint removals of bodies that left, %ld bodies moved to another index, %ld bodies joined\n", removals, moves, adoptions);
    printf("churn: %d frames, %ld structural calls, still on the plan (%zu clusters)\n", frames, calls, live_clusters);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: plan_harness scene.bin [repeats]\n"); return 2; }
    const int repeats = argc > 2 ? atoi(argv[2]) : 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("scene"); return 2; }
    int32_t header[4];  // bundle width, batch count, type batch count, fallback threshold
    if (fread(header, 4, 4, f) != 4) return 2;
    struct Tb { int32_t batch, type, count; std::vector<int32_t> refs; std::vector<float> prestep, accum; };
    std::vector<Tb> tbs(header[2]);
    for (auto& t : tbs) {
        int32_t h[6];  // batch, type, count, refs words, prestep words, accum words
        if (fread(h, 4, 6, f) != 6) return 2;
        t.batch = h[0]; t.type = h[1]; t.count = h[2];
        t.refs.resize(h[3]); t.prestep.resize(h[4]); t.accum.resize(h[5]);
        if (fread(t.refs.data(), 4, h[3], f) != (size_t)h[3] || fread(t.prestep.data(), 4, h[4], f) != (size_t)h[4] || fread(t.accum.data(), 4, h[5], f) != (size_t)h[5]) return 2;
    }
    fclose(f);
    // PLAN_THREADS=N (round 6, the sanitizer runs: hipcc -fsanitize=thread / address -fno-gpu-sanitize): N host threads run the repeats at the same time, each on contexts of
    // its own — two contexts planning at once contend for the parked plan pool exactly as two uploading threads of one process do (the second caller gets threads of its own).
    const int harness_threads = getenv("PLAN_THREADS") ? std::max(1, atoi(getenv("PLAN_THREADS"))) : 1;
    std::atomic<int> verdict_all{0};
    std::mutex print_lock;
    auto run_repeats = [&](int harness_thread) -> int {
    for (int rep = 0; rep < repeats; ++rep) {
        bepuhip_ctx* c = new bepuhip_ctx();
        c->device = 0; c->W = header[0]; c->flags = argc > 3 ? atoi(argv[3]) : 0; c->host_values = true;  // no device here: the values are converted (and permuted by the plan) on the host, as the digest expects
        c->batch_count = header[1]; c->fallback_threshold = header[3]; c->has_fallback = header[1] > header[3]; c->building = true;
        auto t0 = std::chrono::steady_clock::now();
        for (auto& t : tbs)
            if (bepuhip_set_type_batch(c, t.batch, t.type, t.count, t.refs.data(), t.prestep.data(), t.accum.data()) != BEPUHIP_OK) { fprintf(stderr, "set_type_batch: %s\n", bepuhip_last_error()); return 1; }
        if (getenv("PLAN_NO_VALUES")) {  // time the plan as the product runs it: prestep data and impulses stay on the device, only references are permuted here
            for (auto& tb : c->tbs) { std::vector<float>().swap(tb.prestep_soa); std::vector<float>().swap(tb.accum_soa); }
            c->host_values = false;
        }
        auto t1 = std::chrono::steady_clock::now();
        int universe = 0;
        for (auto& tb : c->tbs)
            for (int32_t r : tb.refs_soa)
                if (r >= 0) universe = std::max(universe, (r & kRefMask) + 1);
        c->referenced_bodies = universe;
        c->total_constraints = 0;
        for (auto& tb : c->tbs) c->total_constraints += tb.count;
        c->body_count = universe;
        ClusterPlan plan;
        plan_clusters(c, plan);
        auto t2 = std::chrono::steady_clock::now();
        uint64_t h = 1469598103934665603ull;
        h = fnv_vec(h, plan.items); h = fnv_vec(h, plan.clusters); h = fnv_vec(h, plan.batch_item_begin); h = fnv_vec(h, plan.cluster_bodies); h = fnv_vec(h, plan.clustered_dynamic);
        h = fnv_vec(h, plan.kinlist); h = fnv_vec(h, plan.shared_info);
        for (auto& tb : c->tbs) {
            h = fnv_vec(h, tb.refs_soa); h = fnv_vec(h, tb.prestep_soa); h = fnv_vec(h, tb.accum_soa); h = fnv_vec(h, tb.lrefs_soa); h = fnv_vec(h, tb.perm); h = fnv_vec(h, tb.seg_begin);
            h = fnv(h, &tb.stride, 4); h = fnv(h, &tb.slots, 4);
        }
        size_t shared = 0;
        for (unsigned d : plan.shared_info) shared += d != 0;
        printf("rows %.2f ms, plan %.2f ms | enabled %d shared %d clusters %zu items %zu (max %d per cluster) max slots %d planes %d shared bodies %zu | digest %016llx\n",
               std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), (int)plan.enabled, (int)plan.shared, plan.clusters.size(),
               plan.items.size(), plan.max_items, plan.max_slots, plan.planes, shared, (unsigned long long)h);
        {
            size_t leaders = 0, members = 0;
            for (auto& it : plan.items) { const int fuse = (it.shape >> kItemFuseShift) & 7; leaders += (fuse & 3) != 0; members += (fuse & kItemFuseMember) != 0; }
            {   // the floor: every (cluster, batch, family)'s lanes cut into waves of 64
                std::map<std::tuple<int, int, int>, int> lanes;
                size_t other = 0;
                for (size_t cl = 0; cl < plan.clusters.size(); ++cl)
                    for (int k = 0; k < plan.clusters[cl].item_count; ++k) {
                        const ClusterItem& it = plan.items[plan.clusters[cl].item_begin + k];
                        if (it.type_id > 7) { ++other; continue; }
                        lanes[{(int)cl, it.batch_npred & 0xFFFF, it.type_id / 4}] += it.count;
                    }
                size_t floor_items = other;
                for (auto& kv : lanes) floor_items += (kv.second + 63) / 64;
                printf("lane floor: %zu claimable items per pass if every family's lanes of a batch were cut into full waves\n", floor_items);
            }
            if (leaders) printf("merged manifold groups: %zu groups take %zu typed items: %zu claimable items per pass instead of %zu\n", leaders, leaders + members, plan.items.size() - members, plan.items.size());
        }
        if (getenv("PLAN_DUMP_ITEMS") && plan.enabled) {  // every cluster's work items with their predecessor lists and the bodies they touch
            for (size_t cl = 0; cl < plan.clusters.size(); ++cl) {
                const ClusterDesc& cd = plan.clusters[cl];
                printf("cluster %zu: %d slots, %d items\n", cl, cd.slot_count, cd.item_count);
                for (int k = 0; k < cd.item_count; ++k) {
                    const ClusterItem& it = plan.items[cd.item_begin + k];
                    const HostTypeBatch& tb = c->tbs[it.tb];
                    const int np = (it.batch_npred >> 16) & 0xF, nx = (it.batch_npred >> 20) & 0xF;
                    printf("  item %3d batch %2d type %2d count %2d%s%s pred [", k, it.batch_npred & 0xFFFF, it.type_id, it.count, ((it.batch_npred >> 24) & 1) ? " OVERFLOW" : "", ((it.batch_npred >> 25) & 1) ? " XOVERFLOW" : "");
                    for (int q = 0; q < np; ++q) printf("%s%d", q ? " " : "", it.pred[q]);
                    printf("] xpred [");
                    for (int q = 0; q < nx; ++q) printf("%s%d", q ? " " : "", it.xpred[q]);
                    printf("] bodies");
                    for (int j = it.start; j < it.start + it.count; ++j) {
                        printf(" (");
                        for (int b = 0; b < tb.info.bodies; ++b) { const int32_t r = tb.dev_refs[(size_t)b * tb.stride + j]; printf("%s%d%s", b ? "," : "", r & kRefMask, (uint32_t)r >= kDynamicLimit ? "k" : ""); }
                        printf(")");
                    }
                    printf("\n");
                }
            }
        }
        if (getenv("PLAN_SIMULATE") && plan.enabled) simulate(c, plan);
        if (const char* mutate = getenv("PLAN_MUTATE")) {  // the validator must notice: 1 = a local reference points at the neighbouring slot, 2 = an item loses a row, 3 = a rank is bumped
            const int kind = atoi(mutate);
            for (auto& tb : c->tbs) {
                if (tb.count == 0 || tb.lrefs_soa.empty()) continue;
                if (kind == 1) tb.lrefs_soa[0] ^= 1;
                if (kind == 3 && plan.shared) { const int rows = (tb.info.bodies + 1) / 2; for (int j = 0; j < tb.count; ++j) if (tb.lrefs_soa[(size_t)rows * tb.stride + j] >> 8) { tb.lrefs_soa[(size_t)rows * tb.stride + j] += 1; break; } }
                break;
            }
            if (kind == 2 && !plan.items.empty()) plan.items[0].count -= plan.items[0].count > 1 ? 1 : 0;
        }
        if (getenv("PLAN_VALIDATE") && plan.enabled && validate(c, plan) != 0) return 3;
        if (getenv("PLAN_CHURN") && plan.enabled) { const int verdict = churn(c, plan, atoi(getenv("PLAN_CHURN")) + harness_thread); if (verdict) return verdict; }
        delete c;
    }
    return 0;
    };
    if (harness_threads == 1) return run_repeats(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < harness_threads; ++t) pool.emplace_back([&, t] { const int v = run_repeats(t); if (v) verdict_all.store(v); });
    for (auto& th : pool) th.join();
    return verdict_all.load();
}
