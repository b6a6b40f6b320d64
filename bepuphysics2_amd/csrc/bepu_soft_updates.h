// Structural updates that keep the island-per-workgroup schedule ("ranged ops that patch the cluster plan incrementally").
//
// A whole-island plan lays every type batch out by cluster: one segment of device slots per cluster, live constraints first, free slots behind them (reserved with
// BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS, or left by removals). The caller keeps addressing constraints by the reference's indices (append at ConstraintCount, swap-with-last
// on removal: TypeProcessor.cs:314-334, 695-717); `inv` / `perm` translate them to device slots, so on the device a removal only frees a slot (its local references then name
// a kinematic copy: the lane computes on, and like every kinematic reference writes no body back) and an addition fills a free slot of the segment of the cluster its bodies live in.
// What an update may NOT change is the plan's body sets: an addition whose dynamic bodies are not all in ONE cluster (a contact between two islands that were planned
// into different clusters, or a body that had no constraint), or that needs a kinematic body the cluster holds no copy of, or that finds no free slot, and a
// removal that would leave a body without constraints (the reference then integrates it as an unconstrained body), make the context leave the island schedule as
// before (rows back in the caller's order, launch-per-batch) until the next full upload.
// Predecessor lists: a removal leaves them as they are (a superfluous wait is harmless). An addition makes the lists of its cluster stale; they are rebuilt for
// that cluster alone when the updates are flushed (soft_rebuild_items: the planner's rule over the cluster's live slots, a few tens of microseconds per cluster,
// clusters in parallel on host threads) and the items re-uploaded. The order of constraint applications per body — the only thing results depend on — is the
// batch order either way.
// ("Would leave a body without constraints" is judged per flush, not per call: a pair that is removed and added again in the same frame never leaves the plan.)
#pragma once

// BEPUHIP_PLAN_STATS >= 2: the time the structural calls spend on the island layout's bookkeeping, reported with the next flush.
struct SoftCallTimer {
    bepuhip_ctx* c;
    std::chrono::steady_clock::time_point begin;
    static bool enabled() { static const bool on = env_int("BEPUHIP_PLAN_STATS", 0) >= 2; return on; }
    explicit SoftCallTimer(bepuhip_ctx* ctx) : c(ctx) { if (enabled()) begin = std::chrono::steady_clock::now(); }
    ~SoftCallTimer() { if (enabled()) { c->soft_call_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - begin).count(); ++c->soft_calls; } }
};

static bool soft_refuse(const char* why) {
    if (env_int("BEPUHIP_PLAN_STATS", 0)) fprintf(stderr, "bepuhip: leaving the island schedule: %s\n", why);
    return false;
}
static int soft_cluster_of_slot(const HostTypeBatch& tb, int slot) {
    return (int)(std::upper_bound(tb.seg_begin.begin(), tb.seg_begin.end(), slot) - tb.seg_begin.begin()) - 1;
}

// Constraints per constrained kinematic body, from the device-slot mirrors of the references (first structural update of a plan).
static void soft_ensure_kin_uses(bepuhip_ctx* c) {
    if (c->kin_uses_ready) return;
    c->kin_uses.clear();
    for (auto& tb : c->tbs)
        for (int k = 0; k < tb.info.bodies; ++k)
            for (int d = 0; d < tb.slots; ++d) {
                const int32_t r = tb.dev_refs[(size_t)k * tb.stride + d];
                if (r >= 0 && (uint32_t)r >= kDynamicLimit) ++c->kin_uses[r & kRefMask];
            }
    c->kin_uses_ready = true;
}
static void soft_kinematic_reference(bepuhip_ctx* c, int32_t ref, int delta) {  // a constraint that references kinematic body `ref` comes (+1) or goes (-1)
    if (ref < 0 || (uint32_t)ref < kDynamicLimit) return;
    int32_t& uses = c->kin_uses[ref & kRefMask];
    if ((delta > 0 && uses == 0) || (delta < 0 && uses == 1)) c->kin_touched.push_back(ref & kRefMask);
    uses += delta;
}
// The plan's list of constrained kinematic bodies follows the counts (judged per flush, like a dynamic body's last constraint: a refreshed pair changes nothing).
// true: the list changed (the caller re-uploads it and rebuilds the body flags).
static bool soft_update_kinlist(bepuhip_ctx* c) {
    bool changed = false;
    for (int32_t body : c->kin_touched) {
        const bool wanted = c->kin_uses[body] > 0;
        auto at = std::find(c->kinlist_host.begin(), c->kinlist_host.end(), body);
        if (wanted && at == c->kinlist_host.end()) { c->kinlist_host.push_back(body); changed = true; }
        else if (!wanted && at != c->kinlist_host.end()) { *at = c->kinlist_host.back(); c->kinlist_host.pop_back(); changed = true; }
    }
    c->kin_touched.clear();
    return changed;
}

// ---- what the structural calls note for the next flush (flat tables, see bepuhip_ctx::SoftSlotRecord) ----
static void soft_note_slot(bepuhip_ctx* c, int t, int d, bool live, const uint32_t* words, size_t count) {
    HostTypeBatch& tb = c->tbs[t];
    if (tb.soft_record_of.size() < (size_t)tb.slots) tb.soft_record_of.assign((size_t)tb.slots, -1);
    int32_t& at = tb.soft_record_of[d];
    if (at < 0) { at = (int32_t)c->soft_records.size(); c->soft_records.push_back({t, d, 0u, 0u, 0u}); }
    bepuhip_ctx::SoftSlotRecord& record = c->soft_records[at];
    record.live = live ? 1u : 0u; record.payload_at = (uint32_t)c->soft_payload.size(); record.payload_words = (uint32_t)count;
    c->soft_payload.insert(c->soft_payload.end(), words, words + count);
}
static bool soft_slot_noted(const bepuhip_ctx* c, int t, int d) {
    const HostTypeBatch& tb = c->tbs[t];
    return (size_t)d < tb.soft_record_of.size() && tb.soft_record_of[d] >= 0;
}
static void soft_note_index(bepuhip_ctx* c, HostTypeBatch* tb, int index) { tb->soft_dirty_indices.push_back(index); c->soft_index_dirty = true; }
static void soft_clear_notes(bepuhip_ctx* c) {
    for (auto& record : c->soft_records) c->tbs[record.tb].soft_record_of[record.slot] = -1;
    c->soft_records.clear(); c->soft_payload.clear();
    if (c->soft_index_dirty) { for (auto& tb : c->tbs) tb.soft_dirty_indices.clear(); c->soft_index_dirty = false; }
}
static void split_note_rerank(bepuhip_ctx* c, int32_t body) {
    if (c->split_rerank_flag.size() <= (size_t)body) c->split_rerank_flag.resize((size_t)body + 1, 0);
    if (!c->split_rerank_flag[body]) { c->split_rerank_flag[body] = 1; c->split_rerank_list.push_back(body); }
}

// ---- prefetching for a table of operations (bepuhip_apply_structural_ops) ----
// A frame's structural operations are spread over the scene (every hundredth contact of a pile): each one touches twenty-odd cache lines of mirrors no earlier one
// touched — the caller's index -> device slot, the slot's references / local references / rank words, its bodies' cluster, degree, batch mask, application list —
// and the chain index -> slot -> body -> list is serial. With the whole table in hand the lines of the operations ahead are asked for while the current one runs,
// one link of the chain per stage. Hints only: everything is bounds-checked against the state of the moment, nothing is dereferenced that a later operation may free.
static inline void soft_prefetch_line(const void* p) { __builtin_prefetch(p, 0, 1); }
static inline void soft_prefetch_body(const bepuhip_ctx* c, int32_t r) {
    if (r < 0 || (uint32_t)r >= kDynamicLimit || (size_t)r >= c->body_cluster.size()) return;
    soft_prefetch_line(&c->body_cluster[r]); soft_prefetch_line(&c->body_lref[r]);
    if ((size_t)r < c->body_degree.size()) { soft_prefetch_line(&c->body_degree[r]); soft_prefetch_line(&c->body_batches[r]); }
    if ((size_t)r < c->split_shared.size()) soft_prefetch_line(&c->split_shared[r]);
    if ((size_t)r < c->body_apps.size()) soft_prefetch_line(&c->body_apps[r]);
}
static inline void soft_prefetch_remove(const bepuhip_ctx* c, const HostTypeBatch* tb, int index, int stage) {
    if (!tb || tb->slots == 0 || index < 0 || (size_t)index >= tb->inv.size()) return;
    if (stage == 0) { soft_prefetch_line(&tb->inv[index]); return; }
    const int d = tb->inv[index];
    if (d < 0 || d >= tb->slots) return;
    if (stage == 1) {
        soft_prefetch_line(&tb->perm[d]);
        if ((size_t)d < tb->soft_record_of.size()) soft_prefetch_line(&tb->soft_record_of[d]);
        for (int k = 0; k < tb->info.bodies; ++k) {
            soft_prefetch_line(&tb->dev_refs[(size_t)k * tb->stride + d]);
            if (!tb->plan_lrefs.empty()) { soft_prefetch_line(&tb->plan_lrefs[(size_t)k * tb->stride + d]); soft_prefetch_line(&tb->plan_ranks[(size_t)k * tb->stride + d]); }
        }
        return;
    }
    for (int k = 0; k < tb->info.bodies; ++k) {
        const int32_t r = tb->dev_refs[(size_t)k * tb->stride + d];
        soft_prefetch_body(c, r);
        if (stage == 3 && r >= 0 && (uint32_t)r < kDynamicLimit && (size_t)r < c->body_apps.size() && !c->body_apps[r].empty()) soft_prefetch_line(c->body_apps[r].data());
    }
}
static inline void soft_prefetch_add(const bepuhip_ctx* c, const HostTypeBatch* tb, const int32_t* refs, int stage) {
    if (!tb || tb->slots == 0) return;
    for (int k = 0; k < tb->info.bodies; ++k) {
        const int32_t r = refs[k];
        if (stage == 0) { soft_prefetch_body(c, r); continue; }
        if (r < 0 || (uint32_t)r >= kDynamicLimit || (size_t)r >= c->body_cluster.size()) continue;
        if ((size_t)r < c->body_apps.size() && !c->body_apps[r].empty()) soft_prefetch_line(c->body_apps[r].data());
        const int cl = c->body_cluster[r];
        if (cl < 0 || (size_t)cl + 1 >= tb->seg_begin.size()) continue;
        // the cluster's segment of the type batch: where the free slot is looked for (live slots first, free ones behind them: the tail of the segment)
        const int begin = tb->seg_begin[cl], end = tb->seg_begin[cl + 1];
        for (int at = begin; at < end; at += 16) soft_prefetch_line(&tb->perm[at]);
    }
}

static void soft_setup(bepuhip_ctx* c, ClusterPlan& plan) {
    c->soft_ok = false; c->soft_split = false;
    c->soft_records.clear(); c->soft_payload.clear(); c->soft_index_dirty = false; c->soft_items_dirty = false; c->soft_adds = c->soft_removes = 0;
    for (auto& tb : c->tbs) { tb.soft_record_of.clear(); tb.soft_dirty_indices.clear(); }
    c->body_apps.clear(); c->split_rerank_flag.clear(); c->split_rerank_list.clear(); c->split_patches.clear();
    c->kinlist_host = plan.kinlist; c->kin_uses.clear(); c->kin_touched.clear(); c->kin_uses_ready = false;
    c->free_slots_ready = false; c->cluster_free_slots.clear(); c->body_moves.clear();
    if (!plan.enabled || env_int("BEPUHIP_NO_SOFT_UPDATES", 0)) return;
    if (plan.shared) {  // split-island plan: the second half of this file
        if (env_int("BEPUHIP_NO_SPLIT_SOFT_UPDATES", 0)) return;
        for (auto& tb : c->tbs) if (tb.info.bodies > 2) return;
        c->body_cluster.swap(plan.body_cluster); c->body_lref.swap(plan.body_lref); c->body_degree.clear(); c->body_batches.clear();
        c->split_shared.swap(plan.split_shared); c->cluster_extra.swap(plan.cluster_extra); c->cluster_natural.swap(plan.cluster_natural);
        c->cluster_bodies_host = plan.cluster_bodies; c->split_visit = plan.split_visit;
        c->items_host = plan.items; c->clusters_host = plan.clusters;
        c->cluster_degraded.assign(plan.clusters.size(), 0);
        c->soft_ok = true; c->soft_split = true;
        return;
    }
    c->body_cluster.swap(plan.body_cluster); c->body_lref.swap(plan.body_lref); c->body_degree.clear(); c->body_batches.clear(); c->cluster_kin.swap(plan.cluster_kin);
    c->cluster_bodies_host = plan.cluster_bodies;
    c->items_host = plan.items; c->clusters_host = plan.clusters;
    c->cluster_degraded.assign(plan.clusters.size(), 0);
    c->soft_ok = true;
}

// The mirrors the first structural update of a plan builds lazily (constraint counts, free LDS slots, kinematic uses; on a split plan every body's applications:
// 26 ms for the 100k-box pile), built NOW: by the worker of a background re-plan (bepuhip_replan_begin), whose commit replays a frame's worth of operations at once.
static void soft_ensure_degrees(bepuhip_ctx* c);
static void soft_ensure_free_slots(bepuhip_ctx* c);
static void split_ensure_mirrors(bepuhip_ctx* c);
static void soft_warm(bepuhip_ctx* c) {
    if (!c->soft_ok) return;
    soft_ensure_kin_uses(c);
    if (c->soft_split) split_ensure_mirrors(c);
    else { soft_ensure_degrees(c); soft_ensure_free_slots(c); }
}
// Everything soft_setup and the lazy builders fill, moved from the context it was prepared on (the shadow of a background re-plan; its type batches — which carry the
// per-type-batch half of the state — travel separately) to the context that will run it.
static void soft_move_state(bepuhip_ctx* to, bepuhip_ctx* from) {
#define MV(f) to->f = std::move(from->f)
    MV(soft_ok); MV(soft_split); MV(soft_records); MV(soft_payload); MV(soft_index_dirty); MV(soft_items_dirty); MV(soft_adds); MV(soft_removes);
    MV(body_apps); MV(split_rerank_flag); MV(split_rerank_list); MV(split_patches); MV(kinlist_host); MV(kin_uses); MV(kin_touched); MV(kin_uses_ready);
    MV(free_slots_ready); MV(cluster_free_slots); MV(body_moves); MV(body_cluster); MV(body_lref); MV(body_degree); MV(body_batches); MV(cluster_kin);
    MV(split_shared); MV(cluster_extra); MV(cluster_natural); MV(cluster_extra_uses); MV(cluster_bodies_host); MV(split_visit); MV(items_host); MV(clusters_host);
    MV(cluster_degraded); MV(soft_orphans);
#undef MV
}

// Constraint count of every dynamic body, from the device-slot mirrors of the references: counted when the first structural update arrives (an upload that is never
// followed by one does not pay for it).
static void soft_ensure_degrees(bepuhip_ctx* c) {
    if (!c->body_degree.empty()) return;
    c->body_degree.assign(c->body_cluster.size(), 0);
    c->body_batches.assign(c->body_cluster.size(), 0);
    for (auto& tb : c->tbs)
        for (int k = 0; k < tb.info.bodies; ++k)
            for (int d = 0; d < tb.slots; ++d) {
                const int32_t r = tb.dev_refs[(size_t)k * tb.stride + d];
                if (r >= 0 && (uint32_t)r < kDynamicLimit && (size_t)r < c->body_degree.size()) { ++c->body_degree[r]; if (tb.batch < 64) c->body_batches[r] |= 1ull << tb.batch; }
            }
}

// ---- Bodies join and leave a plan with their first and last constraint (Solver.cs:1025-1030 / 1368-1377 are the reference's bookkeeping of the same events: a body
// without constraints is integrated as an unconstrained body, once per frame) ----
// Every unused LDS slot of every cluster, lowest natural index last (so that pop_back hands out the lowest): scanned from the slot tables' mirror on first use.
static void soft_ensure_free_slots(bepuhip_ctx* c) {
    if (c->free_slots_ready) return;
    c->cluster_free_slots.assign(c->clusters_host.size(), {});
    for (size_t cl = 0; cl < c->clusters_host.size(); ++cl) {
        const ClusterDesc& cd = c->clusters_host[cl];
        for (int natural = std::min(cd.slot_count, 0x3FF0) - 1; natural >= 0; --natural) {
            const int slot = rotated_slot(natural);
            if (slot < cd.slot_count && c->cluster_bodies_host[cd.body_begin + slot] == -1) c->cluster_free_slots[cl].push_back(slot);
        }
    }
    if (c->soft_split) for (size_t cl = 0; cl < c->cluster_natural.size(); ++cl) c->cluster_natural[cl] = c->clusters_host[cl].slot_count;  // the lists hold everything now
    c->free_slots_ready = true;
}
static void soft_patch_slot_table(bepuhip_ctx* c, int cl, int slot, int32_t entry) {
    const size_t at = (size_t)c->clusters_host[cl].body_begin + slot;
    c->cluster_bodies_host[at] = entry;
    c->split_patches.push_back({2, at, -1, 0, 0});
}
static void soft_ensure_body(bepuhip_ctx* c, int32_t body) {  // the per-body tables reach `body`
    if ((size_t)body < c->body_cluster.size()) return;
    const size_t n = (size_t)body + 1;
    c->body_cluster.resize(n, -1); c->body_lref.resize(n, 0);
    if (!c->body_degree.empty()) { c->body_degree.resize(n, 0); c->body_batches.resize(n, 0); }
    if (c->soft_split) { c->split_shared.resize(n, 0); if (!c->body_apps.empty()) c->body_apps.resize(n); }
}
static void soft_clustered_positions(bepuhip_ctx* c) {
    if (!c->clustered_position.empty() || c->clustered_dynamic_host.empty()) return;
    for (size_t i = 0; i < c->clustered_dynamic_host.size(); ++i) c->clustered_position[c->clustered_dynamic_host[i]] = (int32_t)i;
}
// A dynamic body without constraints gets its first one: it becomes a body of cluster `cl` (an unused LDS slot, an entry in the list behind kFlagClustered).
static bool soft_adopt_body(bepuhip_ctx* c, int32_t body, int cl) {
    soft_ensure_free_slots(c);
    if (c->cluster_free_slots[cl].empty()) return soft_refuse("no free LDS slot in the cluster for a body that had no constraints");
    if ((int)c->clustered_dynamic_host.size() >= c->clustered_dynamic_capacity) return soft_refuse("the plan's body list is full");
    if (c->soft_split && (size_t)body >= c->shared_bodies) return soft_refuse("a body beyond the split plan's shared-body tables");
    soft_ensure_body(c, body);
    const int slot = c->cluster_free_slots[cl].back();
    c->cluster_free_slots[cl].pop_back();
    soft_patch_slot_table(c, cl, slot, body);
    c->body_cluster[body] = cl; c->body_lref[body] = slot;
    c->owned_mask_bodies = 0;  // device groups: which member owns the body at the end of a step follows its cluster (owned_body_mask reads the live table)
    soft_clustered_positions(c);
    c->clustered_position[body] = (int32_t)c->clustered_dynamic_host.size();
    c->clustered_dynamic_host.push_back(body);
    c->clustered_dirty = true; c->soft_flags_stale = true;
    return true;
}
// ... and one that lost its last constraint (and did not get one back before the flush) leaves: its slot is free again, the tail integrates it from now on.
static void soft_release_body(bepuhip_ctx* c, int32_t body) {
    const int cl = c->body_cluster[body];
    if (cl < 0) return;
    soft_ensure_free_slots(c);
    const int slot = c->body_lref[body] & 0x3FFF;
    soft_patch_slot_table(c, cl, slot, -1);
    c->cluster_free_slots[cl].push_back(slot);
    c->body_cluster[body] = -1;
    c->owned_mask_bodies = 0;  // device groups: which member owns the body at the end of a step follows its cluster (owned_body_mask reads the live table)
    soft_clustered_positions(c);
    auto at = c->clustered_position.find(body);
    if (at != c->clustered_position.end()) {
        const int32_t position = at->second, last = c->clustered_dynamic_host.back();
        c->clustered_dynamic_host[position] = last;
        c->clustered_position[last] = position;
        c->clustered_dynamic_host.pop_back();
        c->clustered_position.erase(body);
    }
    if (c->soft_split && c->split_shared[body]) { c->split_shared[body] = 0; c->split_patches.push_back({1, (size_t)body, -1, 0, 0}); }
    c->clustered_dirty = true; c->soft_flags_stale = true;
}

static void split_ensure_mirrors(bepuhip_ctx* c);
// Bodies.RemoveAt (BodySet.cs:83-110) moved the last body into a freed slot: body `from` is body `to` from now on. It keeps its cluster, its LDS slots (home, ghosts,
// kinematic copies), its ranks and its constraints; every table that names it by index is re-keyed, the slot tables on the device are patched. The caller patches the
// references constraint by constraint (soft_update_reference below) and sends the body array again.
static bool soft_move_body(bepuhip_ctx* c, int32_t from, int32_t to, bool kinematic) {
    if (kinematic) {
        soft_ensure_kin_uses(c);
        if (c->kin_uses.count(to) && c->kin_uses[to] > 0) return soft_refuse("a body moved onto a kinematic body that still has constraints");
        c->kin_uses[to] = c->kin_uses[from];
        c->kin_uses.erase(from);
        for (int32_t& body : c->kinlist_host) if (body == from) { body = to; c->kinlist_dirty = true; }
        for (size_t cl = 0; cl < c->clusters_host.size(); ++cl) {
            if (c->soft_split) {
                auto found = c->cluster_extra[cl].find(from | kSlotKinematic);
                if (found == c->cluster_extra[cl].end()) continue;
                const int slot = found->second;
                c->cluster_extra[cl].erase(found);
                c->cluster_extra[cl][to | kSlotKinematic] = slot;
                auto uses = c->cluster_extra_uses[cl].find(from | kSlotKinematic);
                if (uses != c->cluster_extra_uses[cl].end()) { const int32_t n = uses->second; c->cluster_extra_uses[cl].erase(uses); c->cluster_extra_uses[cl][to | kSlotKinematic] = n; }
                soft_patch_slot_table(c, (int)cl, slot, to | kSlotKinematic);
            } else {
                auto found = c->cluster_kin[cl].find(from);
                if (found == c->cluster_kin[cl].end()) continue;
                const int slot = found->second;
                c->cluster_kin[cl].erase(found);
                c->cluster_kin[cl][to] = slot;
                soft_patch_slot_table(c, (int)cl, slot, to | kSlotKinematic);
            }
        }
        c->soft_flags_stale = true;
        return true;
    }
    if ((size_t)from >= c->body_cluster.size() || c->body_cluster[from] < 0) return soft_refuse("a moved body that is not part of the plan");
    if (c->soft_split && (size_t)to >= c->shared_bodies) return soft_refuse("a body beyond the split plan's shared-body tables");
    soft_ensure_body(c, to);
    if (c->body_cluster[to] >= 0 && c->body_degree[to] == 0) soft_release_body(c, to);  // the removed body lost its last constraint in this same batch of updates: it leaves now
    if (c->body_cluster[to] >= 0) return soft_refuse("a body moved onto one that still has constraints");
    const int cl = c->body_cluster[from], slot = c->body_lref[from] & 0x3FFF;
    c->body_cluster[to] = cl; c->body_lref[to] = c->body_lref[from]; c->body_cluster[from] = -1;
    c->owned_mask_bodies = 0;  // device groups: which member owns the body at the end of a step follows its cluster (owned_body_mask reads the live table)
    c->body_degree[to] = c->body_degree[from]; c->body_degree[from] = 0;
    c->body_batches[to] = c->body_batches[from]; c->body_batches[from] = 0;
    int32_t entry = to;
    if (c->soft_split) {
        c->split_shared[to] = c->split_shared[from]; c->split_shared[from] = 0;
        c->body_apps[to].swap(c->body_apps[from]); c->body_apps[from].clear();
        if (c->split_shared[to]) {
            entry |= kSlotSharedHome;
            for (size_t q = 0; q < c->clusters_host.size(); ++q) {  // its ghost copies
                auto found = c->cluster_extra[q].find(from | kSlotGhost);
                if (found == c->cluster_extra[q].end()) continue;
                const int ghost_slot = found->second;
                c->cluster_extra[q].erase(found);
                c->cluster_extra[q][to | kSlotGhost] = ghost_slot;
                auto uses = c->cluster_extra_uses[q].find(from | kSlotGhost);
                if (uses != c->cluster_extra_uses[q].end()) { const int32_t n = uses->second; c->cluster_extra_uses[q].erase(uses); c->cluster_extra_uses[q][to | kSlotGhost] = n; }
                soft_patch_slot_table(c, (int)q, ghost_slot, to | kSlotGhost);
            }
            c->split_patches.push_back({1, (size_t)from, -1, 0, 0});  // degrees in shared_info (read from body_apps at the flush)
            c->split_patches.push_back({1, (size_t)to, -1, 0, 0});
        }
        if ((size_t)from < c->split_rerank_flag.size() && c->split_rerank_flag[from]) { c->split_rerank_flag[from] = 0; split_note_rerank(c, to); }  // (the list keeps `from`: skipped at the flush, its flag is down)
    }
    soft_patch_slot_table(c, cl, slot, entry);
    soft_clustered_positions(c);
    auto at = c->clustered_position.find(from);
    if (at != c->clustered_position.end()) { const int32_t position = at->second; c->clustered_dynamic_host[position] = to; c->clustered_position.erase(at); c->clustered_position[to] = position; }
    for (int32_t& orphan : c->soft_orphans) if (orphan == from) orphan = to;
    c->clustered_dirty = true; c->soft_flags_stale = true;
    return true;
}
// TypeProcessor.UpdateForBodyMemoryMove (TypeProcessor.cs:807) on the island layout: one reference of one constraint follows a body that moved.
static bool soft_update_reference(bepuhip_ctx* c, HostTypeBatch* tb, int index, int k, int32_t ref) {
    if (!c->soft_ok || tb->slots == 0 || tb->info.bodies > 2 || env_int("BEPUHIP_SOFT_BODY_MOVES", 1) == 0) return soft_refuse("a body reference changed (a body moved in memory)");
    if (c->soft_split) split_ensure_mirrors(c); else soft_ensure_degrees(c);
    const int d = tb->inv[index];
    int32_t& mirror = tb->dev_refs[(size_t)k * tb->stride + d];
    if (mirror < 0 || ((mirror ^ ref) & ~kRefMask) != 0) return soft_refuse("a body reference changed its kind");
    const int32_t from = mirror & kRefMask, to = ref & kRefMask;
    if (from == to) return true;
    auto known = c->body_moves.find(from);
    if (known == c->body_moves.end() || known->second != to) {  // the first patch of this move brings the body's tables along
        if (!soft_move_body(c, from, to, (uint32_t)ref >= kDynamicLimit)) return false;
        c->body_moves[from] = to;
    }
    mirror = ref;
    c->split_patches.push_back({3, tb->refs_off + (size_t)k * tb->stride + d, (int32_t)(tb - c->tbs.data()), d, k});
    return true;
}

// bepuhip_swap_constraints on an island layout (whole islands or split): the device slots keep their contents, the caller's two indices name each other's slot from now on.
static bool soft_swap(bepuhip_ctx* c, HostTypeBatch* tb, int a, int b) {
    if (!c->soft_ok || tb->slots == 0) return soft_refuse("a swap in a type batch the island layout does not manage");
    const int t = (int)(tb - c->tbs.data());
    const int da = tb->inv[a], db = tb->inv[b];
    tb->inv[a] = db; tb->inv[b] = da;
    tb->perm[db] = a; tb->perm[da] = b;
    soft_note_index(c, tb, a); soft_note_index(c, tb, b);
    return true;
}

// TypeProcessor.Remove on the island layout. false: not possible here (nothing was changed).
static bool split_remove(bepuhip_ctx* c, HostTypeBatch* tb, int index);
static bool split_add(bepuhip_ctx* c, HostTypeBatch* tb, const int32_t* refs, const float* prestep, bool* violation);
static bool soft_remove(bepuhip_ctx* c, HostTypeBatch* tb, int index) {
    if (!c->soft_ok || tb->slots == 0 || tb->info.bodies > 2) return soft_refuse("removal from a type batch the island layout does not manage");
    if (c->soft_split) return split_remove(c, tb, index);
    soft_ensure_degrees(c);
    soft_ensure_kin_uses(c);
    const int t = (int)(tb - c->tbs.data());
    const int d = tb->inv[index], last = tb->count - 1, dl = tb->inv[last];
    for (int k = 0; k < tb->info.bodies; ++k) {
        int32_t& r = tb->dev_refs[(size_t)k * tb->stride + d];
        soft_kinematic_reference(c, r, -1);
        // a body whose last constraint goes would have to leave the plan (the reference integrates it as an unconstrained body from then on) — unless the same
        // batch of updates gives it a constraint again (a refreshed pair): decided when the updates are flushed (soft_bodies_still_constrained)
        if (r >= 0 && (uint32_t)r < kDynamicLimit) {
            if (tb->batch < 64) c->body_batches[r] &= ~(1ull << tb->batch);
            if (--c->body_degree[r] == 0) c->soft_orphans.push_back(r);
        }
        r = -1;
    }
    tb->perm[d] = -1;
    soft_note_slot(c, t, d, false, nullptr, 0);
    if (index != last) {  // TypeProcessor.Move (:578-592): the last constraint takes the removed one's index; on the device it stays where it is
        tb->inv[index] = dl; tb->perm[dl] = index;
        soft_note_index(c, tb, index);
    }
    tb->inv.pop_back();
    tb->count = last;
    ++c->soft_removes;
    return true;
}

// Bodies that lost their last constraint since the last flush and did not get one back leave the plan (soft_release_body). Always true since round 3 (BEPUHIP_SOFT_ORPHANS=0:
// round 2's behaviour, the context leaves the island schedule instead).
static bool soft_bodies_still_constrained(bepuhip_ctx* c) {
    const bool release = env_int("BEPUHIP_SOFT_ORPHANS", 1) != 0;
    bool ok = true;
    for (int32_t body : c->soft_orphans) {
        if (c->body_degree[body] > 0) continue;
        if (release) soft_release_body(c, body); else ok = false;
    }
    c->soft_orphans.clear();
    return ok || soft_refuse("a body lost its last constraint");
}

// TypeProcessor.AllocateInTypeBatch on the island layout. false: not possible here (nothing was changed).
static bool soft_add(bepuhip_ctx* c, HostTypeBatch* tb, const int32_t* refs, const float* prestep, bool* violation) {
    if (!c->soft_ok || tb->slots == 0 || tb->info.bodies > 2) return soft_refuse("addition to a type batch the island layout does not manage");
    if (c->soft_split) return split_add(c, tb, refs, prestep, violation);
    soft_ensure_degrees(c);
    const int t = (int)(tb - c->tbs.data()), nb = tb->info.bodies;
    int cl = -1, dynamic_bodies = 0, newcomers = 0;
    for (int k = 0; k < nb; ++k) {
        if ((uint32_t)refs[k] >= kDynamicLimit) continue;
        ++dynamic_bodies;
        if (refs[k] >= (int)c->body_cluster.size() || c->body_cluster[refs[k]] < 0) { ++newcomers; continue; }  // a body without constraints so far: it joins the cluster below
        if (cl >= 0 && c->body_cluster[refs[k]] != cl) return soft_refuse("the new constraint's bodies live in two clusters");
        cl = c->body_cluster[refs[k]];
    }
    if (dynamic_bodies == 0) return soft_refuse("the new constraint has no dynamic body");
    if (newcomers > 0 && env_int("BEPUHIP_SOFT_ORPHANS", 1) == 0) return soft_refuse("the new constraint's body had no constraints");
    for (int k = 0; k < nb; ++k) {
        if ((uint32_t)refs[k] < kDynamicLimit && (size_t)refs[k] < c->body_batches.size() && tb->batch < 64 && (c->body_batches[refs[k]] >> tb->batch) & 1) { *violation = true; return false; }
        if (nb == 2 && refs[0] == refs[1]) return soft_refuse("a constraint between a body and itself");
    }
    if (cl < 0) {  // only newcomers: a new island — into the first cluster that has a free device slot in this type batch and LDS slots for them
        soft_ensure_free_slots(c);
        for (size_t q = 0; q < c->clusters_host.size() && cl < 0; ++q) {
            bool row = false;
            for (int s0 = tb->seg_begin[q]; s0 < tb->seg_begin[q + 1] && !row; ++s0) row = tb->perm[s0] < 0;
            if (row && (int)c->cluster_free_slots[q].size() >= nb + 1) cl = (int)q;
        }
        if (cl < 0) return soft_refuse("no cluster has room for a new island");
    }
    {   // room for everything before anything is taken: a device slot of the type batch, LDS slots for the newcomers and for kinematic copies the cluster lacks
        bool row = false;
        for (int s0 = tb->seg_begin[cl]; s0 < tb->seg_begin[cl + 1] && !row; ++s0) row = tb->perm[s0] < 0;
        if (!row) return soft_refuse("no free device slot in the cluster's segment of the type batch");
        int needed = newcomers;
        for (int k = 0; k < nb; ++k) needed += (uint32_t)refs[k] >= kDynamicLimit && !c->cluster_kin[cl].count(refs[k] & kRefMask);
        if (needed > 0) {
            soft_ensure_free_slots(c);
            if ((int)c->cluster_free_slots[cl].size() < needed) return soft_refuse("no free LDS slot in the cluster for a new body or kinematic copy");
            if ((int)c->clustered_dynamic_host.size() + newcomers > c->clustered_dynamic_capacity) return soft_refuse("the plan's body list is full");
        }
    }
    for (int k = 0; k < nb; ++k)
        if ((uint32_t)refs[k] < kDynamicLimit && (refs[k] >= (int)c->body_cluster.size() || c->body_cluster[refs[k]] < 0) && !soft_adopt_body(c, refs[k], cl)) return false;
    // The batch invariant the whole solve rests on (Solver.cs:1046-1051, asserted by the reference in debug builds): a dynamic body appears at most once per synchronized
    // batch. A caller that breaks it would get a silent race on the body's velocity inside one work item or launch: refused here, where the host knows the references.
    for (int k = 0; k < nb; ++k)
        if ((uint32_t)refs[k] < kDynamicLimit && tb->batch < 64 && (c->body_batches[refs[k]] >> tb->batch) & 1) { *violation = true; return false; }
    unsigned halves[2] = {0u, 0u};
    for (int k = 0; k < nb; ++k) {
        if ((uint32_t)refs[k] < kDynamicLimit) { halves[k] = (unsigned)c->body_lref[refs[k]]; continue; }
        auto copy = c->cluster_kin[cl].find(refs[k] & kRefMask);
        if (copy == c->cluster_kin[cl].end()) {  // the cluster gets a private copy of the kinematic body (room was checked above)
            const int slot = c->cluster_free_slots[cl].back();
            c->cluster_free_slots[cl].pop_back();
            soft_patch_slot_table(c, cl, slot, (refs[k] & kRefMask) | kSlotKinematic);
            copy = c->cluster_kin[cl].emplace(refs[k] & kRefMask, slot).first;
        }
        halves[k] = (unsigned)copy->second | 0x8000u;
    }
    int d = -1;
    for (int s = tb->seg_begin[cl]; s < tb->seg_begin[cl + 1] && d < 0; ++s) if (tb->perm[s] < 0) d = s;
    if (d < 0) return soft_refuse("no free device slot in the cluster's segment of the type batch");
    uint32_t words[2 + 1 + 64];  // references, the packed local references, the prestep lane (the widest type, Contact4Nonconvex, has 35 prestep floats)
    size_t nwords = 0;
    soft_ensure_kin_uses(c);
    for (int k = 0; k < nb; ++k) {
        words[nwords++] = (uint32_t)refs[k];
        tb->dev_refs[(size_t)k * tb->stride + d] = refs[k];
        soft_kinematic_reference(c, refs[k], +1);
        if ((uint32_t)refs[k] < kDynamicLimit) { ++c->body_degree[refs[k]]; if (tb->batch < 64) c->body_batches[refs[k]] |= 1ull << tb->batch; }
    }
    words[nwords++] = halves[0] | (halves[1] << 16);
    for (int f = 0; f < tb->info.prestep; ++f) { uint32_t w; memcpy(&w, &prestep[f], 4); words[nwords++] = w; }
    soft_note_slot(c, t, d, true, words, nwords);
    tb->perm[d] = tb->count;
    tb->inv.push_back(d);
    soft_note_index(c, tb, tb->count);
    tb->count += 1;
    c->cluster_degraded[cl] = 1;  // its predecessor lists no longer describe it: rebuilt when the updates are flushed (soft_rebuild_items)
    c->soft_items_dirty = true;
    ++c->soft_adds;
    return true;
}

// The predecessor lists of one cluster's work items, from the layout as it is now: the planner's rule (bepu_cluster_plan.h, phase B) over the device slots that are
// live. Item k of the cluster is its k-th item in plan order, which is claim order and batch order.
static void soft_rebuild_items(bepuhip_ctx* c, int cl) {
    const ClusterDesc& cd = c->clusters_host[cl];
    std::vector<int32_t> last_toucher(cd.slot_count + 16, -1);
    std::vector<std::pair<int32_t, int32_t>> first_touch;
    for (int self = 0; self < cd.item_count; ++self) {
        ClusterItem& it = c->items_host[cd.item_begin + self];
        const HostTypeBatch& tb = c->tbs[it.tb];
        const int nb = tb.info.bodies;
        int npred = 0, overflow = 0;
        memset(it.pred, 0, sizeof(it.pred)); memset(it.xpred, 0, sizeof(it.xpred));
        for (int pass = 0; pass < 2; ++pass)  // first the lists (against the touchers so far), then this item becomes the last toucher of its bodies
            for (int j = it.start; j < it.start + it.count; ++j) {
                if (tb.perm[j] < 0) continue;
                for (int k = 0; k < nb; ++k) {
                    const int32_t r = tb.dev_refs[(size_t)k * tb.stride + j];
                    if (r < 0 || (uint32_t)r >= kDynamicLimit) continue;
                    const int lr = c->body_lref[r];
                    if (pass == 1) { last_toucher[lr] = self; continue; }
                    const int pred = last_toucher[lr];
                    if (pred < 0) { first_touch.push_back({self, lr}); continue; }
                    if (pred == self) continue;
                    bool known = false;
                    for (int q = 0; q < npred; ++q) known |= it.pred[q] == pred;
                    if (known) continue;
                    if (npred < kMaxPreds) it.pred[npred++] = (unsigned short)pred; else overflow = 1;
                }
            }
        if (overflow) npred = 0;
        it.batch_npred = (it.batch_npred & 0xFFFF) | (npred << 16) | (overflow << 24);
    }
    for (auto& fs : first_touch) {  // cross-pass predecessors: the last toucher (end of a pass) of every body an item touches first
        ClusterItem& it = c->items_host[cd.item_begin + fs.first];
        const int last = last_toucher[fs.second];
        int nx = (it.batch_npred >> 20) & 0xF;
        if ((it.batch_npred >> 25) & 1) continue;
        bool known = false;
        for (int q = 0; q < nx; ++q) known |= it.xpred[q] == last;
        if (known) continue;
        if (nx < kMaxPreds) { it.xpred[nx++] = (unsigned short)last; it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (nx << 20); }
        else it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (1 << 25);
    }
}

// ======================================================================================================================================================
// Split-island plans. The rows have the same segmented layout, so removals and additions take and free device slots exactly as above. What is
// different: a body may be SHARED — referenced from a cluster that is not its home —, every application on a shared body carries a rank word (rank | degree << 8 |
// hand-off flags) that orders it against the body's other applications, the other cluster holds a ghost slot of the body, and the home cluster's slot table says so.
//  * removal: frees the slot; the shared bodies it touched are RE-RANKED at the flush (their remaining applications renumbered in batch order, hand-off flags
//    recomputed, the degree written to shared_info).
//  * addition: runs in the home cluster of one of its dynamic bodies (the one where the other body already has a slot, if any). A body of another cluster is
//    referenced through a ghost slot there — an existing one, or a free LDS slot of the reserve; a body that becomes shared this way gets the shared bit on all its
//    applications and the kSlotSharedHome flag in its home's slot table. Kinematic bodies get a private copy the same way. The touched shared bodies are re-ranked.
//  * every cluster that holds an application of a re-ranked body has its items' predecessor lists rebuilt (the hand-off flags decide which lanes wait through LDS).
// Refused (the context then leaves the island schedule, as before): no free row slot, no free LDS slot, a body that had no constraint, a body with 255 constraints,
// a removal that leaves a body without constraints.
// ======================================================================================================================================================
constexpr uint32_t kSoftRankPredLocal = 1u << 16, kSoftRankSuccLocal = 1u << 17;

static void split_ensure_mirrors(bepuhip_ctx* c) {
    if (!c->body_apps.empty()) return;
    const size_t universe = c->body_cluster.size();
    c->body_apps.assign(universe, {});
    c->body_degree.assign(universe, 0);
    c->body_batches.assign(universe, 0);
    for (size_t t = 0; t < c->tbs.size(); ++t) {  // type batches in index order = batch order: every body's list ends up sorted by batch
        const HostTypeBatch& tb = c->tbs[t];
        for (int d = 0; d < tb.slots; ++d) {
            if (tb.perm[d] < 0) continue;
            for (int k = 0; k < tb.info.bodies; ++k) {
                const int32_t r = tb.dev_refs[(size_t)k * tb.stride + d];
                if (r < 0 || (uint32_t)r >= kDynamicLimit || (size_t)r >= universe) continue;
                c->body_apps[r].push_back({(int32_t)t, d, k});
                ++c->body_degree[r];
                if (tb.batch < 64) c->body_batches[r] |= 1ull << tb.batch;
            }
        }
    }
    for (auto& apps : c->body_apps) std::sort(apps.begin(), apps.end(), [](const bepuhip_ctx::SplitApp& a, const bepuhip_ctx::SplitApp& b) { return a.tb < b.tb; });
    // how often every ghost / kinematic copy of every cluster is referenced: a copy nothing references any more gives its LDS slot back (split_release_copy)
    c->cluster_extra_uses.assign(c->clusters_host.size(), {});
    c->free_slots_ready = false;
    soft_ensure_free_slots(c);
    for (auto& tb : c->tbs)
        for (int d = 0; d < tb.slots; ++d) {
            if (tb.perm[d] < 0) continue;
            const int cl = soft_cluster_of_slot(tb, d);
            for (int k = 0; k < tb.info.bodies; ++k) {
                const int32_t r = tb.dev_refs[(size_t)k * tb.stride + d];
                if (r < 0) continue;
                if ((uint32_t)r >= kDynamicLimit) ++c->cluster_extra_uses[cl][(r & kRefMask) | kSlotKinematic];
                else if (c->body_cluster[r] != cl) ++c->cluster_extra_uses[cl][r | kSlotGhost];
            }
        }
}
static void split_release_copy(bepuhip_ctx* c, int cl, int32_t key) {  // one reference less to a ghost / kinematic copy; the last one frees its LDS slot
    auto uses = c->cluster_extra_uses[cl].find(key);
    if (uses == c->cluster_extra_uses[cl].end() || --uses->second > 0) return;
    c->cluster_extra_uses[cl].erase(uses);
    auto found = c->cluster_extra[cl].find(key);
    if (found == c->cluster_extra[cl].end()) return;
    const ClusterDesc& cd = c->clusters_host[cl];
    c->cluster_bodies_host[cd.body_begin + found->second] = -1;
    c->split_patches.push_back({2, (size_t)(cd.body_begin + found->second), -1, 0, 0});
    c->cluster_free_slots[cl].push_back(found->second);
    c->cluster_extra[cl].erase(found);
}
static unsigned split_packed_lrefs(const HostTypeBatch& tb, int d, int row) {  // the 16-bit halves of body slots 2 * row and 2 * row + 1 of device slot d
    unsigned word = 0;
    for (int k = 2 * row; k < std::min(tb.info.bodies, 2 * row + 2); ++k) {
        const int32_t lr = tb.plan_lrefs[(size_t)k * tb.stride + d];
        const uint32_t half = ((uint32_t)lr & 0x7FFFu) | (((uint32_t)lr >= kDynamicLimit) ? 0x8000u : 0u);
        word |= half << (16 * (k & 1));
    }
    return word;
}
// A free natural slot index of the cluster's LDS table (rotated into the slot number the kernel uses), or -1.
static int split_take_lds_slot(bepuhip_ctx* c, int cl, int32_t tagged_body) {
    const ClusterDesc& cd = c->clusters_host[cl];
    int slot;
    if (!c->cluster_free_slots[cl].empty()) { slot = c->cluster_free_slots[cl].back(); c->cluster_free_slots[cl].pop_back(); }
    else {
        const int natural = c->cluster_natural[cl];
        if (natural >= cd.slot_count || natural >= 0x3FF0) return -1;
        slot = rotated_slot(natural);
        if (slot >= cd.slot_count) return -1;
        c->cluster_natural[cl] = natural + 1;
    }
    c->cluster_bodies_host[cd.body_begin + slot] = tagged_body;
    c->split_patches.push_back({2, (size_t)(cd.body_begin + slot), -1, 0, 0});
    return slot;
}
static void split_mark_cluster(bepuhip_ctx* c, int cl) { c->cluster_degraded[cl] = 1; c->soft_items_dirty = true; }

static bool split_remove(bepuhip_ctx* c, HostTypeBatch* tb, int index) {
    split_ensure_mirrors(c);
    const int t = (int)(tb - c->tbs.data());
    const int d = tb->inv[index], last = tb->count - 1, dl = tb->inv[last];
    const int cluster_of_slot = soft_cluster_of_slot(*tb, d);
    soft_ensure_kin_uses(c);
    for (int k = 0; k < tb->info.bodies; ++k) {
        int32_t& r = tb->dev_refs[(size_t)k * tb->stride + d];
        soft_kinematic_reference(c, r, -1);
        if (r >= 0 && (uint32_t)r >= kDynamicLimit) split_release_copy(c, cluster_of_slot, (r & kRefMask) | kSlotKinematic);
        else if (r >= 0 && c->body_cluster[r] != cluster_of_slot) split_release_copy(c, cluster_of_slot, r | kSlotGhost);
        if (r >= 0 && (uint32_t)r < kDynamicLimit) {
            auto& apps = c->body_apps[r];
            for (size_t q = 0; q < apps.size(); ++q) if (apps[q].tb == t && apps[q].slot == d) { apps.erase(apps.begin() + q); break; }
            if (tb->batch < 64) c->body_batches[r] &= ~(1ull << tb->batch);
            if (--c->body_degree[r] == 0) c->soft_orphans.push_back(r);
            if (c->split_shared[r]) split_note_rerank(c, r);
        }
        r = -1;
        tb->plan_lrefs[(size_t)k * tb->stride + d] = kPlanDeadLref;
        tb->plan_ranks[(size_t)k * tb->stride + d] = 0u;
    }
    tb->perm[d] = -1;
    soft_note_slot(c, t, d, false, nullptr, 0);
    if (index != last) { tb->inv[index] = dl; tb->perm[dl] = index; soft_note_index(c, tb, index); }
    tb->inv.pop_back();
    tb->count = last;
    ++c->soft_removes;
    return true;
}

static bool split_add(bepuhip_ctx* c, HostTypeBatch* tb, const int32_t* refs, const float* prestep, bool* violation) {
    split_ensure_mirrors(c);
    const int t = (int)(tb - c->tbs.data()), nb = tb->info.bodies;
    int homes[2] = {-1, -1};
    bool newcomer[2] = {false, false};  // a body without constraints so far: it becomes a body of the cluster that runs the constraint
    int dynamic_bodies = 0, newcomers = 0;
    for (int k = 0; k < nb; ++k) {
        if ((uint32_t)refs[k] >= kDynamicLimit) continue;
        ++dynamic_bodies;
        if ((size_t)refs[k] >= c->shared_bodies) return soft_refuse("a body beyond the split plan's shared-body tables");
        soft_ensure_body(c, refs[k]);
        if (c->body_cluster[refs[k]] < 0) {
            if (env_int("BEPUHIP_SOFT_ORPHANS", 1) == 0) return soft_refuse("the new constraint's body had no constraints");
            newcomer[k] = true; ++newcomers;
            continue;
        }
        if (c->body_degree[refs[k]] >= 255) return soft_refuse("a body with 255 constraints (ranks travel as bytes)");
        if (tb->batch < 64 && (c->body_batches[refs[k]] >> tb->batch) & 1) { *violation = true; return false; }
        homes[k] = c->body_cluster[refs[k]];
    }
    if (nb == 2 && refs[0] == refs[1]) return soft_refuse("a constraint between a body and itself");
    // The cluster that runs it: one with a free device slot in its segment of the type batch and LDS slots for the copies it lacks. Tried in this order: the home of one
    // of its dynamic bodies (the one that already holds a ghost of the other first), then any cluster that already runs a constraint of one of the bodies (it holds
    // that body's slot or ghost; the bodies become shared if they were not), then the next clusters by number.
    if (dynamic_bodies == 0) return soft_refuse("the new constraint has no dynamic body");
    if ((int)c->clustered_dynamic_host.size() + newcomers > c->clustered_dynamic_capacity) return soft_refuse("the plan's body list is full");
    int candidates[2 + 32], ncand = 0;
    auto candidate = [&](int cluster) { if (cluster < 0 || ncand == 34) return; for (int q = 0; q < ncand; ++q) if (candidates[q] == cluster) return; candidates[ncand++] = cluster; };
    if (nb == 2 && homes[0] >= 0 && homes[1] >= 0 && homes[0] != homes[1]) {
        const bool b_in_a = c->cluster_extra[homes[0]].count(refs[1] | kSlotGhost) != 0, a_in_b = c->cluster_extra[homes[1]].count(refs[0] | kSlotGhost) != 0;
        if (!b_in_a && a_in_b) candidate(homes[1]);
    }
    candidate(homes[0]); candidate(homes[1]);
    auto free_slot_of = [&](int cluster) { for (int s = tb->seg_begin[cluster]; s < tb->seg_begin[cluster + 1]; ++s) if (tb->perm[s] < 0) return s; return -1; };
    auto copies_missing = [&](int cluster) {
        int missing = 0;
        for (int k = 0; k < nb; ++k) {
            if (newcomer[k]) { ++missing; continue; }  // its home slot, wherever the constraint runs
            const int32_t key = (uint32_t)refs[k] >= kDynamicLimit ? ((refs[k] & kRefMask) | kSlotKinematic) : (homes[k] == cluster ? -1 : (refs[k] | kSlotGhost));
            if (key >= 0 && !c->cluster_extra[cluster].count(key)) ++missing;
        }
        return missing;
    };
    auto lds_room = [&](int cluster, int missing) { return c->cluster_natural[cluster] + missing - (int)c->cluster_free_slots[cluster].size() <= c->clusters_host[cluster].slot_count; };
    int cl = -1, d = -1;
    bool row_room = false;
    for (int q = 0; q < ncand && cl < 0; ++q) {
        const int slot = free_slot_of(candidates[q]);
        if (slot < 0) continue;
        row_room = true;
        if (lds_room(candidates[q], copies_missing(candidates[q]))) { cl = candidates[q]; d = slot; }
    }
    if (cl < 0) {
        const int homes_only = ncand;
        for (int k = 0; k < nb; ++k)
            if ((uint32_t)refs[k] < kDynamicLimit)
                for (auto& app : c->body_apps[refs[k]]) candidate(soft_cluster_of_slot(c->tbs[app.tb], app.slot));
        for (int q = homes_only; q < ncand && cl < 0; ++q) {
            const int slot = free_slot_of(candidates[q]);
            if (slot < 0) continue;
            row_room = true;
            if (lds_room(candidates[q], copies_missing(candidates[q]))) { cl = candidates[q]; d = slot; }
        }
    }
    if (cl < 0) {  // last resort: the next clusters by number run it on ghosts of both bodies (any cluster can; it costs two shared bodies, a lost plan costs the schedule)
        const int first = homes[0] >= 0 ? homes[0] : (homes[1] >= 0 ? homes[1] : 0), nclusters = (int)c->clusters_host.size();
        for (int step = (homes[0] < 0 && homes[1] < 0) ? 0 : 1; step <= std::min(nclusters - 1, 24) && cl < 0; ++step) {  // (only newcomers: a new island, from cluster 0 on)
            const int cluster = (first + step) % nclusters;
            bool tried = false;
            for (int q = 0; q < ncand; ++q) tried |= candidates[q] == cluster;
            if (tried) continue;
            const int slot = free_slot_of(cluster);
            if (slot < 0) continue;
            row_room = true;
            if (lds_room(cluster, copies_missing(cluster))) { cl = cluster; d = slot; }
        }
    }
    if (cl < 0) return soft_refuse(row_room ? "no free LDS slot in the cluster for a ghost or kinematic copy" : "no free device slot in the cluster's segment of the type batch");
    for (int k = 0; k < nb; ++k)
        if (newcomer[k]) { if (!soft_adopt_body(c, refs[k], cl)) return false; homes[k] = cl; }
    int32_t lrefs[2] = {kPlanDeadLref, kPlanDeadLref};
    for (int k = 0; k < nb; ++k) {
        const int32_t r = refs[k];
        if ((uint32_t)r >= kDynamicLimit) {
            const int32_t key = (r & kRefMask) | kSlotKinematic;
            auto found = c->cluster_extra[cl].find(key);
            int slot = found != c->cluster_extra[cl].end() ? found->second : split_take_lds_slot(c, cl, key);
            if (slot < 0) return soft_refuse("no free LDS slot in the cluster for a kinematic copy");
            c->cluster_extra[cl][key] = slot;
            ++c->cluster_extra_uses[cl][key];
            lrefs[k] = slot | (int)kDynamicLimit;
            continue;
        }
        if (homes[k] == cl) { lrefs[k] = c->body_lref[r] | (c->split_shared[r] ? (int)kLrefShared : 0); }
        else {
            const int32_t key = r | kSlotGhost;
            auto found = c->cluster_extra[cl].find(key);
            int slot = found != c->cluster_extra[cl].end() ? found->second : split_take_lds_slot(c, cl, key);
            if (slot < 0) return soft_refuse("no free LDS slot in the cluster for a ghost copy");
            c->cluster_extra[cl][key] = slot;
            ++c->cluster_extra_uses[cl][key];
            if (!c->split_shared[r]) {  // the body becomes shared: its home's slot table says so, every application of it carries the shared bit from now on
                c->split_shared[r] = 1;
                const ClusterDesc& home = c->clusters_host[homes[k]];
                const size_t entry = (size_t)home.body_begin + (size_t)c->body_lref[r];
                c->cluster_bodies_host[entry] |= kSlotSharedHome;
                c->split_patches.push_back({2, entry, -1, 0, 0});
                for (auto& app : c->body_apps[r]) {
                    HostTypeBatch& other = c->tbs[app.tb];
                    other.plan_lrefs[(size_t)app.k * other.stride + app.slot] |= (int)kLrefShared;
                    c->split_patches.push_back({0, other.lrefs_off + (size_t)(app.k / 2) * other.stride + app.slot, app.tb, app.slot, app.k / 2});
                }
                split_mark_cluster(c, homes[k]);
            }
            lrefs[k] = slot | (int)kLrefShared;
        }
        if (c->split_shared[r]) split_note_rerank(c, r);
    }
    // the prestep lane; references, local references and rank words are taken from the mirrors when the updates are flushed
    static_assert(sizeof(float) == sizeof(uint32_t), "the prestep lane is noted as words");
    soft_note_slot(c, t, d, true, reinterpret_cast<const uint32_t*>(prestep), (size_t)tb->info.prestep);
    soft_ensure_kin_uses(c);
    for (int k = 0; k < nb; ++k) {
        tb->dev_refs[(size_t)k * tb->stride + d] = refs[k];
        soft_kinematic_reference(c, refs[k], +1);
        tb->plan_lrefs[(size_t)k * tb->stride + d] = lrefs[k];
        tb->plan_ranks[(size_t)k * tb->stride + d] = 0u;
        if ((uint32_t)refs[k] < kDynamicLimit) {
            auto& apps = c->body_apps[refs[k]];
            auto at = std::upper_bound(apps.begin(), apps.end(), t, [](int value, const bepuhip_ctx::SplitApp& a) { return value < a.tb; });
            apps.insert(at, {(int32_t)t, d, k});
            ++c->body_degree[refs[k]];
            if (tb->batch < 64) c->body_batches[refs[k]] |= 1ull << tb->batch;
        }
    }
    tb->perm[d] = tb->count;
    tb->inv.push_back(d);
    soft_note_index(c, tb, tb->count);
    tb->count += 1;
    split_mark_cluster(c, cl);
    ++c->soft_adds;
    return true;
}

// Renumber the applications of a shared body (they are kept in batch order) and write their rank words: to the mirror, and to the device unless the slot is
// written as a whole by this flush anyway. Every cluster that holds one of them gets its predecessor lists rebuilt.
static void split_rerank_body(bepuhip_ctx* c, int32_t body, bool local_handoff, std::vector<bepuhip_ctx::WordPatch>& patches, std::vector<int>& touched_clusters) {
    auto& apps = c->body_apps[body];
    const uint32_t degree = (uint32_t)apps.size();
    int cluster[256];
    for (size_t q = 0; q < apps.size(); ++q) cluster[q] = soft_cluster_of_slot(c->tbs[apps[q].tb], apps[q].slot);
    for (size_t q = 0; q < apps.size(); ++q) {
        HostTypeBatch& tb = c->tbs[apps[q].tb];
        uint32_t word = (uint32_t)q | (degree << 8);
        if (local_handoff && q > 0 && cluster[q - 1] == cluster[q]) word |= kSoftRankPredLocal;
        if (local_handoff && q + 1 < apps.size() && cluster[q + 1] == cluster[q]) word |= kSoftRankSuccLocal;
        uint32_t& mirror = tb.plan_ranks[(size_t)apps[q].k * tb.stride + apps[q].slot];
        if (mirror != word) patches.push_back({0, tb.lrefs_off + (size_t)((tb.info.bodies + 1) / 2 + apps[q].k) * tb.stride + apps[q].slot, apps[q].tb, apps[q].slot, (tb.info.bodies + 1) / 2 + apps[q].k});
        mirror = word;
        touched_clusters.push_back(cluster[q]);
    }
    patches.push_back({1, (size_t)body, -1, 0, 0});
}

// The predecessor lists of one cluster of a split plan: the planner's rule (plan_split_clusters) over the mirrors.
static void split_rebuild_items(bepuhip_ctx* c, int cl) {
    const ClusterDesc& cd = c->clusters_host[cl];
    std::vector<int32_t> lt(cd.slot_count + 16, -1);
    std::vector<std::pair<int32_t, int32_t>> first_touch;
    for (int self = 0; self < cd.item_count; ++self) {
        ClusterItem& it = c->items_host[cd.item_begin + self];
        const HostTypeBatch& tb = c->tbs[it.tb];
        const int nb = tb.info.bodies;
        int npred = 0, overflow = 0;
        memset(it.pred, 0, sizeof(it.pred)); memset(it.xpred, 0, sizeof(it.xpred));
        for (int j = it.start; j < it.start + it.count; ++j)
            for (int k = 0; k < nb; ++k) {
                int32_t lr = tb.plan_lrefs[(size_t)k * tb.stride + j];
                if ((uint32_t)lr >= kDynamicLimit || tb.perm[j] < 0) continue;
                const bool is_shared = (lr & (int)kLrefShared) != 0;
                if (is_shared && !(tb.plan_ranks[(size_t)k * tb.stride + j] & kSoftRankPredLocal)) continue;
                lr &= ~(int)kLrefShared;
                const int pred = lt[lr];
                if (pred < 0) { if (!is_shared) first_touch.push_back({self, lr}); continue; }
                if (pred == self) continue;
                bool known = false;
                for (int q = 0; q < npred; ++q) known |= it.pred[q] == pred;
                if (known) continue;
                if (npred < kMaxPreds) it.pred[npred++] = (unsigned short)pred; else overflow = 1;
            }
        for (int j = it.start; j < it.start + it.count; ++j)
            for (int k = 0; k < nb; ++k) {
                const int32_t lr = tb.plan_lrefs[(size_t)k * tb.stride + j];
                if ((uint32_t)lr >= kDynamicLimit || tb.perm[j] < 0) continue;
                if (!(lr & (int)kLrefShared)) lt[lr] = self;
                else if (tb.plan_ranks[(size_t)k * tb.stride + j] & kSoftRankSuccLocal) lt[lr & ~(int)kLrefShared] = self;
            }
        if (overflow) npred = 0;
        it.batch_npred = (it.batch_npred & 0xFFFF) | (npred << 16) | (overflow << 24);
    }
    for (auto& fs : first_touch) {
        ClusterItem& it = c->items_host[cd.item_begin + fs.first];
        const int last = lt[fs.second];
        int nx = (it.batch_npred >> 20) & 0xF;
        if ((it.batch_npred >> 25) & 1) continue;
        bool known = false;
        for (int q = 0; q < nx; ++q) known |= it.xpred[q] == last;
        if (known) continue;
        if (nx < kMaxPreds) { it.xpred[nx++] = (unsigned short)last; it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (nx << 20); }
        else it.batch_npred = (it.batch_npred & ~(0xF << 20)) | (1 << 25);
    }
}

// The words of the split plan's tables the flush has to write, with the values the mirrors hold NOW: every word once (the patch kernel writes each patch from its own
// lane), and none of a device slot the flush writes as a whole anyway.
struct ResolvedWord { int table; size_t index; uint32_t value; };
static std::vector<ResolvedWord> split_resolve_patches(bepuhip_ctx* c) {
    std::vector<ResolvedWord> words;
    words.reserve(c->split_patches.size());
    // every word once: an open-addressing set over the patch keys (a node-based set allocated per patch: a third of the listing's time on the pile's churn)
    size_t capacity = 64;
    while (capacity < c->split_patches.size() * 2) capacity *= 2;
    std::vector<uint64_t> seen(capacity, ~0ull);
    auto first_time = [&](uint64_t key) {
        size_t at = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & (capacity - 1);
        while (seen[at] != ~0ull) { if (seen[at] == key) return false; at = (at + 1) & (capacity - 1); }
        seen[at] = key;
        return true;
    };
    for (auto& wp : c->split_patches) {
        if (!first_time(((uint64_t)wp.table << 60) | (uint64_t)wp.index)) continue;
        if (wp.table == 0) {
            if (soft_slot_noted(c, wp.tb, wp.slot)) continue;
            const HostTypeBatch& tb = c->tbs[wp.tb];
            const int lref_rows = (tb.info.bodies + 1) / 2;
            words.push_back({0, wp.index, wp.row < lref_rows ? split_packed_lrefs(tb, wp.slot, wp.row) : tb.plan_ranks[(size_t)(wp.row - lref_rows) * tb.stride + wp.slot]});
        } else if (wp.table == 3) {  // a body reference (applied after the whole-slot writes of the same flush: a slot's payload may still carry the old index)
            const HostTypeBatch& tb = c->tbs[wp.tb];
            words.push_back({0, wp.index, (uint32_t)tb.dev_refs[(size_t)wp.row * tb.stride + wp.slot]});
        } else if (wp.table == 1) words.push_back({1, wp.index, c->split_shared[wp.index] ? (uint32_t)c->body_apps[wp.index].size() : 0u});  // a private body's entry is 0 (an index may change hands)
        else words.push_back({2, wp.index, (uint32_t)c->cluster_bodies_host[wp.index]});
    }
    return words;
}

// Everything the soft updates changed since the last flush, onto the device (both slabs: the snapshot follows, like every other structural update).
// The host half: ranks of the shared bodies whose applications changed, predecessor lists of the clusters that received constraints (after it the mirrors describe the
// layout the device is about to get; tools/plan_harness validates them without a device).
static void flush_soft_host(bepuhip_ctx* c) {
    if (c->soft_split && !c->split_rerank_list.empty()) {  // every body on its own: host threads, their patches and cluster marks merged afterwards
        std::vector<int32_t> bodies;
        bodies.reserve(c->split_rerank_list.size());
        for (int32_t body : c->split_rerank_list) if (c->split_rerank_flag[body]) { c->split_rerank_flag[body] = 0; bodies.push_back(body); }  // (a moved body's old index stays listed with its flag down)
        c->split_rerank_list.clear();
        const bool local_handoff = env_int("BEPUHIP_SPLIT_LOCAL_HANDOFF", 1) != 0;
        const size_t chunk = 256, jobs = (bodies.size() + chunk - 1) / chunk;
        std::vector<std::vector<bepuhip_ctx::WordPatch>> patches(jobs);
        std::vector<std::vector<int>> touched(jobs);
        plan_parallel_for(jobs, [&](size_t j) {
            for (size_t i = j * chunk; i < std::min(bodies.size(), (j + 1) * chunk); ++i) split_rerank_body(c, bodies[i], local_handoff, patches[j], touched[j]);
        });
        for (size_t j = 0; j < jobs; ++j) {
            c->split_patches.insert(c->split_patches.end(), patches[j].begin(), patches[j].end());
            for (int cl : touched[j]) split_mark_cluster(c, cl);
        }
    }
    if (c->soft_items_dirty) {  // clusters that received constraints: their items' predecessor lists, on a few host threads
        std::vector<int> dirty;
        for (size_t cl = 0; cl < c->cluster_degraded.size(); ++cl) if (c->cluster_degraded[cl]) { dirty.push_back((int)cl); c->cluster_degraded[cl] = 0; }
        plan_parallel_for(dirty.size(), [&](size_t i) { if (c->soft_split) split_rebuild_items(c, dirty[i]); else soft_rebuild_items(c, dirty[i]); });
    }
}
static int32_t rebuild_flags(bepuhip_ctx* c);
static int32_t flush_soft(bepuhip_ctx* c) {
    c->body_moves.clear();
    if (c->soft_records.empty() && !c->soft_index_dirty && !c->soft_items_dirty && c->split_rerank_list.empty() && c->split_patches.empty() && c->kin_touched.empty() && !c->clustered_dirty &&
        !c->kinlist_dirty)
        return BEPUHIP_OK;
    if (c->clustered_dirty) {  // bodies joined or left the plan: the list behind kFlagClustered
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (!c->clustered_dynamic_host.empty())
            HIP_TRY(copy_sync(c, c->d_clustered_dynamic, c->clustered_dynamic_host.data(), c->clustered_dynamic_host.size() * 4, hipMemcpyHostToDevice));
        c->clustered_dynamic_count = (int)c->clustered_dynamic_host.size();
        c->clustered_dirty = false;
    }
    if (soft_update_kinlist(c) || c->kinlist_dirty) {
        c->kinlist_dirty = false;  // a kinematic body gained its first or lost its last constraint: the kinematic workgroup's list and the body flags follow
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_kinlist) { hipFree(c->d_kinlist); c->d_kinlist = nullptr; }
        c->kinlist_count = (int)c->kinlist_host.size();
        if (c->kinlist_count > 0) {
            HIP_TRY(hipMalloc((void**)&c->d_kinlist, c->kinlist_host.size() * 4));
            HIP_TRY(copy_sync(c, c->d_kinlist, c->kinlist_host.data(), c->kinlist_host.size() * 4, hipMemcpyHostToDevice));
        }
        c->soft_flags_stale = true;
    }
    const bool timing = env_int("BEPUHIP_PLAN_STATS", 0) >= 2;
    const auto t_begin = std::chrono::steady_clock::now();
    flush_soft_host(c);
    const auto t_host = std::chrono::steady_clock::now();
    std::vector<SoftSlotOp> ops;
    std::vector<uint32_t> payload(1, 0u);
    ops.reserve(c->soft_records.size());
    payload.reserve(c->soft_payload.size() + c->soft_records.size() * 6 + 1);
    for (const auto& record : c->soft_records) {
        const HostTypeBatch& tb = c->tbs[record.tb];
        const int d = record.slot;
        SoftSlotOp op{(unsigned)tb.refs_off, (unsigned)tb.lrefs_off, (unsigned)tb.prestep_off, (unsigned)tb.accum_off, tb.stride, tb.info.bodies, tb.info.prestep, tb.info.impulse,
                      d, (int)record.live, (unsigned)payload.size(), c->soft_split ? tb.info.bodies : 0};
        if (c->soft_split && record.live) {  // references, packed local references and rank words as the mirrors hold them now (the re-ranking above included)
            for (int k = 0; k < tb.info.bodies; ++k) payload.push_back((uint32_t)tb.dev_refs[(size_t)k * tb.stride + d]);
            for (int row = 0; row < (tb.info.bodies + 1) / 2; ++row) payload.push_back(split_packed_lrefs(tb, d, row));
            for (int k = 0; k < tb.info.bodies; ++k) payload.push_back(tb.plan_ranks[(size_t)k * tb.stride + d]);
        }
        if (record.live) payload.insert(payload.end(), c->soft_payload.begin() + record.payload_at, c->soft_payload.begin() + record.payload_at + record.payload_words);
        ops.push_back(op);
    }
    std::vector<IndexPatch> patches;
    if (c->soft_index_dirty)
        for (auto& tb : c->tbs) {  // the caller's indices whose device slot changed: the device copy of `inv` follows `inv` (an index beyond the count was removed since)
            if (!tb.d_device_index) continue;  // built from `inv` on first use: nothing to patch yet
            for (int32_t index : tb.soft_dirty_indices) if (index < tb.count) patches.push_back(IndexPatch{tb.d_device_index, index, tb.inv[index], 0});
        }
    for (auto& word : split_resolve_patches(c)) {  // single words of the split plan's tables (rank words, local references, degrees, slot table entries)
        if (word.table == 0) { for (uint32_t* slab : {c->d_slab, c->d_slab0}) if (slab) patches.push_back(IndexPatch{(int*)slab, (int)word.index, (int)word.value, 0}); }
        else if (word.table == 1) patches.push_back(IndexPatch{(int*)c->d_shared_info, (int)word.index, (int)word.value, 0});
        else patches.push_back(IndexPatch{c->d_cluster_bodies, (int)word.index, (int)word.value, 0});
    }
    c->split_patches.clear();
    const auto t_lists = std::chrono::steady_clock::now();
    // One transfer: [slot operations][word patches][payload][the work items, if their lists changed], built in pinned memory, one copy, the kernels behind it.
    const size_t ops_bytes = ops.size() * sizeof(SoftSlotOp), patch_bytes = patches.size() * sizeof(IndexPatch), payload_bytes = payload.size() * 4;
    const size_t items_bytes = c->soft_items_dirty ? c->items_host.size() * sizeof(ClusterItem) : 0;
    const size_t table_bytes = (ops_bytes + patch_bytes + payload_bytes + 63) / 64 * 64, bytes = table_bytes + items_bytes;
    if (bytes > c->h_flush_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->h_flush) hipHostFree(c->h_flush);
        if (c->d_flush) hipFree(c->d_flush);
        c->h_flush = nullptr; c->d_flush = nullptr; c->h_flush_bytes = c->d_flush_bytes = 0;
        const size_t room = std::max(bytes + bytes / 2, (size_t)1 << 20);
        HIP_TRY(hipHostMalloc((void**)&c->h_flush, room, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void**)&c->d_flush, room));
        c->h_flush_bytes = c->d_flush_bytes = room;
    }
    char* h = c->h_flush;
    char* d = c->d_flush;
    if (ops_bytes) memcpy(h, ops.data(), ops_bytes);
    if (patch_bytes) memcpy(h + ops_bytes, patches.data(), patch_bytes);
    memcpy(h + ops_bytes + patch_bytes, payload.data(), payload_bytes);
    if (items_bytes) memcpy(h + table_bytes, c->items_host.data(), items_bytes);
    SoftSlotOp* d_ops = (SoftSlotOp*)d;
    IndexPatch* d_patches = (IndexPatch*)(d + ops_bytes);
    unsigned* d_payload = (unsigned*)(d + ops_bytes + patch_bytes);
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    if (!ops.empty())
        for (uint32_t* slab : {c->d_slab, c->d_slab0})
            if (slab) hipLaunchKernelGGL(apply_soft_slots_kernel, dim3(((int)ops.size() + 63) / 64), dim3(64), 0, c->stream, slab, (const SoftSlotOp*)d_ops, (int)ops.size(), (const unsigned*)d_payload);
    if (!patches.empty()) hipLaunchKernelGGL(patch_index_kernel, dim3(((int)patches.size() + 63) / 64), dim3(64), 0, c->stream, (const IndexPatch*)d_patches, (int)patches.size());
    if (items_bytes) HIP_TRY(hipMemcpyAsync(c->d_items, d + table_bytes, items_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));  // the staging buffer is the next flush's too (and the caller's next structural calls change the mirrors this flush read)
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "bepuhip flush of structural updates: %ld calls took %.3f ms before it; ranks + predecessor lists %.3f ms, %zu slot writes + %zu word patches listed %.3f ms, device %.3f ms\n",
                c->soft_calls, c->soft_call_ms, ms(t_begin, t_host), ops.size(), patches.size(), ms(t_host, t_lists), ms(t_lists, std::chrono::steady_clock::now()));
        c->soft_calls = 0; c->soft_call_ms = 0.0;
    }
    soft_clear_notes(c); c->soft_items_dirty = false;
    c->total_constraints = 0;
    for (auto& tb : c->tbs) c->total_constraints += tb.count;
    if (c->soft_flags_stale) { c->soft_flags_stale = false; return rebuild_flags(c); }  // after the slots have their references: the flags are derived from them
    return BEPUHIP_OK;
}
