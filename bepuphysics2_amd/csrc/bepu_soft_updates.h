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

static bool soft_refuse(const char* why) {
    if (env_int("BEPUHIP_PLAN_STATS", 0)) fprintf(stderr, "bepuhip: leaving the island schedule: %s\n", why);
    return false;
}
static int soft_cluster_of_slot(const HostTypeBatch& tb, int slot) {
    return (int)(std::upper_bound(tb.seg_begin.begin(), tb.seg_begin.end(), slot) - tb.seg_begin.begin()) - 1;
}

static void soft_setup(bepuhip_ctx* c, ClusterPlan& plan) {
    c->soft_ok = false;
    c->soft_slots.clear(); c->soft_index.clear(); c->soft_items_dirty = false; c->soft_adds = c->soft_removes = 0;
    if (!plan.enabled || plan.shared || env_int("BEPUHIP_NO_SOFT_UPDATES", 0)) return;
    c->body_cluster.swap(plan.body_cluster); c->body_lref.swap(plan.body_lref); c->body_degree.clear(); c->body_batches.clear(); c->cluster_kin.swap(plan.cluster_kin);
    c->items_host = plan.items; c->clusters_host = plan.clusters;
    c->cluster_degraded.assign(plan.clusters.size(), 0);
    c->soft_ok = true;
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

// TypeProcessor.Remove on the island layout. false: not possible here (nothing was changed).
static bool soft_remove(bepuhip_ctx* c, HostTypeBatch* tb, int index) {
    if (!c->soft_ok || tb->slots == 0 || tb->info.bodies > 2) return soft_refuse("removal from a type batch the island layout does not manage");
    soft_ensure_degrees(c);
    const int t = (int)(tb - c->tbs.data());
    const int d = tb->inv[index], last = tb->count - 1, dl = tb->inv[last];
    for (int k = 0; k < tb->info.bodies; ++k) {
        int32_t& r = tb->dev_refs[(size_t)k * tb->stride + d];
        // a body whose last constraint goes would have to leave the plan (the reference integrates it as an unconstrained body from then on) — unless the same
        // batch of updates gives it a constraint again (a refreshed pair): decided when the updates are flushed (soft_bodies_still_constrained)
        if (r >= 0 && (uint32_t)r < kDynamicLimit) {
            if (tb->batch < 64) c->body_batches[r] &= ~(1ull << tb->batch);
            if (--c->body_degree[r] == 0) c->soft_orphans.push_back(r);
        }
        r = -1;
    }
    tb->perm[d] = -1;
    c->soft_slots[{t, d}] = bepuhip_ctx::SoftSlot{false, {}};
    if (index != last) {  // TypeProcessor.Move (:578-592): the last constraint takes the removed one's index; on the device it stays where it is
        tb->inv[index] = dl; tb->perm[dl] = index;
        c->soft_index[{t, index}] = dl;
    }
    c->soft_index.erase({t, last});
    tb->inv.pop_back();
    tb->count = last;
    ++c->soft_removes;
    return true;
}

// True when no body lost its last constraint since the last flush (see soft_remove).
static bool soft_bodies_still_constrained(bepuhip_ctx* c) {
    bool ok = true;
    for (int32_t body : c->soft_orphans) ok &= c->body_degree[body] > 0;
    c->soft_orphans.clear();
    return ok || soft_refuse("a body lost its last constraint");
}

// TypeProcessor.AllocateInTypeBatch on the island layout. false: not possible here (nothing was changed).
static bool soft_add(bepuhip_ctx* c, HostTypeBatch* tb, const int32_t* refs, const float* prestep, bool* violation) {
    if (!c->soft_ok || tb->slots == 0 || tb->info.bodies > 2) return soft_refuse("addition to a type batch the island layout does not manage");
    soft_ensure_degrees(c);
    const int t = (int)(tb - c->tbs.data()), nb = tb->info.bodies;
    int cl = -1;
    for (int k = 0; k < nb; ++k) {
        if ((uint32_t)refs[k] >= kDynamicLimit) continue;
        if (refs[k] >= (int)c->body_cluster.size() || c->body_cluster[refs[k]] < 0) return soft_refuse("the new constraint's body had no constraints");
        if (cl >= 0 && c->body_cluster[refs[k]] != cl) return soft_refuse("the new constraint's bodies live in two clusters");
        cl = c->body_cluster[refs[k]];
    }
    if (cl < 0) return soft_refuse("the new constraint has no dynamic body");
    // The batch invariant the whole solve rests on (Solver.cs:1046-1051, asserted by the reference in debug builds): a dynamic body appears at most once per synchronized
    // batch. A caller that breaks it would get a silent race on the body's velocity inside one work item or launch: refused here, where the host knows the references.
    for (int k = 0; k < nb; ++k)
        if ((uint32_t)refs[k] < kDynamicLimit && tb->batch < 64 && (c->body_batches[refs[k]] >> tb->batch) & 1) { *violation = true; return false; }
    unsigned halves[2] = {0u, 0u};
    for (int k = 0; k < nb; ++k) {
        if ((uint32_t)refs[k] < kDynamicLimit) { halves[k] = (unsigned)c->body_lref[refs[k]]; continue; }
        auto copy = c->cluster_kin[cl].find(refs[k] & kRefMask);
        if (copy == c->cluster_kin[cl].end()) return soft_refuse("the cluster holds no copy of the new constraint's kinematic body");
        halves[k] = (unsigned)copy->second | 0x8000u;
    }
    int d = -1;
    for (int s = tb->seg_begin[cl]; s < tb->seg_begin[cl + 1] && d < 0; ++s) if (tb->perm[s] < 0) d = s;
    if (d < 0) return soft_refuse("no free device slot in the cluster's segment of the type batch");
    bepuhip_ctx::SoftSlot slot{true, {}};
    for (int k = 0; k < nb; ++k) {
        slot.payload.push_back((uint32_t)refs[k]);
        tb->dev_refs[(size_t)k * tb->stride + d] = refs[k];
        if ((uint32_t)refs[k] < kDynamicLimit) { ++c->body_degree[refs[k]]; if (tb->batch < 64) c->body_batches[refs[k]] |= 1ull << tb->batch; }
    }
    slot.payload.push_back(halves[0] | (halves[1] << 16));
    for (int f = 0; f < tb->info.prestep; ++f) { uint32_t w; memcpy(&w, &prestep[f], 4); slot.payload.push_back(w); }
    c->soft_slots[{t, d}] = std::move(slot);
    tb->perm[d] = tb->count;
    tb->inv.push_back(d);
    c->soft_index[{t, tb->count}] = d;
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

// Everything the soft updates changed since the last flush, onto the device (both slabs: the snapshot follows, like every other structural update).
static int32_t flush_soft(bepuhip_ctx* c) {
    if (c->soft_slots.empty() && c->soft_index.empty() && !c->soft_items_dirty) return BEPUHIP_OK;
    if (c->soft_items_dirty) {  // clusters that received constraints: their items' predecessor lists, on a few host threads
        std::vector<int> dirty;
        for (size_t cl = 0; cl < c->cluster_degraded.size(); ++cl) if (c->cluster_degraded[cl]) { dirty.push_back((int)cl); c->cluster_degraded[cl] = 0; }
        const int workers = std::max(1, std::min<int>({env_int("BEPUHIP_PLAN_THREADS", 8), (int)std::thread::hardware_concurrency(), (int)dirty.size()}));
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < dirty.size();) soft_rebuild_items(c, dirty[i]); };
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; ++w) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    std::vector<SoftSlotOp> ops;
    std::vector<uint32_t> payload(1, 0u);
    for (auto& kv : c->soft_slots) {
        const HostTypeBatch& tb = c->tbs[kv.first.first];
        SoftSlotOp op{(unsigned)tb.refs_off, (unsigned)tb.lrefs_off, (unsigned)tb.prestep_off, (unsigned)tb.accum_off, tb.stride, tb.info.bodies, tb.info.prestep, tb.info.impulse,
                      kv.first.second, kv.second.live ? 1 : 0, (unsigned)payload.size(), 0};
        payload.insert(payload.end(), kv.second.payload.begin(), kv.second.payload.end());
        ops.push_back(op);
    }
    std::vector<IndexPatch> patches;
    for (auto& kv : c->soft_index) {
        HostTypeBatch& tb = c->tbs[kv.first.first];
        if (!tb.d_device_index) continue;  // built from `inv` on first use: nothing to patch yet
        patches.push_back(IndexPatch{tb.d_device_index, kv.first.second, kv.second, 0});
    }
    const size_t bytes = ops.size() * sizeof(SoftSlotOp) + payload.size() * 4 + patches.size() * sizeof(IndexPatch) + 64;
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, bytes));
    SoftSlotOp* d_ops = (SoftSlotOp*)d;
    IndexPatch* d_patches = (IndexPatch*)(d + ops.size() * sizeof(SoftSlotOp));
    unsigned* d_payload = (unsigned*)(d + ops.size() * sizeof(SoftSlotOp) + patches.size() * sizeof(IndexPatch));
    if (!ops.empty()) HIP_TRY(hipMemcpyAsync(d_ops, ops.data(), ops.size() * sizeof(SoftSlotOp), hipMemcpyHostToDevice, c->stream));
    if (!patches.empty()) HIP_TRY(hipMemcpyAsync(d_patches, patches.data(), patches.size() * sizeof(IndexPatch), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_payload, payload.data(), payload.size() * 4, hipMemcpyHostToDevice, c->stream));
    if (!ops.empty())
        for (uint32_t* slab : {c->d_slab, c->d_slab0})
            if (slab) hipLaunchKernelGGL(apply_soft_slots_kernel, dim3(((int)ops.size() + 63) / 64), dim3(64), 0, c->stream, slab, (const SoftSlotOp*)d_ops, (int)ops.size(), (const unsigned*)d_payload);
    if (!patches.empty()) hipLaunchKernelGGL(patch_index_kernel, dim3(((int)patches.size() + 63) / 64), dim3(64), 0, c->stream, (const IndexPatch*)d_patches, (int)patches.size());
    if (c->soft_items_dirty) HIP_TRY(hipMemcpyAsync(c->d_items, c->items_host.data(), c->items_host.size() * sizeof(ClusterItem), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(d);
    c->soft_slots.clear(); c->soft_index.clear(); c->soft_items_dirty = false;
    c->total_constraints = 0;
    for (auto& tb : c->tbs) c->total_constraints += tb.count;
    return BEPUHIP_OK;
}
