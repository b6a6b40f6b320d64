// Offline harness of the host-side planner (a developer tool, CPU only, not part of the product): includes the library's translation unit, builds a context by hand
// (no HIP call is made: no device is needed), feeds it a scene dumped by dump_scene.py, runs the host phases of bepuhip_end_constraints — the AOSOA -> row conversion of
// set_type_batch and plan_clusters — and prints their timings, the plan's shape and a 64-bit digest of everything the plan hands to the device, so that a change of the
// planner's implementation can be shown to leave its output byte-identical without a GPU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../include -o plan_harness plan_harness.hip && ./plan_harness scene.bin
#include "../../bepuphysics2_amd/csrc/bepuhip.hip"

#include <cstdio>

// the cluster_kernel variants live in their own translation units; nothing is launched here
#define BEPU_STUB(name) const void* name(bool) { return nullptr; }
BEPU_STUB(bepu_cluster_kernel_hot_1024) BEPU_STUB(bepu_cluster_kernel_hot_768) BEPU_STUB(bepu_cluster_kernel_hot_512)
BEPU_STUB(bepu_cluster_kernel_wide_1024) BEPU_STUB(bepu_cluster_kernel_wide_768) BEPU_STUB(bepu_cluster_kernel_wide_512)
BEPU_STUB(bepu_cluster_kernel_hot_1024n) BEPU_STUB(bepu_cluster_kernel_wide_1024n) BEPU_STUB(bepu_cluster_kernel_hot_512sn) BEPU_STUB(bepu_cluster_kernel_wide_512sn)
BEPU_STUB(bepu_cluster_kernel_hot_1024s) BEPU_STUB(bepu_cluster_kernel_wide_1024s) BEPU_STUB(bepu_cluster_kernel_hot_768s) BEPU_STUB(bepu_cluster_kernel_wide_768s)
BEPU_STUB(bepu_cluster_kernel_hot_512s) BEPU_STUB(bepu_cluster_kernel_wide_512s)

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
template <class T> static uint64_t fnv_vec(uint64_t h, const std::vector<T>& v) { return v.empty() ? h : fnv(h, v.data(), v.size() * sizeof(T)); }

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
    for (int rep = 0; rep < repeats; ++rep) {
        bepuhip_ctx* c = new bepuhip_ctx();
        c->device = 0; c->W = header[0]; c->flags = argc > 3 ? atoi(argv[3]) : 0;
        c->batch_count = header[1]; c->fallback_threshold = header[3]; c->has_fallback = header[1] > header[3]; c->building = true;
        auto t0 = std::chrono::steady_clock::now();
        for (auto& t : tbs)
            if (bepuhip_set_type_batch(c, t.batch, t.type, t.count, t.refs.data(), t.prestep.data(), t.accum.data()) != BEPUHIP_OK) { fprintf(stderr, "set_type_batch: %s\n", bepuhip_last_error()); return 1; }
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
        delete c;
    }
    return 0;
}
