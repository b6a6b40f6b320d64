// libbepuhip — MI355X (gfx950) native constraint solver + pose integrator behind the C ABI of include/bepuhip.h.
//
// Data layout in HBM
//   bodies        : the reference's BodyDynamics AoS, 128 B per body = 8 x float4 (BodyProperties.cs:318-338):
//                   [0] orientation xyzw  [1] position xyz  [2] linear xyz  [3] angular xyz
//                   [4..5] local inverse inertia {xx,yx,yy,zx | zy,zz,invMass}  [6..7] world inverse inertia.
//                   One 128-byte line per gathered body; every access is a 16-byte vector load/store.
//   type batches  : per (batch, type) SoA slabs — refs[slot][stride], prestep[field][stride], accumulated[field][stride],
//                   stride = count rounded up to 64 lanes, so lane i of a wavefront reads consecutive dwords (coalesced 256 B/wave/field).
//                   Converted from the host's AOSOA (BundleIndexing.cs:50-60) on upload and back on download.
// Schedule (no host sync inside a frame; replayed from a hipGraph):
//   per substep: [incremental contact update, all batches, one launch] -> [integrate constrained bodies, one launch]
//                -> warm start: one launch per batch (all constraint types of the batch in one grid)
//                -> velocity iterations: one launch per batch per iteration
//   then one final pose-integration launch over all bodies.
// Integration is hoisted out of the first-touching constraint's warm start into the per-substep body kernel; SURVEY.md A.2
// gives the argument that this is value-identical (nothing reads a body between its integration and its first constraint).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>
#include <atomic>
#include <memory>
#include <thread>

#include <rccl/rccl.h>  // types and enums only: the library is opened at run time (bepuhip_comm_*), the solver itself does not depend on it

#include "../../include/bepuhip.h"
#include "bepu_device_constraints.h"
#include "bepu_device_bounds.h"

#pragma clang fp contract(off)

using namespace bd;

// Split by concern:
//   bepu_device_math.h / bepu_device_constraints.h  scalar-per-lane math and the constraint functions (one lane = one constraint)
//   bepu_kernels_common.h    descriptors, the body record, gather/scatter by access filter
//   bepu_batch_kernels.h     launch-per-batch schedule, per-body integration kernels, boundary exchange, ranged update transposes
//   bepu_cluster_kernel.h    island-per-workgroup schedule (bodies resident in LDS, work-item dataflow); compiled in bepu_cluster_*.hip, one unit per register budget
//   bepu_host_state.h        context, type table, error plumbing;  bepu_cluster_plan.h  host planning of the island schedule
//   this file                the C ABI of include/bepuhip.h: uploads, graph capture / launch sequence, read-backs
#include "bepu_kernels_common.h"
#include "bepu_batch_kernels.h"
#include "bepu_transfer_kernels.h"
#include "bepu_colour_kernels.h"
#include "bepu_host_state.h"
#include "bepu_cluster_plan.h"
#include "bepu_soft_updates.h"
#include "bepu_unit_cache.h"

#include <atomic>
#include <memory>
#include <thread>

// bepuhip_replan_begin / _commit (round 6): a re-plan whose planning runs on a host thread of its own while the frames go on. The job owns a host-only shadow of the
// context — the type batches with the body references as the device held them when the job began, and the handful of fields the planner reads — the plan the worker
// computes for them, and the log of every structural operation the caller has made since (the public calls' own arguments: bepuhip_structural_op + payload words).
struct ReplanJob {
    std::thread worker;
    std::atomic<int> done{0};
    bepuhip_ctx shadow;
    ClusterPlan plan;
    std::vector<std::vector<int32_t>> fallback_refs;  // the fallback type batches' references in the caller's order (build_constraints' level walk wants them unpermuted)
    int batch_count = 0;
    bool has_fallback = false;
    std::vector<bepuhip_structural_op> log;
    std::vector<uint32_t> payload;
    double begin_ms = 0.0, plan_ms = 0.0;
    std::chrono::steady_clock::time_point started;
};

extern "C" {

static void cancel_replan_job(bepuhip_ctx* c);

const char* bepuhip_last_error(void) { return g_last_error.c_str(); }

int32_t bepuhip_type_info(int32_t type_id, int32_t* bodies, int32_t* prestep_floats, int32_t* impulse_floats) {
    TypeInfoH t;
    if (!type_info(type_id, t)) return fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(type_id));
    if (bodies) *bodies = t.bodies;
    if (prestep_floats) *prestep_floats = t.prestep;
    if (impulse_floats) *impulse_floats = t.impulse;
    return BEPUHIP_OK;
}

static int32_t create_device_objects(bepuhip_ctx* c) {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev_start));
    HIP_TRY(hipEventCreate(&c->ev_stop));
    HIP_TRY(hipHostMalloc((void**)&c->d_status, 256, hipHostMallocMapped | hipHostMallocCoherent));  // host-visible while a kernel runs
    HIP_TRY(hipMalloc((void**)&c->d_staged, 4));
    HIP_TRY(fill_async(c, c->d_staged, 0, 4));
    memset(c->d_status, 0, 256);
    return BEPUHIP_OK;
}

int32_t bepuhip_destroy(bepuhip_ctx* c);

// Live contexts per device (each holds one stream): what group_queue_check counts.
constexpr int kMaxCountedDevices = 64;
static std::atomic<int> g_live_contexts[kMaxCountedDevices];

int32_t bepuhip_create(const bepuhip_config* config, bepuhip_ctx** out_ctx) {
    if (!config || !out_ctx) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (config->bundle_width != 4 && config->bundle_width != 8 && config->bundle_width != 16)
        return fail(BEPUHIP_E_INVALID_ARGUMENT, "bundle_width must be 4, 8 or 16");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(BEPUHIP_E_DEVICE, "no HIP device available (the bepuhip product path has no CPU fallback)");
    if (config->device_ordinal < 0 || config->device_ordinal >= n) return fail(BEPUHIP_E_INVALID_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(config->device_ordinal));
    bepuhip_ctx* c = new bepuhip_ctx();
    c->device = config->device_ordinal;
    c->W = config->bundle_width;
    c->flags = config->flags;
    c->specialise_auto = env_int("BEPUHIP_SPECIALISE", 0) != 0;
    const int32_t st = create_device_objects(c);
    if (c->device < kMaxCountedDevices) g_live_contexts[c->device].fetch_add(1);
    if (st != BEPUHIP_OK) { bepuhip_destroy(c); return st; }  // the message of the failing call stays in bepuhip_last_error
    *out_ctx = c;
    return BEPUHIP_OK;
}

static void free_boundary_layout(bepuhip_ctx* c) {
    if (c->d_boundary_rows) hipFree(c->d_boundary_rows);
    if (c->d_boundary_dense) hipFree(c->d_boundary_dense);
    if (c->d_boundary_holders) hipFree(c->d_boundary_holders);
    c->d_boundary_rows = nullptr; c->d_boundary_dense = nullptr; c->d_boundary_holders = nullptr; c->dense_rows = 0;
}
static void release_comm(bepuhip_ctx* c);

static void free_group_records(bepuhip_ctx* c) {
    if (!c->group_records) return;
    if (c->group_records_on_host) hipHostFree(c->group_records); else hipFree(c->group_records);
    c->group_records = nullptr; c->group_records_bodies = 0; c->group_records_on_host = false;
}

int32_t bepuhip_destroy(bepuhip_ctx* c) {
    if (!c) return BEPUHIP_OK;
    hipSetDevice(c->device);
    cancel_replan_job(c);
    if (c->device < kMaxCountedDevices) g_live_contexts[c->device].fetch_sub(1);
    if (c->stream) hipStreamSynchronize(c->stream);
    free_constraints(c);
    if (c->d_bodies) hipFree(c->d_bodies);
    if (c->d_bodies0) hipFree(c->d_bodies0);
    if (c->d_flags) hipFree(c->d_flags);
    if (c->d_kin) hipFree(c->d_kin);
    if (c->d_status) hipHostFree(c->d_status);
    if (c->d_body_gravity) hipFree(c->d_body_gravity);
    if (c->d_staged) hipFree(c->d_staged);
    if (c->d_collidables) hipFree(c->d_collidables);
    if (c->d_hull_points) hipFree(c->d_hull_points);
    if (c->d_hull_begin) hipFree(c->d_hull_begin);
    if (c->d_compound_children) hipFree(c->d_compound_children);
    if (c->d_compound_begin) hipFree(c->d_compound_begin);
    if (c->d_mesh_triangles) hipFree(c->d_mesh_triangles);
    if (c->d_mesh_begin) hipFree(c->d_mesh_begin);
    if (c->d_mesh_scales) hipFree(c->d_mesh_scales);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->h_desc_ring) hipHostFree(c->h_desc_ring);
    for (void* opened : c->peer_opened) if (opened) hipIpcCloseMemHandle(opened);
    free_group_records(c);
    if (c->d_peer_table) hipFree(c->d_peer_table);
    if (c->d_owned_dense) hipFree(c->d_owned_dense);
    if (c->d_owned_mask) hipFree(c->d_owned_mask);
    for (auto& chunk : c->raw_chunks) if (chunk.ptr) hipFree(chunk.ptr);
    if (c->h_staging) hipHostFree(c->h_staging);
    for (uint32_t* spare : c->spare_slab) if (spare) hipFree(spare);
    if (c->h_flush) hipHostFree(c->h_flush);
    if (c->d_flush) hipFree(c->d_flush);
    for (void* p : c->registered_host) hipHostUnregister(p);
    if (c->d_boundary) hipFree(c->d_boundary);
    if (c->d_boundary_snapshot) hipFree(c->d_boundary_snapshot);
    if (c->d_boundary_buf) hipFree(c->d_boundary_buf);
    free_boundary_layout(c);
    release_comm(c);
    for (auto& pair : c->policy_events) { if (pair[0]) hipEventDestroy(pair[0]); if (pair[1]) hipEventDestroy(pair[1]); }
    if (c->ev_start) hipEventDestroy(c->ev_start);
    if (c->ev_stop) hipEventDestroy(c->ev_stop);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return BEPUHIP_OK;
}

static int32_t rebuild_flags(bepuhip_ctx* c) {
    if (!c->d_flags || c->body_count == 0) return BEPUHIP_OK;
    HIP_TRY(hipMemsetAsync(c->d_flags, 0, (size_t)c->body_count * 4, c->stream));
    // Constraints that reference bodies not uploaded yet are not marked (the writes would leave d_flags); validate_solve refuses to run such a
    // state and the set_bodies that repairs it changes the count, which brings us back here.
    const bool marked = c->built && c->referenced_bodies <= c->body_count;
    if (marked) {  // up to kMarkGroupEntries type batches per launch, their descriptions in the kernel's arguments (one launch per type batch was a third of a millisecond of launches)
        MarkGroup group;
        group.count = 0; group.blocks = 0;
        auto launch = [&]() {
            if (group.count > 0) hipLaunchKernelGGL(mark_constrained_group_kernel, dim3(group.blocks), dim3(256), 0, c->stream, (const int*)c->d_slab, group, c->d_flags);
            group.count = 0; group.blocks = 0;
        };
        for (auto& tb : c->tbs) {
            const int extent = tb.device_extent();  // island layouts hold free slots (-1 references) between the live ones
            if (extent == 0) continue;
            MarkGroup::Entry& e = group.entries[group.count++];
            e.refs_off = (unsigned long long)tb.refs_off; e.extent = extent; e.stride = tb.stride; e.bodies = tb.info.bodies; e.block_begin = group.blocks;
            group.blocks += (extent + 255) / 256;
            if (group.count == kMarkGroupEntries) launch();
        }
        launch();
    }
    if (marked && c->clusters_enabled && c->clustered_dynamic_count > 0) {
        hipLaunchKernelGGL(mark_indices_kernel, dim3((c->clustered_dynamic_count + 255) / 256), dim3(256), 0, c->stream, (const int*)c->d_clustered_dynamic,
                           c->clustered_dynamic_count, c->d_flags, (unsigned)kFlagClustered);
    }
    if (marked && c->clusters_enabled && c->kinlist_count > 0) {
        hipLaunchKernelGGL(mark_indices_kernel, dim3((c->kinlist_count + 255) / 256), dim3(256), 0, c->stream, (const int*)c->d_kinlist, c->kinlist_count, c->d_flags,
                           (unsigned)(kFlagClusterKinematic | kFlagConstrained));
    }
    if (c->boundary_count > 0) {  // held by several ranks: always integrated inside the solver, whatever this rank's share of its constraints
        hipLaunchKernelGGL(mark_indices_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, (const int*)c->d_boundary, c->boundary_count, c->d_flags,
                           (unsigned)(kFlagConstrained | kFlagDynamicConstrained));
    }
    if (c->kin_count > 0) {
        hipLaunchKernelGGL(mark_indices_kernel, dim3((c->kin_count + 255) / 256), dim3(256), 0, c->stream, (const int*)c->d_kin, c->kin_count, c->d_flags, (unsigned)(kFlagConstrainedKinematic | kFlagConstrained));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}

int32_t bepuhip_set_bodies(bepuhip_ctx* c, const void* aos, int32_t count) {
    if (!c || (!aos && count > 0) || count < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad bodies argument");
    HIP_TRY(hipSetDevice(c->device));
    const bool count_changed = count != c->body_count;
    if (count_changed || count > c->body_capacity) {
        // Captured graphs bake d_bodies / d_flags / body_count / grid sizes into their kernel arguments (GraphKey holds only the schedule):
        // any change of the body set makes every cached graph stale.
        HIP_TRY(hipStreamSynchronize(c->stream));
        clear_graphs(c);
    }
    if (count > c->body_capacity) {
        if (c->d_bodies) hipFree(c->d_bodies);
        if (c->d_bodies0) hipFree(c->d_bodies0);
        if (c->d_flags) hipFree(c->d_flags);
        c->d_bodies = c->d_bodies0 = nullptr; c->d_flags = nullptr;
        c->body_capacity = c->body_count = 0;  // a failed allocation below leaves the context empty, not half-sized
        void *nb = nullptr, *nb0 = nullptr, *nf = nullptr;
        if (hipMalloc(&nb, (size_t)count * 128) != hipSuccess || hipMalloc(&nb0, (size_t)count * 128) != hipSuccess || hipMalloc(&nf, (size_t)count * 4) != hipSuccess) {
            if (nb) hipFree(nb);
            if (nb0) hipFree(nb0);
            if (nf) hipFree(nf);
            return fail(BEPUHIP_E_DEVICE, "out of device memory for " + std::to_string(count) + " bodies");
        }
        c->d_bodies = (decltype(c->d_bodies))nb; c->d_bodies0 = (decltype(c->d_bodies0))nb0; c->d_flags = (decltype(c->d_flags))nf;
        c->body_capacity = count;
    }
    c->body_count = count;
    if (count > 0) {
        HIP_TRY(hipMemcpyAsync(c->d_bodies, aos, (size_t)count * 128, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_bodies0, c->d_bodies, (size_t)count * 128, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (count_changed) return rebuild_flags(c);
    return BEPUHIP_OK;
}

int32_t bepuhip_begin_constraints(bepuhip_ctx* c, int32_t batch_count, int32_t fallback_batch_threshold) {
    if (!c || batch_count < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad batch count");
    if (fallback_batch_threshold < 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Fallback batch threshold must be positive.");  // SolveDescription.cs:48-51
    if (batch_count > fallback_batch_threshold + 1)
        return fail(BEPUHIP_E_INVALID_ARGUMENT, "more batches than FallbackBatchThreshold + 1 (Solver.cs:1882)");
    HIP_TRY(hipSetDevice(c->device));
    hipStreamSynchronize(c->stream);
    cancel_replan_job(c);  // a plan in the making describes the constraints this upload replaces
    free_constraints(c);
    c->batch_count = batch_count;
    c->fallback_threshold = fallback_batch_threshold;
    c->has_fallback = batch_count > fallback_batch_threshold;  // Batches[FallbackBatchThreshold] is the sequential fallback batch
    c->has_widened_types = false; c->has_joint_types = false; c->last_kernel_family = -1; c->type_mask = 0;
    c->building = true;
    for (auto& chunk : c->raw_chunks) chunk.used = 0;
    return BEPUHIP_OK;
}

// Device memory for the caller's bundles, in chunks that outlive the upload (a simulation uploads again and again): a bump allocator, reset by begin_constraints.
static int32_t raw_reserve(bepuhip_ctx* c, size_t bytes, char** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (c->raw_chunks.empty() || c->raw_chunks.back().used + bytes > c->raw_chunks.back().capacity) {
        for (auto& chunk : c->raw_chunks)  // an earlier chunk with room (after a reset the big ones come first again)
            if (chunk.used + bytes <= chunk.capacity) { *out = chunk.ptr + chunk.used; chunk.used += bytes; return BEPUHIP_OK; }
        bepuhip_ctx::RawChunk chunk{nullptr, std::max<size_t>(bytes, (size_t)64 << 20), 0};
        HIP_TRY(hipMalloc((void**)&chunk.ptr, chunk.capacity));
        c->raw_chunks.push_back(chunk);
    }
    auto& chunk = c->raw_chunks.back();
    *out = chunk.ptr + chunk.used;
    chunk.used += bytes;
    return BEPUHIP_OK;
}
static int32_t staging_reserve(bepuhip_ctx* c, size_t bytes) {  // pinned host memory: H2D copies from it run at the link's rate and asynchronously
    if (bytes <= c->h_staging_bytes) return BEPUHIP_OK;
    if (c->h_staging) hipHostFree(c->h_staging);
    c->h_staging = nullptr; c->h_staging_bytes = 0;
    bytes = std::max(bytes + bytes / 4, (size_t)16 << 20);
    HIP_TRY(hipHostMalloc(&c->h_staging, bytes, hipHostMallocDefault));
    c->h_staging_bytes = bytes;
    return BEPUHIP_OK;
}

int32_t bepuhip_set_type_batch(bepuhip_ctx* c, int32_t batch_index, int32_t type_id, int32_t count, const int32_t* refs, const float* prestep, const float* accum) {
    if (!c || !c->building) return fail(BEPUHIP_E_STATE, "set_type_batch outside begin/end");
    if (batch_index < 0 || batch_index >= c->batch_count || count < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad batch index or count");
    if (!c->tbs.empty() && batch_index < c->tbs.back().batch) return fail(BEPUHIP_E_INVALID_ARGUMENT, "type batches must be supplied in batch order");
    if (count > 0 && (!refs || !prestep || !accum)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null buffer");
    HostTypeBatch tb;
    if (!type_info(type_id, tb.info)) return fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(type_id));
    tb.batch = batch_index; tb.type_id = type_id; tb.count = count;
    tb.stride = ((count + 63) / 64) * 64;
    const int W = c->W;
    const int nb = tb.info.bodies, pf = tb.info.prestep, imf = tb.info.impulse;
    tb.refs_soa.assign((size_t)nb * tb.stride, -1);
    if (c->host_values) {
        tb.prestep_soa.assign((size_t)pf * tb.stride, 0.0f);
        tb.accum_soa.assign((size_t)imf * tb.stride, 0.0f);
    }
    // AOSOA -> SoA (BundleIndexing.cs:50-60, TypeProcessor.cs:269-279): a bundle holds W consecutive constraints of every field, so field f of bundle b is W
    // consecutive words on both sides — copied as such (the last bundle only up to `count`). The host does this for the body references only (the plan needs them);
    // prestep data and accumulated impulses — nine tenths of the bytes — go to the device as they are and are transposed there (end_constraints).
    const bool fallback_batch = batch_index == c->fallback_threshold;
    int highest_reference = -1;
    int64_t live = 0;  // a fallback type batch counts its empty lanes in `count`
    // A synchronized batch has no empty lanes: its references are copied bundle row by bundle row with the checks folded into an OR and a maximum (loops the compiler
    // turns into vector code — this conversion is two of the upload's milliseconds otherwise); the lane-by-lane loop below is the fallback batch's.
    const int plain_bundles = (fallback_batch || c->host_values) ? 0 : count / W;
    if (plain_bundles > 0) {
        int32_t any = 0, highest = 0;
        auto rows_of = [&](auto width) {
            constexpr int kW = decltype(width)::value;
            for (int bundle = 0; bundle < plain_bundles; ++bundle)
                for (int k = 0; k < nb; ++k) {
                    const int32_t* src = refs + ((size_t)bundle * nb + k) * kW;
                    int32_t* dst = tb.refs_soa.data() + (size_t)k * tb.stride + (size_t)bundle * kW;
                    for (int lane = 0; lane < kW; ++lane) { const int32_t r = src[lane]; dst[lane] = r; any |= r; highest = std::max(highest, r & kRefMask); }
                }
        };
        if (W == 8) rows_of(std::integral_constant<int, 8>());
        else if (W == 16) rows_of(std::integral_constant<int, 16>());
        else if (W == 4) rows_of(std::integral_constant<int, 4>());
        else {
            for (int bundle = 0; bundle < plain_bundles; ++bundle)
                for (int k = 0; k < nb; ++k)
                    for (int lane = 0; lane < W; ++lane) {
                        const int32_t r = refs[((size_t)bundle * nb + k) * W + lane];
                        tb.refs_soa[(size_t)k * tb.stride + (size_t)bundle * W + lane] = r; any |= r; highest = std::max(highest, r & kRefMask);
                    }
        }
        if (any < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "empty (-1) body reference inside a synchronized batch");
        highest_reference = highest;
        live = (int64_t)plain_bundles * W;
    }
    for (int b0 = plain_bundles * W; b0 < count; b0 += W) {
        const size_t bundle = (size_t)(b0 / W);
        const int lanes = std::min(W, count - b0);
        for (int k = 0; k < nb; ++k) {
            const int32_t* src = refs + bundle * nb * W + (size_t)k * W;
            int32_t* dst = tb.refs_soa.data() + (size_t)k * tb.stride + b0;
            for (int lane = 0; lane < lanes; ++lane) {
                const int32_t r = src[lane];
                // Empty lanes exist only inside the bundles of the sequential fallback batch (TypeProcessor.cs:451-560): every body slot of the lane is -1.
                if (r < 0 && !(r == -1 && fallback_batch)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "empty (-1) body reference inside a synchronized batch");
                dst[lane] = r;
                if (r >= 0) highest_reference = std::max(highest_reference, r & kRefMask);
                live += (k == 0 && r != -1);
            }
        }
        if (c->host_values) {
            for (int f = 0; f < pf; ++f) memcpy(tb.prestep_soa.data() + (size_t)f * tb.stride + b0, prestep + bundle * pf * W + (size_t)f * W, (size_t)lanes * 4);
            for (int f = 0; f < imf; ++f) memcpy(tb.accum_soa.data() + (size_t)f * tb.stride + b0, accum + bundle * imf * W + (size_t)f * W, (size_t)lanes * 4);
        }
    }
    if (!c->host_values && count > 0) {
        HIP_TRY(hipSetDevice(c->device));  // allocates and copies on the device: contexts of different GPUs may be interleaved on one thread
        const size_t bundles = (size_t)(count + W - 1) / W;
        for (int which = 0; which < 2; ++which) {
            const size_t bytes = bundles * (size_t)(which == 0 ? pf : imf) * W * 4;
            if (bytes == 0) continue;
            char* dst = nullptr;
            const int32_t st = raw_reserve(c, bytes, &dst);
            if (st != BEPUHIP_OK) return st;
            // pageable source: the call returns when the source has been read; registered (pinned) source: asynchronous — hence the header's rule that the buffers stay
            // unchanged until end_constraints returns
            HIP_TRY(hipMemcpyAsync(dst, which == 0 ? (const void*)prestep : (const void*)accum, bytes, hipMemcpyHostToDevice, c->stream));
            (which == 0 ? tb.raw_prestep : tb.raw_accum) = (const float*)dst;
        }
    }
    if (fallback_batch) { tb.occupied.resize((size_t)count); for (int i = 0; i < count; ++i) tb.occupied[i] = tb.refs_soa[i] != -1; }  // (structural updates of the fallback batch keep it current)
    c->has_widened_types = c->has_widened_types || is_widened_type(type_id); if (count > 0) c->type_mask |= 1ull << (type_id & 63);
    c->has_joint_types = c->has_joint_types || type_id > kContact4;
    c->referenced_bodies = std::max(c->referenced_bodies, highest_reference + 1);  // checked against the body count at solve time (validate_solve)
    c->total_constraints += live;
    c->tbs.push_back(std::move(tb));
    return BEPUHIP_OK;
}

static int32_t build_constraints(bepuhip_ctx* c, ClusterPlan* planned = nullptr, std::vector<std::vector<int32_t>>* planned_fallback_refs = nullptr, bepuhip_ctx* soft_from = nullptr);
static int32_t flush_structural(bepuhip_ctx* c);
static HostTypeBatch* find_tb(bepuhip_ctx* c, int batch, int type_id);
static int32_t device_index_of(bepuhip_ctx* c, HostTypeBatch* tb, const int** out);
static int32_t staging_reserve(bepuhip_ctx* c, size_t bytes);
static int32_t group_queue_check(const bepuhip_ctx* c);
static int32_t leave_island_schedule(bepuhip_ctx* c);

int32_t bepuhip_end_constraints(bepuhip_ctx* c) {
    if (!c || !c->building) return fail(BEPUHIP_E_STATE, "end_constraints without begin");
    HIP_TRY(hipSetDevice(c->device));
    c->building = false;
    const int32_t st = build_constraints(c);
    if (st != BEPUHIP_OK) free_constraints(c);  // never a half-built set: the context is back to "no constraints", the error text is kept
    return st;
}

// Launch descriptors from the type batches' current counts / strides / slab offsets (called at end_constraints and after structural updates).
static int32_t build_descriptors(bepuhip_ctx* c, const std::vector<std::vector<int32_t>>& fallback_refs) {
    if (c->d_tbs) { hipFree(c->d_tbs); c->d_tbs = nullptr; }
    if (c->d_inc_tbs) { hipFree(c->d_inc_tbs); c->d_inc_tbs = nullptr; }
    if (c->d_fallback_indices) { hipFree(c->d_fallback_indices); c->d_fallback_indices = nullptr; }
    // Descriptors: per launch grid layout. Synchronized batches: one launch each, all their type batches in one grid. The sequential fallback batch
    // (Solver_Solve.cs:546-583: bundles solved one after the other by one thread, since they may share bodies) becomes one launch per DEPENDENCY LEVEL:
    // walking its type batches and bundles in the reference's order, a constraint's level is one more than the highest level any of its dynamic bodies
    // was last touched at, so every body still meets its constraints in the reference's order and constraints of one level share no dynamic body.
    // The rows stay in the caller's layout (empty lanes included: read-backs and ranged updates are unchanged); a level is a list of row indices.
    const int sync_batches = c->has_fallback ? c->fallback_threshold : c->batch_count;
    std::vector<std::vector<std::vector<int32_t>>> level_rows;  // [level][fallback type batch ordinal] -> rows
    std::vector<size_t> fallback_tbs;
    if (c->has_fallback) {
        for (size_t t = 0; t < c->tbs.size(); ++t) if (c->tbs[t].batch == c->fallback_threshold) fallback_tbs.push_back(t);
        std::vector<int32_t> last_level(c->referenced_bodies, -1);
        for (size_t ord = 0; ord < fallback_tbs.size(); ++ord) {
            const HostTypeBatch& tb = c->tbs[fallback_tbs[ord]];
            for (int i = 0; i < tb.count; ++i) {
                if (fallback_refs[ord][i] == -1) continue;
                int level = 0;
                for (int k = 0; k < tb.info.bodies; ++k) {
                    const int32_t r = fallback_refs[ord][(size_t)k * tb.stride + i];
                    if ((uint32_t)r < kDynamicLimit) level = std::max(level, last_level[r] + 1);  // kinematic bodies are never written: no ordering through them
                }
                for (int k = 0; k < tb.info.bodies; ++k) {
                    const int32_t r = fallback_refs[ord][(size_t)k * tb.stride + i];
                    if ((uint32_t)r < kDynamicLimit) last_level[r] = level;
                }
                if ((size_t)level >= level_rows.size()) level_rows.resize(level + 1, std::vector<std::vector<int32_t>>(fallback_tbs.size()));
                level_rows[level][ord].push_back(tb.perm.empty() ? i : tb.inv[i]);  // on an island layout the constraint's row is its device slot
            }
        }
    }
    c->launch_count = sync_batches + (int)level_rows.size();
    c->batch_begin.assign(c->launch_count + 1, 0);
    c->batch_blocks.assign(c->launch_count, 0);
    std::vector<DevTypeBatch> descs, inc;
    std::vector<int32_t> index_pool;               // all levels' row lists + the occupied rows of every incremental fallback type batch
    std::vector<std::pair<size_t, size_t>> desc_fixups, inc_fixups;  // (descriptor, offset of its row list in index_pool): pointers are set once the pool is on the device
    auto base_desc = [&](const HostTypeBatch& tb) {
        DevTypeBatch d;
        d.type_id = tb.type_id; d.count = tb.count; d.stride = tb.stride; d.block_begin = 0;
        d.refs = (int*)(c->d_slab + tb.refs_off);
        d.prestep = (float*)(c->d_slab + tb.prestep_off);
        d.accum = (float*)(c->d_slab + tb.accum_off);
        d.indices = nullptr;
        return d;
    };
    {
        size_t t = 0;
        for (int b = 0; b < sync_batches; ++b) {
            c->batch_begin[b] = (int)descs.size();
            int blocks = 0;
            while (t < c->tbs.size() && c->tbs[t].batch == b) {
                DevTypeBatch d = base_desc(c->tbs[t]);
                d.block_begin = blocks;
                descs.push_back(d);
                blocks += (c->tbs[t].count + kBlock - 1) / kBlock;
                ++t;
            }
            c->batch_blocks[b] = blocks;
        }
        for (size_t level = 0; level < level_rows.size(); ++level) {
            const int b = sync_batches + (int)level;
            c->batch_begin[b] = (int)descs.size();
            int blocks = 0;
            for (size_t ord = 0; ord < fallback_tbs.size(); ++ord) {
                const std::vector<int32_t>& rows = level_rows[level][ord];
                if (rows.empty()) continue;
                DevTypeBatch d = base_desc(c->tbs[fallback_tbs[ord]]);
                d.count = (int)rows.size(); d.block_begin = blocks;
                desc_fixups.push_back({descs.size(), index_pool.size()});
                index_pool.insert(index_pool.end(), rows.begin(), rows.end());
                descs.push_back(d);
                blocks += ((int)rows.size() + kBlock - 1) / kBlock;
            }
            c->batch_blocks[b] = blocks;
        }
        c->batch_begin[c->launch_count] = (int)descs.size();
    }
    c->inc_blocks = 0;
    for (size_t t = 0; t < c->tbs.size(); ++t) {  // the incremental contact update only writes the constraint's own depths: all batches in one grid, the fallback batch included
        const HostTypeBatch& tb = c->tbs[t];
        if (!tb.info.incremental || tb.count == 0) continue;
        DevTypeBatch d = base_desc(tb);
        if (c->has_fallback && tb.batch == c->fallback_threshold) {  // skip the empty lanes: their references are -1
            const size_t ord = std::find(fallback_tbs.begin(), fallback_tbs.end(), t) - fallback_tbs.begin();
            const size_t begin = index_pool.size();
            for (int i = 0; i < tb.count; ++i) if (fallback_refs[ord][i] != -1) index_pool.push_back(tb.perm.empty() ? i : tb.inv[i]);
            d.count = (int)(index_pool.size() - begin);
            if (d.count == 0) continue;
            inc_fixups.push_back({inc.size(), begin});
        }
        d.block_begin = c->inc_blocks;
        c->inc_blocks += (d.count + kBlock - 1) / kBlock;
        inc.push_back(d);
    }
    c->inc_tb_count = (int)inc.size();
    if (!index_pool.empty()) {
        HIP_TRY(hipMalloc((void**)&c->d_fallback_indices, index_pool.size() * 4));
        HIP_TRY(copy_sync(c, c->d_fallback_indices, index_pool.data(), index_pool.size() * 4, hipMemcpyHostToDevice));
        for (auto& fx : desc_fixups) descs[fx.first].indices = c->d_fallback_indices + fx.second;
        for (auto& fx : inc_fixups) inc[fx.first].indices = c->d_fallback_indices + fx.second;
    }
    if (!descs.empty()) {
        HIP_TRY(hipMalloc((void**)&c->d_tbs, descs.size() * sizeof(DevTypeBatch)));
        HIP_TRY(copy_sync(c, c->d_tbs, descs.data(), descs.size() * sizeof(DevTypeBatch), hipMemcpyHostToDevice));
    }
    if (!inc.empty()) {
        HIP_TRY(hipMalloc((void**)&c->d_inc_tbs, inc.size() * sizeof(DevTypeBatch)));
        HIP_TRY(copy_sync(c, c->d_inc_tbs, inc.data(), inc.size() * sizeof(DevTypeBatch), hipMemcpyHostToDevice));
    }
    return BEPUHIP_OK;
}

// `planned`: the plan was computed beforehand for exactly these type batches (bepuhip_replan_commit: by the job's worker thread, on the shadow the type batches were
// moved out of) — with it the fallback type batches' references as they were before the planner permuted them, and the context (that shadow) on which soft_setup has
// run for the plan already, lazily built mirrors included.
static int32_t build_constraints(bepuhip_ctx* c, ClusterPlan* planned, std::vector<std::vector<int32_t>>* planned_fallback_refs, bepuhip_ctx* soft_from) {
    const bool stats = env_int("BEPUHIP_PLAN_STATS", 0) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!stats) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "bepuhip end_constraints: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    size_t words = 0;  // (total_constraints and referenced_bodies were counted while set_type_batch converted the references)
    c->requirk_stale = true;  // the conserving angular modes' substep-0 lists are built by the first solve that asks for such a mode (build_requirk_lists)
    std::vector<std::vector<int32_t>> fallback_refs;  // the fallback type batches' references (SoA rows), kept past the staging buffers for the level walk below
    if (planned_fallback_refs) fallback_refs = std::move(*planned_fallback_refs);
    else if (c->has_fallback)
        for (auto& tb : c->tbs) if (tb.batch == c->fallback_threshold) fallback_refs.push_back(tb.refs_soa);
    lap("checks, responsibility lists");
    ClusterPlan plan;
    if (planned) plan = std::move(*planned);
    else plan_clusters(c, plan);
    lap(planned ? "cluster plan (adopted)" : "cluster plan (host)");
    // Slab layout: what the host builds (references, local references + ranks) first, in one region that travels through pinned staging in one copy; behind it the
    // prestep and impulse rows, which the device fills itself from the caller's bundles.
    for (auto& tb : c->tbs) {
        tb.refs_off = words; words += tb.refs_soa.size();
        tb.lrefs_off = words; words += tb.lrefs_soa.size();
    }
    const size_t index_words = words;
    for (auto& tb : c->tbs) {
        tb.prestep_off = words; words += (size_t)tb.info.prestep * tb.stride;
        tb.accum_off = words; words += (size_t)tb.info.impulse * tb.stride;
    }
    c->slab_words = words;
    if (words > 0) {
        if (c->spare_slab[0] && c->spare_slab_words >= words) {  // the previous upload's pair (free_constraints): everything below rewrites what it reads
            c->d_slab = c->spare_slab[0]; c->d_slab0 = c->spare_slab[1]; c->slab_alloc_words = c->spare_slab_words;
            c->spare_slab[0] = c->spare_slab[1] = nullptr; c->spare_slab_words = 0;
        } else {
            for (uint32_t*& spare : c->spare_slab) { if (spare) hipFree(spare); spare = nullptr; }
            c->spare_slab_words = 0;
            HIP_TRY(hipMalloc((void**)&c->d_slab, words * 4));
            HIP_TRY(hipMalloc((void**)&c->d_slab0, words * 4));
            c->slab_alloc_words = words;
        }
        // host index -> device slot of every permuted type batch, one pool for all of them (with room for the indices additions on an island layout can create)
        size_t pool_words = 0;
        for (auto& tb : c->tbs)
            if (!tb.perm.empty()) { tb.perm_inverse(0); pool_words += std::max<size_t>(tb.inv.size(), (size_t)tb.device_extent()); }
        { const int32_t st = staging_reserve(c, (index_words + pool_words) * 4); if (st != BEPUHIP_OK) return st; }
        uint32_t* host = (uint32_t*)c->h_staging;
        plan_parallel_for(c->tbs.size(), [&](size_t t) {  // (12 MB for the bench scene: on the plan threads, not one after the other)
            HostTypeBatch& tb = c->tbs[t];
            if (!tb.refs_soa.empty()) memcpy(&host[tb.refs_off], tb.refs_soa.data(), tb.refs_soa.size() * 4);
            if (!tb.lrefs_soa.empty()) memcpy(&host[tb.lrefs_off], tb.lrefs_soa.data(), tb.lrefs_soa.size() * 4);
            std::vector<int32_t>().swap(tb.lrefs_soa);
            std::vector<int32_t>().swap(tb.refs_soa);
        });
        if (pool_words > 0) {
            HIP_TRY(hipMalloc((void**)&c->d_index_pool, pool_words * 4));
            size_t at = 0;
            for (auto& tb : c->tbs) {
                if (tb.perm.empty()) continue;
                const size_t room = std::max<size_t>(tb.inv.size(), (size_t)tb.device_extent());
                memcpy(&host[index_words + at], tb.inv.data(), tb.inv.size() * 4);
                if (room > tb.inv.size()) memset(&host[index_words + at + tb.inv.size()], 0, (room - tb.inv.size()) * 4);
                tb.d_device_index = c->d_index_pool + at;
                tb.index_pooled = true;
                at += room;
            }
            HIP_TRY(hipMemcpyAsync(c->d_index_pool, &host[index_words], pool_words * 4, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipMemsetAsync(c->d_slab + index_words, 0, (words - index_words) * 4, c->stream));  // free slots and the padding behind `count` read as zeros
        if (index_words > 0) HIP_TRY(hipMemcpyAsync(c->d_slab, host, index_words * 4, hipMemcpyHostToDevice, c->stream));
        // AOSOA bundles -> rows, in the plan's device order, on the device (the bundles have been in HBM since set_type_batch)
        for (auto& tb : c->tbs) {
            if (tb.count == 0) continue;
            const int blocks = (tb.count + 255) / 256;
            if (tb.raw_prestep && tb.info.prestep > 0)
                hipLaunchKernelGGL(scatter_bundles_kernel, dim3(blocks), dim3(256), 0, c->stream, tb.raw_prestep, (float*)(c->d_slab + tb.prestep_off), (const int*)tb.d_device_index, 0, tb.count,
                                   tb.info.prestep, tb.stride, c->W);
            if (tb.raw_accum && tb.info.impulse > 0)
                hipLaunchKernelGGL(scatter_bundles_kernel, dim3(blocks), dim3(256), 0, c->stream, tb.raw_accum, (float*)(c->d_slab + tb.accum_off), (const int*)tb.d_device_index, 0, tb.count,
                                   tb.info.impulse, tb.stride, c->W);
            tb.raw_prestep = tb.raw_accum = nullptr;
            if (tb.old_prestep[0] && tb.info.prestep > 0)  // bepuhip_replan: rows of the previous slab, in the caller's order
                hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, tb.old_prestep[0], tb.old_stride, c->d_slab + tb.prestep_off, tb.stride, (const int*)tb.d_device_index,
                                   tb.count, tb.info.prestep);
            if (tb.old_accum[0] && tb.info.impulse > 0)
                hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, tb.old_accum[0], tb.old_stride, c->d_slab + tb.accum_off, tb.stride, (const int*)tb.d_device_index,
                                   tb.count, tb.info.impulse);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->d_slab0, c->d_slab, words * 4, hipMemcpyDeviceToDevice, c->stream));
        for (auto& tb : c->tbs) {  // bepuhip_replan: the snapshot keeps ITS values (what reset_state returns to), in the new layout
            if (tb.count == 0) continue;
            const int blocks = (tb.count + 255) / 256;
            if (tb.old_prestep[1] && tb.info.prestep > 0)
                hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, tb.old_prestep[1], tb.old_stride, c->d_slab0 + tb.prestep_off, tb.stride, (const int*)tb.d_device_index,
                                   tb.count, tb.info.prestep);
            if (tb.old_accum[1] && tb.info.impulse > 0)
                hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, tb.old_accum[1], tb.old_stride, c->d_slab0 + tb.accum_off, tb.stride, (const int*)tb.d_device_index,
                                   tb.count, tb.info.impulse);
            tb.old_prestep[0] = tb.old_prestep[1] = tb.old_accum[0] = tb.old_accum[1] = nullptr;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));  // the caller's buffers (if registered: read asynchronously) and the staging buffer are free again
    }
    lap("slab assembly + upload");
    { const int32_t st = build_descriptors(c, fallback_refs); if (st != BEPUHIP_OK) return st; }
    lap("launch descriptors");
    // cluster path tables
    auto upload_ints = [&](const void* src, size_t bytes, void** dst) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        hipError_t e = hipMalloc(dst, bytes);
        if (e != hipSuccess) return e;
        return copy_sync(c, *dst, src, bytes, hipMemcpyHostToDevice);
    };
    c->kinlist_count = (int)plan.kinlist.size();
    HIP_TRY(upload_ints(plan.kinlist.data(), plan.kinlist.size() * 4, (void**)&c->d_kinlist));
    c->clusters_enabled = plan.enabled;
    if (plan.enabled) {
        c->cluster_count = (int)plan.clusters.size();
        c->cluster_max_slots = plan.max_slots;
        c->cluster_planes = plan.planes;
        HIP_TRY(hipMalloc((void**)&c->d_cycles, plan.clusters.size() * 8));
        HIP_TRY(fill_async(c, c->d_cycles, 0, plan.clusters.size() * 8));
        c->first_cluster = plan.clusters[0];
        c->cluster_max_items = plan.max_items;
        for (auto& it : plan.items) {  // resolve the items' slab offsets now that the slab layout exists
            const HostTypeBatch& tb = c->tbs[it.tb];
            it.lrefs_off = (unsigned)tb.lrefs_off;
            it.prestep_off = (unsigned)tb.prestep_off;
            it.accum_off = (unsigned)tb.accum_off;
        }
        if (soft_from && soft_from->soft_ok) soft_from->items_host = plan.items;  // (the mirror soft_setup took on the worker's thread predates the slab layout)
        c->clustered_dynamic_count = (int)plan.clustered_dynamic.size();
        HIP_TRY(upload_ints(plan.clusters.data(), plan.clusters.size() * sizeof(ClusterDesc), (void**)&c->d_clusters));
        HIP_TRY(upload_ints(plan.items.data(), plan.items.size() * sizeof(ClusterItem), (void**)&c->d_items));
        HIP_TRY(upload_ints(plan.batch_item_begin.data(), plan.batch_item_begin.size() * 4, (void**)&c->d_batch_item_begin));
        HIP_TRY(upload_ints(plan.cluster_bodies.data(), plan.cluster_bodies.size() * 4, (void**)&c->d_cluster_bodies));
        {  // with room for the bodies structural updates bring into the plan (bepu_soft_updates.h: a body's first constraint)
            c->clustered_dynamic_host = plan.clustered_dynamic;
            c->clustered_position.clear();
            c->clustered_dynamic_capacity = (int)plan.clustered_dynamic.size() + std::max(256, (int)plan.clustered_dynamic.size() / 16);
            HIP_TRY(hipMalloc((void**)&c->d_clustered_dynamic, (size_t)c->clustered_dynamic_capacity * 4));
            if (!plan.clustered_dynamic.empty())
                HIP_TRY(copy_sync(c, c->d_clustered_dynamic, plan.clustered_dynamic.data(), plan.clustered_dynamic.size() * 4, hipMemcpyHostToDevice));
        }
        for (int threads : kClusterThreadChoices)
            for (int tr = 0; tr < 2; ++tr)
                for (int wide = 0; wide < 2; ++wide)
                    HIP_TRY(hipFuncSetAttribute(cluster_kernel_variant(threads, tr != 0, wide != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
        for (int tr = 0; tr < 2; ++tr) HIP_TRY(hipFuncSetAttribute(cluster_kernel_variant(768, tr != 0, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
        for (int tr = 0; tr < 2; ++tr)
            for (const void* fn : {bepu_cluster_kernel_contacts_512s(tr != 0), bepu_cluster_kernel_contacts_768s(tr != 0), bepu_cluster_kernel_contacts_1024(tr != 0)})
                HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
        c->group_body_cluster.clear();
        if (c->group_world > 1) c->group_body_cluster = plan.body_cluster;  // (before soft_setup takes the vector: which device owns a body at the end of a step)
        c->owned_mask_bodies = 0;  // (the device copy of the ownership mask follows the plan)
        c->cluster_first = (int)((int64_t)plan.clusters.size() * c->group_rank / c->group_world);
        c->cluster_local = (int)((int64_t)plan.clusters.size() * (c->group_rank + 1) / c->group_world) - c->cluster_first;
        if (soft_from) soft_move_state(c, soft_from); else soft_setup(c, plan);
        for (int tr = 0; tr < 2; ++tr)
            for (int wide = 0; wide < 2; ++wide) {
                HIP_TRY(hipFuncSetAttribute(cluster_kernel_variant(1024, tr != 0, wide != 0, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
                HIP_TRY(hipFuncSetAttribute(cluster_kernel_variant(512, tr != 0, wide != 0, true, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
            }
        for (int wide = 0; wide < 2; ++wide) {
            HIP_TRY(hipFuncSetAttribute(cluster_pass_kernel(wide != 0, false), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
            HIP_TRY(hipFuncSetAttribute(cluster_pass_kernel(wide != 0, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
            HIP_TRY(hipFuncSetAttribute(cluster_kernel_variant(1024, false, wide != 0, false, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
            HIP_TRY(hipFuncSetAttribute(cluster_kernel_variant(512, false, wide != 0, true, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
        }
        c->row_policy = -1; c->policy_samples = 0;  // a new topology is measured afresh
        c->clusters_shared = plan.shared;
        {   // Twelve waves instead of eight for split plans that are short of wave time (enqueue_island_launch): judged by the work items a cluster hands out per pass
            size_t claimable = 0;
            for (auto& it : plan.items) claimable += ((it.shape >> kItemFuseShift) & kItemFuseMember) == 0;
            const double per_cluster = plan.clusters.empty() ? 0.0 : (double)claimable / (double)plan.clusters.size();
            c->split_twelve_waves = plan.shared && getenv("BEPUHIP_SPLIT_THREADS") == nullptr && per_cluster >= (double)env_int("BEPUHIP_SPLIT_TWELVE_WAVES_ITEMS", 56);
        }
        if (plan.shared) {  // split islands: velocity / event tables of the bodies more than one cluster touches (indexed by body, only the shared ones are used)
            // (long enough for the bodies that have no constraints yet: structural updates may bring them into the plan and share them)
            c->shared_bodies = std::max(plan.shared_info.size(), (size_t)std::max(c->body_count, 0)) + 1024;
            if (c->group_world > 1) {
                // The other members of a device group hold this table's address (bepuhip_set_peer_records / import_peer_records) and push records into it: it must not move
                // when this member uploads again or re-plans. Allocated once, with room to grow; a scene that outgrows it gets a new table, and the header says what the
                // members have to do then (exchange the tables again).
                if (c->group_records && c->group_records_bodies < c->shared_bodies) { HIP_TRY(hipStreamSynchronize(c->stream)); free_group_records(c); }
                if (!c->group_records) {
                    c->group_records_bodies = c->shared_bodies + c->shared_bodies / 4;
                    const size_t bytes = c->group_records_bodies * 4 * sizeof(float4);
                    if (env_int("BEPUHIP_GROUP_FAKE_REMOTE", 0) != 0) {
                        // A developer switch for boxes with ONE GPU (VERDICT r5 next #7d): the table in fine-grained, host-coherent memory instead of this device's HBM. Members
                        // that share the device then exchange records through memory that is remote to all of them — the peers' system-scope stores and the owner's agent-scope
                        // polls travel over the host link and have to be coherent there, instead of meeting in the local L2 / HBM where any scope works. Slow, and exactly the
                        // path (minus xGMI) that two devices take.
                        void* host = nullptr;
                        HIP_TRY(hipHostMalloc(&host, bytes, hipHostMallocCoherent | hipHostMallocMapped));
                        void* mapped = nullptr;
                        HIP_TRY(hipHostGetDevicePointer(&mapped, host, 0));
                        c->group_records = (float4*)mapped;
                        c->group_records_on_host = true;
                    } else {
                        HIP_TRY(hipMalloc((void**)&c->group_records, bytes));
                        c->group_records_on_host = false;
                    }
                }
                c->d_shared_vel = c->group_records;
            } else {
                HIP_TRY(hipMalloc((void**)&c->d_shared_vel, c->shared_bodies * 4 * sizeof(float4)));  // two records (substep parity) of two float4 per body
            }
            HIP_TRY(fill_async(c, c->d_shared_vel, 0, c->shared_bodies * 4 * sizeof(float4)));         // cleared here, then never again: every step's event numbers start above the last step's
            c->shared_epoch = 0;
            HIP_TRY(hipMalloc((void**)&c->d_shared_info, c->shared_bodies * 4));
            HIP_TRY(fill_async(c, c->d_shared_info, 0, c->shared_bodies * 4));
            if (!plan.shared_info.empty()) HIP_TRY(copy_sync(c, c->d_shared_info, plan.shared_info.data(), plan.shared_info.size() * 4, hipMemcpyHostToDevice));
            for (int threads : kClusterThreadChoices)
                for (int tr = 0; tr < 2; ++tr)
                    for (int wide = 0; wide < 2; ++wide)
                        HIP_TRY(hipFuncSetAttribute(cluster_kernel_variant(threads, tr != 0, wide != 0, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
        }
    }
    c->built = true;
    lap("plan tables upload");
    const int32_t flags_status = rebuild_flags(c);
    lap("body flags");
    return flags_status;
}

// ---- Re-planning: a new plan for the constraints the device holds NOW (after structural updates the old plan could not absorb, or to refresh a plan's reserves) ----
// The rows of the launch-per-batch layout are the source: body references are read back (in the caller's order), the host plans as bepuhip_end_constraints does, prestep
// data and accumulated impulses — of the working rows and of the snapshot bepuhip_reset_state returns to — move from the old rows to the new layout on the device.
// Nothing crosses PCIe but the references (one way) and the plan's tables (the other).
struct LiveTypeBatch {  // a type batch as the rows hold it: what is needed to find its values again once the context's own bookkeeping has been rebuilt
    int batch, type_id, count, stride;
    TypeInfoH info;
    size_t refs_off, prestep_off, accum_off;
    std::vector<uint8_t> occupied;
};
static std::vector<LiveTypeBatch> live_layout(const bepuhip_ctx* c) {
    std::vector<LiveTypeBatch> live;
    live.reserve(c->tbs.size());
    for (auto& tb : c->tbs) live.push_back(LiveTypeBatch{tb.batch, tb.type_id, tb.count, tb.stride, tb.info, tb.refs_off, tb.prestep_off, tb.accum_off, tb.occupied});
    return live;
}
// Fresh type batches (no plan yet, new strides) with the references `slab` holds for `live`, read through the pinned staging buffer: one wait for all rows.
static int32_t snapshot_type_batches(bepuhip_ctx* c, const uint32_t* slab, const std::vector<LiveTypeBatch>& live, std::vector<HostTypeBatch>& fresh) {
    size_t words = 0;
    for (auto& lt : live) words += (size_t)lt.info.bodies * (size_t)lt.count;
    { const int32_t st = staging_reserve(c, std::max<size_t>(words, 1) * 4); if (st != BEPUHIP_OK) return st; }
    int32_t* staged = (int32_t*)c->h_staging;
    size_t at = 0;
    for (auto& lt : live)
        for (int k = 0; k < lt.info.bodies && lt.count > 0; ++k, at += (size_t)lt.count)
            HIP_TRY(hipMemcpyAsync(staged + at, slab + lt.refs_off + (size_t)k * lt.stride, (size_t)lt.count * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    fresh.clear();
    fresh.reserve(live.size());
    at = 0;
    for (auto& lt : live) {
        HostTypeBatch nt;
        nt.batch = lt.batch; nt.type_id = lt.type_id; nt.count = lt.count; nt.info = lt.info;
        nt.stride = std::max(64, ((lt.count + 63) / 64) * 64);
        nt.refs_off = nt.prestep_off = nt.accum_off = nt.lrefs_off = 0;
        nt.refs_soa.assign((size_t)lt.info.bodies * nt.stride, -1);
        for (int k = 0; k < lt.info.bodies && lt.count > 0; ++k, at += (size_t)lt.count) memcpy(nt.refs_soa.data() + (size_t)k * nt.stride, staged + at, (size_t)lt.count * 4);
        nt.occupied = lt.occupied;
        fresh.push_back(std::move(nt));
    }
    return BEPUHIP_OK;
}
// What begin_constraints / set_type_batch would have counted for these type batches.
static void adopt_type_batches(bepuhip_ctx* c, std::vector<HostTypeBatch>&& fresh, int batch_count, bool has_fallback) {
    c->batch_count = batch_count; c->has_fallback = has_fallback;
    c->has_widened_types = false; c->has_joint_types = false; c->last_kernel_family = -1; c->type_mask = 0;
    c->referenced_bodies = 0; c->total_constraints = 0;
    for (auto& tb : fresh) {
        c->has_widened_types = c->has_widened_types || is_widened_type(tb.type_id);
        c->has_joint_types = c->has_joint_types || tb.type_id > kContact4;
        c->type_mask |= 1ull << (tb.type_id & 63);
        for (int32_t r : tb.refs_soa) if (r >= 0) c->referenced_bodies = std::max(c->referenced_bodies, (r & kRefMask) + 1);
        if (has_fallback && tb.batch == c->fallback_threshold) { for (int i = 0; i < tb.count; ++i) c->total_constraints += tb.refs_soa[i] != -1; }
        else c->total_constraints += tb.count;
    }
    c->tbs = std::move(fresh);
}
// The context's constraints rebuilt from rows it no longer owns (`old_slab` / `old_slab0` in the layout `live`, the caller's order): snapshot, plan, move the values.
// The context must hold no constraints (free_constraints) when this is called; the old slabs stay the caller's.
static int32_t rebuild_from_rows(bepuhip_ctx* c, const std::vector<LiveTypeBatch>& live, const uint32_t* old_slab, const uint32_t* old_slab0, int batch_count, bool has_fallback) {
    std::vector<HostTypeBatch> fresh;
    int32_t st = snapshot_type_batches(c, old_slab, live, fresh);
    if (st != BEPUHIP_OK) return st;
    for (size_t t = 0; t < fresh.size(); ++t) {
        const uint32_t* const slabs[2] = {old_slab, old_slab0};
        for (int which = 0; which < 2; ++which) {
            fresh[t].old_prestep[which] = slabs[which] ? slabs[which] + live[t].prestep_off : nullptr;
            fresh[t].old_accum[which] = slabs[which] ? slabs[which] + live[t].accum_off : nullptr;
        }
        fresh[t].old_stride = live[t].stride;
    }
    adopt_type_batches(c, std::move(fresh), batch_count, has_fallback);
    st = build_constraints(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return st;
}

int32_t bepuhip_replan(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    if (c->building) return fail(BEPUHIP_E_STATE, "replan between begin_constraints and end_constraints");
    if (!c->built) return fail(BEPUHIP_E_STATE, "replan without constraints (begin / set_type_batch / end first)");
    HIP_TRY(hipSetDevice(c->device));
    cancel_replan_job(c);  // (a plan being computed in the background describes what this call is about to replace)
    int32_t st = BEPUHIP_OK;
    if ((st = flush_structural(c)) != BEPUHIP_OK) return st;     // everything the caller has been told is in the rows
    if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;  // ... and the rows are in the caller's order
    HIP_TRY(hipStreamSynchronize(c->stream));
    const std::vector<LiveTypeBatch> live = live_layout(c);
    uint32_t* const old_slab = c->d_slab; uint32_t* const old_slab0 = c->d_slab0;
    c->d_slab = c->d_slab0 = nullptr;  // kept past free_constraints: the new rows are filled from them
    const int batch_count = c->batch_count;
    const bool has_fallback = c->has_fallback;
    free_constraints(c);
    st = rebuild_from_rows(c, live, old_slab, old_slab0, batch_count, has_fallback);
    hipFree(old_slab);
    if (old_slab0) hipFree(old_slab0);
    if (st != BEPUHIP_OK) free_constraints(c);
    return st;
}

// ---- The same, with the planning off the caller's thread (round 6; VERDICT r5 next #6a) ----
// bepuhip_replan costs what the host planner costs — 20-36 ms for the 100k-box pile or the 1M-constraint tube — on the thread that runs the frames. begin takes the
// snapshot (references read back: about a millisecond) and starts a worker that plans it; the frames go on, on the launch-per-batch rows, structural updates included —
// every structural call that succeeds is also written to the job's log. commit (when the worker is done, or waiting for it) makes the new plan the context's: the
// plan is for the constraints of the snapshot, so the context first becomes what it was then — type batches, counts, the plan's tables — and then the log is played
// through the very entry points the caller used (bepuhip_apply_structural_ops on the island layout: the plan absorbs what it can, exactly as it would have frame by
// frame, and a context that an operation of the log drops to the launch-per-batch schedule is simply back where it was); last, prestep data and accumulated impulses of
// every constraint alive NOW move from the old rows (the caller's order, current values: the frames have been solving them) into whatever layout the replay ended in.
// If the replay does not end with the type batches the rows hold — it always should — the commit falls back to the synchronous path from the same rows.
static void cancel_replan_job(bepuhip_ctx* c) {
    if (!c->replan_job) return;
    ReplanJob* job = c->replan_job;
    c->replan_job = nullptr;
    if (job->worker.joinable()) job->worker.join();
    delete job;
}
static void replan_log(bepuhip_ctx* c, int kind, int batch, int type_id, int index, int slot, int reference, const int32_t* refs, const float* prestep) {
    ReplanJob* job = c->replan_job;
    if (!job || c->replan_replaying) return;
    bepuhip_structural_op op = {kind, batch, type_id, index, slot, reference, (int32_t)job->payload.size(), 0};
    if (refs && prestep) {
        TypeInfoH info;
        if (!type_info(type_id, info)) return;
        for (int k = 0; k < info.bodies; ++k) job->payload.push_back((uint32_t)refs[k]);
        for (int f = 0; f < info.prestep; ++f) { uint32_t w; memcpy(&w, &prestep[f], 4); job->payload.push_back(w); }
    }
    job->log.push_back(op);
}

int32_t bepuhip_replan_begin(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    if (c->building) return fail(BEPUHIP_E_STATE, "replan between begin_constraints and end_constraints");
    if (!c->built) return fail(BEPUHIP_E_STATE, "replan without constraints (begin / set_type_batch / end first)");
    if (c->in_substep_event) return fail(BEPUHIP_E_STATE, "replan inside a substep event handler");
    if (c->replan_job) return fail(BEPUHIP_E_STATE, "a re-plan is already in flight (bepuhip_replan_commit)");
    if (c->group_world > 1) return fail(BEPUHIP_E_UNSUPPORTED, "a member of a device group re-plans with bepuhip_replan (every member, at the same frame)");
    HIP_TRY(hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    int32_t st = BEPUHIP_OK;
    if ((st = flush_structural(c)) != BEPUHIP_OK) return st;
    if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;  // the frames in between run on the caller's-order rows, where every structural update is a row operation
    if ((st = flush_structural(c)) != BEPUHIP_OK) return st;       // (descriptors of the launch-per-batch schedule)
    std::unique_ptr<ReplanJob> job(new ReplanJob());
    if ((st = snapshot_type_batches(c, c->d_slab, live_layout(c), job->shadow.tbs)) != BEPUHIP_OK) return st;
    bepuhip_ctx& s = job->shadow;
    s.device = c->device; s.W = c->W; s.flags = c->flags; s.host_values = false;
    s.fallback_threshold = c->fallback_threshold;
    s.group_world = 1; s.group_rank = 0;
    job->batch_count = c->batch_count; job->has_fallback = c->has_fallback;
    if (c->has_fallback)
        for (auto& tb : s.tbs) if (tb.batch == c->fallback_threshold) job->fallback_refs.push_back(tb.refs_soa);
    {   // the planner's inputs besides the type batches (adopt_type_batches, on the shadow)
        std::vector<HostTypeBatch> tbs = std::move(s.tbs);
        adopt_type_batches(&s, std::move(tbs), c->batch_count, c->has_fallback);
    }
    job->begin_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    job->started = std::chrono::steady_clock::now();
    ReplanJob* raw = job.release();
    raw->worker = std::thread([raw] {
        tl_plan_thread_cap = std::max(1, env_int("BEPUHIP_REPLAN_THREADS", 4));  // beside the frames, not instead of them (bepu_cluster_plan.h)
        plan_clusters(&raw->shadow, raw->plan);
        if (raw->plan.enabled) {  // what build_constraints would do with the plan's host half at the commit — and what the first structural update would build lazily
            soft_setup(&raw->shadow, raw->plan);
            soft_warm(&raw->shadow);
        }
        raw->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - raw->started).count();
        raw->done.store(1, std::memory_order_release);
    });
    c->replan_job = raw;
    return BEPUHIP_OK;
}

int32_t bepuhip_replan_poll(bepuhip_ctx* c, int32_t* state_out) {
    if (!c || !state_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *state_out = !c->replan_job ? 0 : (c->replan_job->done.load(std::memory_order_acquire) ? 2 : 1);
    return BEPUHIP_OK;
}

int32_t bepuhip_replan_cancel(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    cancel_replan_job(c);
    return BEPUHIP_OK;
}

int32_t bepuhip_replan_commit(bepuhip_ctx* c, int32_t wait, int32_t* committed_out) {
    if (committed_out) *committed_out = 0;
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    if (!c->replan_job) return fail(BEPUHIP_E_STATE, "no re-plan in flight (bepuhip_replan_begin)");
    if (c->building || c->in_substep_event) return fail(BEPUHIP_E_STATE, "replan_commit between begin_constraints and end_constraints, or inside a substep event handler");
    if (!wait && !c->replan_job->done.load(std::memory_order_acquire)) return BEPUHIP_OK;  // still planning: nothing happened
    HIP_TRY(hipSetDevice(c->device));
    std::unique_ptr<ReplanJob> job(c->replan_job);
    c->replan_job = nullptr;
    if (job->worker.joinable()) job->worker.join();
    const bool stats = env_int("BEPUHIP_PLAN_STATS", 0) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    int32_t st = BEPUHIP_OK;
    if ((st = flush_structural(c)) != BEPUHIP_OK) return st;       // the rows show every operation of the log
    if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;  // (they have been in the caller's order since begin)
    HIP_TRY(hipStreamSynchronize(c->stream));
    const std::vector<LiveTypeBatch> live = live_layout(c);
    const int live_batch_count = c->batch_count;
    const bool live_has_fallback = c->has_fallback;
    uint32_t* const old_slab = c->d_slab; uint32_t* const old_slab0 = c->d_slab0;
    c->d_slab = c->d_slab0 = nullptr;
    free_constraints(c);
    // (1) the context as it was when the job began, on the plan the worker made for it (values: none yet)
    adopt_type_batches(c, std::move(job->shadow.tbs), job->batch_count, job->has_fallback);
    c->referenced_bodies = job->shadow.referenced_bodies; c->total_constraints = job->shadow.total_constraints;  // (counted before the planner permuted the references)
    st = build_constraints(c, &job->plan, &job->fallback_refs, job->plan.enabled ? &job->shadow : nullptr);
    const auto t1 = std::chrono::steady_clock::now();
    // (2) the frames' structural operations, through the public entry points
    int32_t failed = -1;
    if (st == BEPUHIP_OK && !job->log.empty()) {
        c->replan_replaying = true;
        st = bepuhip_apply_structural_ops(c, job->log.data(), (int32_t)job->log.size(), job->payload.empty() ? nullptr : job->payload.data(), (int32_t)job->payload.size(), &failed);
        c->replan_replaying = false;
    }
    if (st == BEPUHIP_OK) st = flush_structural(c);
    const auto t2 = std::chrono::steady_clock::now();
    // (3) the type batches must be the ones the rows hold
    bool same = st == BEPUHIP_OK;
    size_t nonempty = 0;
    for (auto& tb : c->tbs) nonempty += tb.count > 0;
    for (auto& lt : live) {
        if (!same) break;
        if (lt.count == 0) continue;
        --nonempty;
        HostTypeBatch* tb = find_tb(c, lt.batch, lt.type_id);
        same = tb && tb->count == lt.count;
        if (same && live_has_fallback && lt.batch == c->fallback_threshold)
            for (int i = 0; i < lt.count && same; ++i) same = ((size_t)i < lt.occupied.size() && lt.occupied[i]) == ((size_t)i < tb->occupied.size() && tb->occupied[i]);
    }
    same = same && nonempty == 0 && c->has_fallback == live_has_fallback;
    if (!same) {  // never expected: the synchronous path from the same rows
        const std::string why = st != BEPUHIP_OK ? std::string(bepuhip_last_error()) : std::string("the replayed type batches differ from the rows'");
        if (stats || env_int("BEPUHIP_REPLAN_STRICT", 0) != 0) fprintf(stderr, "bepuhip replan_commit: replay failed (%s; operation %d of %zu): re-planning synchronously\n", why.c_str(), failed, job->log.size());
        HIP_TRY(hipStreamSynchronize(c->stream));
        free_constraints(c);
        st = env_int("BEPUHIP_REPLAN_STRICT", 0) != 0 ? fail(BEPUHIP_E_STATE, "replan_commit: replay failed: " + why) : rebuild_from_rows(c, live, old_slab, old_slab0, live_batch_count, live_has_fallback);
    } else {
        // (4) current values into the layout the replay ended in (the island layout's index tables are on the device since the flush; null = the caller's order)
        for (auto& lt : live) {
            if (lt.count == 0) continue;
            HostTypeBatch* tb = find_tb(c, lt.batch, lt.type_id);
            const int* index = nullptr;
            if ((st = device_index_of(c, tb, &index)) != BEPUHIP_OK) break;
            const int blocks = (lt.count + 255) / 256;
            uint32_t* const fresh[2] = {c->d_slab, c->d_slab0};
            const uint32_t* const old[2] = {old_slab, old_slab0};
            for (int which = 0; which < 2; ++which) {
                if (!fresh[which] || !old[which]) continue;
                if (lt.info.prestep > 0)
                    hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, old[which] + lt.prestep_off, lt.stride, fresh[which] + tb->prestep_off, tb->stride, index, lt.count, lt.info.prestep);
                if (lt.info.impulse > 0)
                    hipLaunchKernelGGL(permute_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, old[which] + lt.accum_off, lt.stride, fresh[which] + tb->accum_off, tb->stride, index, lt.count, lt.info.impulse);
            }
        }
        if (st == BEPUHIP_OK) { HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(c->stream)); }
    }
    hipFree(old_slab);
    if (old_slab0) hipFree(old_slab0);
    if (st != BEPUHIP_OK) { free_constraints(c); return st; }
    if (stats) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "bepuhip replan: begin %.2f ms on the caller's thread, planning %.2f ms on the worker, commit %.2f ms on the caller's thread (adopt %.2f, replay of %zu operations + flush %.2f, values %.2f) -> schedule %d\n",
                job->begin_ms, job->plan_ms, ms(t0, std::chrono::steady_clock::now()), ms(t0, t1), job->log.size(), ms(t1, t2), ms(t2, std::chrono::steady_clock::now()),
                !c->clusters_enabled ? 0 : (c->clusters_shared ? 2 : 1));
    }
    if (committed_out) *committed_out = 1;
    return BEPUHIP_OK;
}

int32_t bepuhip_get_schedule(bepuhip_ctx* c, int32_t* schedule_out) {
    if (!c || !schedule_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *schedule_out = !c->clusters_enabled ? 0 : (c->clusters_shared ? 2 : 1);
    return BEPUHIP_OK;
}

int32_t bepuhip_get_kernel_family(bepuhip_ctx* c, int32_t* family_out) {
    if (!c || !family_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *family_out = c->last_kernel_family;
    return BEPUHIP_OK;
}

static UnitKey special_key(const bepuhip_ctx* c, int threads);
static void special_request(bepuhip_ctx* c, const UnitKey& key);
static int island_launch_threads(const bepuhip_ctx* c, bool conserving);
// The island kernel compiled for exactly this context's constraint types (include/bepuhip.h; bepu_unit_cache.h).
int32_t bepuhip_specialise_units(bepuhip_ctx* c, int32_t wait, int32_t* state_out) {
    if (state_out) *state_out = kUnitUnavailable;
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    if (c->building) return fail(BEPUHIP_E_STATE, "specialise_units between begin_constraints and end_constraints");
    c->specialise_auto = true;  // from here on every plan of this context asks for its unit
    if (!c->built || !c->clusters_enabled || c->type_mask == 0) return BEPUHIP_OK;  // nothing to specialise yet (or the launch-per-batch schedule, whose kernels are not per family)
    HIP_TRY(hipSetDevice(c->device));
    special_request(c, special_key(c, island_launch_threads(c, false)));
    if (wait) unit_wait(c->special_unit);
    const int state = c->special_unit->state.load(std::memory_order_acquire);
    if (state_out) *state_out = state;
    if (wait && state == kUnitFailed) return fail(BEPUHIP_E_DEVICE, "specialise_units: " + c->special_unit->why);
    return BEPUHIP_OK;
}

// The same object without a context or a device: found in the unit cache or compiled into it, now, on the caller's thread (bepuphysics2_amd/build.py prebuilds the
// units of the BASELINE.json scenes with it, so that they travel with the tree).
int32_t bepuhip_prebuild_unit(uint64_t type_mask, int32_t threads_budget, int32_t split_plan, char* path_out, int32_t path_capacity) {
    if (type_mask == 0 || (threads_budget != 1024 && threads_budget != 768 && threads_budget != 512)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "prebuild_unit: an empty type mask, or a budget other than 1024 / 768 / 512 threads");
    for (int t = 0; t < 64; ++t) { TypeInfoH info; if (((type_mask >> t) & 1ull) && !type_info(t, info)) return fail(BEPUHIP_E_UNSUPPORTED, "prebuild_unit: unknown constraint type id " + std::to_string(t)); }
    std::string why;
    const std::string path = unit_obtain(UnitKey{(unsigned long long)type_mask, threads_budget, split_plan != 0}, why);
    if (path.empty()) return fail(BEPUHIP_E_UNSUPPORTED, "prebuild_unit: " + why);
    if (path_out && path_capacity > 0) { strncpy(path_out, path.c_str(), (size_t)path_capacity - 1); path_out[path_capacity - 1] = 0; }
    return BEPUHIP_OK;
}

int32_t bepuhip_set_constrained_kinematics(bepuhip_ctx* c, const int32_t* indices, int32_t count) {
    if (!c || count < 0 || (count > 0 && !indices)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad kinematic list");
    for (int i = 0; i < count; ++i)
        if (indices[i] < 0 || indices[i] >= c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "kinematic body index out of range (call set_bodies first)");
    // The list the context already holds (a host that uploads its constraints again hands the same Solver.ConstrainedKinematicHandles over): every call that changes what
    // the flags depend on rebuilds them with the list it finds here, so they are current.
    if (count == c->kin_count && (count == 0 || memcmp(indices, c->kin_indices.data(), (size_t)count * 4) == 0)) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_kin) { hipFree(c->d_kin); c->d_kin = nullptr; }
    c->kin_count = count;
    c->kin_indices.assign(indices, indices + count);
    if (count > 0) {
        HIP_TRY(hipMalloc((void**)&c->d_kin, (size_t)count * 4));
        HIP_TRY(copy_sync(c, c->d_kin, indices, (size_t)count * 4, hipMemcpyHostToDevice));
    }
    return rebuild_flags(c);
}

int32_t bepuhip_sync(bepuhip_ctx* c);

// (`attached`: the one launch inside the scope takes the two events itself — hipExtLaunchKernel(..., a, b, 0): they carry the kernel's own start and end, what a kernel
// trace shows for it. Events recorded on the stream around a launch are marker packets in front of and behind it and add their own few microseconds to what they
// bracket: 54.3 us for a 50 us kernel where the attached pair says 51.6, tools/probes/launch_gap_probe.hip.)
struct Timed {
    bepuhip_ctx* c; int family; hipEvent_t a = nullptr, b = nullptr; bool attached = false;
    Timed(bepuhip_ctx* c_, int f, bool attach = false) : c(c_), family(f), attached(attach && c_->profiling) {
        if (c->profiling) { hipEventCreate(&a); hipEventCreate(&b); if (!attached) hipEventRecord(a, c->stream); }
    }
    ~Timed() {
        if (c->profiling) {
            if (!attached) hipEventRecord(b, c->stream);
            hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            c->prof_ms[family] += ms; c->prof_launches[family] += 1;
            hipEventDestroy(a); hipEventDestroy(b);
        }
    }
};

static StepParams make_params(const bepuhip_ctx* c, const bepuhip_integrator* in, float dt_for_callbacks, float dt, float inv_dt) {
    // DemoPoseIntegratorCallbacks.PrepareForIntegration (Demos/DemoCallbacks.cs:79-86)
    StepParams sp;
    sp.velocity_model = c->velocity_model.model;  // bepuhip_set_velocity_model: which IntegrateVelocity the device evaluates
    sp.callback_dt = dt_for_callbacks;
    sp.cx = c->velocity_model.center[0]; sp.cy = c->velocity_model.center[1]; sp.cz = c->velocity_model.center[2];
    sp.radial = dt_for_callbacks * c->velocity_model.gravity;  // PlanetDemo.cs:39: gravityDt = dt * Gravity
    sp.body_gravity = c->d_body_gravity;
    float l = 1 - in->linear_damping, a = 1 - in->angular_damping;
    l = l < 0 ? 0 : (l > 1 ? 1 : l);
    a = a < 0 ? 0 : (a > 1 ? 1 : a);
    sp.lin_damp = powf(l, dt_for_callbacks);
    sp.ang_damp = powf(a, dt_for_callbacks);
    sp.gx = in->gravity[0] * dt_for_callbacks; sp.gy = in->gravity[1] * dt_for_callbacks; sp.gz = in->gravity[2] * dt_for_callbacks;
    sp.dt = dt; sp.inv_dt = inv_dt;
    sp.angular_mode = in->angular_integration_mode;
    return sp;
}

// Bodies the reference re-transforms in substep 0 of the conserving angular modes (see momentum_requirk_kernel): lists per batch, built when a solve first asks for
// such a mode and again after structural updates (the lists describe the topology). Bundles are W consecutive constraints in the CALLER's order, so the references
// are read back from the device rows and brought into that order through the type batch's index tables (island layouts are permuted by cluster).
static int32_t build_requirk_lists(bepuhip_ctx* c) {
    if (c->d_requirk) { hipFree(c->d_requirk); c->d_requirk = nullptr; }
    c->requirk_begin.clear();
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int W = c->W;
    std::vector<std::vector<int32_t>> host_refs(c->tbs.size());  // [type batch][body slot * count + caller's index]
    int universe = 0;
    for (size_t t = 0; t < c->tbs.size(); ++t) {
        HostTypeBatch& tb = c->tbs[t];
        const int nb = tb.info.bodies;
        if (tb.count == 0) continue;
        std::vector<int32_t> rows((size_t)nb * tb.stride);
        HIP_TRY(copy_sync(c, rows.data(), c->d_slab + tb.refs_off, rows.size() * 4, hipMemcpyDeviceToHost));
        host_refs[t].assign((size_t)nb * tb.count, -1);
        for (int i = 0; i < tb.count; ++i) {
            const int d = tb.perm.empty() ? i : tb.perm_inverse(i);
            for (int k = 0; k < nb; ++k) {
                const int32_t r = rows[(size_t)k * tb.stride + d];
                host_refs[t][(size_t)k * tb.count + i] = r;
                if (r >= 0) universe = std::max(universe, (r & kRefMask) + 1);
            }
        }
    }
    std::vector<int32_t> first_batch(universe, INT32_MAX);
    for (size_t t = 0; t < c->tbs.size(); ++t)
        for (int32_t r : host_refs[t])
            if ((uint32_t)r < kDynamicLimit) first_batch[r] = std::min(first_batch[r], c->tbs[t].batch);
    // Lists per LAUNCH: launch b < sync_batches is batch b; the sequential fallback batch follows as one launch per dependency level (build_descriptors). There a body
    // may sit in several bundles: the one at its earliest slot (type batch, index) integrates it if the batch is the first to see it (Solver_Solve.cs:1003-1019), every
    // other occurrence is an ordinary non-integrating lane — re-transformed like any other when its bundle integrates somebody, right before ITS constraint runs, i.e.
    // at that row's level (the levels are walked exactly as build_descriptors walks them).
    const int sync_batches = c->has_fallback ? c->fallback_threshold : c->batch_count;
    std::vector<std::vector<int32_t>> lists(std::max(c->launch_count, sync_batches));
    std::vector<uint64_t> earliest(c->has_fallback ? universe : 0, UINT64_MAX);  // per body: its earliest (fallback type batch ordinal << 32 | index) in the fallback batch
    std::vector<std::vector<int32_t>> row_level;                                 // per fallback type batch ordinal: the level of every row
    if (c->has_fallback) {
        std::vector<int32_t> last_level(universe, -1);
        size_t ord = 0;
        for (size_t t = 0; t < c->tbs.size(); ++t) {
            const HostTypeBatch& tb = c->tbs[t];
            if (tb.batch != c->fallback_threshold) continue;
            row_level.emplace_back(tb.count, -1);
            for (int i = 0; i < tb.count; ++i) {
                if (tb.count == 0 || host_refs[t][i] == -1) continue;
                int level = 0;
                for (int k = 0; k < tb.info.bodies; ++k) {
                    const int32_t r = host_refs[t][(size_t)k * tb.count + i];
                    if ((uint32_t)r < kDynamicLimit) { level = std::max(level, last_level[r] + 1); earliest[r] = std::min(earliest[r], ((uint64_t)ord << 32) | (uint32_t)i); }
                }
                for (int k = 0; k < tb.info.bodies; ++k) {
                    const int32_t r = host_refs[t][(size_t)k * tb.count + i];
                    if ((uint32_t)r < kDynamicLimit) last_level[r] = level;
                }
                row_level.back()[i] = level;
            }
            ++ord;
        }
    }
    std::vector<BitMark> marks;
    size_t fallback_ord = 0;
    for (size_t t = 0; t < c->tbs.size(); ++t) {
        const HostTypeBatch& tb = c->tbs[t];
        const bool fallback = c->has_fallback && tb.batch == c->fallback_threshold;
        const size_t ord = fallback ? fallback_ord++ : 0;
        if (tb.batch == 0 || tb.count == 0) continue;  // batch 0 always integrates (Solver_Solve.cs:188-194): no conditional bundles
        for (int k = 0; k < tb.info.bodies; ++k) {
            const int32_t* refs = host_refs[t].data() + (size_t)k * tb.count;
            auto integrates = [&](int i) {  // this lane is the one that integrates its body
                const int32_t r = refs[i];
                if ((uint32_t)r >= kDynamicLimit || first_batch[r] != tb.batch) return false;
                return !fallback || earliest[r] == (((uint64_t)ord << 32) | (uint32_t)i);
            };
            for (int b0 = 0; b0 < tb.count; b0 += W) {
                const int b1 = std::min(tb.count, b0 + W);
                bool any = false;
                for (int i = b0; i < b1; ++i) any |= integrates(i);
                if (!any) continue;
                for (int i = b0; i < b1; ++i) {
                    if ((uint32_t)refs[i] >= kDynamicLimit || integrates(i)) continue;
                    const int launch = fallback ? sync_batches + row_level[ord][i] : tb.batch;
                    if (launch >= 0 && (size_t)launch < lists.size()) lists[launch].push_back(refs[i]);
                    if (c->clusters_enabled) {  // island layouts: the lane that holds the body does it (kLrefRequirk / kRankRequirk, bepu_cluster_kernel.h) — the sequential
                        // fallback batch's lanes too (round 5): its items wait for every earlier item of their batch, so "right before ITS constraint runs" is the lane's own gate
                        const int d = tb.perm.empty() ? i : c->tbs[t].perm_inverse(i);
                        const int lref_rows = (tb.info.bodies + 1) / 2;
                        if (c->clusters_shared) marks.push_back({tb.lrefs_off + (size_t)(lref_rows + k) * tb.stride + d, 1u << 18});
                        else marks.push_back({tb.lrefs_off + (size_t)(k / 2) * tb.stride + d, (1u << 14) << (16 * (k & 1))});
                    }
                }
            }
        }
    }
    std::vector<int32_t> flat;
    c->requirk_begin.assign(lists.size() + 1, 0);
    for (size_t b = 0; b < lists.size(); ++b) { c->requirk_begin[b] = (int)flat.size(); flat.insert(flat.end(), lists[b].begin(), lists[b].end()); }
    c->requirk_begin[lists.size()] = (int)flat.size();
    if (!flat.empty()) {
        HIP_TRY(hipMalloc((void**)&c->d_requirk, flat.size() * 4));
        HIP_TRY(copy_sync(c, c->d_requirk, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    }
    // the island layouts' bits: the old ones go (their words may belong to other constraints by now), the new ones come — in the working rows and in the snapshot
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<BitMark>& list = pass == 0 ? c->requirk_marks : marks;
        if (list.empty() || !c->d_slab) continue;
        BitMark* d_marks = nullptr;
        HIP_TRY(hipMalloc((void**)&d_marks, list.size() * sizeof(BitMark)));
        HIP_TRY(copy_sync(c, d_marks, list.data(), list.size() * sizeof(BitMark), hipMemcpyHostToDevice));
        for (uint32_t* slab : {c->d_slab, c->d_slab0})
            if (slab) hipLaunchKernelGGL(mark_bits_kernel, dim3(((int)list.size() + 255) / 256), dim3(256), 0, c->stream, slab, (const BitMark*)d_marks, (int)list.size(), pass);
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(d_marks);
    }
    c->requirk_marks.swap(marks);
    c->requirk_stale = false;
    clear_graphs(c);  // graphs captured for a conserving mode hold the old lists
    return BEPUHIP_OK;
}

static void enqueue_requirk(bepuhip_ctx* c, int substep, int batch, const StepParams& sp) {
    if (substep != 0 || sp.angular_mode == 0 || c->requirk_begin.empty() || batch + 1 >= (int)c->requirk_begin.size()) return;  // `batch`: launch index (a level of the fallback batch counts)
    const int n = c->requirk_begin[batch + 1] - c->requirk_begin[batch];
    if (n > 0)
        hipLaunchKernelGGL(momentum_requirk_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_bodies, (const int*)(c->d_requirk + c->requirk_begin[batch]), n, sp);
}

// Launch policy of the island schedules (bepu_host_state.h). Candidates — all bit-identical in their results:
//   0 plain constraint-row accesses, 1 non-temporal row accesses, 2 plain rows + one 8 KB span of code touched per work item (two spans measured no better on
//   either box class: profiles/r03_code_touch_ab_*).
// Which one is fastest depends on the box class (on the slow class an instruction fetch that misses L2 is what costs; the touch keeps the code there).
// The first kPolicySamples solves after an upload cycle through the candidates, each launch under its own event pair. Once the last of them has FINISHED
// (hipEventQuery: a solve never blocks for the policy's sake) the medians are compared: the fastest candidate stays if it beats plain by more than 2 %, else plain.
// The decision is kept per (device, plan kind, workgroup size) for the life of the process, so that hosts that upload every frame settle once.
constexpr int kPolicyCandidates = 3, kPolicyRounds = 5, kPolicySamples = kPolicyCandidates * kPolicyRounds;
static std::mutex g_policy_mutex;
static std::map<std::tuple<int, int, int, int>, int> g_policy_cache;
static std::tuple<int, int, int, int> policy_key(const bepuhip_ctx* c, int threads) { return {c->device, c->clusters_shared ? 1 : 0, c->has_widened_types ? 2 : (c->has_joint_types ? 1 : 0), threads}; }  // (the type-set family: each has its own units)
static void settle_row_policy(bepuhip_ctx* c, int threads, bool may_block) {
    const int pinned = env_int("BEPUHIP_ROW_POLICY", -1);
    if (pinned >= 0 && pinned < kPolicyCandidates) { c->row_policy = pinned; return; }
    if (env_int("BEPUHIP_POLICY_CACHE", 1) != 0 && c->policy_samples == 0) {
        std::lock_guard<std::mutex> lock(g_policy_mutex);
        auto found = g_policy_cache.find(policy_key(c, threads));
        if (found != g_policy_cache.end()) { c->row_policy = found->second; return; }
    }
    if (!c->policy_events[0][0])
        for (auto& pair : c->policy_events) { hipEventCreate(&pair[0]); hipEventCreate(&pair[1]); }
    if (c->policy_samples < kPolicySamples) return;
    hipEvent_t last = c->policy_events[kPolicySamples - 1][1];
    if (may_block) hipEventSynchronize(last);
    else if (hipEventQuery(last) != hipSuccess) { (void)hipGetLastError(); return; }  // still running: this solve launches plain and is not a sample
    float median[kPolicyCandidates];
    for (int cand = 0; cand < kPolicyCandidates; ++cand) {
        std::vector<float> ms;
        for (int round = 1; round < kPolicyRounds; ++round) {  // the first round warms clocks and caches
            float t = 0;
            const int i = round * kPolicyCandidates + cand;
            if (hipEventElapsedTime(&t, c->policy_events[i][0], c->policy_events[i][1]) == hipSuccess) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        median[cand] = ms.empty() ? 1e30f : ms[ms.size() / 2];
    }
    int best = 0;
    for (int cand = 1; cand < kPolicyCandidates; ++cand) if (median[cand] < median[best]) best = cand;
    if (!(median[best] < 0.98f * median[0])) best = 0;  // within the noise of plain: keep plain
    c->row_policy = best;
    if (env_int("BEPUHIP_POLICY_CACHE", 1) != 0) {
        std::lock_guard<std::mutex> lock(g_policy_mutex);
        g_policy_cache[policy_key(c, threads)] = best;
    }
    if (env_int("BEPUHIP_PLAN_STATS", 0))
        fprintf(stderr, "bepuhip launch policy: plain %.4f, non-temporal rows %.4f, code touch %.4f ms per launch (medians of %d) -> %d\n", median[0], median[1], median[2],
                kPolicyRounds - 1, best);
}

// BEPUHIP_DEBUG_JITTER=<seed>: schedule fuzzing of the island kernels (bepu_cluster_kernel.h, jitter_nap) — a different nap pattern for every launch of the process,
// reproducible per seed as long as the launches come in the same order. 0 / unset: off.
static unsigned debug_jitter_seed() {
    static std::atomic<unsigned> launches{0};
    const unsigned seed = (unsigned)env_int("BEPUHIP_DEBUG_JITTER", 0);
    if (seed == 0u) return 0u;
    const unsigned mixed = seed * 0x9E3779B1u + launches.fetch_add(1u) * 0x85EBCA77u;
    return mixed ? mixed : 1u;
}

// Cooperative launches are issued by one host thread at a time: two contexts of one process launching them concurrently from two threads (the in-process lattice
// harness) leave the runtime in a state that crashes at process exit (ROCm 7.0, observed: every test passes, then a segmentation fault in the exit handlers).
static std::mutex g_cooperative_launch;

// Threads per cluster workgroup: 16 waves for whole-island plans, 8 for split plans (BEPUHIP_CLUSTER_THREADS / BEPUHIP_SPLIT_THREADS override).
static int cluster_threads(const bepuhip_ctx* c) {
    const int req = c->clusters_shared ? env_int("BEPUHIP_SPLIT_THREADS", kSplitClusterThreads) : env_int("BEPUHIP_CLUSTER_THREADS", kClusterThreads);
    // (split plans: the units that exist are compiled for 512 threads — and 768 for the hot and contacts families; the 1024-thread split units spilled 372 - 3,437 VGPRs,
    // were never a default and are no longer built)
    // (BEPUHIP_SPLIT_ALLOW_1024=1, experiments: sixteen waves on a split plan — only a unit compiled for the scene's exact types exists for that, bepuhip_specialise_units first)
    const int most = c->clusters_shared ? (c->has_widened_types ? 512 : (env_int("BEPUHIP_SPLIT_ALLOW_1024", 0) ? 1024 : 768)) : 1024;
    return std::max(64, std::min(most, req / 64 * 64));
}
// The momentum-conserving angular modes run the island schedule through the kernel units that carry their code (round 3; BEPUHIP_CONSERVING_CLUSTERS=0: launch-per-batch
// as in round 2), which exist for the default workgroup sizes.
static bool group_chain_hazard(const bepuhip_ctx* c, int launches) { return c->group_world > 1 && c->clusters_shared && launches > 1; }
static bool island_schedule_applies(const bepuhip_ctx* c, int substeps, const bepuhip_integrator* in) {
    // (round 5: a sequential fallback batch runs the island schedule under the conserving modes as well — the substep-0 re-transformation of a non-integrating lane is a
    // bit on that lane like in every other batch, build_requirk_lists; BEPUHIP_FALLBACK_CONSERVING_CLUSTERS=0: the per-level lists of the launch-per-batch schedule)
    // (round 5: any number of substeps — a step of more than kMaxClusterSubsteps is a chain of launches, enqueue_island_launches)
    // (round 6, ADVICE r5: NOT in a device group on a split plan — between two launches of a chain every slot, ghost copies included, is staged again from THIS device's
    // HBM, and a ghost's home cluster may have run on another device: such a step runs the launch-per-batch schedule, on every member over the whole scene — identical
    // results on all of them, so the owners' merge is unaffected)
    if (group_chain_hazard(c, (substeps + kMaxClusterSubsteps - 1) / kMaxClusterSubsteps)) return false;
    return c->clusters_enabled && !(c->has_fallback && in->angular_integration_mode != 0 && env_int("BEPUHIP_FALLBACK_CONSERVING_CLUSTERS", 1) == 0) &&
           cluster_lds_bytes(c->cluster_planes, c->cluster_max_slots, c->cluster_max_items, c->clusters_shared) <= kLdsBudgetBytes &&
           (in->angular_integration_mode == 0 || (conserving_variant_exists(cluster_threads(c), c->clusters_shared) && c->d_trace == nullptr && env_int("BEPUHIP_CONSERVING_CLUSTERS", 1) != 0));
}
// The island schedule's launch for substeps [base, base + count) of a step of `substeps` substeps (the whole step when that is one launch: the usual case). A step is a
// CHAIN of such launches when it has more substeps than a launch's arguments carry (kMaxClusterSubsteps), or when the host wants to be called between substeps
// (bepuhip_solve_with_substep_events: one substep per launch). Between two launches of a chain the bodies are in HBM with the pose of the last substep run and their
// velocities; "first substep of the step" rules and the trailing pose integration follow the STEP (ClusterParams.substep_base / final_launch).
// The specialised unit a launch of `threads` threads per cluster would use: the context's type set, the register budget of the workgroup size, the plan kind.
static UnitKey special_key(const bepuhip_ctx* c, int threads) {
    const int budget = c->clusters_shared ? (threads <= 512 ? 512 : (threads <= 768 ? 768 : 1024)) : cluster_variant_threads(threads);
    return UnitKey{c->type_mask, budget, c->clusters_shared};
}
static void special_request(bepuhip_ctx* c, const UnitKey& key) {
    c->special_unit = unit_request(key, c->device, kLdsBudgetBytes);
    c->special_mask = key.mask; c->special_budget = key.budget; c->special_shared = key.shared;
}
// The events behind bepuhip_last_solve_ms. Off unless bepuhip_set_solve_timing asked for them (round 6, last session): an event recorded on a stream is a marker packet of
// its own with a completion signal, and one in front of and one behind the launch put 7 us between two back-to-back solves of the island schedule — 11 us from the end of
// one launch to the start of the next where two plain launches of the same shape take 2.3 (tools/probes/launch_gap_probe.hip, profiles/r06_s54_launch_gap_probe.txt:
// hipExtLaunchKernel's own event pair costs 4.4). A solve that is not timed enqueues its kernels and nothing else.
static hipError_t solve_event(bepuhip_ctx* c, bool start) {
    if (start) c->solve_timed = c->solve_timing;
    if (!c->solve_timed) return hipSuccess;
    return hipEventRecord(start ? c->ev_start : c->ev_stop, c->stream);
}
static int island_launch_threads(const bepuhip_ctx* c, bool conserving) {
    int threads = cluster_threads(c);
    if (c->split_twelve_waves && c->clusters_shared && threads == kSplitClusterThreads && !conserving && !c->has_widened_types) threads = 768;
    return threads;
}
static void enqueue_island_launch(bepuhip_ctx* c, float dt, int substeps, int base, int count, const int32_t* iterations, const bepuhip_integrator* in, const StepParams& sp) {
    const float substep_dt = dt / substeps;
    // (whole-island plans: the slot -> body table goes to LDS too when the workgroup has the room for it; the planner never counts on it)
    const bool slot_table = c->clusters_shared || (env_int("BEPUHIP_SLOT_TABLE_IN_LDS", 1) != 0 && cluster_lds_bytes(c->cluster_planes, c->cluster_max_slots, c->cluster_max_items, true) <= kLdsBudgetBytes);
    const size_t lds_bytes = cluster_lds_bytes(c->cluster_planes, c->cluster_max_slots, c->cluster_max_items, slot_table);
    const bool whole_step = base == 0 && count == substeps;
    ClusterParams cp;
    cp.substeps = count; cp.batch_count = c->batch_count; cp.integrate_velocity_for_kinematics = in->integrate_velocity_for_kinematics;
    cp.planes = c->cluster_planes;
    cp.fallback_batch = c->has_fallback ? c->fallback_threshold : -1;
    cp.pass_stage = 0; cp.pass_substep = 0;
    cp.substep_base = base; cp.final_launch = base + count == substeps ? 1 : 0;
    for (int s = 0; s < kMaxClusterSubsteps; ++s) cp.iters[s] = s < count ? iterations[base + s] : 0;
    cp.sp = sp;
    {
        const bool cooperative = c->clusters_shared && env_int("BEPUHIP_COOPERATIVE", (c->flags & BEPUHIP_FLAG_EXCLUSIVE_DEVICE) ? 0 : 1) != 0;
        Timed t(c, 5, !cooperative);  // (a plain launch carries its own event pair when the context is profiling)
        // Waves per cluster: 16 by default (four per SIMD; 12 is as fast when memory latency is low, 8 is slower everywhere); BEPUHIP_CLUSTER_THREADS overrides.
        // Scenes with SURVEY 8(f) types take the 512-thread build: at 128 VGPRs per wave the widened types spill hundreds of registers.
        // (the widened variant too: its 1024-thread build spills 700 VGPRs and is still the faster one — 0.222 against 0.325 ms on the bench graph with widened joints
        // on the pool's slow class of box, 0.195 against 0.193 on the fast one, profiles/r03_s13_widened_slowbox.txt: sixteen waves hide what the scratch traffic costs)
        const bool conserving = in->angular_integration_mode != 0;
        int threads = island_launch_threads(c, conserving);
        // Split plans with many work items per cluster (the ragdoll crowd: 77 - 96 per pass on eight waves, every wave busy, no item waiting for more than its flag's
        // round trip: profiles/r05_s20_crowd_trace.txt) run twelve waves per cluster: the 768-thread unit has 168 VGPRs per wave and, since the manifolds' tails were
        // looked at ("spills on the chain"), no reload in a joint's tail — crowd 0.3784 -> 0.3490 ms same box. Plans with few, heavy items (the pile:
        // 22 - 33 per pass, on its chain of batch steps) lose 5 % there and keep eight waves. Hot types, nonconserving mode: the units that exist at 768 threads.
        const size_t launch_lds = lds_bytes;
        // One launch per step: workgroups [0, clusters) run the islands, the next `body_blocks` integrate the bodies no cluster owns, the last one the
        // constrained kinematic bodies (the per-substep kinematic prepass and the final pass, folded in).
        // (a device group: this context's contiguous range of the plan's clusters; cluster descriptors name items, slots and rows by their absolute positions, so the
        // range is just an offset into the descriptor and cycle arrays)
        const int local_clusters = c->group_world > 1 ? c->cluster_local : c->cluster_count;
        const ClusterDesc* d_my_clusters = c->d_clusters + (c->group_world > 1 ? c->cluster_first : 0);
        unsigned long long* d_my_cycles = c->d_cycles + (c->group_world > 1 ? c->cluster_first : 0);
        TailParams tp;
        tp.flags = c->d_flags; tp.kinlist = c->d_kinlist; tp.staged = c->d_staged;
        tp.body_count = c->body_count; tp.kin_count = c->kinlist_count; tp.cluster_count = local_clusters;
        // IntegrateAfterSubstepping of the bodies no cluster owns: the step's last launch — unless the plan owns every body (the list behind kFlagClustered is a set: as long as
        // the body count, no index is left for these workgroups; a pile of boxes in contact: 196 workgroups that read a flag and leave, a launch of their own on the cooperative path)
        const bool every_body_clustered = c->clustered_dynamic_count == c->body_count && !c->clustered_dirty && c->group_world <= 1;
        tp.body_blocks = cp.final_launch && !every_body_clustered ? (c->body_count + threads - 1) / threads : 0;
        tp.block_offset = 0;
        tp.dt = dt; tp.substep_dt = substep_dt; tp.substep_count = substeps;
        tp.substep_base = base; tp.launch_substeps = count; tp.final_launch = cp.final_launch;
        tp.allow_substeps_for_unconstrained = in->allow_substeps_for_unconstrained; tp.integrate_velocity_for_kinematics = in->integrate_velocity_for_kinematics;
        const float vdt = in->allow_substeps_for_unconstrained ? substep_dt : dt;
        tp.final_sp = make_params(c, in, vdt, vdt, 1.0f / vdt);
        SharedTables st = {c->d_shared_vel, c->d_shared_info, env_int("BEPUHIP_SHARED_POLL", 1), 0u, (int)c->peer_records.size(), c->d_peer_table};
        if (c->clusters_shared) {
            // Event numbers of this launch: [base, base + span). A body sees at most substeps + 255 x passes events per launch; the span is kept even (record parity).
            unsigned passes = 0;
            for (int s = 0; s < count; ++s) passes += 1u + (unsigned)iterations[base + s];
            const unsigned long long span = ((unsigned long long)count + 255ull * passes + 3ull) & ~1ull;
            if ((unsigned long long)c->shared_epoch + 2ull * span > 0xFFFFFFFFull) {  // once in ~2 M steps: start over from cleared records
                hipMemsetAsync(c->d_shared_vel, 0, c->shared_bodies * 4 * sizeof(float4), c->stream);
                c->shared_epoch = 0;
            }
            st.base = c->shared_epoch;
            c->shared_epoch += (unsigned)span;
        }
        void* args[] = {(void*)&d_my_clusters, (void*)&c->d_items, (void*)&c->d_batch_item_begin, (void*)&c->d_cluster_bodies, (void*)&c->d_bodies, (void*)&c->d_slab,
                        (void*)&cp, (void*)&c->cluster_max_slots, (void*)&c->cluster_max_items, (void*)&c->d_trace, (void*)&c->d_status, (void*)&d_my_cycles, (void*)&tp, (void*)&st};
        const bool tr = c->d_trace != nullptr;
        const bool policy_applies = cluster_variant_threads(threads) == (c->clusters_shared ? 512 : 1024) && !conserving;  // the variants that exist in both row policies (a conserving solve neither measures nor follows the policy beyond the code touch)
        int sample = -1, candidate = 0;
        if (policy_applies) {
            if (c->row_policy < 0) settle_row_policy(c, threads, false);
            if (c->row_policy >= 0) candidate = c->row_policy;
            else if (whole_step && c->policy_samples < kPolicySamples) { sample = c->policy_samples++; candidate = sample % kPolicyCandidates; c->policy_threads = threads; }  // each under its own event pair (whole steps only: a link of a chain is not what the policy is for)
        }
        if (conserving && c->row_policy == 2) candidate = 2;
        const bool nt = candidate == 1;
        // (split plans take two spans: their work items run more code — pile 0.3795 -> 0.3695 ms, crowd 0.3976 -> 0.3899 on a slow-class box, the whole-island
        // kernel 0.1875 -> 0.1897 with two; profiles/r04_s30_code_touch_spans_slowbox.txt)
        cp.code_touch = candidate == 2 ? (c->clusters_shared ? 2 : 1) : 0;
        if (const int forced = env_int("BEPUHIP_CODE_TOUCH", -1); forced >= 0) cp.code_touch = std::min(kCodeTouchMaxSpans, forced);  // never beyond the padding behind the unit's kernels
        cp.jitter = debug_jitter_seed();
        // (split plans: every item is a contact item or nearly so and their incremental update waits for memory — crowd 0.3425 -> 0.3373 ms, pile 0.3424 -> 0.3398 with the
        // halves; whole-island plans: the boundary is arithmetic either way, 0.1563 / 0.1564 ms on the headline — profiles/r06_s38_ab_slot_table_halves.txt)
        cp.split_integration = env_int("BEPUHIP_SPLIT_INTEGRATION", c->clusters_shared ? 1 : 0);
        cp.slot_table_in_lds = slot_table && !c->clusters_shared;
        // (BEPUHIP_FORCE_WIDE_FAMILY=1, a developer switch: the all-44 unit for a scene that does not need it — what a family's extra code costs the types both carry, tools/ab_scene.py)
        const bool wide_family = c->has_widened_types || env_int("BEPUHIP_FORCE_WIDE_FAMILY", 0) != 0;
        const void* fn = cluster_kernel_variant(threads, tr, wide_family, c->clusters_shared, nt, conserving, !c->has_joint_types && !wide_family);  // the register budget that matches the workgroup size, the type set that matches the scene
        c->last_kernel_family = wide_family ? kFamilyWide : ((!c->has_joint_types && contacts_family_enabled() && !nt && !conserving && fn == contacts_kernel_variant(threads, tr, c->clusters_shared)) ? kFamilyContacts : kFamilyHot);
        // A unit compiled for exactly this scene's type set (bepu_unit_cache.h), once it is loaded: the plain-row kernel of the same register budget, nothing else differs.
        // (not with the code touch: that policy reads up to 32 KB ahead of a wave's PC as data and relies on the padding build.py verifies behind a prebuilt unit's kernels)
        if (!nt && !conserving && cp.code_touch == 0 && !env_int("BEPUHIP_FORCE_WIDE_FAMILY", 0)) {
            const UnitKey want = special_key(c, threads);
            if (c->specialise_auto && (!c->special_unit || c->special_mask != want.mask || c->special_budget != want.budget || c->special_shared != want.shared)) special_request(c, want);
            if (c->special_unit && c->special_mask == want.mask && c->special_budget == want.budget && c->special_shared == want.shared && c->special_unit->state.load(std::memory_order_acquire) == kUnitLoaded) {
                fn = tr ? c->special_unit->traced : c->special_unit->kernel;
                c->last_kernel_family = kFamilySpecial;
            }
        }
        if (sample >= 0) hipEventRecord(c->policy_events[sample][0], c->stream);
        const int tail_blocks = tp.body_blocks + (c->kinlist_count > 0 ? 1 : 0);
        bool launched = false;
        if (cooperative) {
            // The clusters of a split plan wait for each other: they must all be resident at once. A cooperative launch of exactly the clusters makes the runtime
            // guarantee that (or refuse), whatever else runs on the device; the per-body tail follows as an ordinary launch of the same kernel.
            std::lock_guard<std::mutex> one_at_a_time(g_cooperative_launch);
            if (hipLaunchCooperativeKernel(fn, dim3(local_clusters), dim3(threads), args, (unsigned)launch_lds, c->stream) == hipSuccess) {
                launched = true;
                if (tail_blocks > 0) {
                    tp.block_offset = local_clusters;
                    hipLaunchKernel(fn, dim3(tail_blocks), dim3(threads), args, 0, c->stream);
                }
            } else {
                (void)hipGetLastError();  // e.g. hipErrorCooperativeLaunchTooLarge: fall back to the plain launch below
            }
        }
        if (!launched) {
            if (t.attached) hipExtLaunchKernel(fn, dim3(local_clusters + tail_blocks), dim3(threads), args, launch_lds, c->stream, t.a, t.b, 0);
            else hipLaunchKernel(fn, dim3(local_clusters + tail_blocks), dim3(threads), args, launch_lds, c->stream);
        }
        if (sample >= 0) hipEventRecord(c->policy_events[sample][1], c->stream);
    }
}

// Enqueue every kernel of one Simulation.Solve on the context's stream (Solver_Solve.cs:1415-1479 + PoseIntegrator.cs:707-726).
static void enqueue_solve(bepuhip_ctx* c, float dt, int substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    const float substep_dt = dt / substeps;          // Solver_Solve.cs:1417
    const float inv_dt = 1.0f / substep_dt;          // :1421
    const StepParams sp = make_params(c, in, substep_dt, substep_dt, inv_dt);
    const int body_blocks = (c->body_count + 255) / 256;
    const bool use_clusters = island_schedule_applies(c, substeps, in);
    const int skip_clustered = use_clusters ? 1 : 0;
    if (!use_clusters) c->last_kernel_family = -1;  // (the launch-per-batch kernels carry every type: bepuhip_get_kernel_family)
    if (use_clusters) {
        // Every constraint belongs to an island a workgroup holds (or to a cluster of a cut island): the whole substep loop runs in ONE launch — or, past
        // kMaxClusterSubsteps substeps (SolveDescription.SubstepCount is unbounded in the reference, SolveDescription.cs:16-136), in a chain of them.
        for (int base = 0; base < substeps; base += kMaxClusterSubsteps)
            enqueue_island_launch(c, dt, substeps, base, std::min(kMaxClusterSubsteps, substeps - base), iterations, in, sp);
    }
    for (int s = 0; s < substeps && !use_clusters; ++s) {
        if (s > 0 && c->inc_blocks > 0) {             // :1427-1439 (all batches in one grid: it reads velocities and writes only prestep depths)
            Timed t(c, 0);
            hipLaunchKernelGGL(batch_kernel<kStageIncremental>, dim3(c->inc_blocks), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_inc_tbs, 0, c->inc_tb_count, c->d_bodies, substep_dt, inv_dt);
        }
        if (body_blocks > 0) {                        // :1440-1445 + the integration half of GatherAndIntegrate
            Timed t(c, 1);
            hipLaunchKernelGGL(substep_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, s > 0 ? 1 : 0,
                               in->integrate_velocity_for_kinematics, skip_clustered, sp);
        }
        for (int b = 0; b < c->launch_count; ++b) {   // :1447-1463 (+ the fallback batch's levels, :546-563)
            if (c->batch_blocks[b] == 0) continue;
            enqueue_requirk(c, s, b, sp);
            Timed t(c, 2);
            hipLaunchKernelGGL(batch_kernel<kStageWarmStart>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                               c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
        }
        for (int it = 0; it < iterations[s]; ++it) {  // :1464-1476
            for (int b = 0; b < c->launch_count; ++b) {  // (+ the fallback batch's levels, :574-583)
                if (c->batch_blocks[b] == 0) continue;
                Timed t(c, 3);
                hipLaunchKernelGGL(batch_kernel<kStageSolve>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                                   c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
            }
        }
    }
    if (body_blocks > 0 && !use_clusters) {           // PoseIntegrator.cs:707-726 (the island schedule's launch includes it)
        const float vdt = in->allow_substeps_for_unconstrained ? substep_dt : dt;
        const StepParams fsp = make_params(c, in, vdt, vdt, 1.0f / vdt);
        Timed t(c, 4);
        hipLaunchKernelGGL(final_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, dt, substep_dt, substeps,
                           in->allow_substeps_for_unconstrained, in->integrate_velocity_for_kinematics, skip_clustered, fsp);
    }
}

// `referenced_bodies` only ever grows while references are patched (bepuhip_update_body_reference lowers them: the last body moved into a freed slot and the body
// array shrank): counted again from the rows, with everything the caller has been told applied, before a solve is refused for it.
static int32_t recount_referenced_bodies(bepuhip_ctx* c) {
    HIP_TRY(hipSetDevice(c->device));
    int32_t st = flush_structural(c);
    if (st != BEPUHIP_OK) return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    int highest = -1;
    std::vector<int32_t> row;
    for (auto& tb : c->tbs) {
        const int extent = tb.device_extent();
        if (extent == 0) continue;
        row.resize(extent);
        for (int k = 0; k < tb.info.bodies; ++k) {
            HIP_TRY(copy_sync(c, row.data(), c->d_slab + tb.refs_off + (size_t)k * tb.stride, (size_t)extent * 4, hipMemcpyDeviceToHost));
            for (int32_t r : row) if (r >= 0) highest = std::max(highest, r & kRefMask);
        }
    }
    c->referenced_bodies = highest + 1;
    return rebuild_flags(c);  // (they were skipped while the references seemed to point past the body array)
}

static int32_t validate_solve(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    if (!c || !in || !iterations) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (!(dt > 0)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Timestep duration must be positive.");                  // Simulation.cs:318-319
    if (substeps < 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Substep count must be positive.");                    // SolveDescription.cs:42-47
    for (int s = 0; s < substeps; ++s)
        if (iterations[s] < 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Velocity iteration count must be positive.");
    if (in->angular_integration_mode < 0 || in->angular_integration_mode > 2) return fail(BEPUHIP_E_INVALID_ARGUMENT, "unknown AngularIntegrationMode");
    if (c->building) return fail(BEPUHIP_E_STATE, "solve between begin_constraints and end_constraints");
    if (c->velocity_model.model == BEPUHIP_VELOCITY_PER_BODY_GRAVITY && c->body_gravity_count < c->body_count)
        return fail(BEPUHIP_E_STATE, "the per-body gravity table holds " + std::to_string(c->body_gravity_count) + " values for " + std::to_string(c->body_count) + " bodies (set_velocity_model)");
    if (c->built && c->referenced_bodies > c->body_count) {
        const int32_t st = recount_referenced_bodies(c);
        if (st != BEPUHIP_OK) return st;
    }
    if (c->built && c->referenced_bodies > c->body_count)
        return fail(BEPUHIP_E_STATE, "a constraint references body " + std::to_string(c->referenced_bodies - 1) + " but only " + std::to_string(c->body_count) +
                                         " bodies are uploaded (set_bodies)");
    return BEPUHIP_OK;
}

// BEPUHIP_SOLVE_STATS=1 (developer switch): a solve call that takes more than 2 ms on the HOST says where (a launch is ~10 us; round 6's hunt for the 37 ms call)
struct SolveLaps {
    const bool on = env_int("BEPUHIP_SOLVE_STATS", 0) != 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    std::string text;
    void lap(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof(buf), " %s %.3f", what, std::chrono::duration<double, std::milli>(now - last).count());
        text += buf; last = now;
    }
    ~SolveLaps() { if (on && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0) fprintf(stderr, "bepuhip solve call, ms:%s\n", text.c_str()); }
};
int32_t bepuhip_solve_async(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    SolveLaps laps;
    int32_t st = validate_solve(c, dt, substeps, iterations, in);
    if (st != BEPUHIP_OK) return st;
    laps.lap("validate");
    HIP_TRY(hipSetDevice(c->device));
    if ((st = group_queue_check(c)) != BEPUHIP_OK) return st;
    if ((st = flush_structural(c)) != BEPUHIP_OK) return st;
    laps.lap("flush");
    if (c->clusters_enabled && !island_schedule_applies(c, substeps, in)) {
        // the launch-per-batch kernels address rows [0, count): an island layout with free slots between the live ones has to be brought back into the caller's order first
        bool gaps = false;
        for (auto& tb : c->tbs) gaps |= tb.slots > 0 && tb.slots != tb.count;
        for (auto& tb : c->tbs) if (tb.slots > 0 && !gaps) for (int d = 0; d < tb.count && !gaps; ++d) gaps = tb.perm[d] < 0;
        if (gaps) {
            if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;
            if ((st = flush_structural(c)) != BEPUHIP_OK) return st;
        }
    }
    if (in->angular_integration_mode != 0 && c->requirk_stale && c->built && (st = build_requirk_lists(c)) != BEPUHIP_OK) return st;  // first conserving solve of this topology
    int64_t iters = 0;
    for (int s = 0; s < substeps; ++s) iters += c->total_constraints * (int64_t)(1 + iterations[s]);
    c->last_constraint_iterations = iters;
    if (c->profiling) { for (int i = 0; i < 6; ++i) { c->prof_ms[i] = 0; c->prof_launches[i] = 0; } }
    laps.lap("checks");
    HIP_TRY(solve_event(c, true));
    laps.lap("event record");
    // A graph pays for the launch-per-batch schedule's 100+ launches; the island schedule is ONE kernel, which a plain launch starts sooner (6-7 us per step).
    const bool use_graph = !(c->flags & BEPUHIP_FLAG_NO_GRAPH) && !c->profiling && !island_schedule_applies(c, substeps, in);
    if (!use_graph) c->graphs_cleared_by_structure = false;
    if (use_graph) {
        GraphKey key;
        key.iterations.assign(iterations, iterations + substeps);
        key.dt = dt; key.integ = *in;
        auto it = c->graphs.find(key);
        if (it == c->graphs.end() && c->graphs_cleared_by_structure) {  // topology changed since the last solve: capturing + instantiating (~1.5 ms) would not pay for itself
            c->graphs_cleared_by_structure = false;
            enqueue_solve(c, dt, substeps, iterations, in);
            HIP_TRY(hipGetLastError());
            HIP_TRY(solve_event(c, false));
            return BEPUHIP_OK;
        }
        if (it == c->graphs.end()) {
            if (c->graphs.size() >= kMaxCachedGraphs) {  // a caller with a variable time step produces a new key every frame: keep the cache bounded
                HIP_TRY(hipStreamSynchronize(c->stream));
                for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
                c->graphs.clear();
                HIP_TRY(solve_event(c, true));
            }
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            enqueue_solve(c, dt, substeps, iterations, in);
            // Whatever happened while capturing, the stream must leave capture mode and the graph must not leak: a context that failed once stays usable.
            const hipError_t launch_err = hipGetLastError();
            const hipError_t end_err = hipStreamEndCapture(c->stream, &graph);
            hipGraphExec_t exec = nullptr;
            hipError_t inst_err = hipSuccess;
            if (launch_err == hipSuccess && end_err == hipSuccess && graph != nullptr) inst_err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (graph != nullptr) hipGraphDestroy(graph);
            if (launch_err != hipSuccess || end_err != hipSuccess || inst_err != hipSuccess || exec == nullptr) {
                if (exec != nullptr) hipGraphExecDestroy(exec);
                // A capture can be invalidated from outside the library (another thread's legacy-stream call, ROCm 7.2: bepu_host_state.h, copy_sync): make sure the
                // stream has left capture mode and no error of the dead capture is left behind for the eager launches to trip over.
                hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(c->stream, &status) == hipSuccess && status != hipStreamCaptureStatusNone) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(c->stream, &dead); if (dead) hipGraphDestroy(dead); }
                for (int k = 0; k < 4 && hipGetLastError() != hipSuccess; ++k) {}
                HIP_TRY(solve_event(c, true));
                enqueue_solve(c, dt, substeps, iterations, in);  // eager fallback for this call; the next call tries to capture again
                HIP_TRY(hipGetLastError());
                HIP_TRY(solve_event(c, false));
                return BEPUHIP_OK;
            }
            it = c->graphs.emplace(key, exec).first;
            // the event recorded before capture began is still first in stream order
        }
        HIP_TRY(hipGraphLaunch(it->second, c->stream));
    } else {
        enqueue_solve(c, dt, substeps, iterations, in);
    }
    laps.lap("enqueue");
    HIP_TRY(hipGetLastError());
    HIP_TRY(solve_event(c, false));
    laps.lap("stop event");
    return BEPUHIP_OK;
}

// Simulation.Solve with Solver.SubstepStarted / SubstepEnded raised (include/bepuhip.h): the launch-per-batch kernels of one substep at a time, the host's handlers
// between them with the stream drained.
int32_t bepuhip_solve_with_substep_events(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in, bepuhip_substep_fn started,
                                          bepuhip_substep_fn ended, void* user) {
    int32_t st = validate_solve(c, dt, substeps, iterations, in);
    if (st != BEPUHIP_OK) return st;
    HIP_TRY(hipSetDevice(c->device));
    if ((st = flush_structural(c)) != BEPUHIP_OK) return st;
    // Round 5: a context on an island plan raises the events BETWEEN launches of the island kernel — one substep per launch (enqueue_island_launch), the bodies in HBM
    // with the substep's pose and velocities while the handler runs, so that what it rewrites through the update_* entry points is what the next launch stages.
    const bool islands = island_schedule_applies(c, substeps, in) && env_int("BEPUHIP_EVENT_CLUSTERS", 1) != 0 && !group_chain_hazard(c, substeps);  // (one launch per substep: a chain)
    if (c->clusters_enabled && !islands) {  // the launch-per-batch kernels address rows [0, count): an island layout with free slots goes back into the caller's order first
        bool gaps = false;
        for (auto& tb : c->tbs) gaps |= tb.slots > 0 && tb.slots != tb.count;
        for (auto& tb : c->tbs) if (tb.slots > 0 && !gaps) for (int d = 0; d < tb.count && !gaps; ++d) gaps = tb.perm[d] < 0;
        if (gaps) {
            if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;
            if ((st = flush_structural(c)) != BEPUHIP_OK) return st;
        }
    }
    if (in->angular_integration_mode != 0 && c->requirk_stale && c->built && (st = build_requirk_lists(c)) != BEPUHIP_OK) return st;
    int64_t iters = 0;
    for (int s = 0; s < substeps; ++s) iters += c->total_constraints * (int64_t)(1 + iterations[s]);
    c->last_constraint_iterations = iters;
    HIP_TRY(solve_event(c, true));
    const float substep_dt = dt / substeps, inv_dt = 1.0f / substep_dt;  // Solver_Solve.cs:1417, :1421
    const StepParams sp = make_params(c, in, substep_dt, substep_dt, inv_dt);
    const int body_blocks = (c->body_count + 255) / 256;
    auto raise = [&](bepuhip_substep_fn fn, int s) -> int32_t {
        if (!fn) return BEPUHIP_OK;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));  // the handler sees (and may rewrite, through the update_* entry points) the state the substep starts from / ended with
        c->in_substep_event = true;
        fn(user, s);
        c->in_substep_event = false;
        return BEPUHIP_OK;
    };
    for (int s = 0; s < substeps && islands; ++s) {
        if ((st = raise(started, s)) != BEPUHIP_OK) return st;  // OnSubstepStarted(substepIndex), Solver_Solve.cs:1425
        if (!island_schedule_applies(c, substeps, in)) return fail(BEPUHIP_E_STATE, "a substep event handler made the context leave its island plan");
        enqueue_island_launch(c, dt, substeps, s, 1, iterations, in, sp);
        if ((st = raise(ended, s)) != BEPUHIP_OK) return st;    // OnSubstepEnded(substepIndex), :1478
    }
    if (islands) {
        HIP_TRY(hipGetLastError());
        HIP_TRY(solve_event(c, false));
        return bepuhip_sync(c);
    }
    for (int s = 0; s < substeps; ++s) {
        if ((st = raise(started, s)) != BEPUHIP_OK) return st;  // OnSubstepStarted(substepIndex), Solver_Solve.cs:1425
        if (s > 0 && c->inc_blocks > 0)
            hipLaunchKernelGGL(batch_kernel<kStageIncremental>, dim3(c->inc_blocks), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_inc_tbs, 0, c->inc_tb_count, c->d_bodies, substep_dt, inv_dt);
        if (body_blocks > 0)
            hipLaunchKernelGGL(substep_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, s > 0 ? 1 : 0,
                               in->integrate_velocity_for_kinematics, 0, sp);
        for (int b = 0; b < c->launch_count; ++b) {
            if (c->batch_blocks[b] == 0) continue;
            enqueue_requirk(c, s, b, sp);
            hipLaunchKernelGGL(batch_kernel<kStageWarmStart>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                               c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
        }
        for (int it = 0; it < iterations[s]; ++it)
            for (int b = 0; b < c->launch_count; ++b) {
                if (c->batch_blocks[b] == 0) continue;
                hipLaunchKernelGGL(batch_kernel<kStageSolve>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                                   c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
            }
        if ((st = raise(ended, s)) != BEPUHIP_OK) return st;    // OnSubstepEnded(substepIndex), :1478
    }
    if (body_blocks > 0) {  // PoseIntegrator.IntegrateAfterSubstepping (PoseIntegrator.cs:707-726)
        const float vdt = in->allow_substeps_for_unconstrained ? substep_dt : dt;
        const StepParams fsp = make_params(c, in, vdt, vdt, 1.0f / vdt);
        hipLaunchKernelGGL(final_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, dt, substep_dt, substeps,
                           in->allow_substeps_for_unconstrained, in->integrate_velocity_for_kinematics, 0, fsp);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(solve_event(c, false));
    return bepuhip_sync(c);
}

int32_t bepuhip_set_boundary_bodies(bepuhip_ctx* c, const int32_t* indices, int32_t count) {
    if (!c || count < 0 || (count > 0 && !indices)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad boundary list");
    HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < count; ++i)
        if (indices[i] < 0 || indices[i] >= c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "boundary body index out of range (call set_bodies first)");
    if (c->d_boundary) { hipFree(c->d_boundary); hipFree(c->d_boundary_snapshot); hipFree(c->d_boundary_buf); c->d_boundary = nullptr; c->d_boundary_snapshot = nullptr; c->d_boundary_buf = nullptr; }
    free_boundary_layout(c);  // rows belong to a boundary list
    c->boundary_count = count;
    if (count > 0) {
        HIP_TRY(hipMalloc((void**)&c->d_boundary, (size_t)count * 4));
        HIP_TRY(hipMalloc((void**)&c->d_boundary_snapshot, (size_t)count * 32));
        HIP_TRY(hipMalloc((void**)&c->d_boundary_buf, (size_t)count * 24));
        HIP_TRY(copy_sync(c, c->d_boundary, indices, (size_t)count * 4, hipMemcpyHostToDevice));
    }
    return rebuild_flags(c);
}

int32_t bepuhip_boundary_deltas(bepuhip_ctx* c, float* out, int32_t out_is_device) {
    if (!c || (c->boundary_count > 0 && !out)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (c->boundary_count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    float* dst = out_is_device ? out : c->d_boundary_buf;
    hipLaunchKernelGGL(boundary_deltas_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, (const float4*)c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                       (const float4*)c->d_boundary_snapshot, dst, (const int*)nullptr, c->exchange_mode == BEPUHIP_EXCHANGE_PER_BATCH_EXACT ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if (!out_is_device) HIP_TRY(hipMemcpyAsync(out, dst, (size_t)c->boundary_count * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the caller hands the buffer to a collective on another stream / the host next
    return BEPUHIP_OK;
}

int32_t bepuhip_boundary_apply(bepuhip_ctx* c, const float* sums, int32_t in_is_device) {
    if (!c || (c->boundary_count > 0 && !sums)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (c->boundary_count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const float* src = sums;
    if (!in_is_device) {
        HIP_TRY(hipMemcpyAsync(c->d_boundary_buf, sums, (size_t)c->boundary_count * 24, hipMemcpyHostToDevice, c->stream));
        src = c->d_boundary_buf;
    }
    hipLaunchKernelGGL(boundary_apply_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                       c->d_boundary_snapshot, src, (const int*)nullptr, (const float*)nullptr, c->exchange_mode == BEPUHIP_EXCHANGE_PER_BATCH_EXACT ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}

int32_t bepuhip_colour_constraints(int32_t device, const int32_t* refs, int32_t count, int32_t body_count, int32_t order, int32_t fallback_batch_threshold, int32_t* colours_out,
                                   int32_t* batch_count_out, int32_t* rounds_out) {
    if (count < 0 || body_count < 0 || (count > 0 && (!refs || !colours_out)) || order < 0 || order > 1 || fallback_batch_threshold < 1 || fallback_batch_threshold > 64)
        return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad colour_constraints argument");
    for (int64_t i = 0; i < (int64_t)count * kColourBodies; ++i)
        if (refs[i] != -1 && (refs[i] < 0 || (refs[i] & kRefMask) >= body_count)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "body reference out of range");
    if (batch_count_out) *batch_count_out = 0;
    if (rounds_out) *rounds_out = 0;
    if (count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(device));
    int* d_refs = nullptr; int* d_colour = nullptr; unsigned* d_degree = nullptr; unsigned long long* d_words = nullptr; unsigned* d_remaining = nullptr;
    hipStream_t stream = nullptr;  // a stream of its own, non-blocking: nothing of the library runs on the legacy stream (bepu_host_state.h, copy_sync)
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    auto release = [&]() { hipStreamSynchronize(stream); for (void* p : {(void*)d_refs, (void*)d_colour, (void*)d_degree, (void*)d_words, (void*)d_remaining}) if (p) hipFree(p); hipStreamDestroy(stream); };
    const size_t nb = (size_t)std::max(body_count, 1);
    hipError_t e = hipMalloc((void**)&d_refs, (size_t)count * kColourBodies * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_colour, (size_t)count * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_degree, nb * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_words, ((size_t)count + 2 * nb) * 8);  // priority per constraint, best bid and used-batch mask per body
    if (e == hipSuccess) e = hipMalloc((void**)&d_remaining, 4);
    if (e != hipSuccess) { release(); return fail(BEPUHIP_E_DEVICE, std::string("colour_constraints: ") + hipGetErrorString(e)); }
    unsigned long long* d_priority = d_words; unsigned long long* d_best = d_words + count; unsigned long long* d_used = d_best + nb;
    hipMemcpyAsync(d_refs, refs, (size_t)count * kColourBodies * 4, hipMemcpyHostToDevice, stream);
    hipMemsetAsync(d_colour, 0xFF, (size_t)count * 4, stream);
    hipMemsetAsync(d_degree, 0, nb * 4, stream);
    hipMemsetAsync(d_used, 0, nb * 8, stream);
    const dim3 grid((count + 255) / 256), block(256);
    hipLaunchKernelGGL(colour_degree_kernel, grid, block, 0, stream, (const int*)d_refs, count, d_degree);
    hipLaunchKernelGGL(colour_priority_kernel, grid, block, 0, stream, (const int*)d_refs, count, (const unsigned*)d_degree, order, d_priority);
    // Rounds are enqueued eight at a time; the host only looks at the count of uncoloured constraints behind each group (a round after the last useful one finds
    // nothing to do and costs two near-empty launches), so the loop runs on the device and the host reads four bytes every eight rounds.
    constexpr int kRoundsPerCheck = 8;
    unsigned* d_remaining_slots = nullptr;  // one counter per round of a group: [k] = constraints still uncoloured after round k
    e = hipMalloc((void**)&d_remaining_slots, kRoundsPerCheck * 4);
    if (e != hipSuccess) { release(); return fail(BEPUHIP_E_DEVICE, std::string("colour_constraints: ") + hipGetErrorString(e)); }
    int rounds = 0;
    for (bool done = false; !done;) {
        if (rounds > count + kRoundsPerCheck) { hipFree(d_remaining_slots); release(); return fail(BEPUHIP_E_DEVICE, "colour_constraints made no progress"); }  // every round colours at least the highest bid
        hipMemsetAsync(d_remaining_slots, 0, kRoundsPerCheck * 4, stream);
        for (int k = 0; k < kRoundsPerCheck; ++k) {
            hipMemsetAsync(d_best, 0, nb * 8, stream);
            hipLaunchKernelGGL(colour_bid_kernel, grid, block, 0, stream, (const int*)d_refs, count, (const int*)d_colour, (const unsigned long long*)d_priority, d_best);
            hipLaunchKernelGGL(colour_pick_kernel, grid, block, 0, stream, (const int*)d_refs, count, d_colour, (const unsigned long long*)d_priority, (const unsigned long long*)d_best, d_used,
                               fallback_batch_threshold, d_remaining_slots + k);
        }
        unsigned remaining[kRoundsPerCheck];
        e = hipMemcpyAsync(remaining, d_remaining_slots, sizeof(remaining), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { hipFree(d_remaining_slots); release(); return fail(BEPUHIP_E_DEVICE, std::string("colour_constraints: ") + hipGetErrorString(e)); }
        for (int k = 0; k < kRoundsPerCheck && !done; ++k) { ++rounds; done = remaining[k] == 0; }  // rounds = the rounds that were needed
    }
    hipFree(d_remaining_slots);
    e = hipMemcpyAsync(colours_out, d_colour, (size_t)count * 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    release();
    if (e != hipSuccess) return fail(BEPUHIP_E_DEVICE, std::string("colour_constraints: ") + hipGetErrorString(e));
    int highest = -1;
    for (int i = 0; i < count; ++i) highest = std::max(highest, colours_out[i]);
    if (batch_count_out) *batch_count_out = highest + 1;
    if (rounds_out) *rounds_out = rounds;
    return BEPUHIP_OK;
}

int32_t bepuhip_set_exchange_mode(bepuhip_ctx* c, int32_t mode) {
    if (!c || (mode != BEPUHIP_EXCHANGE_PER_PASS_AVERAGE && mode != BEPUHIP_EXCHANGE_PER_BATCH_EXACT)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "unknown exchange mode");
    c->exchange_mode = mode;
    return BEPUHIP_OK;
}

int32_t bepuhip_set_boundary_layout(bepuhip_ctx* c, const int32_t* dense_rows, int32_t dense_row_count, const float* holders) {
    if (!c || dense_row_count < 0 || (c->boundary_count > 0 && !dense_rows)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad boundary layout");
    for (int i = 0; i < c->boundary_count; ++i)
        if (dense_rows[i] < 0 || dense_rows[i] >= dense_row_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "dense row out of range");
    if (holders)
        for (int i = 0; i < dense_row_count; ++i) if (!(holders[i] >= 1.0f)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "holder count below one");
    // everything is validated before the context is touched; from here on a failure leaves NO layout (never a partial one: rows without holders would let the
    // averaged mode apply un-averaged sums)
    HIP_TRY(hipSetDevice(c->device));
    free_boundary_layout(c);
    if (dense_row_count == 0) return BEPUHIP_OK;
    hipError_t e = hipMalloc((void**)&c->d_boundary_dense, (size_t)dense_row_count * 24);
    if (e == hipSuccess && c->boundary_count > 0) {
        e = hipMalloc((void**)&c->d_boundary_rows, (size_t)c->boundary_count * 4);
        if (e == hipSuccess) e = copy_sync(c, c->d_boundary_rows, dense_rows, (size_t)c->boundary_count * 4, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && holders) {
        e = hipMalloc((void**)&c->d_boundary_holders, (size_t)dense_row_count * 4);
        if (e == hipSuccess) e = copy_sync(c, c->d_boundary_holders, holders, (size_t)dense_row_count * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        free_boundary_layout(c);
        return fail(BEPUHIP_E_DEVICE, std::string("set_boundary_layout: ") + hipGetErrorString(e));
    }
    c->dense_rows = dense_row_count;
    return BEPUHIP_OK;
}

// ---- RCCL, opened at run time ----
// torch ships its own librccl; a C# host gets /opt/rocm/lib's. Opening by soname returns whichever the process already has, so there is never a second copy.
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl* rccl() {
    static Rccl lib;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib.handle) break;
        }
        if (lib.handle) {
            lib.GetUniqueId = (decltype(lib.GetUniqueId))dlsym(lib.handle, "ncclGetUniqueId");
            lib.CommInitRank = (decltype(lib.CommInitRank))dlsym(lib.handle, "ncclCommInitRank");
            lib.CommDestroy = (decltype(lib.CommDestroy))dlsym(lib.handle, "ncclCommDestroy");
            lib.AllReduce = (decltype(lib.AllReduce))dlsym(lib.handle, "ncclAllReduce");
            lib.GetErrorString = (decltype(lib.GetErrorString))dlsym(lib.handle, "ncclGetErrorString");
            if (!lib.GetUniqueId || !lib.CommInitRank || !lib.CommDestroy || !lib.AllReduce) { dlclose(lib.handle); lib.handle = nullptr; }
        }
    }
    return lib.handle ? &lib : nullptr;
}
static int32_t rccl_fail(const char* what, ncclResult_t r) {
    Rccl* lib = rccl();
    return fail(BEPUHIP_E_DEVICE, std::string(what) + ": " + (lib && lib->GetErrorString ? lib->GetErrorString(r) : "RCCL error") + " (" + std::to_string((int)r) + ")");
}
static void release_comm(bepuhip_ctx* c) {
    if (c->comm && c->comm_owned) { Rccl* lib = rccl(); if (lib) lib->CommDestroy((ncclComm_t)c->comm); }
    c->comm = nullptr; c->comm_owned = false; c->comm_world = 1;
}

int32_t bepuhip_comm_unique_id(void* id_out) {
    if (!id_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    Rccl* lib = rccl();
    if (!lib) return fail(BEPUHIP_E_UNSUPPORTED, "librccl.so could not be opened (needed only for the on-stream exchange of a split scene)");
    static_assert(sizeof(ncclUniqueId) == BEPUHIP_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    const ncclResult_t r = lib->GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id_out, &id, sizeof(id));
    return BEPUHIP_OK;
}

int32_t bepuhip_comm_init(bepuhip_ctx* c, const void* id, int32_t rank, int32_t world) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad communicator arguments");
    Rccl* lib = rccl();
    if (!lib) return fail(BEPUHIP_E_UNSUPPORTED, "librccl.so could not be opened (needed only for the on-stream exchange of a split scene)");
    HIP_TRY(hipSetDevice(c->device));
    release_comm(c);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = lib->CommInitRank(&comm, world, uid, rank);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    c->comm = comm; c->comm_owned = true; c->comm_world = world;
    return BEPUHIP_OK;
}

int32_t bepuhip_comm_adopt(bepuhip_ctx* c, void* nccl_comm, int32_t world) {
    if (!c || world < 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad communicator arguments");
    if (nccl_comm && !rccl()) return fail(BEPUHIP_E_UNSUPPORTED, "librccl.so could not be opened");
    release_comm(c);
    c->comm = nccl_comm; c->comm_owned = false; c->comm_world = nccl_comm ? world : 1;
    return BEPUHIP_OK;
}

// One exchange on the solver's stream, no host involvement: deltas into the dense buffer (rows this rank does not hold stay zero), all-reduce in place, apply.
static int32_t enqueue_exchange(bepuhip_ctx* c) {
    if (c->dense_rows == 0) return BEPUHIP_OK;
    const int exact = c->exchange_mode == BEPUHIP_EXCHANGE_PER_BATCH_EXACT ? 1 : 0;
    HIP_TRY(hipMemsetAsync(c->d_boundary_dense, 0, (size_t)c->dense_rows * 24, c->stream));
    if (c->boundary_count > 0)
        hipLaunchKernelGGL(boundary_deltas_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, (const float4*)c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                           (const float4*)c->d_boundary_snapshot, c->d_boundary_dense, (const int*)c->d_boundary_rows, exact);
    if (c->comm) {
        const ncclResult_t r = rccl()->AllReduce(c->d_boundary_dense, c->d_boundary_dense, (size_t)c->dense_rows * 6, exact ? ncclUint32 : ncclFloat32, ncclSum, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    }
    if (c->boundary_count > 0)
        hipLaunchKernelGGL(boundary_apply_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                           c->d_boundary_snapshot, (const float*)c->d_boundary_dense, (const int*)c->d_boundary_rows, exact ? (const float*)nullptr : (const float*)c->d_boundary_holders, exact);
    return BEPUHIP_OK;
}

// Simulation.Solve with exchange points: the launch-per-batch schedule, eager. `fn` != null: host-synchronised call-backs (any transport); null: the exchange
// is enqueued on the stream (enqueue_exchange) and the host does not wait for anything before the end of the frame. BEPUHIP_EXCHANGE_PER_PASS_AVERAGE exchanges
// after every pass, BEPUHIP_EXCHANGE_PER_BATCH_EXACT after every batch of every pass.
// One sweep of the island schedule as a launch of its own (the kPass units of bepu_cluster_kernel.h): all batches of one warm start or one velocity iteration.
static void enqueue_cluster_pass(bepuhip_ctx* c, int stage, int substep, const StepParams& sp) {
    ClusterParams cp;
    memset(&cp, 0, sizeof(cp));
    cp.substeps = 1; cp.batch_count = c->batch_count; cp.planes = c->cluster_planes;
    cp.pass_stage = stage; cp.pass_substep = substep;
    cp.fallback_batch = -1;  // (exchanged solves refuse a fallback batch)
    cp.sp = sp;
    cp.code_touch = c->row_policy == 2 ? (c->clusters_shared ? 2 : 1) : 0;
    cp.jitter = debug_jitter_seed();
    const int threads = cluster_threads(c);
    TailParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.flags = c->d_flags; tp.staged = c->d_staged; tp.cluster_count = c->cluster_count; tp.body_count = c->body_count;
    SharedTables st = {c->d_shared_vel, c->d_shared_info, env_int("BEPUHIP_SHARED_POLL", 1), 0u, 0, nullptr};
    if (c->clusters_shared) {  // a pass is a one-substep, one-pass step of its own as far as the event numbers go
        const unsigned span = 260u;
        if ((unsigned long long)c->shared_epoch + 2ull * span > 0xFFFFFFFFull) { hipMemsetAsync(c->d_shared_vel, 0, c->shared_bodies * 4 * sizeof(float4), c->stream); c->shared_epoch = 0; }
        st.base = c->shared_epoch;
        c->shared_epoch += span;
    }
    const size_t lds_bytes = cluster_lds_bytes(c->cluster_planes, c->cluster_max_slots, c->cluster_max_items, c->clusters_shared);
    void* args[] = {(void*)&c->d_clusters, (void*)&c->d_items, (void*)&c->d_batch_item_begin, (void*)&c->d_cluster_bodies, (void*)&c->d_bodies, (void*)&c->d_slab,
                    (void*)&cp, (void*)&c->cluster_max_slots, (void*)&c->cluster_max_items, (void*)&c->d_trace, (void*)&c->d_status, (void*)&c->d_cycles, (void*)&tp, (void*)&st};
    const void* fn = cluster_pass_kernel(c->has_widened_types, c->clusters_shared);
    bool launched = false;
    if (c->clusters_shared && env_int("BEPUHIP_COOPERATIVE", (c->flags & BEPUHIP_FLAG_EXCLUSIVE_DEVICE) ? 0 : 1) != 0) {
        std::lock_guard<std::mutex> one_at_a_time(g_cooperative_launch);
        launched = hipLaunchCooperativeKernel(fn, dim3(c->cluster_count), dim3(threads), args, (unsigned)lds_bytes, c->stream) == hipSuccess;
        if (!launched) (void)hipGetLastError();
    }
    if (!launched) hipLaunchKernel(fn, dim3(c->cluster_count), dim3(threads), args, lds_bytes, c->stream);
}

static int32_t run_exchanged(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in, bepuhip_exchange_fn fn, void* user) {
    int32_t st = validate_solve(c, dt, substeps, iterations, in);
    if (st != BEPUHIP_OK) return st;
    // A context on an island plan runs the sweeps between two exchanges as ONE launch each (round 3): possible where the exchange comes per pass, not per batch, the
    // angular mode is the nonconserving one and the rows hold no free slots (the integration kernels around the sweeps address rows [0, count)).
    bool island_passes = false;
    if (c->clusters_enabled) {
        bool gaps = false;
        for (auto& tb : c->tbs) gaps |= tb.slots > 0 && tb.slots != tb.count;
        island_passes = c->exchange_mode != BEPUHIP_EXCHANGE_PER_BATCH_EXACT && in->angular_integration_mode == 0 && !gaps && env_int("BEPUHIP_EXCHANGED_CLUSTERS", 1) != 0 &&
                        conserving_variant_exists(cluster_threads(c), c->clusters_shared) &&
                        cluster_lds_bytes(c->cluster_planes, c->cluster_max_slots, c->cluster_max_items, c->clusters_shared) <= kLdsBudgetBytes;
        if (!island_passes)
            return fail(BEPUHIP_E_STATE, "this exchanged solve needs a context created with BEPUHIP_FLAG_NO_CLUSTERS: an island plan runs exchanged solves only with one exchange per pass, "
                                         "the nonconserving angular mode and no reserved update slots");
    }
    // The fallback batch runs as one launch per dependency level of THIS share: the ranks would disagree on the number of exchange points (a hang in RCCL or in the
    // host barrier) and, in the exact mode, on who touched a body since the last exchange (its level numbering is rank-local). Shares keep the global batch indices,
    // so a scene whose colouring reaches the fallback threshold has to be recoloured (bepuhip_colour_constraints) or solved unsplit.
    if (c->has_fallback) return fail(BEPUHIP_E_UNSUPPORTED, "an exchanged solve of a share that contains a sequential fallback batch: the exchange points of the ranks would not pair up");
    if (!fn && c->boundary_count > 0 && !c->d_boundary_rows) return fail(BEPUHIP_E_STATE, "bepuhip_solve_lattice needs bepuhip_set_boundary_layout after bepuhip_set_boundary_bodies");
    HIP_TRY(hipSetDevice(c->device));
    if ((st = flush_structural(c)) != BEPUHIP_OK) return st;
    if (in->angular_integration_mode != 0 && c->requirk_stale && c->built && (st = build_requirk_lists(c)) != BEPUHIP_OK) return st;
    int64_t iters = 0;
    for (int s = 0; s < substeps; ++s) iters += c->total_constraints * (int64_t)(1 + iterations[s]);
    c->last_constraint_iterations = iters;
    HIP_TRY(solve_event(c, true));
    const float substep_dt = dt / substeps, inv_dt = 1.0f / substep_dt;
    const StepParams sp = make_params(c, in, substep_dt, substep_dt, inv_dt);
    const int body_blocks = (c->body_count + 255) / 256;
    const bool per_batch = c->exchange_mode == BEPUHIP_EXCHANGE_PER_BATCH_EXACT;
    auto exchange = [&](int s, int pass, int launch) -> int32_t {
        if (!fn) return enqueue_exchange(c);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int32_t r = fn(user, s, pass | ((launch + 1) << 16));
        if (r != 0) return fail(BEPUHIP_E_STATE, "exchange call-back failed with status " + std::to_string(r));
        return BEPUHIP_OK;
    };
    for (int s = 0; s < substeps; ++s) {
        if (s > 0 && c->inc_blocks > 0)
            hipLaunchKernelGGL(batch_kernel<kStageIncremental>, dim3(c->inc_blocks), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_inc_tbs, 0, c->inc_tb_count, c->d_bodies, substep_dt, inv_dt);
        if (body_blocks > 0)
            hipLaunchKernelGGL(substep_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, s > 0 ? 1 : 0,
                               in->integrate_velocity_for_kinematics, 0, sp);
        if (c->boundary_count > 0)  // deltas of this substep are relative to the integrated velocities (identical on every holder)
            hipLaunchKernelGGL(boundary_snapshot_kernel, dim3((c->boundary_count + 255) / 256), dim3(256), 0, c->stream, (const float4*)c->d_bodies, (const int*)c->d_boundary, c->boundary_count,
                               c->d_boundary_snapshot);
        if (island_passes) enqueue_cluster_pass(c, kStageWarmStart, s, sp);
        for (int b = 0; b < c->launch_count && !island_passes; ++b) {
            if (c->batch_blocks[b] > 0) {
                enqueue_requirk(c, s, b, sp);
                hipLaunchKernelGGL(batch_kernel<kStageWarmStart>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                                   c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
            }
            if (per_batch && (st = exchange(s, 0, b)) != BEPUHIP_OK) return st;  // also after a batch this rank has nothing in: the ranks' collectives pair up by count
        }
        if (!per_batch && (st = exchange(s, 0, -1)) != BEPUHIP_OK) return st;
        for (int it = 0; it < iterations[s]; ++it) {
            if (island_passes) enqueue_cluster_pass(c, kStageSolve, s, sp);
            for (int b = 0; b < c->launch_count && !island_passes; ++b) {
                if (c->batch_blocks[b] > 0)
                    hipLaunchKernelGGL(batch_kernel<kStageSolve>, dim3(c->batch_blocks[b]), dim3(kBlock), 0, c->stream, (const DevTypeBatch*)c->d_tbs, c->batch_begin[b],
                                       c->batch_begin[b + 1] - c->batch_begin[b], c->d_bodies, substep_dt, inv_dt);
                if (per_batch && (st = exchange(s, 1 + it, b)) != BEPUHIP_OK) return st;
            }
            if (!per_batch && (st = exchange(s, 1 + it, -1)) != BEPUHIP_OK) return st;
        }
    }
    if (body_blocks > 0) {
        const float vdt = in->allow_substeps_for_unconstrained ? substep_dt : dt;
        const StepParams fsp = make_params(c, in, vdt, vdt, 1.0f / vdt);
        hipLaunchKernelGGL(final_integrate_kernel, dim3(body_blocks), dim3(256), 0, c->stream, c->d_bodies, (const unsigned*)c->d_flags, c->body_count, dt, substep_dt, substeps,
                           in->allow_substeps_for_unconstrained, in->integrate_velocity_for_kinematics, 0, fsp);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(solve_event(c, false));
    return bepuhip_sync(c);
}

int32_t bepuhip_solve_exchanged(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in, bepuhip_exchange_fn fn, void* user) {
    if (!fn) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null exchange call-back");
    return run_exchanged(c, dt, substeps, iterations, in, fn, user);
}

int32_t bepuhip_solve_lattice(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    return run_exchanged(c, dt, substeps, iterations, in, nullptr, nullptr);
}

// ---- One scene on several devices, exact (include/bepuhip.h: device groups) ----
static HostTypeBatch* find_tb(bepuhip_ctx* c, int batch, int type_id);
int32_t bepuhip_set_device_group(bepuhip_ctx* c, int32_t world, int32_t rank) {
    if (!c || world < 1 || world > kMaxPeers + 1 || rank < 0 || rank >= world) return fail(BEPUHIP_E_INVALID_ARGUMENT, "a device group has 1 to 8 members and ranks 0 .. world - 1");
    if (c->building) return fail(BEPUHIP_E_STATE, "set_device_group between begin_constraints and end_constraints");
    if (c->built && (world != c->group_world || rank != c->group_rank)) return fail(BEPUHIP_E_STATE, "the constraints were planned for another device group: call set_device_group before begin_constraints");
    c->group_world = world; c->group_rank = rank;
    return BEPUHIP_OK;
}
int32_t bepuhip_get_shared_records(bepuhip_ctx* c, void** records_out, int64_t* bytes_out) {
    if (!c || !records_out || !bytes_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *records_out = c->clusters_shared ? (void*)c->d_shared_vel : nullptr;  // (a plan without shared bodies — whole islands only — has no records to exchange)
    *bytes_out = c->clusters_shared ? (int64_t)(c->shared_bodies * 4 * sizeof(float4)) : 0;
    return BEPUHIP_OK;
}
static int32_t upload_peer_table(bepuhip_ctx* c) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->d_peer_table) HIP_TRY(hipMalloc((void**)&c->d_peer_table, kMaxPeers * sizeof(void*)));
    if (!c->peer_records.empty()) HIP_TRY(copy_sync(c, c->d_peer_table, c->peer_records.data(), c->peer_records.size() * sizeof(void*), hipMemcpyHostToDevice));
    return BEPUHIP_OK;
}
int32_t bepuhip_set_peer_records(bepuhip_ctx* c, int32_t peer, void* records) {
    if (!c || peer < 0 || peer >= c->group_world - 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "a group of N devices has N - 1 peers (ordinals 0 .. N - 2)");
    if ((int)c->peer_records.size() <= peer) c->peer_records.resize(peer + 1, nullptr);
    c->peer_records[peer] = records;
    if ((int)c->peer_on_this_device.size() <= peer) c->peer_on_this_device.resize(peer + 1, 0);
    hipPointerAttribute_t where;
    c->peer_on_this_device[peer] = records && hipPointerGetAttributes(&where, records) == hipSuccess && (where.device == c->device || where.type == hipMemoryTypeHost) ? 1 : 0;  // (a table in host memory: BEPUHIP_GROUP_FAKE_REMOTE, members of one device)
    (void)hipGetLastError();
    for (void* p : c->peer_records) if (!p) return BEPUHIP_OK;  // incomplete: the table is uploaded with the last entry
    if ((int)c->peer_records.size() != c->group_world - 1) return BEPUHIP_OK;
    return upload_peer_table(c);
}
// Members of a device group wait for each other INSIDE their kernels, so their kernels must run at the same time. One member per device (the deployment) has a device's
// queues to itself. Members that share a device (a test configuration: this pool has single-GPU boxes) depend on the runtime giving each of their streams a hardware queue
// of its own: ROCm multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues per device, the null stream included, and two members whose
// streams share a queue run one after the other — each waits for the other's records until the watchdog words report a stall (found by tests/test_gpu_soak.py: two members
// beside two idle contexts). Refused up front instead: more live contexts on the device than hardware queues left.
static int32_t group_queue_check(const bepuhip_ctx* c) {
    if (c->group_world <= 1 || !c->clusters_shared || c->device >= kMaxCountedDevices || env_int("BEPUHIP_GROUP_QUEUE_CHECK", 1) == 0) return BEPUHIP_OK;
    bool shares_device = false;
    for (uint8_t same : c->peer_on_this_device) shares_device |= same != 0;
    if (!shares_device) return BEPUHIP_OK;
    static const int hw_queues = std::max(1, env_int("GPU_MAX_HW_QUEUES", 4));
    const int live = g_live_contexts[c->device].load();
    if (live + 1 > hw_queues)
        return fail(BEPUHIP_E_STATE, "device group members share device " + std::to_string(c->device) + " with " + std::to_string(live) + " live contexts, but the runtime has " + std::to_string(hw_queues) +
                    " hardware queues per device (null stream included): members whose streams share a queue cannot run at the same time. Destroy idle contexts, or start the process "
                    "with GPU_MAX_HW_QUEUES=" + std::to_string(live + 1) + " or more (one member per device needs neither)");
    return BEPUHIP_OK;
}
int32_t bepuhip_export_shared_records(bepuhip_ctx* c, void* ipc_handle_out) {
    if (!c || !ipc_handle_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == BEPUHIP_IPC_HANDLE_BYTES, "IPC handle size");
    if (!c->clusters_shared) { memset(ipc_handle_out, 0, BEPUHIP_IPC_HANDLE_BYTES); return BEPUHIP_OK; }
    HIP_TRY(hipSetDevice(c->device));
    hipIpcMemHandle_t handle;
    HIP_TRY(hipIpcGetMemHandle(&handle, c->d_shared_vel));
    memcpy(ipc_handle_out, &handle, sizeof(handle));
    return BEPUHIP_OK;
}
int32_t bepuhip_import_peer_records(bepuhip_ctx* c, int32_t peer, const void* ipc_handle) {
    if (!c || !ipc_handle) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (peer < 0 || peer >= c->group_world - 1) return fail(BEPUHIP_E_INVALID_ARGUMENT, "a group of N devices has N - 1 peers (ordinals 0 .. N - 2)");
    HIP_TRY(hipSetDevice(c->device));
    hipIpcMemHandle_t handle;
    memcpy(&handle, ipc_handle, sizeof(handle));
    void* opened = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&opened, handle, hipIpcMemLazyEnablePeerAccess));
    // A peer whose scene outgrew its table frees it and exports a new one (the header's re-exchange): the mapping of the old table is closed here, not at destroy
    if ((int)c->peer_opened.size() <= peer) c->peer_opened.resize(peer + 1, nullptr);
    if (c->peer_opened[peer] && c->peer_opened[peer] != opened) { HIP_TRY(hipStreamSynchronize(c->stream)); hipIpcCloseMemHandle(c->peer_opened[peer]); }
    c->peer_opened[peer] = opened;
    return bepuhip_set_peer_records(c, peer, opened);
}
// 1 for the bodies whose state this context leaves final at the end of a solve: the bodies of its own clusters, and — on rank 0 — the bodies of no cluster
// (unconstrained and kinematic ones: every device integrates them identically, one speaks for all).
static void owned_body_mask(const bepuhip_ctx* c, std::vector<uint8_t>& mask, int count) {
    mask.assign((size_t)count, 0);
    for (int i = 0; i < count; ++i) {
        // the LIVE body -> cluster table when the context keeps one (structural updates move bodies between clusters, bring them into the plan and take them out:
        // bepu_soft_updates.h), else the upload's snapshot — such a context leaves the island schedule at its first structural update
        const std::vector<int32_t>& body_cluster = c->soft_ok ? c->body_cluster : c->group_body_cluster;
        const int cl = (size_t)i < body_cluster.size() ? body_cluster[i] : -1;
        mask[i] = cl < 0 ? (c->group_rank == 0) : (cl >= c->cluster_first && cl < c->cluster_first + c->cluster_local);
    }
}
int32_t bepuhip_get_owned_bodies(bepuhip_ctx* c, uint8_t* mask_out, int32_t count) {
    if (!c || (!mask_out && count > 0) || count < 0 || count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad get_owned_bodies argument");
    if (c->group_world <= 1 || !c->clusters_enabled) { if (count > 0) memset(mask_out, 1, (size_t)count); return BEPUHIP_OK; }
    std::vector<uint8_t> mask;
    owned_body_mask(c, mask, count);
    if (count > 0) memcpy(mask_out, mask.data(), (size_t)count);
    return BEPUHIP_OK;
}
int32_t bepuhip_get_owned_constraints(bepuhip_ctx* c, int32_t batch, int32_t type_id, uint8_t* mask_out) {
    if (!c || !mask_out || !c->built) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad argument or no constraints");
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb) return fail(BEPUHIP_E_INVALID_ARGUMENT, "no such type batch");
    if (c->group_world <= 1 || !c->clusters_enabled || tb->seg_begin.empty()) { if (tb->count > 0) memset(mask_out, 1, (size_t)tb->count); return BEPUHIP_OK; }
    for (int i = 0; i < tb->count; ++i) {  // the constraint's device slot lies in the segment of the cluster that runs it
        const int d = tb->perm_inverse(i);
        const int cl = (int)(std::upper_bound(tb->seg_begin.begin(), tb->seg_begin.end(), d) - tb->seg_begin.begin()) - 1;
        mask_out[i] = cl >= c->cluster_first && cl < c->cluster_first + c->cluster_local;
    }
    return BEPUHIP_OK;
}
// The end-of-step exchange of a device group on the solver's stream: every device contributes the MotionState halves of the bodies it owns (zeros elsewhere), one
// unsigned-integer all-reduce returns every owner's bit patterns exactly, and every device writes what it does not own. One collective per step.
__global__ __launch_bounds__(256) void owned_pack_kernel(const float4* __restrict__ bodies, const uint8_t* __restrict__ owned, uint4* __restrict__ dense, int count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)count * 4) return;
    const float4 v = bodies[(i >> 2) * 8 + (i & 3)];
    dense[i] = owned[i >> 2] ? make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)) : make_uint4(0u, 0u, 0u, 0u);
}
__global__ __launch_bounds__(256) void owned_unpack_kernel(float4* __restrict__ bodies, const uint8_t* __restrict__ owned, const uint4* __restrict__ dense, int count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)count * 4 || owned[i >> 2]) return;
    const uint4 v = dense[i];
    bodies[(i >> 2) * 8 + (i & 3)] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
int32_t bepuhip_sync_owned_bodies(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    if (c->group_world <= 1 || c->body_count == 0) return BEPUHIP_OK;
    if (!c->comm) return fail(BEPUHIP_E_STATE, "a device group exchanges its bodies over the context's communicator (bepuhip_comm_init / comm_adopt)");
    HIP_TRY(hipSetDevice(c->device));
    if (c->owned_mask_bodies != c->body_count) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_owned_mask) hipFree(c->d_owned_mask);
        if (c->d_owned_dense) hipFree(c->d_owned_dense);
        c->d_owned_mask = nullptr; c->d_owned_dense = nullptr; c->owned_mask_bodies = 0;
        HIP_TRY(hipMalloc((void**)&c->d_owned_mask, (size_t)c->body_count));
        HIP_TRY(hipMalloc((void**)&c->d_owned_dense, (size_t)c->body_count * 64));
        std::vector<uint8_t> mask;
        owned_body_mask(c, mask, c->body_count);
        HIP_TRY(copy_sync(c, c->d_owned_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
        c->owned_mask_bodies = c->body_count;
    }
    const unsigned blocks = (unsigned)(((size_t)c->body_count * 4 + 255) / 256);
    owned_pack_kernel<<<blocks, 256, 0, c->stream>>>((const float4*)c->d_bodies, c->d_owned_mask, (uint4*)c->d_owned_dense, c->body_count);
    const ncclResult_t r = rccl()->AllReduce(c->d_owned_dense, c->d_owned_dense, (size_t)c->body_count * 16, ncclUint32, ncclSum, (ncclComm_t)c->comm, c->stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    owned_unpack_kernel<<<blocks, 256, 0, c->stream>>>(c->d_bodies, c->d_owned_mask, (const uint4*)c->d_owned_dense, c->body_count);
    HIP_TRY(hipGetLastError());
    return BEPUHIP_OK;
}

int32_t bepuhip_sync(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    if (!c->solve_timed) {}  // (bepuhip_set_solve_timing is off: no events were recorded)
    else if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) == hipSuccess) { c->last_ms = ms; c->last_ms_valid = true; }
    else (void)hipGetLastError();  // no solve has been enqueued yet (the events were never recorded): not an error of THIS call — and it must not stay behind as the thread's last error, where the next launch check would find it (round 5: a sync between an upload's row transfers and the first solve made the following transfer fail with "invalid resource handle")
    c->desc_ring_used = 0;  // every transfer_rows descriptor table has been consumed
    if (c->clusters_enabled) {
        unsigned st[8];
        memcpy(st, c->d_status, sizeof(st));
        if (st[0] != 0) {
            memset(c->d_status, 0, 256);
            char msg[256];
            snprintf(msg, sizeof(msg), "cluster schedule stalled: cluster %u kind %u item %u waiting on %u, wanted %u saw %u, claim counter %u", st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
            return fail(BEPUHIP_E_DEVICE, msg);
        }
    }
    return BEPUHIP_OK;
}

int32_t bepuhip_solve(bepuhip_ctx* c, float dt, int32_t substeps, const int32_t* iterations, const bepuhip_integrator* in) {
    int32_t st = bepuhip_solve_async(c, dt, substeps, iterations, in);
    if (st != BEPUHIP_OK) return st;
    return bepuhip_sync(c);
}

int32_t bepuhip_get_bodies(bepuhip_ctx* c, void* out, int32_t count) {
    if (!c || (!out && count > 0) || count < 0 || count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad get_bodies argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (count > 0) HIP_TRY(copy_sync(c, out, c->d_bodies, (size_t)count * 128, hipMemcpyDeviceToHost));  // an empty simulation is a valid one
    return BEPUHIP_OK;
}

// BufferPool blocks are pinned, unmanaged memory that lives as long as the simulation (BufferPool.cs:42,83): registering them with the HIP runtime once makes every
// copy from / to them a DMA at the link's rate and truly asynchronous (pageable memory is staged through the runtime's own buffers at a fraction of that).
int32_t bepuhip_register_host_memory(bepuhip_ctx* c, void* memory, int64_t bytes) {
    if (!c || !memory || bytes <= 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad host memory range");
    HIP_TRY(hipSetDevice(c->device));
    for (void* p : c->registered_host) if (p == memory) return BEPUHIP_OK;
    const hipError_t e = hipHostRegister(memory, (size_t)bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) {
        // Somebody else's registration (another context's, or a neighbouring sub-allocation of the same BufferPool block) overlaps this range. That is fine only when it
        // covers the range WHOLE: a buffer pinned in part would still be read by asynchronous DMA as if it were pinned throughout. Not recorded: not ours to unpin.
        // Every page is asked for: two neighbouring registrations with an unpinned hole between them have a pinned first and last byte. Rare path, once per
        // buffer: a microsecond per page.
        (void)hipGetLastError();
        bool covered = true;
        const uintptr_t first_page = (uintptr_t)memory & ~(uintptr_t)4095, last_byte = (uintptr_t)memory + (uintptr_t)bytes - 1;
        for (uintptr_t page = first_page; covered && page <= last_byte; page += 4096) {
            hipPointerAttribute_t at{};
            const void* probe = (const void*)(page < (uintptr_t)memory ? (uintptr_t)memory : page);
            covered = hipPointerGetAttributes(&at, probe) == hipSuccess && at.type == hipMemoryTypeHost;
        }
        (void)hipGetLastError();
        if (covered) return BEPUHIP_OK;
        return fail(BEPUHIP_E_INVALID_ARGUMENT, "host range overlaps an existing registration that does not cover it: register whole allocation blocks (BufferPool blocks), not sub-ranges");
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(BEPUHIP_E_DEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
    c->registered_host.push_back(memory);
    c->registered_bytes.push_back((size_t)bytes);
    return BEPUHIP_OK;
}
int32_t bepuhip_unregister_host_memory(bepuhip_ctx* c, void* memory) {
    if (!c || !memory) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    for (size_t i = 0; i < c->registered_host.size(); ++i)
        if (c->registered_host[i] == memory) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipHostUnregister(memory));
            c->registered_host.erase(c->registered_host.begin() + i);
            c->registered_bytes.erase(c->registered_bytes.begin() + i);
            return BEPUHIP_OK;
        }
    return fail(BEPUHIP_E_INVALID_ARGUMENT, "this memory was not registered through this context");
}

// Host memory this context registered is mapped into the device's address space: a kernel reads and writes it directly over the link — one launch where a copy per
// type batch (or a strided 2-D copy, which the runtime runs at a fifth of the link's rate: tools/probes/pcie_frame_probe.hip, profiles/r05_s1_pcie_frame_probe.txt)
// would be needed. nullptr: the range is not wholly inside one of OUR registrations (pageable memory, somebody else's registration): the caller stages.
static void* mapped_pointer(bepuhip_ctx* c, const void* host, size_t bytes) {
    if (!host || bytes == 0 || env_int("BEPUHIP_NO_ZERO_COPY", 0)) return nullptr;
    for (size_t i = 0; i < c->registered_host.size(); ++i) {
        const char* base = (const char*)c->registered_host[i];
        if ((const char*)host < base || (const char*)host + bytes > base + c->registered_bytes[i]) continue;
        void* device = nullptr;
        if (hipHostGetDevicePointer(&device, c->registered_host[i], 0) != hipSuccess || !device) { (void)hipGetLastError(); return nullptr; }
        return (char*)device + ((const char*)host - base);
    }
    return nullptr;
}
// MotionState halves of `count` BodyDynamics records straight into the host's array: lane quartets move one body's four float4 (64 of every 128 bytes).
__global__ __launch_bounds__(256) void poses_out_kernel(const float4* __restrict__ bodies, float4* __restrict__ host, int count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)count * 4) return;
    const size_t at = (i >> 2) * 8 + (i & 3);
    host[at] = bodies[at];
}

// What a host needs back after a solve: poses and velocities — the first 64 bytes of every 128-byte BodyDynamics (MotionState: orientation, position, linear,
// angular; BodyProperties.cs:318-338). The inertia half is the host's own (local inertia) or only valid inside the frame (world inertia, BodyProperties.cs:291-297).
// The async form is enqueued behind the solve on the context's stream; bepuhip_sync waits for it.
int32_t bepuhip_get_poses_and_velocities_async(bepuhip_ctx* c, void* body_dynamics_aos, int32_t count) {
    if (!c || (!body_dynamics_aos && count > 0) || count < 0 || count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad get_poses_and_velocities argument");
    HIP_TRY(hipSetDevice(c->device));
    if (count == 0) return BEPUHIP_OK;
    // (16-byte stores: a registered buffer that is not 16-byte aligned takes the 2-D copy — ADVICE r5; BodyDynamics arrays of the reference's BufferPool are)
    void* mapped = ((uintptr_t)body_dynamics_aos & 15) == 0 ? mapped_pointer(c, body_dynamics_aos, (size_t)count * 128) : nullptr;
    if (mapped) {  // 15 MB in 0.34 ms; the 2-D copy below takes 1.4 ms
        poses_out_kernel<<<(unsigned)(((size_t)count * 4 + 255) / 256), 256, 0, c->stream>>>((const float4*)c->d_bodies, (float4*)mapped, count);
        HIP_TRY(hipGetLastError());
        return BEPUHIP_OK;
    }
    HIP_TRY(hipMemcpy2DAsync(body_dynamics_aos, 128, c->d_bodies, 128, 64, (size_t)count, hipMemcpyDeviceToHost, c->stream));
    return BEPUHIP_OK;
}
int32_t bepuhip_get_poses_and_velocities(bepuhip_ctx* c, void* body_dynamics_aos, int32_t count) {
    const int32_t st = bepuhip_get_poses_and_velocities_async(c, body_dynamics_aos, count);
    if (st != BEPUHIP_OK) return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}

static HostTypeBatch* find_tb(bepuhip_ctx* c, int batch, int type_id) {
    const uint64_t key = ((uint64_t)(uint32_t)batch << 32) | (uint32_t)type_id;
    auto hit = c->tb_lookup.find(key);
    if (hit != c->tb_lookup.end() && (size_t)hit->second < c->tbs.size() && c->tbs[hit->second].batch == batch && c->tbs[hit->second].type_id == type_id) return &c->tbs[hit->second];
    for (size_t t = 0; t < c->tbs.size(); ++t)
        if (c->tbs[t].batch == batch && c->tbs[t].type_id == type_id) { c->tb_lookup[key] = (int)t; return &c->tbs[t]; }
    return nullptr;
}
static int32_t download_aosoa(bepuhip_ctx* c, HostTypeBatch* tb, size_t off, int fields, float* out) {
    if (tb->count == 0) return BEPUHIP_OK;
    std::vector<float> soa((size_t)fields * tb->stride);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(copy_sync(c, soa.data(), c->d_slab + off, soa.size() * 4, hipMemcpyDeviceToHost));
    const int W = c->W;
    for (int i = 0; i < tb->count; ++i) {
        const size_t bundle = (size_t)(i / W), lane = (size_t)(i % W);
        const int d = tb->perm.empty() ? i : tb->perm_inverse(i);
        for (int f = 0; f < fields; ++f) out[bundle * fields * W + (size_t)f * W + lane] = soa[(size_t)f * tb->stride + d];
    }
    return BEPUHIP_OK;
}
int32_t bepuhip_get_accumulated_impulses(bepuhip_ctx* c, int32_t batch, int32_t type_id, float* out) {
    if (!c || !out || !c->built) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad argument or no constraints");
    HIP_TRY(hipSetDevice(c->device));
    { const int32_t fs = flush_structural(c); if (fs != BEPUHIP_OK) return fs; }
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb) return fail(BEPUHIP_E_INVALID_ARGUMENT, "no such type batch");
    return download_aosoa(c, tb, tb->accum_off, tb->info.impulse, out);
}
int32_t bepuhip_get_prestep(bepuhip_ctx* c, int32_t batch, int32_t type_id, float* out) {
    if (!c || !out || !c->built) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad argument or no constraints");
    HIP_TRY(hipSetDevice(c->device));
    { const int32_t fs = flush_structural(c); if (fs != BEPUHIP_OK) return fs; }
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb) return fail(BEPUHIP_E_INVALID_ARGUMENT, "no such type batch");
    return download_aosoa(c, tb, tb->prestep_off, tb->info.prestep, out);
}


// ---- Structural updates: queue, apply, re-derive ----
static int32_t apply_pending_ops(bepuhip_ctx* c) {
    if (c->pending_ops.empty()) return BEPUHIP_OK;
    std::stable_sort(c->pending_ops.begin(), c->pending_ops.end(), [](const bepuhip_ctx::PendingOp& a, const bepuhip_ctx::PendingOp& b) { return a.tb < b.tb; });
    std::vector<StructuralOp> ops;
    std::vector<int> group_begin;
    for (size_t i = 0; i < c->pending_ops.size(); ++i) {
        if (i == 0 || c->pending_ops[i].tb != c->pending_ops[i - 1].tb) group_begin.push_back((int)i);
        ops.push_back(c->pending_ops[i].op);
    }
    group_begin.push_back((int)ops.size());
    const int groups = (int)group_begin.size() - 1;
    if (c->pending_payload.empty()) c->pending_payload.push_back(0u);
    const size_t bytes = ops.size() * sizeof(StructuralOp) + group_begin.size() * 4 + c->pending_payload.size() * 4;
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, bytes));
    StructuralOp* d_ops = (StructuralOp*)d;
    int* d_groups = (int*)(d + ops.size() * sizeof(StructuralOp));
    unsigned* d_payload = (unsigned*)(d_groups + group_begin.size());
    HIP_TRY(hipMemcpyAsync(d_ops, ops.data(), ops.size() * sizeof(StructuralOp), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_groups, group_begin.data(), group_begin.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_payload, c->pending_payload.data(), c->pending_payload.size() * 4, hipMemcpyHostToDevice, c->stream));
    for (uint32_t* slab : {c->d_slab, c->d_slab0})  // the snapshot follows: reset_state restores "what the set_* / update_* / structural calls produced"
        if (slab) hipLaunchKernelGGL(apply_structural_ops_kernel, dim3(groups), dim3(64), 0, c->stream, slab, (const StructuralOp*)d_ops, (const int*)d_groups, (const unsigned*)d_payload);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(d);
    c->pending_ops.clear();
    c->pending_payload.clear();
    return BEPUHIP_OK;
}

static int32_t flush_structural(bepuhip_ctx* c) {
    int32_t st = BEPUHIP_OK;
    if (c->soft_ok && !soft_bodies_still_constrained(c) && (st = leave_island_schedule(c)) != BEPUHIP_OK) return st;  // a body lost its last constraint: not this plan's scene any more
    if ((st = flush_soft(c)) != BEPUHIP_OK) return st;
    if ((st = apply_pending_ops(c)) != BEPUHIP_OK) return st;
    if (!c->structure_dirty) return BEPUHIP_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    clear_graphs(c);  // grids and descriptor pointers are baked into captured launches
    c->graphs_cleared_by_structure = true;  // the solve that follows launches eagerly: a graph captured now would be thrown away by the next frame's updates
    c->total_constraints = 0;
    for (auto& tb : c->tbs) {  // (a type batch of the sequential fallback batch counts its empty lanes in `count`)
        if (c->has_fallback && tb.batch == c->fallback_threshold && !tb.occupied.empty()) { for (int i = 0; i < tb.count && (size_t)i < tb.occupied.size(); ++i) c->total_constraints += tb.occupied[i]; }
        else c->total_constraints += tb.count;
    }
    // (a sequential fallback batch: its dependency levels are rebuilt from its references as the device holds them — rows in the caller's order here: a context
    // with a fallback batch leaves the island schedule for its first structural update, structural_preamble)
    std::vector<std::vector<int32_t>> fallback_refs;
    if (c->has_fallback)
        for (auto& tb : c->tbs) {
            if (tb.batch != c->fallback_threshold) continue;
            std::vector<int32_t> rows((size_t)tb.info.bodies * tb.stride, -1);
            if (!rows.empty() && c->d_slab) HIP_TRY(copy_sync(c, rows.data(), c->d_slab + tb.refs_off, rows.size() * 4, hipMemcpyDeviceToHost));
            fallback_refs.push_back(std::move(rows));
        }
    if ((st = build_descriptors(c, fallback_refs)) != BEPUHIP_OK) return st;
    c->structure_dirty = false;
    return rebuild_flags(c);
}

// Re-lays the slab (and its snapshot) out for the type batches' current `stride`s; `old` holds their previous layout. Rows are copied device to device.
struct OldLayout { size_t refs_off, prestep_off, accum_off; int stride, count; };
static int32_t relayout_slab(bepuhip_ctx* c, const std::vector<OldLayout>& old) {
    size_t words = 0;
    for (auto& tb : c->tbs) {
        tb.refs_off = words; words += (size_t)tb.info.bodies * tb.stride;
        tb.prestep_off = words; words += (size_t)tb.info.prestep * tb.stride;
        tb.accum_off = words; words += (size_t)tb.info.impulse * tb.stride;
        tb.lrefs_off = words;
    }
    uint32_t* fresh[2] = {nullptr, nullptr};
    uint32_t* prev[2] = {c->d_slab, c->d_slab0};
    for (int k = 0; k < 2 && words > 0; ++k) {
        HIP_TRY(hipMalloc((void**)&fresh[k], words * 4));
        HIP_TRY(hipMemsetAsync(fresh[k], 0xFF, words * 4, c->stream));  // unused lanes read as -1 references / NaN floats: never touched by a launch (count bounds them)
        for (size_t t = 0; t < c->tbs.size(); ++t) {
            const HostTypeBatch& tb = c->tbs[t];
            if (t >= old.size() || old[t].count == 0 || old[t].stride == 0 || !prev[k]) continue;
            const OldLayout& o = old[t];
            const size_t src[3] = {o.refs_off, o.prestep_off, o.accum_off}, dst[3] = {tb.refs_off, tb.prestep_off, tb.accum_off};
            const int rows[3] = {tb.info.bodies, tb.info.prestep, tb.info.impulse};
            for (int r = 0; r < 3; ++r)
                HIP_TRY(hipMemcpy2DAsync(fresh[k] + dst[r], (size_t)tb.stride * 4, prev[k] + src[r], (size_t)o.stride * 4, (size_t)o.count * 4, rows[r], hipMemcpyDeviceToDevice, c->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_slab) hipFree(c->d_slab);
    if (c->d_slab0) hipFree(c->d_slab0);
    c->d_slab = fresh[0]; c->d_slab0 = fresh[1];
    c->slab_words = words; c->slab_alloc_words = words;
    c->structure_dirty = true;
    return BEPUHIP_OK;
}
static std::vector<OldLayout> current_layout(const bepuhip_ctx* c) {
    std::vector<OldLayout> old;
    for (auto& tb : c->tbs) old.push_back({tb.refs_off, tb.prestep_off, tb.accum_off, tb.stride, tb.count});
    return old;
}

// The island schedule stores every type batch permuted by cluster; structural updates address constraints by the caller's indices. The first one
// brings the rows back into the caller's order on the device and leaves the island schedule (planning clusters needs the whole topology).
static int32_t leave_island_schedule(bepuhip_ctx* c) {
    bool permuted = c->clusters_enabled;
    for (auto& tb : c->tbs) permuted |= !tb.perm.empty();
    if (!permuted) return BEPUHIP_OK;
    { const int32_t fs = flush_soft(c); if (fs != BEPUHIP_OK) return fs; }  // the device has to show what the caller has been told
    c->soft_ok = false; c->soft_split = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    clear_graphs(c);
    {   // every permuted type batch's device slot -> caller's index table in ONE device buffer, sent once (round 6; a hipMalloc, a copy and a wait per type batch and slab until
        // then: 9 ms for the headline scene's 32 type batches, on the caller's thread in front of a structural frame or a bepuhip_replan_begin)
        size_t perm_words = 0;
        for (auto& tb : c->tbs) if (!tb.perm.empty() && tb.device_extent() > 0) perm_words += tb.perm.size();
        int* d_perms = nullptr;
        if (perm_words > 0) {
            { const int32_t st = staging_reserve(c, perm_words * 4); if (st != BEPUHIP_OK) return st; }
            int32_t* staged = (int32_t*)c->h_staging;
            size_t at = 0;
            for (auto& tb : c->tbs) if (!tb.perm.empty() && tb.device_extent() > 0) { memcpy(staged + at, tb.perm.data(), tb.perm.size() * 4); at += tb.perm.size(); }
            HIP_TRY(hipMalloc((void**)&d_perms, perm_words * 4));
            HIP_TRY(hipMemcpyAsync(d_perms, staged, perm_words * 4, hipMemcpyHostToDevice, c->stream));
        }
        for (int k = 0; k < 2; ++k) {
            uint32_t*& slab = k == 0 ? c->d_slab : c->d_slab0;
            if (!slab || c->slab_words == 0) continue;
            uint32_t* fresh = nullptr;
            HIP_TRY(hipMalloc((void**)&fresh, c->slab_words * 4));
            HIP_TRY(hipMemcpyAsync(fresh, slab, c->slab_words * 4, hipMemcpyDeviceToDevice, c->stream));
            size_t at = 0;
            for (auto& tb : c->tbs) {
                const int extent = tb.device_extent();
                if (tb.perm.empty() || extent == 0) continue;
                const size_t offs[3] = {tb.refs_off, tb.prestep_off, tb.accum_off};
                const int rows[3] = {tb.info.bodies, tb.info.prestep, tb.info.impulse};
                for (int r = 0; r < 3; ++r)
                    if (rows[r] > 0)
                        hipLaunchKernelGGL(unpermute_rows_kernel, dim3((extent + 255) / 256), dim3(256), 0, c->stream, (const unsigned*)(slab + offs[r]), (unsigned*)(fresh + offs[r]), (const int*)(d_perms + at),
                                           extent, tb.stride, rows[r]);
                at += tb.perm.size();
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
            hipFree(slab);
            slab = fresh;
        }
        if (d_perms) hipFree(d_perms);
    }
    for (auto& tb : c->tbs) {
        tb.perm.clear(); tb.inv.clear();
        tb.slots = 0; tb.seg_begin.clear(); std::vector<int32_t>().swap(tb.dev_refs);
        if (tb.d_device_index && !tb.index_pooled) hipFree(tb.d_device_index);
        tb.d_device_index = nullptr; tb.index_pooled = false;
    }
    for (void** p : {(void**)&c->d_clusters, (void**)&c->d_items, (void**)&c->d_batch_item_begin, (void**)&c->d_cluster_bodies, (void**)&c->d_clustered_dynamic, (void**)&c->d_kinlist,
                     (void**)&c->d_cycles}) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    c->clusters_enabled = false; c->cluster_count = 0; c->clustered_dynamic_count = 0; c->kinlist_count = 0;
    c->requirk_marks.clear(); c->requirk_stale = true;  // (the marks lived in rows that no longer exist; the launch-per-batch lists are rebuilt by the next conserving solve)
    c->structure_dirty = true;
    return BEPUHIP_OK;
}

// `stay`: the caller will try the update on the island layout first (bepu_soft_updates.h) and leaves the island schedule itself if that is not possible.
static int32_t structural_preamble(bepuhip_ctx* c, bool stay = false) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    if (c->building) return fail(BEPUHIP_E_STATE, "structural update between begin_constraints and end_constraints");
    if (c->in_substep_event) return fail(BEPUHIP_E_STATE, "structural update inside a substep event handler: the substep loop runs on the constraint set the solve started with");
    // A sequential fallback batch (round 5; UNSUPPORTED until then): constraints of the synchronized batches come and go on the launch-per-batch rows — the context leaves
    // its island layout for the first such update —, the fallback batch's own rows stay as uploaded (additions to it and removals from it are refused by the calls below;
    // its references may be patched when a body moves in memory: the dependency levels are rebuilt at the flush, flush_structural).
    if (stay && c->soft_ok && !c->has_fallback) return BEPUHIP_OK;  // bookkeeping on the host only: no device call on this path (a structural call per changed contact per frame)
    HIP_TRY(hipSetDevice(c->device));
    return leave_island_schedule(c);
}

int32_t bepuhip_get_constraint_count(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t* out) {
    if (!c || !out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    *out = tb ? tb->count : 0;
    return BEPUHIP_OK;
}

// ---- The sequential fallback batch's own additions and removals (round 6; TypeProcessor.cs:451-571, :633-694), on the launch-per-batch rows: the rows are in the caller's
// layout there, empty lanes included, and the dependency levels the batch is solved in are rebuilt from the references at the flush (flush_structural). ----
static void queue_fallback_op(bepuhip_ctx* c, HostTypeBatch* tb, int kind, int src, int dst, int lanes) {
    bepuhip_ctx::PendingOp p;
    p.tb = (int)(tb - c->tbs.data());
    p.op = StructuralOp{(unsigned)tb->refs_off, (unsigned)tb->prestep_off, (unsigned)tb->accum_off, tb->stride, tb->info.bodies, tb->info.prestep, tb->info.impulse, kind, src, dst, 0u, lanes};
    c->pending_ops.push_back(p);
}
static int32_t remove_from_fallback(bepuhip_ctx* c, HostTypeBatch* tb, int index) {
    // (structural_preamble has brought the rows into the caller's order: a context with a fallback batch leaves its island layout for any structural update)
    if (tb->occupied.size() < (size_t)tb->count) tb->occupied.resize((size_t)tb->count, 0);
    if (!tb->occupied[index]) return fail(BEPUHIP_E_INVALID_ARGUMENT, "the lane of the sequential fallback batch is empty already");
    const int W = c->W;
    queue_fallback_op(c, tb, 4, 1, index, 0);  // RemoveBodyReferencesLane (:301-311)
    tb->occupied[index] = 0;
    const int bundle = index / W;
    bool empty = true;
    for (int i = bundle * W; i < std::min(tb->count, (bundle + 1) * W); ++i) empty = empty && !tb->occupied[i];
    if (empty) {  // :650-681
        int last_bundle = (tb->count + W - 1) / W - 1;
        if (bundle != last_bundle) {  // the last bundle's prestep data, accumulated impulses and references overwrite the dead bundle's (whole bundles: its empty lanes too)
            queue_fallback_op(c, tb, 5, last_bundle * W, bundle * W, W);
            queue_fallback_op(c, tb, 4, W, last_bundle * W, 0);  // (the reference leaves the moved-from bundle's references behind ConstraintCount and clears them when a new bundle is opened there; cleared now)
            if (tb->occupied.size() < (size_t)(last_bundle + 1) * W) tb->occupied.resize((size_t)(last_bundle + 1) * W, 0);
            for (int l = 0; l < W; ++l) { tb->occupied[(size_t)bundle * W + l] = tb->occupied[(size_t)last_bundle * W + l]; tb->occupied[(size_t)last_bundle * W + l] = 0; }
            --last_bundle;
        }
        int inner = 0;  // BundleIndexing.GetLastSetLaneCount of the new last bundle's occupied lanes
        if (last_bundle >= 0) for (int l = 0; l < W; ++l) if ((size_t)last_bundle * W + l < tb->occupied.size() && tb->occupied[(size_t)last_bundle * W + l]) inner = l + 1;
        tb->count = std::max(0, last_bundle * W + inner);
        tb->occupied.resize((size_t)tb->count);
    }
    c->structure_dirty = true; c->requirk_stale = true;
    return BEPUHIP_OK;
}

static int32_t add_constraint_at_impl(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t index, const int32_t* refs, const float* prestep) {
    int32_t st = structural_preamble(c, false);
    if (st != BEPUHIP_OK) return st;
    if (!refs || !prestep || index < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad add_constraint_at argument");
    if (batch != c->fallback_threshold) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bepuhip_add_constraint_at places constraints of the sequential fallback batch (batch index == FallbackBatchThreshold); synchronized batches append: bepuhip_add_constraint");
    TypeInfoH info;
    if (!type_info(type_id, info)) return fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(type_id));
    for (int k = 0; k < info.bodies; ++k)
        if (refs[k] < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "empty body reference");
    const int W = c->W;
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    const int count = tb ? tb->count : 0;
    const int bundles = (count + W - 1) / W;
    if (tb && tb->occupied.size() < (size_t)count) tb->occupied.resize((size_t)count, 0);
    const bool into_hole = index < bundles * W && (index >= count || !tb->occupied[index]);
    const bool new_bundle = index == bundles * W;
    if (!into_hole && !new_bundle)
        return fail(BEPUHIP_E_INVALID_ARGUMENT, "index " + std::to_string(index) + " is neither an empty lane of one of the type batch's " + std::to_string(bundles) + " bundles nor lane 0 of a new one (TypeProcessor.cs:451-571)");
    if (!tb || index >= tb->stride) {  // a new type batch, or rows that have to grow (InternalResize: capacity doubles)
        if ((st = apply_pending_ops(c)) != BEPUHIP_OK) return st;
        std::vector<OldLayout> old = current_layout(c);
        if (!tb) {
            HostTypeBatch fresh;
            fresh.batch = batch; fresh.type_id = type_id; fresh.count = 0; fresh.stride = 64; fresh.info = info;
            fresh.refs_off = fresh.prestep_off = fresh.accum_off = fresh.lrefs_off = 0;
            size_t pos = 0;
            while (pos < c->tbs.size() && c->tbs[pos].batch <= batch) ++pos;
            c->tbs.insert(c->tbs.begin() + pos, fresh);
            old.insert(old.begin() + pos, OldLayout{0, 0, 0, 0, 0});
            c->batch_count = std::max(c->batch_count, batch + 1);
            c->has_fallback = true;  // (Batches[FallbackBatchThreshold] exists from now on)
            c->has_widened_types = c->has_widened_types || is_widened_type(type_id); c->type_mask |= 1ull << (type_id & 63);
            c->has_joint_types = c->has_joint_types || type_id > kContact4;
            c->built = true;
        } else {
            while (index >= tb->stride) tb->stride = std::max(64, tb->stride * 2);
        }
        if ((st = relayout_slab(c, old)) != BEPUHIP_OK) return st;
        tb = find_tb(c, batch, type_id);
    }
    if (new_bundle) queue_fallback_op(c, tb, 4, W, index, 0);  // AddBodyReferencesLane with innerIndex 0 (:287-296): the bundle's lanes start empty
    bepuhip_ctx::PendingOp p;
    p.tb = (int)(tb - c->tbs.data());
    p.op = StructuralOp{(unsigned)tb->refs_off, (unsigned)tb->prestep_off, (unsigned)tb->accum_off, tb->stride, tb->info.bodies, tb->info.prestep, tb->info.impulse, 1, 0, index,
                        (unsigned)c->pending_payload.size(), 0};
    for (int k = 0; k < info.bodies; ++k) {
        c->pending_payload.push_back((uint32_t)refs[k]);
        c->referenced_bodies = std::max(c->referenced_bodies, (refs[k] & kRefMask) + 1);
    }
    for (int f = 0; f < info.prestep; ++f) { uint32_t w; memcpy(&w, &prestep[f], 4); c->pending_payload.push_back(w); }
    c->pending_ops.push_back(p);
    tb->count = std::max(tb->count, index + 1);
    tb->occupied.resize((size_t)tb->count, 0);
    tb->occupied[index] = 1;
    c->structure_dirty = true; c->requirk_stale = true;
    return BEPUHIP_OK;
}

static int32_t add_constraint_impl(bepuhip_ctx* c, int32_t batch, int32_t type_id, const int32_t* refs, const float* prestep, int32_t* index_out) {
    int32_t st = structural_preamble(c, true);
    if (st != BEPUHIP_OK) return st;
    if (batch < 0 || !refs || !prestep) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad add_constraint argument");
    if (batch >= c->fallback_threshold) return fail(BEPUHIP_E_UNSUPPORTED, "the constraint belongs to the sequential fallback batch (batch index >= FallbackBatchThreshold): its lane is the reference's choice, bepuhip_add_constraint_at");
    TypeInfoH info;
    if (!type_info(type_id, info)) return fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(type_id));
    for (int k = 0; k < info.bodies; ++k)
        if (refs[k] < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "empty body reference");
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (c->soft_ok) {  // on the island layout: a free slot in the segment of the cluster the bodies live in
        for (int k = 0; k < info.bodies; ++k) c->referenced_bodies = std::max(c->referenced_bodies, (refs[k] & kRefMask) + 1);  // (a solve before the matching set_bodies is refused, validate_solve)
        bool violation = false;
        SoftCallTimer timer(c);
        if (tb && soft_add(c, tb, refs, prestep, &violation)) {
            if (index_out) *index_out = tb->count - 1;
            c->requirk_stale = true;
            return BEPUHIP_OK;
        }
        if (violation) return fail(BEPUHIP_E_INVALID_ARGUMENT, "a dynamic body of the new constraint is already referenced in this batch (a body appears at most once per synchronized batch, Solver.cs:1046-1051)");
        if (!tb) soft_refuse("the new constraint opens a type batch");
        HIP_TRY(hipSetDevice(c->device));
        if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;
        tb = find_tb(c, batch, type_id);
    }
    if (!tb || tb->count == tb->stride) {  // a new type batch (ConstraintBatch.GetOrCreateTypeBatch) or a full one (InternalResize: capacity doubles, TypeProcessor.cs:317-320)
        if ((st = apply_pending_ops(c)) != BEPUHIP_OK) return st;  // queued operations carry offsets of the layout that is about to change
        std::vector<OldLayout> old = current_layout(c);
        if (!tb) {
            HostTypeBatch fresh;
            fresh.batch = batch; fresh.type_id = type_id; fresh.count = 0; fresh.stride = 64; fresh.info = info;
            fresh.refs_off = fresh.prestep_off = fresh.accum_off = fresh.lrefs_off = 0;
            size_t pos = 0;
            while (pos < c->tbs.size() && c->tbs[pos].batch <= batch) ++pos;  // type batches of a batch in creation order, batches in index order
            c->tbs.insert(c->tbs.begin() + pos, fresh);
            old.insert(old.begin() + pos, OldLayout{0, 0, 0, 0, 0});
            c->batch_count = std::max(c->batch_count, batch + 1);
            c->has_widened_types = c->has_widened_types || is_widened_type(type_id); c->type_mask |= 1ull << (type_id & 63);
            c->has_joint_types = c->has_joint_types || type_id > kContact4;
            c->built = true;
        } else {
            tb->stride = std::max(64, tb->stride * 2);
        }
        if ((st = relayout_slab(c, old)) != BEPUHIP_OK) return st;
        tb = find_tb(c, batch, type_id);
    }
    bepuhip_ctx::PendingOp p;
    p.tb = (int)(tb - c->tbs.data());
    p.op = StructuralOp{(unsigned)tb->refs_off, (unsigned)tb->prestep_off, (unsigned)tb->accum_off, tb->stride, tb->info.bodies, tb->info.prestep, tb->info.impulse, 1, 0, tb->count,
                        (unsigned)c->pending_payload.size(), 0};
    for (int k = 0; k < info.bodies; ++k) {
        c->pending_payload.push_back((uint32_t)refs[k]);
        c->referenced_bodies = std::max(c->referenced_bodies, (refs[k] & kRefMask) + 1);
    }
    for (int f = 0; f < info.prestep; ++f) { uint32_t w; memcpy(&w, &prestep[f], 4); c->pending_payload.push_back(w); }
    c->pending_ops.push_back(p);
    if (index_out) *index_out = tb->count;
    tb->count += 1;
    c->structure_dirty = true; c->requirk_stale = true;
    return BEPUHIP_OK;
}

static int32_t remove_constraint_impl(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t index) {
    int32_t st = structural_preamble(c, true);
    if (st != BEPUHIP_OK) return st;
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb || index < 0 || index >= tb->count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "Can only remove elements that are actually in the batch!");  // TypeProcessor.cs:636
    if (c->has_fallback && batch >= c->fallback_threshold) return remove_from_fallback(c, tb, index);
    if (c->soft_ok) {  // on the island layout: the slot is freed where it is, the caller's indices are remapped
        SoftCallTimer timer(c);
        if (soft_remove(c, tb, index)) { c->requirk_stale = true; return BEPUHIP_OK; }
        HIP_TRY(hipSetDevice(c->device));
        if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;
        tb = find_tb(c, batch, type_id);
    }
    const int last = tb->count - 1;
    if (index < last) {  // TypeProcessor.cs:702-714
        bepuhip_ctx::PendingOp p;
        p.tb = (int)(tb - c->tbs.data());
        p.op = StructuralOp{(unsigned)tb->refs_off, (unsigned)tb->prestep_off, (unsigned)tb->accum_off, tb->stride, tb->info.bodies, tb->info.prestep, tb->info.impulse, 0, last, index, 0u, 0};
        c->pending_ops.push_back(p);
    }
    tb->count = last;
    c->structure_dirty = true; c->requirk_stale = true;
    return BEPUHIP_OK;
}

static int32_t update_body_reference_impl(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t index, int32_t slot, int32_t ref) {
    int32_t st = structural_preamble(c, true);
    if (st != BEPUHIP_OK) return st;
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb || index < 0 || index >= tb->count || slot < 0 || slot >= tb->info.bodies || ref < 0) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad update_body_reference argument");
    if (c->soft_ok) {  // on the island layout: the body keeps its place in the plan under its new index
        c->referenced_bodies = std::max(c->referenced_bodies, (ref & kRefMask) + 1);
        if (soft_update_reference(c, tb, index, slot, ref)) { c->requirk_stale = true; return BEPUHIP_OK; }
        HIP_TRY(hipSetDevice(c->device));
        if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;
        tb = find_tb(c, batch, type_id);
    }
    bepuhip_ctx::PendingOp p;
    p.tb = (int)(tb - c->tbs.data());
    p.op = StructuralOp{(unsigned)tb->refs_off, (unsigned)tb->prestep_off, (unsigned)tb->accum_off, tb->stride, tb->info.bodies, tb->info.prestep, tb->info.impulse, 2, slot, index,
                        (unsigned)c->pending_payload.size(), 0};
    c->pending_payload.push_back((uint32_t)ref);
    c->pending_ops.push_back(p);
    c->referenced_bodies = std::max(c->referenced_bodies, (ref & kRefMask) + 1);
    c->structure_dirty = true; c->requirk_stale = true;  // constrained flags follow the references
    return BEPUHIP_OK;
}

// Two constraints of a type batch change places (include/bepuhip.h: what a host that diffs the reference's type batches needs besides append and swap-with-last).
static int32_t swap_constraints_impl(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t a, int32_t b) {
    int32_t st = structural_preamble(c, true);
    if (st != BEPUHIP_OK) return st;
    HostTypeBatch* tb = find_tb(c, batch, type_id);
    if (!tb || a < 0 || b < 0 || a >= tb->count || b >= tb->count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad swap_constraints argument");
    if (a == b) return BEPUHIP_OK;
    if (c->has_fallback && batch >= c->fallback_threshold) return fail(BEPUHIP_E_UNSUPPORTED, "a swap inside the sequential fallback batch would change the order its bundles are solved in: re-upload with begin/set/end");
    if (c->soft_ok) {  // on an island layout the rows stay where they are: two entries of the index tables change hands
        if (soft_swap(c, tb, a, b)) { c->requirk_stale = true; return BEPUHIP_OK; }
        HIP_TRY(hipSetDevice(c->device));
        if ((st = leave_island_schedule(c)) != BEPUHIP_OK) return st;
        tb = find_tb(c, batch, type_id);
    }
    bepuhip_ctx::PendingOp p;
    p.tb = (int)(tb - c->tbs.data());
    p.op = StructuralOp{(unsigned)tb->refs_off, (unsigned)tb->prestep_off, (unsigned)tb->accum_off, tb->stride, tb->info.bodies, tb->info.prestep, tb->info.impulse, 3, a, b, 0u, 0};
    c->pending_ops.push_back(p);
    c->structure_dirty = true; c->requirk_stale = true;
    return BEPUHIP_OK;
}

// The public calls: the implementations above, and — while a re-plan is in flight (bepuhip_replan_begin) — a line in its log for every call that succeeded.
int32_t bepuhip_add_constraint(bepuhip_ctx* c, int32_t batch, int32_t type_id, const int32_t* refs, const float* prestep, int32_t* index_out) {
    int32_t index = -1;
    const int32_t st = add_constraint_impl(c, batch, type_id, refs, prestep, &index);
    if (index_out) *index_out = index;
    if (st == BEPUHIP_OK && c->replan_job) replan_log(c, 0, batch, type_id, index, 0, 0, refs, prestep);
    return st;
}
int32_t bepuhip_add_constraint_at(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t index, const int32_t* refs, const float* prestep) {
    const int32_t st = add_constraint_at_impl(c, batch, type_id, index, refs, prestep);
    if (st == BEPUHIP_OK && c->replan_job) replan_log(c, 4, batch, type_id, index, 0, 0, refs, prestep);
    return st;
}
int32_t bepuhip_remove_constraint(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t index) {
    const int32_t st = remove_constraint_impl(c, batch, type_id, index);
    if (st == BEPUHIP_OK && c->replan_job) replan_log(c, 1, batch, type_id, index, 0, 0, nullptr, nullptr);
    return st;
}
int32_t bepuhip_update_body_reference(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t index, int32_t slot, int32_t ref) {
    const int32_t st = update_body_reference_impl(c, batch, type_id, index, slot, ref);
    if (st == BEPUHIP_OK && c->replan_job) replan_log(c, 2, batch, type_id, index, slot, ref, nullptr, nullptr);
    return st;
}
int32_t bepuhip_swap_constraints(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t a, int32_t b) {
    const int32_t st = swap_constraints_impl(c, batch, type_id, a, b);
    if (st == BEPUHIP_OK && a != b && c->replan_job) replan_log(c, 3, batch, type_id, a, b, 0, nullptr, nullptr);
    return st;
}

// A frame's structural changes in one call (include/bepuhip.h), in order, each exactly as the call of the same name.
int32_t bepuhip_apply_structural_ops(bepuhip_ctx* c, const bepuhip_structural_op* ops, int32_t count, const uint32_t* payload, int32_t payload_words, int32_t* failed_op_out) {
    if (failed_op_out) *failed_op_out = -1;
    if (!c || count < 0 || (count > 0 && !ops) || payload_words < 0 || (payload_words > 0 && !payload)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad apply_structural_ops argument");
    // On an island layout the bookkeeping of one operation is twenty-odd cache lines nothing before it touched (bepu_soft_updates.h, soft_prefetch_*): the lines of the
    // operations ahead are asked for while the current one runs. Hints only: the type batch is looked up afresh (one entry cached: operations come in runs of a type batch).
    struct { int batch = -1, type_id = -1; const HostTypeBatch* base = nullptr; HostTypeBatch* tb = nullptr; } ahead_cache;
    auto prefetch_ahead = [&](int32_t at, int stage) {
        if (at >= count || !c->soft_ok) return;
        const bepuhip_structural_op& op = ops[at];
        if (op.kind != 0 && op.kind != 1) return;
        if (ahead_cache.batch != op.batch_index || ahead_cache.type_id != op.type_id || ahead_cache.base != c->tbs.data()) {
            ahead_cache.batch = op.batch_index; ahead_cache.type_id = op.type_id; ahead_cache.base = c->tbs.data(); ahead_cache.tb = find_tb(c, op.batch_index, op.type_id);
        }
        HostTypeBatch* tb = ahead_cache.tb;
        if (!tb) return;
        if (op.kind == 1) { soft_prefetch_remove(c, tb, op.index, stage); return; }
        if (stage >= 2 || op.payload_offset < 0 || (int64_t)op.payload_offset + tb->info.bodies > (int64_t)payload_words) return;
        soft_prefetch_add(c, tb, (const int32_t*)(payload + op.payload_offset), stage);
    };
    for (int32_t i = 0; i < count; ++i) {
        const bepuhip_structural_op& op = ops[i];
        prefetch_ahead(i + 12, 0); prefetch_ahead(i + 8, 1); prefetch_ahead(i + 5, 2); prefetch_ahead(i + 2, 3);
        int32_t st = BEPUHIP_OK;
        switch (op.kind) {
            case 0: {
                TypeInfoH info;
                if (!type_info(op.type_id, info)) { st = fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(op.type_id)); break; }
                if (op.payload_offset < 0 || (int64_t)op.payload_offset + info.bodies + info.prestep > (int64_t)payload_words) { st = fail(BEPUHIP_E_INVALID_ARGUMENT, "an addition's payload lies outside the payload array"); break; }
                int32_t index = -1;
                st = bepuhip_add_constraint(c, op.batch_index, op.type_id, (const int32_t*)(payload + op.payload_offset), (const float*)(payload + op.payload_offset + info.bodies), &index);
                if (st == BEPUHIP_OK && op.index >= 0 && index != op.index)
                    st = fail(BEPUHIP_E_STATE, "the device's type batch is out of step with the caller's: the added constraint got index " + std::to_string(index) + ", expected " + std::to_string(op.index));
                break;
            }
            case 1: st = bepuhip_remove_constraint(c, op.batch_index, op.type_id, op.index); break;
            case 2: st = bepuhip_update_body_reference(c, op.batch_index, op.type_id, op.index, op.slot, op.reference); break;
            case 3: st = bepuhip_swap_constraints(c, op.batch_index, op.type_id, op.index, op.slot); break;
            case 4: {
                TypeInfoH info;
                if (!type_info(op.type_id, info)) { st = fail(BEPUHIP_E_UNSUPPORTED, "unknown constraint type id " + std::to_string(op.type_id)); break; }
                if (op.payload_offset < 0 || (int64_t)op.payload_offset + info.bodies + info.prestep > (int64_t)payload_words) { st = fail(BEPUHIP_E_INVALID_ARGUMENT, "an addition's payload lies outside the payload array"); break; }
                st = bepuhip_add_constraint_at(c, op.batch_index, op.type_id, op.index, (const int32_t*)(payload + op.payload_offset), (const float*)(payload + op.payload_offset + info.bodies));
                break;
            }
            default: st = fail(BEPUHIP_E_INVALID_ARGUMENT, "unknown structural operation kind " + std::to_string(op.kind));
        }
        if (st != BEPUHIP_OK) {
            if (failed_op_out) *failed_op_out = i;
            const std::string why = bepuhip_last_error();
            return fail(st, "structural operation " + std::to_string(i) + " of " + std::to_string(count) + ": " + why);
        }
    }
    return BEPUHIP_OK;
}

// IPoseIntegratorCallbacks.IntegrateVelocity as data (include/bepuhip.h): the model stays with the context until the next call.
int32_t bepuhip_set_velocity_model(bepuhip_ctx* c, const bepuhip_velocity_model* model, const float* per_body_gravity, int32_t body_count) {
    if (!c || !model) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (model->model < 0 || model->model > BEPUHIP_VELOCITY_RADIAL_GRAVITY) return fail(BEPUHIP_E_UNSUPPORTED, "unknown velocity model " + std::to_string(model->model) + ": keep simulation.Solve for arbitrary IntegrateVelocity code");
    if (model->model == BEPUHIP_VELOCITY_PER_BODY_GRAVITY && (!per_body_gravity || body_count <= 0)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "the per-body gravity model needs one value per body");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (model->model == BEPUHIP_VELOCITY_PER_BODY_GRAVITY) {
        if (body_count > c->body_gravity_capacity) {
            if (c->d_body_gravity) hipFree(c->d_body_gravity);
            c->d_body_gravity = nullptr; c->body_gravity_capacity = 0;
            HIP_TRY(hipMalloc((void**)&c->d_body_gravity, (size_t)body_count * 4));
            c->body_gravity_capacity = body_count;
        }
        HIP_TRY(copy_sync(c, c->d_body_gravity, per_body_gravity, (size_t)body_count * 4, hipMemcpyHostToDevice));
        c->body_gravity_count = body_count;
    }
    c->velocity_model = *model;
    clear_graphs(c);  // captured launch sequences carry the model in their kernel arguments
    return BEPUHIP_OK;
}

// ---- Device-resident incremental updates: ranged rewrites of what already lives in HBM, no re-plan, no full re-upload ----
static int32_t stage_reserve(bepuhip_ctx* c, size_t floats) {
    if (floats <= c->stage_floats) return BEPUHIP_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_stage) hipFree(c->d_stage);
    c->d_stage = nullptr; c->stage_floats = 0;
    HIP_TRY(hipMalloc((void**)&c->d_stage, floats * 4));
    c->stage_floats = floats;
    return BEPUHIP_OK;
}
static int32_t device_index_of(bepuhip_ctx* c, HostTypeBatch* tb, const int** out) {
    *out = nullptr;
    if (tb->perm.empty()) return BEPUHIP_OK;  // launch-per-batch layout: host order
    if (!tb->d_device_index) {
        tb->perm_inverse(0);
        HIP_TRY(hipMalloc((void**)&tb->d_device_index, std::max<size_t>(tb->inv.size(), (size_t)tb->device_extent()) * 4));  // room for every index additions on the island layout can create
        if (!tb->inv.empty()) HIP_TRY(copy_sync(c, tb->d_device_index, tb->inv.data(), tb->inv.size() * 4, hipMemcpyHostToDevice));
    }
    *out = tb->d_device_index;
    return BEPUHIP_OK;
}
// A frame's row traffic in one call (include/bepuhip.h). Consecutive items of one direction form a run: a run of updates is N host-to-device copies into the staging
// buffer and ONE scatter launch (two with the snapshot), a run of gets ONE gather launch and N device-to-host copies — where the single calls cost two or three stream
// operations per type batch each (96 type batches on the bench scene). Everything is enqueued on the context's stream; the staging buffer is reused in stream order.
static int32_t transfer_run(bepuhip_ctx* c, const bepuhip_row_transfer* items, int count) {
    const bool update = items[0].kind == BEPUHIP_ROWS_UPDATE_PRESTEP || items[0].kind == BEPUHIP_ROWS_UPDATE_IMPULSES;
    std::vector<RowTransferDesc> descs;
    struct Copy { size_t stage_off, floats; void* host; };
    std::vector<Copy> copies;  // the items that go through the staging buffer (host memory this context has not registered)
    size_t staged_floats = 0;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const bepuhip_row_transfer& it = items[i];
        HostTypeBatch* tb = find_tb(c, it.batch_index, it.type_id);
        if (!tb) return fail(BEPUHIP_E_INVALID_ARGUMENT, "transfer_rows item " + std::to_string(i) + ": no such type batch");
        const int bundles = (tb->count + c->W - 1) / c->W;
        const int first_bundle = it.first_bundle, bundle_count = it.bundle_count < 0 ? bundles - first_bundle : it.bundle_count;
        if (first_bundle < 0 || bundle_count < 0 || (int64_t)first_bundle + bundle_count > bundles || (!it.bundles && bundle_count > 0))
            return fail(BEPUHIP_E_INVALID_ARGUMENT, "transfer_rows item " + std::to_string(i) + ": bundle range exceeds the type batch");
        const int first = first_bundle * c->W, n = std::min(bundle_count * c->W, tb->count - first);
        if (n <= 0) continue;
        const bool prestep = it.kind == BEPUHIP_ROWS_UPDATE_PRESTEP || it.kind == BEPUHIP_ROWS_GET_PRESTEP;
        const int fields = prestep ? tb->info.prestep : tb->info.impulse;
        const int* index;
        int32_t st = device_index_of(c, tb, &index);
        if (st != BEPUHIP_OK) return st;
        const size_t floats = (size_t)bundle_count * fields * c->W;
        float* mapped = ((uintptr_t)it.bundles & 15u) ? nullptr : (float*)mapped_pointer(c, it.bundles, floats * 4);  // the kernel moves 16-byte quads
        if (!mapped) { copies.push_back(Copy{staged_floats, floats, it.bundles}); staged_floats += (floats + 3) / 4 * 4; }
        // (a staged item's pointer is filled in below, once the staging buffer has its final address)
        descs.push_back(RowTransferDesc{mapped, prestep ? tb->prestep_off : tb->accum_off, index, first, n, fields, tb->stride, blocks, mapped ? 0 : (int)copies.size()});
        blocks += (int)((floats / 4 + 255) / 256);  // a thread moves four consecutive floats of the bundles: four lanes of one field
    }
    if (descs.empty()) return BEPUHIP_OK;
    const size_t table_bytes = descs.size() * sizeof(RowTransferDesc), table_floats = (table_bytes + 3) / 4;
    int32_t st = stage_reserve(c, staged_floats + table_floats + 64);
    if (st != BEPUHIP_OK) return st;
    for (RowTransferDesc& d : descs) if (d.staged) d.bundles = c->d_stage + copies[d.staged - 1].stage_off;
    // the descriptor table travels through pinned memory that stays untouched until the next bepuhip_sync (the copy is asynchronous)
    if (c->desc_ring_used + table_bytes > c->desc_ring_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->desc_ring_used = 0;
        if (table_bytes > c->desc_ring_bytes) {
            if (c->h_desc_ring) hipHostFree(c->h_desc_ring);
            c->h_desc_ring = nullptr; c->desc_ring_bytes = 0;
            const size_t want = std::max<size_t>(table_bytes * 2, 1u << 18);
            HIP_TRY(hipHostMalloc((void**)&c->h_desc_ring, want, hipHostMallocDefault));
            c->desc_ring_bytes = want;
        }
    }
    char* slot = c->h_desc_ring + c->desc_ring_used;
    c->desc_ring_used += (table_bytes + 63) / 64 * 64;
    memcpy(slot, descs.data(), table_bytes);
    float* d_table = c->d_stage + ((staged_floats + 15) / 16) * 16;
    HIP_TRY(hipMemcpyAsync(d_table, slot, table_bytes, hipMemcpyHostToDevice, c->stream));
    if (update) {
        for (const Copy& copy : copies) HIP_TRY(hipMemcpyAsync(c->d_stage + copy.stage_off, copy.host, copy.floats * 4, hipMemcpyHostToDevice, c->stream));
        // (the pristine snapshot follows, as with every update_* call: same launch, the bundles cross the link once)
        transfer_rows_kernel<true><<<blocks, 256, 0, c->stream>>>((const RowTransferDesc*)d_table, (int)descs.size(), (float*)c->d_slab, (float*)c->d_slab0, c->W);
        HIP_TRY(hipGetLastError());
    } else {
        transfer_rows_kernel<false><<<blocks, 256, 0, c->stream>>>((const RowTransferDesc*)d_table, (int)descs.size(), (float*)c->d_slab, nullptr, c->W);
        HIP_TRY(hipGetLastError());
        for (const Copy& copy : copies) HIP_TRY(hipMemcpyAsync(copy.host, c->d_stage + copy.stage_off, copy.floats * 4, hipMemcpyDeviceToHost, c->stream));
    }
    return BEPUHIP_OK;
}
int32_t bepuhip_transfer_rows_async(bepuhip_ctx* c, const bepuhip_row_transfer* items, int32_t count) {
    if (!c || count < 0 || (count > 0 && !items)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad transfer_rows argument");
    if (!c->built) return fail(BEPUHIP_E_STATE, "no constraints uploaded");
    if (count == 0) return BEPUHIP_OK;
    for (int i = 0; i < count; ++i)
        if (items[i].kind < BEPUHIP_ROWS_UPDATE_PRESTEP || items[i].kind > BEPUHIP_ROWS_GET_IMPULSES) return fail(BEPUHIP_E_INVALID_ARGUMENT, "transfer_rows item " + std::to_string(i) + ": unknown kind");
    HIP_TRY(hipSetDevice(c->device));
    { const int32_t fs = flush_structural(c); if (fs != BEPUHIP_OK) return fs; }
    int begin = 0;
    while (begin < count) {
        const bool update = items[begin].kind <= BEPUHIP_ROWS_UPDATE_IMPULSES;
        int end = begin + 1;
        while (end < count && (items[end].kind <= BEPUHIP_ROWS_UPDATE_IMPULSES) == update) ++end;
        const int32_t st = transfer_run(c, items + begin, end - begin);
        if (st != BEPUHIP_OK) return st;
        begin = end;
    }
    return BEPUHIP_OK;
}

// The single ranged calls are one-item transfers (synchronous unless _async).
static int32_t transfer_one(bepuhip_ctx* c, int kind, int batch, int type_id, int first_bundle, int bundle_count, const void* bundles, bool wait) {
    if (!c || !c->built) return fail(BEPUHIP_E_STATE, "no constraints uploaded");
    if (first_bundle < 0 || bundle_count < 0 || (!bundles && bundle_count > 0)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad bundle range");
    const bepuhip_row_transfer item{kind, batch, type_id, first_bundle, bundle_count, 0, const_cast<void*>(bundles)};
    const int32_t st = bepuhip_transfer_rows_async(c, &item, 1);
    if (st != BEPUHIP_OK) return st;
    if (wait) HIP_TRY(hipStreamSynchronize(c->stream));  // the caller's buffer is free again on return
    return BEPUHIP_OK;
}
int32_t bepuhip_update_prestep(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, const float* prestep_bundles) {
    return transfer_one(c, BEPUHIP_ROWS_UPDATE_PRESTEP, batch, type_id, first_bundle, bundle_count, prestep_bundles, true);
}
int32_t bepuhip_update_prestep_async(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, const float* prestep_bundles) {
    return transfer_one(c, BEPUHIP_ROWS_UPDATE_PRESTEP, batch, type_id, first_bundle, bundle_count, prestep_bundles, false);
}
int32_t bepuhip_update_accumulated_impulses(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, const float* impulse_bundles) {
    return transfer_one(c, BEPUHIP_ROWS_UPDATE_IMPULSES, batch, type_id, first_bundle, bundle_count, impulse_bundles, true);
}
int32_t bepuhip_update_accumulated_impulses_async(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, const float* impulse_bundles) {
    return transfer_one(c, BEPUHIP_ROWS_UPDATE_IMPULSES, batch, type_id, first_bundle, bundle_count, impulse_bundles, false);
}
int32_t bepuhip_get_prestep_range(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, float* prestep_bundles_out) {
    return transfer_one(c, BEPUHIP_ROWS_GET_PRESTEP, batch, type_id, first_bundle, bundle_count, prestep_bundles_out, true);
}
int32_t bepuhip_get_accumulated_impulses_range(bepuhip_ctx* c, int32_t batch, int32_t type_id, int32_t first_bundle, int32_t bundle_count, float* impulse_bundles_out) {
    return transfer_one(c, BEPUHIP_ROWS_GET_IMPULSES, batch, type_id, first_bundle, bundle_count, impulse_bundles_out, true);
}

int32_t bepuhip_update_bodies(bepuhip_ctx* c, const void* aos, int32_t first, int32_t count) {
    if (!c || first < 0 || count < 0 || (!aos && count > 0) || (int64_t)first + count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad body range");
    if (count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->d_bodies + (size_t)first * 8, aos, (size_t)count * 128, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_bodies0 + (size_t)first * 8, c->d_bodies + (size_t)first * 8, (size_t)count * 128, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}
int32_t bepuhip_get_bodies_range(bepuhip_ctx* c, void* aos_out, int32_t first, int32_t count) {
    if (!c || first < 0 || count < 0 || (!aos_out && count > 0) || (int64_t)first + count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad body range");
    if (count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(copy_sync(c, aos_out, c->d_bodies + (size_t)first * 8, (size_t)count * 128, hipMemcpyDeviceToHost));
    return BEPUHIP_OK;
}

// ---- PredictBoundingBoxes on the device ----
static_assert(sizeof(bepuhip_collidable) == sizeof(CollidableIn) && sizeof(bepuhip_collidable) == 64, "bepuhip_collidable layout");
static_assert(sizeof(bepuhip_predicted_bounds) == sizeof(PredictedBounds) && sizeof(bepuhip_predicted_bounds) == 32, "bepuhip_predicted_bounds layout");
constexpr int kBoundsWaveThreshold = 64;  // points / children / triangles above which a body's bounds are computed by a whole wave (predict_heavy_bounds_kernel) instead of its lane
static int32_t check_collidables(const bepuhip_ctx* c, const bepuhip_collidable* collidables, int32_t count) {
    for (int i = 0; i < count; ++i) {
        if (collidables[i].shape_type < -1 || collidables[i].shape_type > 8)
            return fail(BEPUHIP_E_UNSUPPORTED, "shape type " + std::to_string(collidables[i].shape_type) + " is not one of the library's nine (Sphere.Id 0 ... Mesh.Id 8)");
        if (collidables[i].shape_type >= 6) {
            const bool mesh = collidables[i].shape_type == 8;
            const int table = mesh ? c->mesh_count : c->compound_count;
            const float k = collidables[i].shape[0];
            if (!(k >= 0) || k != (float)(int)k || (int)k >= table)
                return fail(BEPUHIP_E_INVALID_ARGUMENT, std::string(mesh ? "mesh" : "compound") + " index " + std::to_string(k) + " of collidable " + std::to_string(i) + " is not one of the " +
                                                             std::to_string(table) + " entries of " + (mesh ? "bepuhip_set_meshes" : "bepuhip_set_compounds"));
            if (!mesh && c->compound_hulls_needed > c->hull_count)
                return fail(BEPUHIP_E_INVALID_ARGUMENT, "a compound child names convex hull " + std::to_string(c->compound_hulls_needed - 1) + " but bepuhip_set_convex_hulls holds " +
                                                             std::to_string(c->hull_count));
        }
        if (collidables[i].shape_type == 5) {
            const float h = collidables[i].shape[0];
            if (!(h >= 0) || h != (float)(int)h || (int)h >= c->hull_count)
                return fail(BEPUHIP_E_INVALID_ARGUMENT, "convex hull index " + std::to_string(h) + " of collidable " + std::to_string(i) + " is not one of the " +
                                                             std::to_string(c->hull_count) + " hulls of bepuhip_set_convex_hulls");
        }
    }
    return BEPUHIP_OK;
}
int32_t bepuhip_set_convex_hulls(bepuhip_ctx* c, const float* points, const int32_t* point_begin, int32_t hull_count) {
    if (!c || hull_count < 0 || (hull_count > 0 && (!points || !point_begin))) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad set_convex_hulls argument");
    for (int h = 0; h < hull_count; ++h)
        if (point_begin[h] < 0 || point_begin[h + 1] <= point_begin[h]) return fail(BEPUHIP_E_INVALID_ARGUMENT, "hull " + std::to_string(h) + " has no points (or the offsets decrease)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_hull_points) hipFree(c->d_hull_points);
    if (c->d_hull_begin) hipFree(c->d_hull_begin);
    c->d_hull_points = nullptr; c->d_hull_begin = nullptr; c->hull_count = 0;
    if (hull_count == 0) return BEPUHIP_OK;
    const size_t total = (size_t)point_begin[hull_count];
    HIP_TRY(hipMalloc((void**)&c->d_hull_points, total * 12));
    HIP_TRY(hipMalloc((void**)&c->d_hull_begin, ((size_t)hull_count + 1) * 4));
    HIP_TRY(copy_sync(c, c->d_hull_points, points, total * 12, hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(c, c->d_hull_begin, point_begin, ((size_t)hull_count + 1) * 4, hipMemcpyHostToDevice));
    c->hull_count = hull_count;
    return BEPUHIP_OK;
}
static_assert(sizeof(bepuhip_compound_child) == sizeof(CompoundChildIn) && sizeof(bepuhip_compound_child) == 68, "bepuhip_compound_child layout");
int32_t bepuhip_set_compounds(bepuhip_ctx* c, const bepuhip_compound_child* children, const int32_t* child_begin, int32_t compound_count) {
    if (!c || compound_count < 0 || (compound_count > 0 && (!children || !child_begin))) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad set_compounds argument");
    int hulls_needed = 0;
    for (int k = 0; k < compound_count; ++k) {
        if (child_begin[k] < 0 || child_begin[k + 1] <= child_begin[k]) return fail(BEPUHIP_E_INVALID_ARGUMENT, "compound " + std::to_string(k) + " has no children (or the offsets decrease)");
        for (int j = child_begin[k]; j < child_begin[k + 1]; ++j) {
            const int t = children[j].shape_type;
            if (t < 0 || t > 5) return fail(BEPUHIP_E_INVALID_ARGUMENT, "child " + std::to_string(j - child_begin[k]) + " of compound " + std::to_string(k) + " has shape type " + std::to_string(t) +
                                                                          ": compound children are convex (Compound.cs:182)");
            if (t == 5) {
                const float h = children[j].shape[0];
                if (!(h >= 0) || h != (float)(int)h) return fail(BEPUHIP_E_INVALID_ARGUMENT, "child of compound " + std::to_string(k) + " names convex hull " + std::to_string(h));
                hulls_needed = std::max(hulls_needed, (int)h + 1);
            }
        }
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_compound_children) hipFree(c->d_compound_children);
    if (c->d_compound_begin) hipFree(c->d_compound_begin);
    c->d_compound_children = nullptr; c->d_compound_begin = nullptr; c->compound_count = 0; c->compound_hulls_needed = 0;
    if (compound_count == 0) return BEPUHIP_OK;
    const size_t total = (size_t)child_begin[compound_count];
    HIP_TRY(hipMalloc((void**)&c->d_compound_children, total * sizeof(CompoundChildIn)));
    HIP_TRY(hipMalloc((void**)&c->d_compound_begin, ((size_t)compound_count + 1) * 4));
    HIP_TRY(copy_sync(c, c->d_compound_children, children, total * sizeof(CompoundChildIn), hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(c, c->d_compound_begin, child_begin, ((size_t)compound_count + 1) * 4, hipMemcpyHostToDevice));
    c->compound_count = compound_count;
    c->compound_hulls_needed = hulls_needed;
    return BEPUHIP_OK;
}
int32_t bepuhip_set_meshes(bepuhip_ctx* c, const float* triangles, const int32_t* triangle_begin, const float* scales, int32_t mesh_count) {
    if (!c || mesh_count < 0 || (mesh_count > 0 && (!triangles || !triangle_begin || !scales))) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad set_meshes argument");
    for (int m = 0; m < mesh_count; ++m)
        if (triangle_begin[m] < 0 || triangle_begin[m + 1] <= triangle_begin[m]) return fail(BEPUHIP_E_INVALID_ARGUMENT, "mesh " + std::to_string(m) + " has no triangles (or the offsets decrease)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_mesh_triangles) hipFree(c->d_mesh_triangles);
    if (c->d_mesh_begin) hipFree(c->d_mesh_begin);
    if (c->d_mesh_scales) hipFree(c->d_mesh_scales);
    c->d_mesh_triangles = nullptr; c->d_mesh_begin = nullptr; c->d_mesh_scales = nullptr; c->mesh_count = 0;
    if (mesh_count == 0) return BEPUHIP_OK;
    const size_t total = (size_t)triangle_begin[mesh_count];
    HIP_TRY(hipMalloc((void**)&c->d_mesh_triangles, total * 36));
    HIP_TRY(hipMalloc((void**)&c->d_mesh_begin, ((size_t)mesh_count + 1) * 4));
    HIP_TRY(hipMalloc((void**)&c->d_mesh_scales, (size_t)mesh_count * 12));
    HIP_TRY(copy_sync(c, c->d_mesh_triangles, triangles, total * 36, hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(c, c->d_mesh_begin, triangle_begin, ((size_t)mesh_count + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(c, c->d_mesh_scales, scales, (size_t)mesh_count * 12, hipMemcpyHostToDevice));
    c->mesh_count = mesh_count;
    return BEPUHIP_OK;
}
int32_t bepuhip_set_collidables(bepuhip_ctx* c, const bepuhip_collidable* collidables, int32_t count) {
    if (!c || count < 0 || (count > 0 && !collidables)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad set_collidables argument");
    int32_t st = check_collidables(c, collidables, count);
    if (st != BEPUHIP_OK) return st;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_collidables) { hipFree(c->d_collidables); c->d_collidables = nullptr; }
    c->collidable_count = count;
    c->resident_hulls_needed = c->resident_compounds_needed = c->resident_meshes_needed = 0;
    for (int i = 0; i < count; ++i) {
        const int t = collidables[i].shape_type, k = (int)collidables[i].shape[0] + 1;
        if (t == 5) c->resident_hulls_needed = std::max(c->resident_hulls_needed, k);
        else if (t == 6 || t == 7) c->resident_compounds_needed = std::max(c->resident_compounds_needed, k);
        else if (t == 8) c->resident_meshes_needed = std::max(c->resident_meshes_needed, k);
    }
    if (count > 0) {
        HIP_TRY(hipMalloc((void**)&c->d_collidables, (size_t)count * sizeof(CollidableIn)));
        HIP_TRY(copy_sync(c, c->d_collidables, collidables, (size_t)count * sizeof(CollidableIn), hipMemcpyHostToDevice));
    }
    return BEPUHIP_OK;
}
int32_t bepuhip_predict_bounding_boxes(bepuhip_ctx* c, float dt, const bepuhip_integrator* in, const bepuhip_collidable* collidables, int32_t count, bepuhip_predicted_bounds* out) {
    if (!c || !in || count < 0 || count > c->body_count || (count > 0 && !out)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad predict_bounding_boxes argument");
    if (!(dt > 0)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "dt must be positive");
    if (c->velocity_model.model == BEPUHIP_VELOCITY_PER_BODY_GRAVITY && c->body_gravity_count < count)  // the kernels read gravity[body] for every body they bound
        return fail(BEPUHIP_E_STATE, "the per-body gravity table holds " + std::to_string(c->body_gravity_count) + " values for " + std::to_string(count) + " bodies (set_velocity_model)");
    const bool resident = collidables == nullptr;
    if (resident && count > c->collidable_count) return fail(BEPUHIP_E_STATE, "no collidables given and fewer resident ones than bodies (bepuhip_set_collidables)");
    if (!resident) { int32_t st = check_collidables(c, collidables, count); if (st != BEPUHIP_OK) return st; }
    if (resident && (c->resident_hulls_needed > c->hull_count || c->resident_compounds_needed > c->compound_count || c->resident_meshes_needed > c->mesh_count ||
                     (c->resident_compounds_needed > 0 && c->compound_hulls_needed > c->hull_count)))
        return fail(BEPUHIP_E_STATE, "the resident collidables name hulls, compounds or meshes that the shape tables no longer hold (they were replaced after bepuhip_set_collidables)");
    if (count == 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    // bodies that get a wave of their own (compounds, meshes, large hulls) are queued by the per-body kernel: {body, bundle integrates} pairs behind a counter
    const bool heavy_pass = (c->compound_count > 0 || c->mesh_count > 0 || c->hull_count > 0) && !getenv("BEPUHIP_BOUNDS_ONE_LANE");
    const size_t in_floats = resident ? 0 : (size_t)count * 16, out_floats = (size_t)count * 8, queue_floats = heavy_pass ? (size_t)count * 2 + 4 : 0;
    int32_t st = stage_reserve(c, in_floats + out_floats + queue_floats);
    if (st != BEPUHIP_OK) return st;
    CollidableIn* d_in = resident ? c->d_collidables : (CollidableIn*)c->d_stage;
    PredictedBounds* d_out = (PredictedBounds*)(c->d_stage + in_floats);
    int* d_heavy_count = heavy_pass ? (int*)(c->d_stage + in_floats + out_floats) : nullptr;
    int2* d_heavy_queue = heavy_pass ? (int2*)(c->d_stage + in_floats + out_floats + 4) : nullptr;
    if (heavy_pass) HIP_TRY(hipMemsetAsync(d_heavy_count, 0, 4, c->stream));
    if (!resident) HIP_TRY(hipMemcpyAsync(d_in, collidables, in_floats * 4, hipMemcpyHostToDevice, c->stream));
    const StepParams sp = make_params(c, in, dt, dt, 1.0f / dt);  // Callbacks.PrepareForIntegration(dt): the full frame step
    hipLaunchKernelGGL(predict_bounds_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, (const float4*)c->d_bodies, count, d_in, resident ? 1 : 0, d_out, dt,
                       in->integrate_velocity_for_kinematics, sp,
                       ShapeTables{HullTable{c->d_hull_points, c->d_hull_begin, c->hull_count}, c->d_compound_children, c->d_compound_begin, c->compound_count, c->d_mesh_triangles,
                                   c->d_mesh_begin, c->d_mesh_scales, c->mesh_count},
                       c->W, d_heavy_queue, d_heavy_count, getenv("BEPUHIP_BOUNDS_WAVE_THRESHOLD") ? atoi(getenv("BEPUHIP_BOUNDS_WAVE_THRESHOLD")) : kBoundsWaveThreshold);
    HIP_TRY(hipGetLastError());
    if (heavy_pass) {
        const int waves = std::min(count, 4096);  // 16 single-wave workgroups on each of the 256 CUs; they loop over the queue
        hipLaunchKernelGGL(predict_heavy_bounds_kernel, dim3(waves), dim3(64), 0, c->stream, (const float4*)c->d_bodies, (const CollidableIn*)d_in, d_out, dt, sp,
                           ShapeTables{HullTable{c->d_hull_points, c->d_hull_begin, c->hull_count}, c->d_compound_children, c->d_compound_begin, c->compound_count, c->d_mesh_triangles,
                                       c->d_mesh_begin, c->d_mesh_scales, c->mesh_count},
                           (const int2*)d_heavy_queue, (const int*)d_heavy_count);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(out, d_out, out_floats * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BEPUHIP_OK;
}

int32_t bepuhip_get_constrained_flags(bepuhip_ctx* c, uint8_t* out, int32_t count) {
    if (!c || !out || count < 0 || count > c->body_count) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<unsigned> f((size_t)count);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (count > 0) HIP_TRY(copy_sync(c, f.data(), c->d_flags, (size_t)count * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < count; ++i) out[i] = (uint8_t)(f[i] & 3u);
    return BEPUHIP_OK;
}

int32_t bepuhip_last_solve_ms(bepuhip_ctx* c, float* ms) {
    if (!c || !ms) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (!c->last_ms_valid) return fail(BEPUHIP_E_STATE, "no timed solve has completed: bepuhip_set_solve_timing(ctx, 1), solve, bepuhip_sync");
    *ms = c->last_ms;
    return BEPUHIP_OK;
}
int32_t bepuhip_set_solve_timing(bepuhip_ctx* c, int32_t enabled) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    c->solve_timing = enabled != 0;
    if (!c->solve_timing) c->last_ms_valid = false;
    return BEPUHIP_OK;
}
int32_t bepuhip_set_profiling(bepuhip_ctx* c, int32_t enabled) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    c->profiling = enabled != 0;
    return BEPUHIP_OK;
}
int32_t bepuhip_get_profile(bepuhip_ctx* c, int32_t family, float* ms, int32_t* launches) {
    if (!c || family < 0 || family > 5) return fail(BEPUHIP_E_INVALID_ARGUMENT, "bad family");
    if (ms) *ms = c->prof_ms[family];
    if (launches) *launches = c->prof_launches[family];
    return BEPUHIP_OK;
}
int32_t bepuhip_set_cluster_trace(bepuhip_ctx* c, int32_t enabled) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    clear_graphs(c);  // captured launches bake the trace pointer in
    if (c->d_trace) { hipFree(c->d_trace); c->d_trace = nullptr; c->trace_words = 0; }
    if (enabled && c->clusters_enabled) {
        const ClusterDesc first = c->first_cluster;
        c->trace_words = (size_t)first.item_count * 8 * (size_t)kClusterTracePasses;  // up to 128 passes of cluster 0; the kernel drops later ones
        HIP_TRY(hipMalloc((void**)&c->d_trace, c->trace_words * 8));
        HIP_TRY(fill_async(c, c->d_trace, 0, c->trace_words * 8));
    }
    return BEPUHIP_OK;
}
int32_t bepuhip_get_cluster_trace(bepuhip_ctx* c, uint64_t* out, int64_t capacity_words, int32_t* items_out) {
    if (!c || !out || !items_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (!c->d_trace) return fail(BEPUHIP_E_STATE, "cluster trace is not enabled (or the scene does not use the cluster schedule)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t n = std::min<size_t>(c->trace_words, (size_t)std::max<int64_t>(capacity_words, 0));
    HIP_TRY(copy_sync(c, out, c->d_trace, n * 8, hipMemcpyDeviceToHost));
    *items_out = c->first_cluster.item_count;
    return BEPUHIP_OK;
}
int32_t bepuhip_get_row_policy(bepuhip_ctx* c, int32_t* policy_out) {
    if (!c || !policy_out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    if (c->row_policy < 0 && c->policy_samples >= kPolicySamples) settle_row_policy(c, c->policy_threads, true);
    *policy_out = c->row_policy;
    return BEPUHIP_OK;
}

int32_t bepuhip_get_cluster_cycles(bepuhip_ctx* c, uint64_t* out, int32_t capacity, int32_t* count_out) {
    if (!c || !count_out || (capacity > 0 && !out)) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *count_out = c->clusters_enabled ? c->cluster_count : 0;
    if (!c->clusters_enabled || capacity <= 0) return BEPUHIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(copy_sync(c, out, c->d_cycles, (size_t)std::min(capacity, c->cluster_count) * 8, hipMemcpyDeviceToHost));
    return BEPUHIP_OK;
}
int32_t bepuhip_debug_status(bepuhip_ctx* c, uint32_t* out16) {
    if (!c || !out16) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    memcpy(out16, c->d_status, 64);
    return BEPUHIP_OK;
}
int32_t bepuhip_last_constraint_iterations(bepuhip_ctx* c, int64_t* out) {
    if (!c || !out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *out = c->last_constraint_iterations;
    return BEPUHIP_OK;
}
int32_t bepuhip_get_stream(bepuhip_ctx* c, void** out) {
    if (!c || !out) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null argument");
    *out = (void*)c->stream;
    return BEPUHIP_OK;
}
int32_t bepuhip_reset_state(bepuhip_ctx* c) {
    if (!c) return fail(BEPUHIP_E_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { const int32_t fs = flush_structural(c); if (fs != BEPUHIP_OK) return fs; }
    if (c->body_count > 0) HIP_TRY(hipMemcpyAsync(c->d_bodies, c->d_bodies0, (size_t)c->body_count * 128, hipMemcpyDeviceToDevice, c->stream));
    if (c->slab_words > 0) HIP_TRY(hipMemcpyAsync(c->d_slab, c->d_slab0, c->slab_words * 4, hipMemcpyDeviceToDevice, c->stream));
    return BEPUHIP_OK;
}

}  // extern "C"
