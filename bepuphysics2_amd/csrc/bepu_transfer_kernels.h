// Kernels of the boundary that only bepuhip.hip launches (the cluster units do not include this file: editing it does not rebuild them).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// bepuhip_transfer_rows_async: the ranged kernels above for MANY (type batch, bundle range) pairs in one launch. Item t's bundles (the caller's AOSOA bundles, `fields`
// floats per lane) are at `bundles` — a range of the staging buffer, or the HOST's own registered buffer, which the kernel then reads / writes over the link itself — its
// rows at slab + rows_off; workgroup b serves the item whose [block_begin, next block_begin) holds b. A thread moves four consecutive floats of the bundles (four lanes
// of one field: the bundle width is 4, 8 or 16), so that a wave touches the bundles — the side that may be host memory — as one contiguous kilobyte; the rows take
// scattered dwords, which HBM forgives. Lanes of the last bundle beyond the type batch's count read back as zero.
struct RowTransferDesc { float* bundles; unsigned long long rows_off; const int* device_index; int first, n, fields, stride, block_begin, staged; };
template <bool SCATTER>
__global__ __launch_bounds__(256) void transfer_rows_kernel(const RowTransferDesc* __restrict__ descs, int count, float* __restrict__ slab, float* __restrict__ snapshot, int W) {
    int lo = 0, hi = count - 1;  // the last item whose block_begin <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const RowTransferDesc d = descs[lo];
    const size_t quad = (size_t)((int)blockIdx.x - d.block_begin) * blockDim.x + threadIdx.x;  // floats [4 quad, 4 quad + 4) of the item's bundles
    const int per_bundle = d.fields * W;
    const int bundle_count = (d.n + W - 1) / W;
    if (quad * 4 >= (size_t)bundle_count * per_bundle) return;
    const int bundle = (int)((quad * 4) / per_bundle), within = (int)((quad * 4) % per_bundle), f = within / W, lane0 = within % W;
    float4* at = reinterpret_cast<float4*>(d.bundles + quad * 4);
    float* rows = slab + d.rows_off + (size_t)f * d.stride;
    const int j0 = bundle * W + lane0;
    if (SCATTER) {
        // (the bundles are read ONCE — they may live in host memory — and written to the working rows and, when there is one, to the pristine snapshot
        // bepuhip_reset_state returns to: two launches would pull them over the link twice, 0.45 instead of 0.23 ms for the bench scene's contacts)
        const float4 v = *at;
        const float part[4] = {v.x, v.y, v.z, v.w};
        float* rows2 = snapshot ? snapshot + d.rows_off + (size_t)f * d.stride : nullptr;
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {
            if (j0 + q >= d.n) break;
            const int h = d.first + j0 + q;
            const int row = d.device_index ? d.device_index[h] : h;
            rows[row] = part[q];
            if (rows2) rows2[row] = part[q];
        }
    } else {
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {
            if (j0 + q >= d.n) break;
            const int h = d.first + j0 + q;
            part[q] = rows[d.device_index ? d.device_index[h] : h];
        }
        *at = make_float4(part[0], part[1], part[2], part[3]);
    }
}

}  // namespace
