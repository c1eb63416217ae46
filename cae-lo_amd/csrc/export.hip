// export.hip -- AllVoxels0/1/2 in the reference's exact order (off the hot path; API parity for
// Voxelization(PC), Voxel.py:100-173).
//
// The reference appends a voxel the first time a point touches it: AllVoxels1/2 are in first-touch
// order (Voxel.py:153-158); AllVoxels0 is grouped by 64^3 block, blocks in first-touch order, voxels
// in first-touch order inside a block (Voxel.py:126-143,:161-165).  The voxel map already records
// the smallest point index touching every voxel, so the order is a sort by that index -- by
// (first index of the block, first index of the voxel) for scale 0.  The sort itself is rocPRIM's
// device radix sort (a library primitive, like a memcpy; none of the hot-path kernels use a library).
#include <cstring>

#include "caelo_internal.h"

#include <rocprim/rocprim.hpp>

// forward: helpers shared with voxel.hip are small enough to restate
__device__ static inline int exp_table_insert(unsigned long long *keys, uint32_t mask, unsigned long long key) {
    uint32_t h = caelo_hash64(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        unsigned long long k = keys[h];
        if (k == key) return (int)h;
        if (k == CAELO_EMPTY_KEY) {
            k = atomicCAS(&keys[h], CAELO_EMPTY_KEY, key);
            if (k == CAELO_EMPTY_KEY || k == key) return (int)h;
        }
        h = (h + 1) & mask;
    }
    return -1;
}

__device__ static inline unsigned long long block_of(unsigned long long vkey) {
    const unsigned x = (unsigned)(vkey >> 40) & 0xFFFFF, y = (unsigned)(vkey >> 20) & 0xFFFFF, z = (unsigned)vkey & 0xFFFFF;
    return caelo_pack3((int)(x >> 6), (int)(y >> 6), (int)(z >> 6));
}

__global__ void __launch_bounds__(256) k_exp_block_first(const unsigned long long *__restrict__ vkeys,
                                                         const uint32_t *__restrict__ vfirst, uint32_t vmask,
                                                         unsigned long long *bkeys, uint32_t *bfirst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > vmask) return;
    const unsigned long long k = vkeys[i];
    if (k == CAELO_EMPTY_KEY) return;
    const int s = exp_table_insert(bkeys, vmask, block_of(k));
    if (s >= 0) atomicMin(&bfirst[s], vfirst[i]);
}

__global__ void __launch_bounds__(256) k_exp_compact(const unsigned long long *__restrict__ vkeys,
                                                     const uint32_t *__restrict__ vfirst, uint32_t vmask,
                                                     const unsigned long long *__restrict__ bkeys,
                                                     const uint32_t *__restrict__ bfirst, int use_block,
                                                     unsigned long long *skeys, unsigned long long *svals, int32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > vmask) return;
    const unsigned long long k = vkeys[i];
    if (k == CAELO_EMPTY_KEY) return;
    unsigned long long sk = (unsigned long long)(unsigned)vfirst[i];
    if (use_block) {
        const unsigned long long bk = block_of(k);
        uint32_t h = caelo_hash64(bk) & vmask;
        while (bkeys[h] != bk) h = (h + 1) & vmask;
        sk |= (unsigned long long)(unsigned)bfirst[h] << 32;
    }
    const int p = atomicAdd(count, 1);
    skeys[p] = sk;
    svals[p] = k;
}

// the sorted run is `*count` long (a device word: the call never waits for it); the rest of the padded sort is 0xFF.. keys
__global__ void __launch_bounds__(256) k_exp_write(const unsigned long long *__restrict__ svals, const int32_t *__restrict__ count,
                                                   int64_t capacity, int16_t *__restrict__ out, int64_t *__restrict__ count_out64,
                                                   int32_t *__restrict__ count_out32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *count;
    if (i == 0) {
        if (count_out64) *count_out64 = n;
        if (count_out32) *count_out32 = n;
    }
    if (i >= n || i >= capacity) return;
    const unsigned long long k = svals[i];
    out[3 * i] = (int16_t)((k >> 40) & 0xFFFFF);
    out[3 * i + 1] = (int16_t)((k >> 20) & 0xFFFFF);
    out[3 * i + 2] = (int16_t)(k & 0xFFFFF);
}

// The lists of the scales in `mask`, each in first-touch order, written to outs[scale] (device), their lengths to counts64[scale] /
// counts32[scale] (device; either may be null).  No host round trip: every scale is sorted at its table's size with 0xFF.. keys
// behind the `count` real ones (a few hundred microseconds of sorting more than the exact length would take, against four stream
// synchronisations per call -- which is what serialised the tie redo of many frames on side streams, Engine.resolve_ties_many).
static int export_scales(caelo_voxmap *m, int mask, int16_t *const outs[3], int64_t capacity, int64_t *counts64, int32_t *counts32, hipStream_t s) {
    const size_t vs = (size_t)m->vmask[0] + 1;
    // scratch: block table (keys + first) | sort keys in/out | sort vals in/out | count | rocprim temp
    size_t temp_bytes = 0;
    CAELO_HIP(rocprim::radix_sort_pairs((void *)nullptr, temp_bytes, (unsigned long long *)nullptr,
                                        (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                        (unsigned long long *)nullptr, vs, 0, 64, s));
    const size_t need = vs * (8 + 4) + 4 * vs * 8 + 64 + temp_bytes;
    if ((size_t)m->scratch_bytes < need) {
        if (m->scratch) CAELO_HIP(hipFree(m->scratch));
        CAELO_HIP(hipMalloc(&m->scratch, need));
        m->scratch_bytes = (int64_t)need;
    }
    char *base = (char *)m->scratch;
    unsigned long long *bkeys = (unsigned long long *)base;
    unsigned long long *k_in = bkeys + vs, *k_out = k_in + vs, *v_in = k_out + vs, *v_out = v_in + vs;
    uint32_t *bfirst = (uint32_t *)(v_out + vs);
    int32_t *count = (int32_t *)(bfirst + vs);
    void *temp = (void *)(count + 16);
    for (int sc = 0; sc < 3; ++sc) {
        if (!(mask >> sc & 1)) continue;
        const size_t vsc = (size_t)m->vmask[sc] + 1;
        const unsigned grid = (unsigned)((vsc + 255) / 256);
        CAELO_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
        if (sc == 0) {
            CAELO_HIP(hipMemsetAsync(bkeys, 0xFF, vs * 8, s));
            CAELO_HIP(hipMemsetAsync(bfirst, 0xFF, vs * 4, s));
            k_exp_block_first<<<grid, 256, 0, s>>>(m->vkeys[0], m->vfirst[0], m->vmask[0], bkeys, bfirst);
            CAELO_LAUNCH_CHECK();
        }
        CAELO_HIP(hipMemsetAsync(k_in, 0xFF, vsc * 8, s));
        k_exp_compact<<<grid, 256, 0, s>>>(m->vkeys[sc], m->vfirst[sc], m->vmask[sc], bkeys, bfirst, sc == 0, k_in, v_in, count);
        CAELO_LAUNCH_CHECK();
        size_t tb = temp_bytes;
        CAELO_HIP(rocprim::radix_sort_pairs(temp, tb, k_in, k_out, v_in, v_out, vsc, 0, 64, s));
        k_exp_write<<<grid, 256, 0, s>>>(v_out, count, capacity, outs[sc], counts64 ? counts64 + sc : nullptr, counts32 ? counts32 + sc : nullptr);
        CAELO_LAUNCH_CHECK();
    }
    return CAELO_OK;
}

CAELO_API int caelo_voxmap_export(caelo_ctx *c, caelo_voxmap *m, int16_t *all0, int16_t *all1, int16_t *all2,
                                  int64_t capacity, int64_t *counts, void *stream) {
    CAELO_REQUIRE(c && m && all0 && all1 && all2 && counts, "null argument");
    CAELO_REQUIRE(m->order_tracked, "caelo_voxmap_export: the map was not filled by caelo_voxelize (no first-touch order)");
    int16_t *const outs[3] = {all0, all1, all2};
    return export_scales(m, 7, outs, capacity, counts, nullptr, caelo_stream(stream));
}

CAELO_API int caelo_voxmap_order(caelo_ctx *c, caelo_voxmap *m, int scale_mask, void *stream) {
    CAELO_REQUIRE(c && m, "null argument");
    CAELO_REQUIRE(m->order_tracked, "caelo_voxmap_order: the map was not filled by caelo_voxelize (no first-touch order)");
    hipStream_t s = caelo_stream(stream);
    int16_t *outs[3];
    int32_t *n_out = nullptr;
    const int rc = kd_begin_device_lists(m, outs, &n_out, s);
    if (rc != CAELO_OK) return rc;
    return export_scales(m, scale_mask & 7, outs, m->max_points, nullptr, n_out, s);
}
