// voxel.hip -- multi-resolution voxelization and voxel-patch gather.
//
// Reference behaviour restated here (never its code):
//   Voxel constants      Voxel.py:15-52
//   FilterOutTooFarPts   Voxel.py:89-97
//   Voxelization         Voxel.py:100-173   (Python per-point loop, 4 s/frame on the CPU)
//   GetPatchesList       Voxel.py:177-216   (3x sklearn kd-tree 496-NN + Python scatter)
//
// MI355X design: the occupancy of each scale lives in an open-addressing hash table of 8x8x8-voxel
// bricks; a brick is exactly one 64-byte line (8 u64 words: word = x&7, bit = (y&7)*8 + (z&7)).
// A 16^3 patch window touches <= 27 bricks, the 496-NN ball (radius sqrt(192) voxels) <= 125, so a
// wavefront stages them in LDS with <= 125 coalesced 64-byte reads and assembles the bit-packed patch
// (one u64 word per lane) with shifts -- no kd-tree, no 50 MB of dense f32 patches.  HBM-bound
// integer work: nothing here is reshaped into a GEMM.
#include "caelo_internal.h"

#define VOX_SIZE 0.02
#define BLOCK_REAL 1.28
#define VIS_L 99.84
#define VIS_W 99.84
#define VIS_H 14.72
#define INT_BIG 0x7F7F7F7F

// ------------------------------------------------------------------------------------------------
// map lifetime
// ------------------------------------------------------------------------------------------------
static uint32_t pow2ceil(uint64_t v) {
    uint32_t p = 1024;
    while (p < v) p <<= 1;
    return p;
}

CAELO_API int caelo_voxmap_create(caelo_ctx *c, int64_t max_points, caelo_voxmap **out) {
    CAELO_REQUIRE(c && out && max_points > 0, "bad argument");
    caelo_voxmap *m = new caelo_voxmap();
    memset(m, 0, sizeof(*m));
    m->max_points = max_points;
    // Open addressing wants load factors well under 1/2 (at 3/4 the longest probe chain in a wavefront
    // set the pace: 190 us for the insert kernel).  Scale 0 can hold one brick per point; scales 1 and 2
    // are sized for 1/4 and 1/16 of that (a LiDAR scan fills ~1/5 and ~1/30); a cloud that overflows
    // them reports CAELO_ST_MAP_FULL and the caller retries with a larger map.
    // >= 1.5 slots per point (distinct voxels <= points: load <= 0.67; 160 000 points -- a HDL-64E scan never exceeds ~131 k -- still
    // get the 2^18 slots of rounds 1-3, so the tables' cache footprint does not move)
    const size_t vslots = pow2ceil(((uint64_t)max_points * 3 + 1) / 2);
    const size_t bslots[3] = {vslots, vslots / 4, vslots / 16};
    size_t off = 0;
    size_t o_bkeys[3], o_vkeys[3], o_vfirst[3], o_bits[3];
    for (int s = 0; s < 3; ++s) { o_bkeys[s] = off; off += bslots[s] * 8; }
    m->ff_bytes_keys = off;
    o_vkeys[0] = off; off += vslots * 8;
    o_vfirst[0] = off; off += vslots * 4;
    m->ff_bytes_min = off;
    for (int s = 1; s < 3; ++s) { o_vkeys[s] = off; off += vslots * 8; }
    for (int s = 1; s < 3; ++s) { o_vfirst[s] = off; off += vslots * 4; }
    m->ff_bytes_all = off;
    m->zero_off = off;
    for (int s = 0; s < 3; ++s) { o_bits[s] = off; off += bslots[s] * 64; }
    const size_t o_counts = off;
    off += 64;
    m->zero_bytes = off - m->zero_off;
    const size_t o_list0 = off; off += bslots[0] * 4;
    const size_t o_list1 = off; off += bslots[1] * 4;
    // suspect-voxel tables of the fused build (one 0xFF region, wiped entry by entry between builds) + their list
    m->sp_off = off;
    const size_t o_spk = off; off += vslots * 8;
    const size_t o_sbk = off; off += vslots * 8;
    const size_t o_spf = off; off += vslots * 4;
    const size_t o_sbc = off; off += vslots * 4;
    m->sp_bytes = off - m->sp_off;
    const size_t o_splist = off; off += (size_t)max_points * 16;
    m->total_bytes = off;
    CAELO_HIP(hipMalloc(&m->base, m->total_bytes));
    CAELO_HIP(hipMemset(m->base + m->sp_off, 0xFF, m->sp_bytes));
    m->sp_keys = (unsigned long long *)(m->base + o_spk);
    m->sb_keys = (unsigned long long *)(m->base + o_sbk);
    m->sp_first = (uint32_t *)(m->base + o_spf);
    m->sb_cnt = (uint32_t *)(m->base + o_sbc);
    m->sp_list = (uint4 *)(m->base + o_splist);
    m->sp_mask = (uint32_t)(vslots - 1);
    for (int s = 0; s < 3; ++s) {
        m->brick[s].mask = (uint32_t)(bslots[s] - 1);
        m->brick[s].keys = (unsigned long long *)(m->base + o_bkeys[s]);
        m->brick[s].bits = (unsigned long long *)(m->base + o_bits[s]);
        m->vmask[s] = (uint32_t)(vslots - 1);
        m->vkeys[s] = (unsigned long long *)(m->base + o_vkeys[s]);
        m->vfirst[s] = (uint32_t *)(m->base + o_vfirst[s]);
    }
    m->counts = (int32_t *)(m->base + o_counts);
    m->list0 = (uint32_t *)(m->base + o_list0);
    m->list1 = (uint32_t *)(m->base + o_list1);
    *out = m;
    return CAELO_OK;
}

CAELO_API void caelo_voxmap_destroy(caelo_voxmap *m) {
    if (!m) return;
    kd_destroy(m);
    if (m->base) (void)hipFree(m->base);
    if (m->scratch) (void)hipFree(m->scratch);
    delete m;
}

void vox_clear_items(caelo_voxmap *m, int level, caelo_clear_list &list) {
    list.item[list.n++] = {m->base, level >= 2 ? m->ff_bytes_all : (level == 1 ? m->ff_bytes_min : m->ff_bytes_keys), 0xFFFFFFFFu};
    list.item[list.n++] = {m->base + m->zero_off, m->zero_bytes, 0u};
}

// wipe the bricks of the previous fused build: key -> empty, 8 payload words -> 0 (entry e, word w per thread)
void frame_dev_set_map(caelo_frame_dev &d, const caelo_voxmap *m) {
    for (int i = 0; i < 3; ++i) {
        d.brick[i] = m->brick[i];
        d.vkeys[i] = m->vkeys[i];
        d.vfirst[i] = m->vfirst[i];
    }
    d.vmask0 = m->vmask[0];
    d.vmask12 = m->vmask[1];
    d.counts = m->counts;
    d.list0 = m->list0;
    d.list1 = m->list1;
    d.sp = suspect_tables(m);
}

// blockIdx.z = frame; frames whose map is cleared whole this time (clear_mask bit) have nothing to do here
__global__ void __launch_bounds__(256) k_vox_clear_lists(const caelo_frame_set fs, unsigned int skip_mask) {
    if ((skip_mask >> blockIdx.z) & 1u) return;
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const caelo_brick_table b0 = F.brick[0], b1 = F.brick[1];
    const uint32_t *__restrict__ list0 = F.list0, *__restrict__ list1 = F.list1;
    const int32_t *__restrict__ counts = F.counts;
    const SuspectTables sp = F.sp;
    const int n0 = counts[4], n1 = counts[5], nsp = counts[6];
    const long long total = (long long)(n0 + n1) * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i >> 3), w = (int)(i & 7);
        const caelo_brick_table &t = e < n0 ? b0 : b1;
        const uint32_t slot = e < n0 ? list0[e] : list1[e - n0];
        if (slot == 0xFFFFFFFFu) continue;  // (a hole of list0: a run of points whose brick another run had created)
        t.bits[(size_t)slot * 8 + w] = 0ull;
        if (w == 0) t.keys[slot] = CAELO_EMPTY_KEY;
    }
    // the suspect-voxel tables of the previous build, through its list of inconsistent points
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nsp; e += gridDim.x * blockDim.x) {
        const uint4 ent = sp.list[e];
        sp.sp_keys[ent.y] = CAELO_EMPTY_KEY;
        sp.sp_first[ent.y] = 0xFFFFFFFFu;
        if (ent.z != 0xFFFFFFFFu) {
            sp.sb_keys[ent.z] = CAELO_EMPTY_KEY;
            sp.sb_cnt[ent.z] = 0xFFFFFFFFu;
        }
    }
}

int vox_clear_for_fast_build(caelo_voxmap *m, caelo_clear_list &list, hipStream_t s) {
    caelo_voxmap *maps[1] = {m};
    return vox_clear_for_fast_build_set(maps, 1, &list, s);
}

// maps[i] / lists[i]: frame i of a set.  One launch wipes the listed bricks of every map that holds nothing but its
// previous fused build; the others (first use, or an exact build in between) get their whole tables on the clear list.
int vox_clear_for_fast_build_set(caelo_voxmap *const *maps, int n, caelo_clear_list *lists, hipStream_t s) {
    caelo_frame_set fs = {};
    fs.n = n;
    unsigned int skip = 0;
    for (int i = 0; i < n; ++i) {
        caelo_voxmap *m = maps[i];
        frame_dev_set_map(fs.f[i], m);
        caelo_clear_list &list = lists[i];
        if (!m->lists_valid) {
            skip |= 1u << i;
            vox_clear_items(m, 0, list);
            list.item[list.n++] = {m->base + m->sp_off, m->sp_bytes, 0xFFFFFFFFu};  // the suspect-voxel tables, whole
            continue;
        }
        // scale 2 has no list: its 16 k-slot table is cleared whole (keys, then payload + the counters right behind it)
        const size_t slots2 = (size_t)m->brick[2].mask + 1;
        list.item[list.n++] = {m->brick[2].keys, slots2 * 8, 0xFFFFFFFFu};
        list.item[list.n++] = {m->brick[2].bits, slots2 * 64 + 64, 0u};
    }
    if (skip != (1u << n) - 1u) {
        k_vox_clear_lists<<<dim3(256, 1, n), 256, 0, s>>>(fs, skip);
        CAELO_LAUNCH_CHECK();
    }
    return CAELO_OK;
}

static int voxmap_clear(caelo_voxmap *m, bool track_order, hipStream_t s) {
    m->lists_valid = false;
    m->order_tracked = track_order;
    m->kd_lists = false;   // (whatever fills the map next: only caelo_voxmap_from_lists / caelo_voxmap_order know the reference's list order)
    caelo_clear_list list;
    list.n = 0;
    vox_clear_items(m, track_order ? 2 : 1, list);
    return caelo_clear_many(list, s);
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// find-or-insert a key; returns slot or -1 when the table is full
__device__ inline int table_insert(unsigned long long *keys, uint32_t mask, unsigned long long key) {
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

__device__ inline int table_find(const unsigned long long *keys, uint32_t mask, unsigned long long key) {
    uint32_t h = caelo_hash64(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const unsigned long long k = keys[h];
        if (k == key) return (int)h;
        if (k == CAELO_EMPTY_KEY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

// set the voxel's bit in its brick; returns 1 if the bit was newly set, 0 if present, -1 if full
__device__ inline int brick_set(caelo_brick_table t, int x, int y, int z) {
    const int slot = table_insert(t.keys, t.mask, caelo_pack3(x >> 3, y >> 3, z >> 3));
    if (slot < 0) return -1;
    const unsigned long long bit = 1ull << (((y & 7) << 3) | (z & 7));
    unsigned long long *w = &t.bits[(size_t)slot * 8 + (x & 7)];
    // Thousands of near-range points share one coarse voxel: test before the atomic so the word is
    // not hammered by read-modify-writes that change nothing (a stale 0 only costs one extra atomic).
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return 0;
    const unsigned long long old = atomicOr(w, bit);
    return (old & bit) ? 0 : 1;
}

// find-or-insert that also reports whether THIS call created the entry
__device__ inline int table_insert_new(unsigned long long *keys, uint32_t mask, unsigned long long key, bool *is_new) {
    uint32_t h = caelo_hash64(key) & mask;
    *is_new = false;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        unsigned long long k = keys[h];
        if (k == key) return (int)h;
        if (k == CAELO_EMPTY_KEY) {
            k = atomicCAS(&keys[h], CAELO_EMPTY_KEY, key);
            if (k == CAELO_EMPTY_KEY) { *is_new = true; return (int)h; }
            if (k == key) return (int)h;
        }
        h = (h + 1) & mask;
    }
    return -1;
}

// brick_mark that reports a newly created brick (the caller appends its slot to a compact list, one
// global atomic per workgroup, so the next pass runs on dense wavefronts)
__device__ inline bool brick_mark_new(caelo_brick_table t, int x, int y, int z, int *slot_out, bool *is_new) {
    const int slot = table_insert_new(t.keys, t.mask, caelo_pack3(x >> 3, y >> 3, z >> 3), is_new);
    *slot_out = slot;
    if (slot < 0) return false;
    const unsigned long long bit = 1ull << (((y & 7) << 3) | (z & 7));
    unsigned long long *w = &t.bits[(size_t)slot * 8 + (x & 7)];
    if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit))
        (void)__hip_atomic_fetch_or(w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// fire-and-forget variant: no returned value -> a non-returning L2 atomic, no round trip.
// returns false only when the table is full.
__device__ inline bool brick_mark(caelo_brick_table t, int x, int y, int z) {
    const int slot = table_insert(t.keys, t.mask, caelo_pack3(x >> 3, y >> 3, z >> 3));
    if (slot < 0) return false;
    const unsigned long long bit = 1ull << (((y & 7) << 3) | (z & 7));
    unsigned long long *w = &t.bits[(size_t)slot * 8 + (x & 7)];
    if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit))
        (void)__hip_atomic_fetch_or(w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

struct VoxIdx {
    int g[3];   // scale-0 global voxel index
    int v1[3];  // scale-1
    int v2[3];  // scale-2
    bool ok, oob, nonfinite;
};

// int(p / d) exactly as the reference's float64 division + truncation gives it (p >= 0, quotient < 1e4), without the division
// for all but a few points: t = p * (1 / d) is within 3e-12 of the true quotient, and so is the correctly rounded p / d, so
// whenever t sits further than 1e-6 from an integer both truncate alike; only a point within a micro-voxel of a voxel face (the
// points that make a metrically quantised scan interesting, 14 of 126 k) pays for v_div_f64's ~40 instructions.
__device__ inline int vox_trunc_div(double p, double d, double inv_d) {
    const double t = p * inv_d;
    const int n = (int)t;
    const double frac = t - (double)n;
    if (frac > 1e-6 && frac < 1.0 - 1e-6) return n;
    return (int)(p / d);
}

// Voxel.py:89-97,:118-152 for one point; f64 index math
__device__ inline VoxIdx voxel_indices(float fx, float fy, float fz) {
    VoxIdx r;
    r.ok = false;
    r.oob = false;
    r.nonfinite = fx != fx || fy != fy || fz != fz;   // NaN passes the filter below (abs(nan) > L is False) and int(nan) raises: Voxel.py:122
    if (r.nonfinite) return r;
    if (fabsf(fx) > (float)VIS_L || fabsf(fy) > (float)VIS_W || fabsf(fz) > (float)VIS_H) return r;  // :89-97
    const double p[3] = {(double)fx + VIS_L, (double)fy + VIS_W, (double)fz + VIS_H};                 // :118-120
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int b = vox_trunc_div(p[a], BLOCK_REAL, 1.0 / BLOCK_REAL);                        // :122-124
        const double rem = p[a] - b * BLOCK_REAL;                                               // :136-138 (may be -ulp: the division below)
        const int v = rem >= 0.0 ? vox_trunc_div(rem, VOX_SIZE, 1.0 / VOX_SIZE) : (int)(rem / VOX_SIZE);
        if (v < 0 || v >= 64) r.oob = true;
        r.g[a] = v + b * 64;                                                                    // :143
        r.v1[a] = vox_trunc_div(p[a], VOX_SIZE * 8, 1.0 / (VOX_SIZE * 8));                      // :147-149
        r.v2[a] = vox_trunc_div(p[a], VOX_SIZE * 32, 1.0 / (VOX_SIZE * 32));                    // :150-152
    }
    r.ok = !r.oob;
    return r;
}

// ------------------------------------------------------------------------------------------------
// K5: voxelization.  Pass 1 records, per scale-0 voxel, the smallest point index touching it; pass
// 2 lets exactly that point (the reference's first touch, Voxel.py:139-141) set the scale-0/1/2 bits
// -- a later duplicate never reaches layers 1/2 in the reference either (`continue` at :140).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vox_first(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const float *__restrict__ pc = F.pc;
    const int64_t n = F.n;
    const int stride = F.pc_stride;
    unsigned long long *vkeys = F.vkeys[0];
    uint32_t *vfirst = F.vfirst[0];
    const uint32_t vmask = F.vmask0;
    int32_t *status = F.status;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pc + i * stride;
    const VoxIdx v = voxel_indices(p[0], p[1], p[2]);
    if (v.oob) atomicOr(status, CAELO_ST_VOXEL_OOB);
    if (v.nonfinite) atomicOr(status, CAELO_ST_NONFINITE);
    if (!v.ok) return;
    const int slot = table_insert(vkeys, vmask, caelo_pack3(v.g[0], v.g[1], v.g[2]));
    if (slot < 0) { atomicOr(status, CAELO_ST_MAP_FULL); return; }
    atomicMin(&vfirst[slot], (uint32_t)i);
}

// one atomic per wavefront instead of one per lane (120k same-address atomics serialise in L2)
__device__ inline void wave_count(int32_t *counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (pred && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(counter, __popcll(m));
}

__global__ void __launch_bounds__(256) k_vox_insert(const caelo_frame_set fs, int track_order) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const float *__restrict__ pc = F.pc;
    const int64_t n = F.n;
    const int stride = F.pc_stride;
    const unsigned long long *__restrict__ vkeys0 = F.vkeys[0];
    const uint32_t *__restrict__ vfirst0 = F.vfirst[0];
    const uint32_t vmask0 = F.vmask0, vmask12 = F.vmask12;
    const caelo_brick_table b0 = F.brick[0], b1 = F.brick[1], b2 = F.brick[2];
    unsigned long long *vkeys1 = F.vkeys[1], *vkeys2 = F.vkeys[2];
    uint32_t *vfirst1 = F.vfirst[1], *vfirst2 = F.vfirst[2];
    int32_t *counts = F.counts, *status = F.status;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool first = false;
    VoxIdx v;
    v.ok = false;
    if (i < n) {
        const float *p = pc + i * stride;
        v = voxel_indices(p[0], p[1], p[2]);
        if (v.ok) {
            const int slot = table_find(vkeys0, vmask0, caelo_pack3(v.g[0], v.g[1], v.g[2]));
            first = slot >= 0 && vfirst0[slot] == (uint32_t)i;  // the reference's first touch
        }
    }
    int full = 0, r0 = 0, r1 = 0, r2 = 0;
    if (first) {
        r0 = brick_set(b0, v.g[0], v.g[1], v.g[2]);
        r1 = brick_set(b1, v.v1[0], v.v1[1], v.v1[2]);
        r2 = brick_set(b2, v.v2[0], v.v2[1], v.v2[2]);
        full = (r0 < 0) | (r1 < 0) | (r2 < 0);
        if (track_order) {
            const int s1 = table_insert(vkeys1, vmask12, caelo_pack3(v.v1[0], v.v1[1], v.v1[2]));
            if (s1 >= 0) atomicMin(&vfirst1[s1], (uint32_t)i); else full = 1;
            const int s2 = table_insert(vkeys2, vmask12, caelo_pack3(v.v2[0], v.v2[1], v.v2[2]));
            if (s2 >= 0) atomicMin(&vfirst2[s2], (uint32_t)i); else full = 1;
        }
    }
    wave_count(&counts[0], first && r0 >= 0);
    wave_count(&counts[1], first && r1 > 0);
    wave_count(&counts[2], first && r2 > 0);
    if (full) atomicOr(status, CAELO_ST_MAP_FULL);
}

__global__ void k_or_status(int32_t *status, int32_t bit) { atomicOr(status, bit); }

__global__ void k_vox_check(const int32_t *counts, int32_t *status) {
    if (counts[0] < 496 || counts[1] < 496 || counts[2] < 496) atomicOr(status, CAELO_ST_FEW_VOXELS);
}

// ------------------------------------------------------------------------------------------------
// K5 fast path (fused extract): one pass over the points + one pass over the scale-0 bricks.
// A scale-0 brick (8 x 0.02 m) IS a scale-1 voxel (0.16 m) and 4 of those a scale-2 voxel
// (Voxel.py:15-31), so scales 1/2 follow from the set of scale-0 bricks -- for every scale-0 voxel whose
// FIRST point (the only one that reaches layers 1/2, Voxel.py:139-158) has its own int(x_/0.16), int(x_/0.64)
// (Voxel.py:147-152) equal to its scale-0 index >> 3, >> 5.  A point with x_ within an ulp of a voxel face
// breaks that (x = 4.0 exactly; a few of every metrically quantised scan: 14 of 126 k points at mm resolution).
// Such points are handled exactly, without leaving the one-pass scheme:
//   * k_vox_points enters the voxel of every inconsistent point into a small "suspect" table (and counts the
//     suspect voxels of each brick);
//   * k_vox_coarse derives a scale-1 voxel from a brick only if the brick holds a NON-suspect voxel (all points of
//     such a voxel are consistent, so its first one is), k_vox_coarse2 derives scale 2 from those;
//   * k_vox_suspects_first / _resolve (empty launches when the frame has no inconsistent point) find the first point
//     of every suspect voxel (atomicMin over all points) and insert that point's own scale-1 / 2 indices -- exactly
//     what the reference's loop does when it meets the voxel for the first time.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vox_points(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const float *__restrict__ pc = F.pc;
    const int64_t n = F.n;
    const int stride = F.pc_stride;
    const caelo_brick_table b0 = F.brick[0];
    uint32_t *list0 = F.list0;
    int32_t *counts = F.counts, *status = F.status;
    const SuspectTables sp = F.sp;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    VoxIdx v;
    v.ok = false;
    v.oob = false;
    v.nonfinite = false;
    if (i < n) {
        const float *p = pc + i * stride;
        v = voxel_indices(p[0], p[1], p[2]);
    }
    int st = (v.oob ? CAELO_ST_VOXEL_OOB : 0) | (v.nonfinite ? CAELO_ST_NONFINITE : 0);
    unsigned long long key = CAELO_EMPTY_KEY;
    // every point's scale-0 voxel, for k_vox_suspects_first (the fused build leaves the exact build's first-touch key table
    // free; it holds >= 1.5 entries per point, checked against the point count by the caller: indexed by the point here)
    if (i < n) F.vkeys[0][i] = v.ok ? caelo_pack3(v.g[0], v.g[1], v.g[2]) : CAELO_EMPTY_KEY;
    if (v.ok) {
        bool consistent = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) consistent &= (v.v1[a] == (v.g[a] >> 3)) && (v.v2[a] == (v.g[a] >> 5));
        key = caelo_pack3(v.g[0] >> 3, v.g[1] >> 3, v.g[2] >> 3);
        if (!consistent) {  // rare: its voxel becomes a suspect (tables hold >= 1.5 slots per point and at most one entry per point: load <= 0.67, never full)
            bool sp_new = false;
            const int ss = table_insert_new(sp.sp_keys, sp.mask, caelo_pack3(v.g[0], v.g[1], v.g[2]), &sp_new);
            uint32_t sb = 0xFFFFFFFFu;
            if (sp_new) {
                sb = (uint32_t)table_insert(sp.sb_keys, sp.mask, key);
                atomicAdd(&sp.sb_cnt[sb], 1u);  // 0xFFFFFFFF + 1 = 0: the entry holds (suspect voxels of the brick) - 1
            }
            sp.list[atomicAdd(&counts[6], 1)] = make_uint4((uint32_t)i, (uint32_t)ss, sb, 0u);
        }
    }
    // Neighbouring points of a scan line fall into the same 16 cm brick: only the first lane of each run of
    // equal keys walks the hash table (memory-side atomics cost microseconds), the run reuses its slot.
    // The chain of dependent memory round trips is what this kernel costs (86 % of its wave cycles wait), so it is kept to two:
    // the scan's point, then ONE compare-and-swap on the key's home slot (no look first: most heads create their brick) with the
    // workgroup's list reservation in flight beside it -- a slot for every head, reserved before anyone knows which heads are
    // new (the others leave holes, 0xFFFFFFFF, which the two readers of the list skip); the bit goes out as an atomic OR
    // nobody waits for.  (Round 2: load key -> CAS -> reserve -> load word -> OR, 44 us per 8 frames.)
    const unsigned long long prev = __shfl_up(key, 1);
    const bool head = v.ok && (lane == 0 || prev != key);
    const uint32_t h0 = caelo_hash64(key) & b0.mask;
    unsigned long long old = CAELO_EMPTY_KEY;
    if (head) old = atomicCAS(&b0.keys[h0], CAELO_EMPTY_KEY, key);
    __shared__ int s_tmp[2];
    const int lpos = caelo_block_reserve_async(&counts[4], head, s_tmp);  // ONE global atomic per workgroup
    int slot = -1;
    if (head) {
        bool is_new = old == CAELO_EMPTY_KEY;
        slot = (int)h0;
        if (!is_new && old != key) slot = table_insert_new(b0.keys, b0.mask, key, &is_new);  // home slot taken: probe on
        if (slot < 0) st |= CAELO_ST_MAP_FULL;
        list0[lpos] = is_new ? (uint32_t)slot : 0xFFFFFFFFu;
    }
    // The bits.  A memory-side atomic moves a 64-byte line there and back whatever it changes (one OR per point: 76 MB of
    // traffic for 16 MB of points, and the kernel's time): the wave first ORs its points into an LDS image of the bricks it
    // touches -- row = run of equal keys, 8 words each -- and then issues one atomic per non-empty word.
    __shared__ unsigned long long s_agg[4][64][8];
    __shared__ int s_slot[4][64];
    const int wave = threadIdx.x >> 6;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));  // heads at or below this lane
    const int run = __popcll(upto) - 1;                                                        // (-1: no head yet -> not ok either)
    const int nruns = __popcll(heads);
#pragma unroll
    for (int w = 0; w < 8; w += 2) *(ulonglong2 *)&s_agg[wave][lane][w] = make_ulonglong2(0ull, 0ull);
    if (head) s_slot[wave][run] = slot;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (v.ok) atomicOr(&s_agg[wave][run][v.g[0] & 7], 1ull << (((v.g[1] & 7) << 3) | (v.g[2] & 7)));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {   // lane (r, w) = (lane >> 3, lane & 7) flushes word w of runs r, r + 8, ...
        const int w = lane & 7;
        for (int r = lane >> 3; r < nruns; r += 8) {
            const unsigned long long bitsw = s_agg[wave][r][w];
            const int sl = s_slot[wave][r];
            if (bitsw && sl >= 0)
                (void)__hip_atomic_fetch_or(&b0.bits[(size_t)sl * 8 + w], bitsw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (st) atomicOr(status, st);
}

// The suspect voxels of the frame (see the header of this section) take two steps that have nothing to do -- one word read
// per workgroup -- when no point of the frame was inconsistent.  (A single kernel whose last workgroup resolves would need an
// agent-scope release per workgroup: on gfx950 that is an L2 write-back walk, 433 us for 8 frames.)
// (1) first point of every suspect voxel: smallest index over ALL points of the voxel, consistent ones included.  Runs in extra
//     workgroups of k_vox_coarse's launch (both only need k_vox_points done; both are chains of dependent memory round trips
//     that leave the CUs idle: side by side they cost the longer of the two, 32 us, instead of 32 + 14)
#define SUSPECT_LDS 128
__device__ inline void vox_suspects_first(const caelo_frame_dev &F, int block) {
    const int nsp = F.counts[6];
    if (nsp == 0) return;
    // A handful of voxels (14 of a 126 k-point scan quantised to 1 mm): the workgroup stages their keys in LDS and every point
    // compares the key k_vox_points stored for it against them -- no index arithmetic, no table probe per point (both together
    // were 15.7 us per 8 frames for those 14 voxels).  More inconsistent points than the LDS list holds: probe the table.
    __shared__ unsigned long long s_key[SUSPECT_LDS];
    __shared__ uint32_t s_slot[SUSPECT_LDS];
    const bool listed = nsp <= SUSPECT_LDS;
    if (listed) {
        if ((int)threadIdx.x < nsp) {
            const uint32_t ss = F.sp.list[threadIdx.x].y;
            s_slot[threadIdx.x] = ss;
            s_key[threadIdx.x] = F.sp.sp_keys[ss];
        }
        __syncthreads();
    }
    const int64_t i = (int64_t)block * blockDim.x + threadIdx.x;
    if (i >= F.n) return;
    const unsigned long long key = F.vkeys[0][i];
    if (key == CAELO_EMPTY_KEY) return;
    if (listed) {
        for (int e = 0; e < nsp; ++e)  // (several inconsistent points of one voxel repeat its key: the same atomicMin twice)
            if (s_key[e] == key) { atomicMin(&F.sp.sp_first[s_slot[e]], (uint32_t)i); break; }
    } else {
        const int ss = table_find(F.sp.sp_keys, F.sp.mask, key);
        if (ss >= 0) atomicMin(&F.sp.sp_first[ss], (uint32_t)i);
    }
}

// the occupied scale-0 bricks (256 per workgroup iteration): a brick is a scale-1 voxel, and counts its voxels
#define COARSE_WGS 256
__global__ void __launch_bounds__(256) k_vox_coarse(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    if (blockIdx.x >= COARSE_WGS) {  // (workgroup-uniform)
        vox_suspects_first(F, (int)blockIdx.x - COARSE_WGS);
        return;
    }
    const caelo_brick_table b0 = F.brick[0], b1 = F.brick[1];
    const uint32_t *list0 = F.list0;
    uint32_t *list1 = F.list1;
    int32_t *counts = F.counts, *status = F.status;
    const SuspectTables sp = F.sp;
    __shared__ int s_tmp[2];
    const int nb = counts[4];
    const bool any_suspect = counts[6] > 0;
    int pop = 0;
    for (int i0 = blockIdx.x * blockDim.x; i0 < nb; i0 += COARSE_WGS * blockDim.x) {  // uniform trip count per workgroup
        const int i = i0 + threadIdx.x;
        bool is_new = false;
        int slot1 = -1;
        const uint32_t slot = i < nb ? list0[i] : 0xFFFFFFFFu;
        if (slot != 0xFFFFFFFFu) {  // (list0 has holes, see k_vox_points)
            const unsigned long long k = b0.keys[slot];
            const ulonglong2 *w = (const ulonglong2 *)(b0.bits + (size_t)slot * 8);
            int mine = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const ulonglong2 u = w[q]; mine += __popcll(u.x) + __popcll(u.y); }
            pop += mine;
            if (any_suspect) {  // voxels whose first point may disagree with the brick are left to k_vox_suspects
                const int sb = table_find(sp.sb_keys, sp.mask, k);
                if (sb >= 0) mine -= (int)(sp.sb_cnt[sb] + 1u);
            }
            const int x = (int)((k >> 40) & 0xFFFFF), y = (int)((k >> 20) & 0xFFFFF), z = (int)(k & 0xFFFFF);
            if (mine > 0 && !brick_mark_new(b1, x, y, z, &slot1, &is_new)) atomicOr(status, CAELO_ST_MAP_FULL);
        }
        const int lpos = caelo_block_reserve(&counts[5], is_new, s_tmp);
        if (is_new) list1[lpos] = (uint32_t)slot1;
    }
    caelo_block_add(&counts[0], pop, s_tmp);
}

// the occupied scale-1 bricks: count their voxels and mark the scale-2 voxels under them
// (a scale-1 brick spans 2x2x2 scale-2 voxels: scale-2 index = scale-1 index >> 2)
__global__ void __launch_bounds__(256) k_vox_coarse2(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const caelo_brick_table b1 = F.brick[1], b2 = F.brick[2];
    const uint32_t *list1 = F.list1;
    int32_t *counts = F.counts, *status = F.status;
    __shared__ int s_tmp[2];
    const int nb = counts[5];
    int pop = 0, pop2 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const uint32_t slot = list1[i];
        const unsigned long long k = b1.keys[slot];
        const int bx = (int)((k >> 40) & 0xFFFFF) << 3, by = (int)((k >> 20) & 0xFFFFF) << 3, bz = (int)(k & 0xFFFFF) << 3;
        const unsigned long long *w = b1.bits + (size_t)slot * 8;
        // The 2x2x2 scale-2 voxels under this brick all live in ONE scale-2 brick (index >> 2): one table
        // insert, then one OR per x-half with the (hy, hz) bits gathered -- not eight insert+OR chains of
        // memory-side atomics.  Octant (hx, hy, hz): words x in [4hx, 4hx+4), bits y in [4hy, ..), z in [4hz, ..)
        unsigned long long orv[2] = {0ull, 0ull};
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            const unsigned long long m = w[4 * hx] | w[4 * hx + 1] | w[4 * hx + 2] | w[4 * hx + 3];
            pop += __popcll(w[4 * hx]) + __popcll(w[4 * hx + 1]) + __popcll(w[4 * hx + 2]) + __popcll(w[4 * hx + 3]);
#pragma unroll
            for (int hy = 0; hy < 2; ++hy)
#pragma unroll
                for (int hz = 0; hz < 2; ++hz) {
                    const unsigned long long sel = (0x0F0F0F0Full << (4 * hz)) << (32 * hy);
                    const int y2 = (by >> 2) + hy, z2 = (bz >> 2) + hz;
                    if (m & sel) orv[hx] |= 1ull << (((y2 & 7) << 3) | (z2 & 7));
                }
        }
        if (orv[0] | orv[1]) {
            const int x2 = bx >> 2;  // even: x2 and x2 + 1 share the scale-2 brick
            const int slot2 = table_insert(b2.keys, b2.mask, caelo_pack3(x2 >> 3, by >> 5, bz >> 5));
            if (slot2 < 0) atomicOr(status, CAELO_ST_MAP_FULL);
            else {
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) {
                    if (!orv[hx]) continue;
                    unsigned long long *wd = &b2.bits[(size_t)slot2 * 8 + ((x2 + hx) & 7)];
                    if ((__hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & orv[hx]) != orv[hx]) {
                        // every scale-2 voxel is counted by the one atomic that sets its bit (no separate counting pass)
                        const unsigned long long old = __hip_atomic_fetch_or(wd, orv[hx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pop2 += __popcll(orv[hx] & ~old);
                    }
                }
            }
        }
    }
    caelo_block_add(&counts[1], pop, s_tmp);
    __syncthreads();  // s_tmp is reused
    caelo_block_add(&counts[2], pop2, s_tmp);
}

// (2) that point's own scale-1 / 2 indices are inserted, exactly what the reference's loop does when it meets the voxel
__global__ void __launch_bounds__(256) k_vox_suspects_resolve(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const float *__restrict__ pc = F.pc;
    const int stride = F.pc_stride;
    const caelo_brick_table b1 = F.brick[1], b2 = F.brick[2];
    uint32_t *list1 = F.list1;
    int32_t *counts = F.counts, *status = F.status;
    const SuspectTables sp = F.sp;
    const int nsp = counts[6];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nsp; e += gridDim.x * blockDim.x) {
        const uint4 ent = sp.list[e];
        const uint32_t j = sp.sp_first[ent.y];
        const float *p = pc + (int64_t)j * stride;
        const VoxIdx v = voxel_indices(p[0], p[1], p[2]);  // the voxel's first point: its own scale-1 / 2 indices count
        // several inconsistent points of one voxel repeat this: idempotent, the counters only see bits that were new
        bool is_new = false;
        const int s1 = table_insert_new(b1.keys, b1.mask, caelo_pack3(v.v1[0] >> 3, v.v1[1] >> 3, v.v1[2] >> 3), &is_new);
        if (s1 < 0) atomicOr(status, CAELO_ST_MAP_FULL);
        else {
            if (is_new) list1[atomicAdd(&counts[5], 1)] = (uint32_t)s1;
            const unsigned long long bit = 1ull << (((v.v1[1] & 7) << 3) | (v.v1[2] & 7));
            if (!(atomicOr(&b1.bits[(size_t)s1 * 8 + (v.v1[0] & 7)], bit) & bit)) atomicAdd(&counts[1], 1);
        }
        const int r2 = brick_set(b2, v.v2[0], v.v2[1], v.v2[2]);
        if (r2 < 0) atomicOr(status, CAELO_ST_MAP_FULL);
        else if (r2) atomicAdd(&counts[2], 1);
    }
}

static int64_t vox_set_max_points(const caelo_frame_set &fs) {
    int64_t n = 0;
    for (int i = 0; i < fs.n; ++i) n = fs.f[i].n > n ? fs.f[i].n : n;
    return n;
}

static void vox_single_set(caelo_frame_set &fs, const caelo_voxmap *m, const float *pc, int64_t n, int stride, int32_t *status) {
    fs.n = 1;
    frame_dev_set_map(fs.f[0], m);
    fs.f[0].pc = pc; fs.f[0].n = n; fs.f[0].pc_stride = stride; fs.f[0].status = status;
}

int vox_build_fast_launch(caelo_voxmap *m, const float *pc, int64_t n, int stride, int32_t *status, hipStream_t s) {
    caelo_frame_set fs = {};
    vox_single_set(fs, m, pc, n, stride, status);
    caelo_voxmap *maps[1] = {m};
    return vox_build_fast_set(maps, fs, s);
}

// fused build of every frame of the set (fs.f[i] carries map i's tables, the scan and the status word)
int vox_build_fast_set(caelo_voxmap *const *maps, const caelo_frame_set &fs, hipStream_t s) {
    for (int i = 0; i < fs.n; ++i) { maps[i]->lists_valid = false; maps[i]->kd_lists = false; maps[i]->order_tracked = false; }  // until every kernel of the build is enqueued
    const unsigned gp = (unsigned)((vox_set_max_points(fs) + 255) / 256);
    k_vox_points<<<dim3(gp, 1, fs.n), 256, 0, s>>>(fs);
    CAELO_LAUNCH_CHECK();
    k_vox_coarse<<<dim3(COARSE_WGS + gp, 1, fs.n), 256, 0, s>>>(fs);  // + the suspect voxels' first points
    CAELO_LAUNCH_CHECK();
    k_vox_coarse2<<<dim3(64, 1, fs.n), 256, 0, s>>>(fs);
    CAELO_LAUNCH_CHECK();
    k_vox_suspects_resolve<<<dim3(4, 1, fs.n), 256, 0, s>>>(fs);
    CAELO_LAUNCH_CHECK();
    for (int i = 0; i < fs.n; ++i) maps[i]->lists_valid = true;
    return CAELO_OK;
}

int vox_build_launch(caelo_voxmap *m, const float *pc, int64_t n, int stride, bool track_order, int32_t *status,
                     hipStream_t s) {
    caelo_frame_set fs = {};
    vox_single_set(fs, m, pc, n, stride, status);
    caelo_voxmap *maps[1] = {m};
    return vox_build_set(maps, fs, track_order, s);
}

int vox_build_set(caelo_voxmap *const *maps, const caelo_frame_set &fs, bool track_order, hipStream_t s) {
    for (int i = 0; i < fs.n; ++i) { maps[i]->lists_valid = false; maps[i]->kd_lists = false; maps[i]->order_tracked = track_order; }
    const unsigned grid = (unsigned)((vox_set_max_points(fs) + 255) / 256);
    k_vox_first<<<dim3(grid, 1, fs.n), 256, 0, s>>>(fs);
    CAELO_LAUNCH_CHECK();
    k_vox_insert<<<dim3(grid, 1, fs.n), 256, 0, s>>>(fs, track_order ? 1 : 0);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_voxelize(caelo_ctx *c, caelo_voxmap *m, const float *pc, int64_t n, int stride, int32_t *status,
                             void *stream) {
    CAELO_REQUIRE(c && m && pc && status, "null argument");
    CAELO_REQUIRE(stride >= 3, "points need >= 3 columns");
    if (n > m->max_points) {
        caelo_set_error("caelo_voxelize: %lld points exceed the map capacity %lld", (long long)n, (long long)m->max_points);
        return CAELO_ERR_CAPACITY;
    }
    hipStream_t s = caelo_stream(stream);
    int rc = voxmap_clear(m, true, s);
    if (rc) return rc;
    rc = vox_build_launch(m, pc, n, stride, true, status, s);
    if (rc) return rc;
    k_vox_check<<<1, 1, 0, s>>>(m->counts, status);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_voxelize_fast(caelo_ctx *c, caelo_voxmap *m, const float *pc, int64_t n, int stride, int32_t *status,
                                  void *stream) {
    CAELO_REQUIRE(c && m && pc && status, "null argument");
    CAELO_REQUIRE(stride >= 3, "points need >= 3 columns");
    if (n > m->max_points) {
        caelo_set_error("caelo_voxelize_fast: %lld points exceed the map capacity %lld", (long long)n, (long long)m->max_points);
        return CAELO_ERR_CAPACITY;
    }
    hipStream_t s = caelo_stream(stream);
    caelo_clear_list list;
    list.n = 0;
    int rc = vox_clear_for_fast_build(m, list, s);
    if (rc) return rc;
    if ((rc = caelo_clear_many(list, s))) return rc;
    if ((rc = vox_build_fast_launch(m, pc, n, stride, status, s))) return rc;
    k_vox_check<<<1, 1, 0, s>>>(m->counts, status);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

__global__ void __launch_bounds__(256) k_vox_dump(caelo_brick_table t, unsigned long long *keys, unsigned long long *bits,
                                                  int64_t capacity, int32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > t.mask) return;
    const unsigned long long k = t.keys[i];
    if (k == CAELO_EMPTY_KEY) return;
    const int p = atomicAdd(count, 1);
    if (p >= capacity) return;
    keys[p] = k;
    for (int w = 0; w < 8; ++w) bits[(size_t)p * 8 + w] = t.bits[(size_t)i * 8 + w];
}

CAELO_API int caelo_voxmap_dump(caelo_ctx *c, const caelo_voxmap *m, int scale, uint64_t *keys, uint64_t *bits, int64_t capacity,
                                int32_t *count, void *stream) {
    CAELO_REQUIRE(c && m && keys && bits && count && capacity > 0, "bad argument");
    CAELO_REQUIRE(scale >= 0 && scale < 3, "scale must be 0, 1 or 2");
    hipStream_t s = caelo_stream(stream);
    CAELO_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    const caelo_brick_table t = m->brick[scale];
    k_vox_dump<<<(t.mask + 256) / 256, 256, 0, s>>>(t, (unsigned long long *)keys, (unsigned long long *)bits, capacity, count);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// map from reference-format voxel lists (GetPatchesList called with AllVoxels0/1/2 arrays)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vox_from_list(const int16_t *__restrict__ vox, int64_t n, caelo_brick_table b,
                                                       int32_t *count, int32_t *status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = brick_set(b, vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]);
    if (r < 0) atomicOr(status, CAELO_ST_MAP_FULL);
    else if (r) atomicAdd(count, 1);
}

CAELO_API int caelo_voxmap_from_lists(caelo_ctx *c, caelo_voxmap *m, const int16_t *a0, int64_t n0, const int16_t *a1,
                                      int64_t n1, const int16_t *a2, int64_t n2, int32_t *status, void *stream) {
    CAELO_REQUIRE(c && m && a0 && a1 && a2 && status, "null argument");
    if (n0 > m->max_points || n1 > m->max_points || n2 > m->max_points) {
        caelo_set_error("caelo_voxmap_from_lists: list longer than the map capacity %lld", (long long)m->max_points);
        return CAELO_ERR_CAPACITY;
    }
    hipStream_t s = caelo_stream(stream);
    int rc = voxmap_clear(m, false, s);
    if (rc) return rc;
    const int16_t *lists[3] = {a0, a1, a2};
    const int64_t ns[3] = {n0, n1, n2};
    for (int i = 0; i < 3; ++i) {
        if (ns[i] == 0) continue;
        k_vox_from_list<<<(unsigned)((ns[i] + 255) / 256), 256, 0, s>>>(lists[i], ns[i], m->brick[i], m->counts + i, status);
        CAELO_LAUNCH_CHECK();
    }
    // list lengths (n_samples), not unique counts, decide sklearn's ValueError (Voxel.py:195-196)
    if (n0 < 496 || n1 < 496 || n2 < 496) {
        k_or_status<<<1, 1, 0, s>>>(status, CAELO_ST_FEW_VOXELS);
        CAELO_LAUNCH_CHECK();
    }
    // the lists in the caller's order: what scikit-learn's kd-tree is built on when the 496-nearest cut splits a tie class (kdorder.hip)
    return kd_store_lists(m, lists, ns, s);
}

// ------------------------------------------------------------------------------------------------
// K6: patch gather.  One wavefront per (keypoint, scale).
// ------------------------------------------------------------------------------------------------
#define PW_WAVES 4
#ifndef PW_OCC
#define PW_OCC 6       // waves per SIMD the register allocation must allow (70 registers: 7 fit; 8 would spill)
#endif
#define BALL_R 13      // |d| <= 13 per axis covers every voxel with d2 <= 192
#define BALL_D2 192    // farthest in-window offset (-8,-8,-8)
#define NN_CAP 496     // Voxel.py:182
#define CLASS_CAP 192   // max #lattice points on a sphere x^2+y^2+z^2 = n, n <= 192

// debug aid: timestamps (100 MHz) of workgroup 0 / wave 0 of the last k_patches launch
__device__ unsigned long long g_patch_stamp[8];
#define PATCH_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_patch_stamp[i] = wall_clock64(); } while (0)
int patch_debug_copy(unsigned long long *out_host) {
    CAELO_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_patch_stamp), sizeof(unsigned long long) * 8));
    return CAELO_OK;
}

// Per wavefront: only the 3x3x3 bricks the 16^3 window can touch are staged in LDS (1.7 KB); the other
// bricks of the 5x5x5 ball cube stay in the registers of the lanes that fetched them, a quarter brick per
// lane (they only matter for the 496-NN count).  Small LDS footprint = many resident wavefronts: the kernel is latency bound
// (two dependent random accesses into tables that live at the memory side after the atomic build).
struct PatchWaveLds {
    // The window bricks are read once (into each lane's output word) BEFORE the rare 496-nearest path starts: its histogram and the
    // members of its cut class live in the same bytes (round 4: 4.5 -> 2.8 KB per wavefront -- beside the encoder's persistent
    // workgroups the LDS that is left decides how many of this kernel's workgroups a CU takes).
    union {
        unsigned long long win[27 * 8];
        struct {
            unsigned long long cls[CLASS_CAP];
            unsigned int hist[BALL_D2 + 1];
        };
    };
    unsigned int found[128];  // (table slot << 7 | ball-cube brick index) of the bricks that exist
    int ncls, cut, room;
};

__device__ inline int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(64 * PW_WAVES, PW_OCC) k_patches(const caelo_frame_set fs, int64_t k_max, int check_counts,
                                                           unsigned long long dd_mask) {
    unsigned fz, bx, per_frame_wgs;   // frame <-> XCD (caelo_frame_block)
    caelo_frame_block(fs.n, fz, bx, per_frame_wgs);
    const caelo_frame_dev &F = fs.f[fz];
    const float *__restrict__ pts = F.key_pts;
    const int pts_ld = F.kp_ld;
    const int32_t *__restrict__ n_key = F.n_key;
    const caelo_brick_table t0 = F.brick[0], t1 = F.brick[1], t2 = F.brick[2];
    unsigned long long *__restrict__ bits = F.bits;
    uint8_t *__restrict__ flags = F.flags;
    const int32_t *counts = check_counts ? F.counts : nullptr;
    int32_t *status = F.status;
    DedupScratch *dd = F.dd;
    if (counts && bx == 0 && threadIdx.x == 0 && (counts[0] < 496 || counts[1] < 496 || counts[2] < 496))
        atomicOr(status, CAELO_ST_FEW_VOXELS);  // sklearn ValueError at Voxel.py:195-196
    __shared__ PatchWaveLds lds_all[PW_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    PATCH_STAMP(0);
    PatchWaveLds &L = lds_all[wave];
    const int64_t pw = (int64_t)bx * PW_WAVES + wave;
    if (pw >= k_max * 3) return;
    const int64_t kp = pw / 3;
    const int scale = (int)(pw % 3);
    unsigned long long *out = bits + pw * 64;
    const int K = n_key ? *n_key : (int)k_max;
    if (kp >= K) {
        out[lane] = 0ull;
        if (lane == 0) flags[pw] = 0;
        if (dd) caelo_dedup_insert(0ull, lane, (int)pw, (int)(fz * CAELO_FRAME_PATCHES + pw), dd, fs.f[0].dd, dedup_slot_mask(fs.n), dd_mask);
        return;
    }
    const caelo_brick_table tab = scale == 0 ? t0 : (scale == 1 ? t1 : t2);
    const double vs = scale == 0 ? VOX_SIZE : (scale == 1 ? VOX_SIZE * 8 : VOX_SIZE * 32);  // Voxel.py:31
    const double ivs = scale == 0 ? 1.0 / VOX_SIZE : (scale == 1 ? 1.0 / (VOX_SIZE * 8) : 1.0 / (VOX_SIZE * 32));
    // Voxel.py:185,:193  KeyVoxels = int32((Pts + Visible*) / VoxelSizes[s])  (f64; the division only for a point within a
    // micro-voxel of a voxel face, vox_trunc_div)
    const int kx = vox_trunc_div((double)pts[(size_t)pts_ld * kp] + VIS_L, vs, ivs);
    const int ky = vox_trunc_div((double)pts[(size_t)pts_ld * kp + 1] + VIS_W, vs, ivs);
    const int kz = vox_trunc_div((double)pts[(size_t)pts_ld * kp + 2] + VIS_H, vs, ivs);
    const int bx0 = (kx - BALL_R) >> 3, by0 = (ky - BALL_R) >> 3, bz0 = (kz - BALL_R) >> 3;
    const int nbx = ((kx + BALL_R) >> 3) - bx0 + 1, nby = ((ky + BALL_R) >> 3) - by0 + 1, nbz = ((kz + BALL_R) >> 3) - bz0 + 1;
    // window bricks: first brick of the window per axis, relative to the ball cube (0 or 1)
    const int wx0 = ((kx - 8) >> 3) - bx0, wy0 = ((ky - 8) >> 3) - by0, wz0 = ((kz - 8) >> 3) - bz0;
    // ---- fetch <= 125 bricks in two dependent round trips.
    // (1) lane l probes the keys of bricks l and l + 64 (both in flight; the tables are <= 10 % full);
    // (2) the bricks that exist are compacted, and each 64-byte payload is fetched by FOUR lanes (16 B each):
    //     a wave-wide load touches 16 cache lines instead of 64, and absent bricks cost nothing -- the kernel
    //     is bound by the number of line requests the vector L1 can keep in flight, not by bytes.
    int nfound;
    for (int i = lane; i < 27 * 8; i += 64) L.win[i] = 0ull;  // absent window bricks read as empty
    {
        unsigned long long key[2];
        uint32_t h[2];
        bool want[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int l = lane + 64 * u;
            const int bix = l / 25, biy = (l / 5) % 5, biz = l % 5;
            want[u] = l < 125 && bix < nbx && biy < nby && biz < nbz && bx0 + bix >= 0 && by0 + biy >= 0 && bz0 + biz >= 0;
            key[u] = caelo_pack3(bx0 + bix, by0 + biy, bz0 + biz);
            h[u] = caelo_hash64(key[u]) & tab.mask;
        }
        const unsigned long long k0v = want[0] ? tab.keys[h[0]] : CAELO_EMPTY_KEY;
        const unsigned long long k1v = want[1] ? tab.keys[h[1]] : CAELO_EMPTY_KEY;
        int slot[2];
        slot[0] = !want[0] || k0v == CAELO_EMPTY_KEY ? -1 : (k0v == key[0] ? (int)h[0] : -2);
        slot[1] = !want[1] || k1v == CAELO_EMPTY_KEY ? -1 : (k1v == key[1] ? (int)h[1] : -2);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (slot[u] == -2) slot[u] = table_find(tab.keys, tab.mask, key[u]);  // collided: probe on
        const unsigned long long b0 = __ballot(slot[0] >= 0), b1 = __ballot(slot[1] >= 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int n0 = __popcll(b0);
        nfound = n0 + __popcll(b1);
        if (slot[0] >= 0) L.found[__popcll(b0 & below)] = ((unsigned)slot[0] << 7) | (unsigned)lane;
        if (slot[1] >= 0) L.found[n0 + __popcll(b1 & below)] = ((unsigned)slot[1] << 7) | (unsigned)(lane + 64);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // lane owns quarter `part` (x planes 2 part, 2 part + 1) of found brick it * 16 + (lane >> 2), it = 0..7
    const int part = lane & 3;
    // (the payloads are not kept: the rare wave that needs the 496-NN cut below fetches them again -- from L2 by then -- and
    //  every other wave runs in 64 registers, eight to a SIMD instead of four: this kernel is bound by memory round trips)
    int pop = 0;
    {
        ulonglong2 pay[8];
        int bl[8];  // ball-cube brick index of that brick, -1 = none
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = it * 16 + (lane >> 2);
            const bool ok = e < nfound;
            const unsigned ent = ok ? L.found[e] : 0u;
            bl[it] = ok ? (int)(ent & 127u) : -1;
            pay[it] = make_ulonglong2(0ull, 0ull);
            if (it * 16 < nfound && ok) pay[it] = ((const ulonglong2 *)(tab.bits + (size_t)(ent >> 7) * 8))[part];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            pop += __popcll(pay[it].x) + __popcll(pay[it].y);
            if (bl[it] >= 0) {
                const int wx = bl[it] / 25 - wx0, wy = (bl[it] / 5) % 5 - wy0, wz = bl[it] % 5 - wz0;
                if (wx >= 0 && wx < 3 && wy >= 0 && wy < 3 && wz >= 0 && wz < 3) {
                    L.win[((wx * 3 + wy) * 3 + wz) * 8 + 2 * part] = pay[it].x;
                    L.win[((wx * 3 + wy) * 3 + wz) * 8 + 2 * part + 1] = pay[it].y;
                }
            }
        }
    }
    PATCH_STAMP(1);
    pop = wave_sum(pop);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- window assembly: lane w owns output word w = ix*4 + iy/4 (Voxel.py:204-214 incl. wrap-around)
    unsigned long long word = 0ull;
    {
        const int ix = lane >> 2;
        const int x = kx + (ix < 8 ? ix : ix - 16);
        const int bxl = (x >> 3) - bx0 - wx0, xw = x & 7;
        const int z0 = kz - 8;
        const int zsh = z0 & 7;  // the window's first z brick is window-local index 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int iy = (lane & 3) * 4 + q;
            const int y = ky + (iy < 8 ? iy : iy - 16);
            const int byl = (y >> 3) - by0 - wy0, ysh = (y & 7) << 3;
            unsigned int str = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned int byte = (unsigned int)(L.win[((bxl * 3 + byl) * 3 + j) * 8 + xw] >> ysh) & 0xFFu;
                str |= byte << (8 * j);
            }
            const unsigned int r16 = (str >> zsh) & 0xFFFFu;           // bit t <-> dz = t - 8
            const unsigned int o16 = ((r16 >> 8) | (r16 << 8)) & 0xFFFFu;  // iz = dz mod 16
            word |= (unsigned long long)o16 << (16 * q);
        }
    }
    PATCH_STAMP(2);
    unsigned int fl = 0;
    if (pop > NN_CAP) {
        // ---- exact 496-NN semantics (Voxel.py:182,:195-196): histogram the ball by squared distance,
        //      every lane walking the set bits of the two bricks it holds in registers
        for (int i = lane; i <= BALL_D2; i += 64) L.hist[i] = 0u;
        if (lane == 0) L.ncls = 0;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll 1
        for (int it = 0; it * 16 < nfound; ++it) {
            const int e = it * 16 + (lane >> 2);
            if (e >= nfound) continue;
            const unsigned ent = L.found[e];
            const int bli = (int)(ent & 127u);
            const ulonglong2 pay = ((const ulonglong2 *)(tab.bits + (size_t)(ent >> 7) * 8))[part];
            const int xb = (bx0 + bli / 25) * 8 - kx, ybase = (by0 + (bli / 5) % 5) * 8 - ky, zbase = (bz0 + bli % 5) * 8 - kz;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned long long v = q ? pay.y : pay.x;
                const int dx = xb + 2 * part + q;
                while (v) {
                    const int t = __ffsll((long long)v) - 1;
                    v &= v - 1;
                    const int dy = ybase + (t >> 3), dz = zbase + (t & 7);
                    const int d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 <= BALL_D2) atomicAdd(&L.hist[d2], 1u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        {
            // cut class = first d2 whose cumulative count exceeds 496: lane l owns classes 4l .. 4l+3, a wave
            // prefix sum replaces the serial walk over 193 LDS words (which alone cost ~6 us on the few
            // wavefronts that take this path -- the tail of the whole kernel)
            int hh[4], local = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d2 = 4 * lane + q;
                hh[q] = d2 <= BALL_D2 ? (int)L.hist[d2] : 0;
                local += hh[q];
            }
            int incl = local;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            int cum = incl - local, mycut = -1, myroom = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mycut < 0 && 4 * lane + q <= BALL_D2 && cum + hh[q] > NN_CAP) { mycut = 4 * lane + q; myroom = NN_CAP - cum; }
                cum += hh[q];
            }
            const unsigned long long hit = __ballot(mycut >= 0);
            if (hit == 0ull) {
                if (lane == 0) { L.cut = BALL_D2 + 1; L.room = 0; }
            } else if (lane == __ffsll((long long)hit) - 1) {
                L.cut = mycut;
                L.room = myroom;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int cut = L.cut, room = L.room;
        if (cut <= BALL_D2) {
            if (room > 0) {
                // members of the cut class, for the canonical tie rule (ascending (x,y,z) key)
#pragma unroll 1
                for (int it = 0; it * 16 < nfound; ++it) {
                    const int e = it * 16 + (lane >> 2);
                    if (e >= nfound) continue;
                    const unsigned ent = L.found[e];
                    const int bli = (int)(ent & 127u);
                    const ulonglong2 pay = ((const ulonglong2 *)(tab.bits + (size_t)(ent >> 7) * 8))[part];
                    const int xb = (bx0 + bli / 25) * 8, yb = (by0 + (bli / 5) % 5) * 8, zb = (bz0 + bli % 5) * 8;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        unsigned long long v = q ? pay.y : pay.x;
                        const int x = xb + 2 * part + q;
                        while (v) {
                            const int t = __ffsll((long long)v) - 1;
                            v &= v - 1;
                            const int y = yb + (t >> 3), z = zb + (t & 7);
                            const int dx = x - kx, dy = y - ky, dz = z - kz;
                            if (dx * dx + dy * dy + dz * dz == cut) {
                                const int p = atomicAdd(&L.ncls, 1);
                                if (p < CLASS_CAP) L.cls[p] = caelo_pack3(x, y, z);
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            const int ncls = L.ncls < CLASS_CAP ? L.ncls : CLASS_CAP;
            // filter this lane's word
            const int ix = lane >> 2;
            const int dx = ix < 8 ? ix : ix - 16;
            unsigned long long v = word;
            while (v) {
                const int t = __ffsll((long long)v) - 1;
                v &= v - 1;
                const int iy = (lane & 3) * 4 + (t >> 4), iz = t & 15;
                const int dy = iy < 8 ? iy : iy - 16, dz = iz < 8 ? iz : iz - 16;
                const int d2 = dx * dx + dy * dy + dz * dz;
                bool keep = d2 < cut;
                if (d2 == cut) {
                    fl |= (room > 0) ? 2u : 0u;
                    if (room > 0) {
                        const unsigned long long kk = caelo_pack3(kx + dx, ky + dy, kz + dz);
                        int rank = 0;
                        for (int q = 0; q < ncls; ++q) rank += (L.cls[q] < kk) ? 1 : 0;
                        keep = rank < room;
                    }
                }
                if (!keep) {
                    word &= ~(1ull << t);
                    fl |= 1u;
                }
            }
        }
    }
    // OR the flags across the wave (two ballots)
    fl = (__ballot(fl & 1u) ? 1u : 0u) | (__ballot(fl & 2u) ? 2u : 0u);
    out[lane] = word;
    if (dd)  // equal patches of the launch set are encoded once (dedup.hip)
        caelo_dedup_insert(word, lane, (int)pw, (int)(fz * CAELO_FRAME_PATCHES + pw), dd, fs.f[0].dd, dedup_slot_mask(fs.n), dd_mask);
    if (lane == 0) flags[pw] = (uint8_t)fl;
    PATCH_STAMP(3);
}

int vox_patches_launch(const caelo_voxmap *m, const float *pts, int pts_ld, int64_t k_max, const int32_t *n_key,
                       uint64_t *bits, uint8_t *flags, int32_t *status, bool check_counts, hipStream_t s, void *dedup_scratch) {
    caelo_frame_set fs = {};
    fs.n = 1;
    caelo_frame_dev &d = fs.f[0];
    frame_dev_set_map(d, m);
    d.key_pts = const_cast<float *>(pts); d.kp_ld = pts_ld; d.n_key = const_cast<int32_t *>(n_key);
    d.bits = (unsigned long long *)bits; d.flags = flags; d.status = status; d.dd = (DedupScratch *)dedup_scratch;
    return vox_patches_set(fs, k_max, check_counts, s);
}

int vox_patches_set(const caelo_frame_set &fs, int64_t k_max, bool check_counts, hipStream_t s) {
    const int64_t waves = k_max * 3;
    k_patches<<<dim3((unsigned)((waves + PW_WAVES - 1) / PW_WAVES) * (unsigned)fs.n), 64 * PW_WAVES, 0, s>>>(fs, k_max, check_counts ? 1 : 0,
                                                                                                   dedup_hash_mask());
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_patches(caelo_ctx *c, const caelo_voxmap *m, const float *pts, int64_t k_max, const int32_t *n_key,
                            uint64_t *bits, uint8_t *flags, int32_t *status, void *stream) {
    CAELO_REQUIRE(c && m && pts && bits && flags && status, "null argument");
    CAELO_REQUIRE(k_max > 0, "k_max must be positive");
    const int rc = vox_patches_launch(m, pts, 3, k_max, n_key, bits, flags, status, false, caelo_stream(stream));
    if (rc) return rc;
    return kd_resolve(m, pts, 3, k_max, n_key, bits, flags, caelo_stream(stream));   // tie-split patches in the library's order (kdorder.hip)
}

// caelo_patches for n maps / key point sets (n <= 8): the patch gathers one after the other, then the kd-tree redo of ALL of them behind
// one launch of each kd kernel (kdorder.hip, kd_resolve_many) -- the tie redo of the frames of a chunk
CAELO_API int caelo_patches_many(caelo_ctx *c, int n, const caelo_voxmap *const *maps, const float *const *pts, int64_t k_max,
                                 const int32_t *const *n_key, uint64_t *const *bits, uint8_t *const *flags, int32_t *const *status, void *stream) {
    CAELO_REQUIRE(c && maps && pts && n_key && bits && flags && status && n >= 1 && n <= CAELO_FB_MAX && k_max > 0, "caelo_patches_many: bad argument");
    for (int i = 0; i < n; ++i) {
        CAELO_REQUIRE(maps[i] && pts[i] && bits[i] && flags[i] && status[i], "caelo_patches_many: null entry");
        const int rc = vox_patches_launch(maps[i], pts[i], 3, k_max, n_key[i], bits[i], flags[i], status[i], false, caelo_stream(stream));
        if (rc) return rc;
    }
    return kd_resolve_many(n, maps, pts, 3, k_max, n_key, bits, flags, caelo_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// dense <-> packed patches (API parity with the reference's [K,16,16,16,1] f32 arrays)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unpack(const unsigned long long *__restrict__ bits, int64_t n,
                                                float *__restrict__ dense) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 4 voxels
    if (i >= n * 1024) return;
    const int64_t p = i >> 10;
    const int lin = (int)(i & 1023) * 4;
    const unsigned long long w = bits[p * 64 + (lin >> 6)];
    const unsigned int nib = (unsigned int)(w >> (lin & 63)) & 0xFu;
    ((float4 *)dense)[i] = make_float4((float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u),
                                       (float)((nib >> 3) & 1u));
}

__global__ void __launch_bounds__(256) k_pack(const float *__restrict__ dense, int64_t n,
                                              unsigned long long *__restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per output word
    if (i >= n * 64) return;
    const float4 *src = (const float4 *)(dense + i * 64);
    unsigned long long w = 0ull;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 v = src[q];
        w |= (unsigned long long)((v.x != 0.f) | ((v.y != 0.f) << 1) | ((v.z != 0.f) << 2) | ((v.w != 0.f) << 3)) << (4 * q);
    }
    bits[i] = w;
}

CAELO_API int caelo_unpack_patches(caelo_ctx *c, const uint64_t *bits, int64_t n, float *dense, void *stream) {
    CAELO_REQUIRE(c && bits && dense && n > 0, "bad argument");
    k_unpack<<<(unsigned)((n * 1024 + 255) / 256), 256, 0, caelo_stream(stream)>>>((const unsigned long long *)bits, n, dense);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_pack_patches(caelo_ctx *c, const float *dense, int64_t n, uint64_t *bits, void *stream) {
    CAELO_REQUIRE(c && bits && dense && n > 0, "bad argument");
    k_pack<<<(unsigned)((n * 64 + 255) / 256), 256, 0, caelo_stream(stream)>>>(dense, n, (unsigned long long *)bits);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
