// kdorder.hip -- the 496-nearest cut of GetPatchesList in scikit-learn's own order, for the patches where it matters.
//
// Reference behaviour restated (never its code): Voxel.py:182,195-196 --
//     NearestNeighbors(n_neighbors=496, radius=14, algorithm='auto').fit(AllVoxels_s).kneighbors(KeyVoxels)
// followed by the window test of :204-210.  When more than 496 voxels lie within the window's reach AND the 496th place falls
// inside a class of EQUIDISTANT voxels that has members in the window, which of them are returned is decided by the library's
// kd-tree (scikit-learn 0.24.2: `auto` = kd_tree for n_samples // 2 > n_neighbors, i.e. n >= 994 voxels; leaf_size 30).  The
// algorithm below is the one oracle/caelo_oracle.c restates and tools/make_goldens.py validates against the library itself
// (index array of the tree, 920 lattice queries, every truncated patch of the fixtures):
//   build   n_levels = int(log2(max(1, (n - 1) / 30)) + 1); node i owns idx[start, end); inner nodes split on the dimension of
//           largest spread (first of equals) at n / 2 by a quickselect whose partition is Lomuto's with the LAST element as the
//           pivot (strict <): the resulting order of idx is part of the contract, so the partition is run as written --
//           sequentially, one thread per node, level by level (a node's points sit in a (coordinate, index) key array so that
//           the scan reads consecutive addresses);
//   query   depth first; a node whose bounding-box distance exceeds the heap's largest is skipped; leaves push their points in
//           idx order; the child with the smaller lower bound first (<=); max-heap of 496 squared distances (integers: the
//           coordinates are voxel indices), a candidate >= the largest is rejected, otherwise it replaces the root and is
//           sifted down (first child when dist[c1] >= dist[c2]).
// k_patches (voxel.hip) has already produced every patch under the canonical rule and flagged those whose cut splits a tie class
// (flag bit 2); only THOSE are redone here, and only when the voxel lists are known in the reference's order -- a map filled by
// caelo_voxmap_from_lists (the staged API: api.GetPatchesList) -- and long enough for the kd-tree.  A redone patch carries flag 4
// instead of 2.  Rare by construction: 0 of 614 400 patches on the KITTI-shaped scene, 115 on the clutter scene (200 frames each).
// The fused path (caelo_extract, caelo_pipeline) builds voxel SETS, not lists in first-touch order: its flagged patches keep the
// canonical rule and bit 2 (INTEGRATION.md says what a caller who needs them exact does).
//
// Lists of 496 .. 993 voxels (round 6): `auto` is BRUTE FORCE there -- squared distances to every list entry in list order (exact
// integers), then np.argpartition(dist, 495)[:496] (scikit-learn 0.24.2 _kneighbors_reduce_func; the sort that follows only reorders
// the 496, and a patch is a set).  Which members of a split tie class come out is decided by NumPy's introselect (1.18 .. 1.26:
// median of three with the 3-lowest moved to low + 1, unguarded Hoare partition, median of medians of five after 2 * msb(n)
// partitions, selection sort for kth - low < 3): k_brute_query runs it as written, one lane per tie-split patch over a packed
// (distance, index) array in LDS -- at most 993 entries and a few thousand steps; such lists are a degenerate scan's.  Restated in
// oracle/caelo_oracle.c (orc_argpartition) and pinned there against the library; tests/golden/patch_brute.npz.
#include "caelo_internal.h"

#define KD_K 496
#define KD_LEAF 30
#define KD_MIN_N 994          // 'auto' -> kd_tree iff n // 2 > 496
#define KD_MAX_NODES 16384
#define KD_INF 0x7FFFFFFF

struct caelo_kd_scale {
    int16_t *vox;             // [cap][3] the list in the caller's (= the reference's) order
    int32_t *idx;             // [cap]
    unsigned long long *keys; // [cap] quickselect scratch: (coordinate + 32768) << 32 | index
    unsigned long long *keys2;// [cap] the partition's output before it is copied back
    int32_t *ev;              // [cap] the partition's event list (see k_kd_build)
    int32_t *start, *end;     // [KD_MAX_NODES]
    int16_t *lo, *hi;         // [KD_MAX_NODES][3]
    int32_t *queue;           // [k_cap] key points whose patch of this scale is tie-split
};   // (the list's length is a device word, kd.state[16 + scale]: a list ordered on the device never tells the host)
struct caelo_kd {
    caelo_kd_scale s[3];
    int32_t *state;           // device words: see k_kd_build_top
    int64_t cap, k_cap;
    char *base;
};

// Up to CAELO_FB_MAX maps behind ONE launch of each kd kernel (blockIdx.z = map): a redone frame is a chain of ~150 dependent quickselect
// passes, a millisecond or two whatever the GPU has free -- eight frames' chains side by side cost what one does (round 6; the tie redo
// of many frames used to issue the four kernels frame by frame on side streams: Engine.resolve_ties_many).
struct caelo_kd_set {
    caelo_kd k[CAELO_FB_MAX];
    const float *pts[CAELO_FB_MAX];
    const int32_t *n_key[CAELO_FB_MAX];
    unsigned long long *bits[CAELO_FB_MAX];
    uint8_t *flags[CAELO_FB_MAX];
    int n, pts_ld;
};
static_assert(sizeof(caelo_kd_set) <= 3800, "caelo_kd_set must fit the kernel argument segment");

namespace {

__global__ void __launch_bounds__(256) k_kd_collect(const caelo_kd_set S, int64_t k0, int64_t k_max) {
    const caelo_kd &kd = S.k[blockIdx.z];
    const uint8_t *__restrict__ flags = S.flags[blockIdx.z];
    const int32_t *__restrict__ n_key = S.n_key[blockIdx.z];
    const int64_t pw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (key point of this chunk, scale)
    if (pw >= kd.k_cap * 3) return;
    const int64_t kp = k0 + pw / 3;
    const int sc = (int)(pw % 3);
    const int64_t K = n_key ? min((int64_t)*n_key, k_max) : k_max;
    if (kp >= K || !(flags[kp * 3 + sc] & 2) || kd.state[16 + sc] < KD_K) return;   // (fewer than 496 voxels: an error of the call)
    const int q = atomicAdd(&kd.state[sc], 1);
    kd.s[sc].queue[q] = (int32_t)kp;   // (at most k_cap entries per scale and chunk)
}

// The build.  Level by level; `tpn` threads share a node of the upper levels for the parallel parts (bounding box, key array) and
// for the quickselect's partitions.  Two kernels: k_kd_build_top runs the levels whose nodes are long (one workgroup of 1024 threads
// per scale, key arrays in global memory) down to the level LT where every node holds at most KD_SUBCAP voxels; from there the 2^LT
// subtrees are independent and k_kd_build_sub gives each its own workgroup with the key arrays in LDS.  (One workgroup for the whole
// tree took 9 ms for a 35 k-voxel list, three quarters of it below level 4: dozens of quickselect passes per level, each a handful of
// barriers and dependent L2 round trips.)
constexpr int KD_T = 1024;        // threads of the top kernel: the upper levels' partitions are chains of dependent L2 reads, more of them in flight
constexpr int KD_TS = 256;        // threads of a subtree's workgroup
constexpr int KD_SUBCAP = 2048;   // voxels of a subtree (keys + keys2 + ev in LDS: 40 KB)

template <int NT>
struct KdShared {
    int lo[NT][3], hi[NT][3];
    int left[NT], right[NT], act[NT], lf[NT], cl[NT], cg[NT];   // per node of a pass / per unit
    long long budget[NT];
    int gave_up;   // a node's quickselect exceeded its budget (below): the tree is not built, the canonical rule stays
};

// n_levels = int(log2(max(1, (n - 1) / 30)) + 1) in integers: the largest L with 30 * 2^L <= n - 1, plus one
__host__ __device__ inline int kd_levels_of(int64_t n) {
    if (n < KD_MIN_N) return 0;
    int L = 0;
    while (((int64_t)KD_LEAF << (L + 1)) <= n - 1) ++L;
    return L + 1;
}
// the first level whose nodes hold at most KD_SUBCAP voxels (a node of level L holds at most ceil(n / 2^L))
__host__ __device__ inline int kd_top_levels_of(int64_t n) {
    int L = 0;
    while (((n + ((int64_t)1 << L) - 1) >> L) > KD_SUBCAP) ++L;
    return L;
}

// levels [level0, level1) of the subtree under node `root` of level0 (root = its index among that level's nodes); the key arrays
// A0 / KB0 / EV0 are addressed by (position in idx) - off
template <int NT>
__device__ __forceinline__ void kd_build_levels(const caelo_kd_scale &T, const int n_nodes, KdShared<NT> &S, const int level0, const int level1,
                                                const int root, unsigned long long *A0, unsigned long long *KB0, int32_t *EV0, const int off,
                                                const int scale_for_print, const int n_for_print) {
    const int tid = threadIdx.x;
    for (int level = level0; level < level1; ++level) {
#ifdef KD_PROFILE
        const long long t_level = wall_clock64();
        int n_pass = 0;
        long long t_count = 0, t_write = 0, t_resolve = 0, t_copy = 0;
#endif
        const int d = level - level0, count = 1 << d;           // nodes of this (sub)tree on the level
        const int first = (1 << level) - 1 + (root << d);
        const int tpn = count >= NT ? 1 : NT >> d;              // threads per node
        const int per_pass = NT / tpn;                       // nodes in flight
        for (int base = 0; base < count; base += per_pass) {
            const int local = tid / tpn, sub = tid % tpn;
            const int W = tpn < 64 ? tpn : 64, nu = tpn / W;          // lanes per unit, units per node (see the partition below)
            const int u = sub / W, lu = sub % W, ug = tid / W;
            const int ushift = (tid & 63) / W * W;
            const unsigned long long umask = W == 64 ? ~0ull : (1ull << W) - 1ull;
            const int node = first + base + local;
            const bool live = base + local < count;
            const int s = live ? T.start[node] : 0, e = live ? T.end[node] : 0;
            // ---- bounding box of the node's points
            int lo[3] = {32767, 32767, 32767}, hi[3] = {-32768, -32768, -32768};
            for (int i = s + sub; i < e; i += tpn) {
                const int16_t *p = T.vox + 3 * (int64_t)T.idx[i];
#pragma unroll
                for (int j = 0; j < 3; ++j) { lo[j] = min(lo[j], (int)p[j]); hi[j] = max(hi[j], (int)p[j]); }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) { S.lo[tid][j] = lo[j]; S.hi[tid][j] = hi[j]; }
            __syncthreads();
            for (int o = tpn >> 1; o > 0; o >>= 1) {
                if (sub < o)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        S.lo[tid][j] = min(S.lo[tid][j], S.lo[tid + o][j]);
                        S.hi[tid][j] = max(S.hi[tid][j], S.hi[tid + o][j]);
                    }
                __syncthreads();
            }
            const int lead = tid - sub;
            int jmax = 0, spread = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lo[j] = S.lo[lead][j]; hi[j] = S.hi[lead][j];
                if (hi[j] - lo[j] > spread) { spread = hi[j] - lo[j]; jmax = j; }
            }
            if (live && sub == 0)
#pragma unroll
                for (int j = 0; j < 3; ++j) { T.lo[3 * node + j] = (int16_t)lo[j]; T.hi[3 * node + j] = (int16_t)hi[j]; }
            const bool split = live && 2 * node + 1 < n_nodes && e - s >= 2;
            // ---- the node's (coordinate, index) keys, consecutive
            if (split)
                for (int i = s + sub; i < e; i += tpn) {
                    const int32_t p = T.idx[i];
                    A0[i - off] = ((unsigned long long)(unsigned)((int)T.vox[3 * (int64_t)p + jmax] + 32768) << 32) | (unsigned)p;
                }
            __syncthreads();
            // ---- quickselect around position n / 2: Lomuto, last element as the pivot, strict <.  The ORDER the partition leaves behind
            // is the contract (the library's tree has it), and Lomuto's swaps have a closed form: the elements below the pivot end up in
            // front in their original order; the others form a QUEUE between `mid` and the scan position -- an element >= pivot is
            // appended to it, an element < pivot rotates it (the front element goes to the back) when it is not empty -- and the final
            // swap with the pivot is one more rotation.  Number the enqueue events in time order (a push of element i, or the re-enqueue
            // of whatever rotation r dequeued = the element of enqueue event r): with G pushes and R + 1 rotations the queue at the end
            // holds the events R + 1 .. G + R, in that order, and an event that is a re-enqueue is resolved by following its reference
            // (strictly decreasing) to a push.  Every event index is a prefix sum, so the node's `tpn` threads partition it together:
            // count (chunk per thread), scan, write the stable front and the event list, resolve, copy back.  (Rounds 4's one thread per
            // node took 9 ms for a 9 k-voxel list -- 85 % of what re-doing a frame's tie-split patches cost; validated against the serial
            // form on random cases in Python and by the patches of the truncation fixtures.)
            unsigned long long *a = A0 + (s - off), *kb = KB0 + (s - off);
            int32_t *ev = EV0 + (s - off);
            const int m = e - s, nmid = m / 2;
            if (sub == 0) {
                S.left[local] = 0; S.right[local] = m - 1; S.act[local] = split ? 1 : 0;
                // Lomuto with the last element as the pivot is quadratic on a list sorted along the split dimension (the library pays
                // that too).  Lists in first-touch order are PARTLY sorted (a scan line sweeps the azimuth): measured on the clutter
                // scene's lists the worst node needs 87 passes' worth of its length (a first budget of 48 gave up on real frames -- the
                // 600-frame soak caught it by the descriptors of the patches it left on the canonical rule; round 6's soak over 600
                // STRUCTURED clutter frames caught 256 the same way: frame 103's 16 cm list, 25 326 voxels, one patch).  A give-up is a
                // WRONG patch (flag 2 stays set, the caller can see it), a long build is only slow: 4096 passes' worth -- tens of
                // milliseconds for the longest node of a scan's list in the worst case, which the library would pay as well.
                S.budget[local] = 4096ll * m + 65536;
            }
            for (;;) {
                __syncthreads();   // (the reduction below is a barrier, not a fence: the q_* words written at the end of the previous pass must have landed)
                if (!__syncthreads_or(sub == 0 && S.act[local] != 0)) break;
#ifdef KD_PROFILE
                ++n_pass;
                long long t_ph = wall_clock64();
#endif
                const bool act = S.act[local] != 0;
                const int left = act ? S.left[local] : 0, right = act ? S.right[local] : -1;
                const int len = right - left;                            // elements in front of the pivot
                // The node's threads work in UNITS of W = min(tpn, 64) lanes of one wavefront: a unit owns a contiguous chunk and walks it
                // W consecutive elements at a time (coalesced; a thread walking its own chunk element by element made every load of a
                // wavefront touch 64 cache lines), positions inside a step come from the ballot, positions across steps are carried.
                const int chunk = len > 0 ? (len + nu - 1) / nu : 0;
                const int i0 = left + u * chunk, i1 = min(i0 + chunk, right);
                const unsigned pvv = act ? (unsigned)(a[right] >> 32) : 0u;
                int cl = 0, fg = -1;
                for (int b0 = i0; b0 < i1; b0 += W) {
                    const int i = b0 + lu;
                    const bool valid = i < i1;
                    const bool less = valid && (unsigned)(a[valid ? i : i0] >> 32) < pvv;
                    const unsigned long long ml = (__ballot(less) >> ushift) & umask, mg = (__ballot(valid && !less) >> ushift) & umask;
                    cl += __popcll(ml);
                    if (fg < 0 && mg) fg = b0 + (int)__builtin_ctzll(mg);
                }
                const int cg = (i1 > i0 ? i1 - i0 : 0) - cl;
                if (lu == 0) { S.cl[ug] = cl; S.cg[ug] = cg; }
                __syncthreads();
#ifdef KD_PROFILE
                t_count += wall_clock64() - t_ph; t_ph = wall_clock64();
#endif
                for (int o = 1; o < nu; o <<= 1) {                       // inclusive scans over the node's units (nu is the same for all)
                    int x = 0, y = 0;
                    if (u >= o) { x = S.cl[ug - o]; y = S.cg[ug - o]; }
                    __syncthreads();
                    if (lu == 0) { S.cl[ug] += x; S.cg[ug] += y; }
                    __syncthreads();
                }
                const int lead2 = ug - u;
                const int lb = S.cl[ug] - cl, gb = S.cg[ug] - cg;        // below / not below the pivot in front of this unit's chunk
                const int L = S.cl[lead2 + nu - 1], G = S.cg[lead2 + nu - 1];
                if (act && fg >= 0 && gb == 0 && lu == 0) S.lf[local] = lb + (fg - i0);   // elements below the pivot in front of the FIRST one that is not
                __syncthreads();
                const int Lf = G > 0 ? S.lf[local] : L, R = L - Lf;
                {
                    int pq0 = gb, l0 = lb;
                    const unsigned long long below = (1ull << lu) - 1ull;
                    for (int b0 = i0; b0 < i1; b0 += W) {
                        const int i = b0 + lu;
                        const bool valid = i < i1;
                        const unsigned long long v = a[valid ? i : i0];
                        const bool less = valid && (unsigned)(v >> 32) < pvv;
                        const unsigned long long ml = (__ballot(less) >> ushift) & umask, mg = (__ballot(valid && !less) >> ushift) & umask;
                        const int l = l0 + __popcll(ml & below), pq = pq0 + __popcll(mg & below);   // of the elements in front of i
                        if (less) {
                            kb[left + l] = v;
                            if (pq > 0) ev[left + pq + (l - Lf)] = -((l - Lf) + 1);   // rotation l - Lf: re-enqueue of the element of event l - Lf
                        } else if (valid) {
                            ev[left + pq + (l - Lf)] = i;                            // push of element i
                        }
                        l0 += __popcll(ml); pq0 += __popcll(mg);
                    }
                }
                if (act && sub == 0) {
                    kb[left + L] = a[right];
                    if (G > 0) ev[left + G + R] = -(R + 1);                          // the final swap with the pivot
                }
                __syncthreads();
#ifdef KD_PROFILE
                t_write += wall_clock64() - t_ph; t_ph = wall_clock64();
#endif
                for (int k = sub; act && k < G; k += tpn) {
                    int ee = R + 1 + k, x, steps = 0;
                    while ((x = ev[left + ee]) < 0) {
                        ee = -x - 1;
                        if (++steps > (1 << 22)) { S.gave_up = 1; x = left; break; }
                    }
                    kb[left + L + 1 + k] = a[x];
                }
                __syncthreads();
#ifdef KD_PROFILE
                t_resolve += wall_clock64() - t_ph; t_ph = wall_clock64();
#endif
                for (int i = left + sub; act && i <= right; i += tpn) a[i] = kb[i];
                __syncthreads();
#ifdef KD_PROFILE
                t_copy += wall_clock64() - t_ph;
#endif
#ifdef KD_DEBUG
                if (act && sub == 0) {
                    int bad = 0;
                    for (int i = left; i <= right; ++i) {
                        const unsigned long long v = a[i];
                        const unsigned kv = (unsigned)(v >> 32), id = (unsigned)v;
                        if (id >= (unsigned)n_for_print) bad |= 1;
                        if (i < left + L && !(kv < pvv)) bad |= 2;
                        if (i > left + L && (kv < pvv)) bad |= 4;
                        if (i == left + L && kv != pvv) bad |= 8;
                    }
                    if (bad) printf("kd build: level %d node %d tpn %d left %d right %d L %d G %d R %d Lf %d bad %d\n", level, node, tpn, left, right, L, G, R, Lf, bad);
                }
                __syncthreads();
#endif
                if (act && sub == 0) {
                    const int mid = left + L;
                    S.budget[local] -= (long long)len + 1;
                    if (mid == nmid) S.act[local] = 0;
                    else if (S.budget[local] < 0) { S.gave_up = 1; S.act[local] = 0; }
                    else if (mid < nmid) S.left[local] = mid + 1;
                    else S.right[local] = mid - 1;
                }
            }
            if (split && sub == 0) {
                T.start[2 * node + 1] = s; T.end[2 * node + 1] = s + nmid;
                T.start[2 * node + 2] = s + nmid; T.end[2 * node + 2] = e;
            }
            __syncthreads();
            if (split)
                for (int i = s + sub; i < e; i += tpn) T.idx[i] = (int32_t)(unsigned)A0[i - off];
            __syncthreads();
        }
#ifdef KD_PROFILE
        if (tid == 0 && root == 0) printf("kd build scale %d n %d level %2d (from %d): %7.1f us, %4d passes (count %.1f write %.1f resolve %.1f copy %.1f us)\n",
                                          scale_for_print, n_for_print, level, level0, (wall_clock64() - t_level) * 0.01, n_pass, t_count * 0.01,
                                          t_write * 0.01, t_resolve * 0.01, t_copy * 0.01);
#endif
    }
}

// kd.state: [0..2] queue lengths | [4..6] 0 no tree, 1 built, 2 not built (a quickselect gave up / too many nodes: the canonical rule stays),
// 3 upper levels built, subtrees pending | [8..10] a subtree gave up | [12..14] subtrees done | [16..18] list lengths
__global__ void __launch_bounds__(KD_T) k_kd_build_top(const caelo_kd_set S_) {
    const caelo_kd &kd = S_.k[blockIdx.z];
    const int sc = blockIdx.x;
    const caelo_kd_scale T = kd.s[sc];
    if (kd.state[sc] == 0 || kd.state[4 + sc] != 0) return;   // no tie-split patch of this scale / tree already built (or given up)
    __shared__ KdShared<KD_T> S;
    const int tid = threadIdx.x;
    const int n = kd.state[16 + sc];
    if (n < KD_MIN_N) return;   // brute force in the library: k_brute_query
    const int n_levels = kd_levels_of(n), n_nodes = (1 << n_levels) - 1;
    if (n_nodes > KD_MAX_NODES || n > kd.cap) { if (tid == 0) kd.state[4 + sc] = 2; return; }
    if (tid == 0) S.gave_up = 0;
    for (int i = tid; i < n; i += KD_T) T.idx[i] = i;
    for (int i = tid; i < n_nodes; i += KD_T) { T.start[i] = 0; T.end[i] = 0; }   // (children of a node that did not split stay empty)
    __syncthreads();
    if (tid == 0) { T.start[0] = 0; T.end[0] = n; }
    __syncthreads();
    const int lt = min(kd_top_levels_of(n), n_levels);
    kd_build_levels<KD_T>(T, n_nodes, S, 0, lt, 0, T.keys, T.keys2, T.ev, 0, sc, n);
    __syncthreads();
    __threadfence();
    if (tid == 0) kd.state[4 + sc] = S.gave_up ? 2 : 3;
}

__global__ void __launch_bounds__(KD_TS) k_kd_build_sub(const caelo_kd_set S_) {
    const caelo_kd &kd = S_.k[blockIdx.z];
    const int sc = blockIdx.y;
    const caelo_kd_scale T = kd.s[sc];
    if (kd.state[4 + sc] != 3) return;   // (stays 3 until the LAST subtree of this scale is done)
    const int n = kd.state[16 + sc];
    const int n_levels = kd_levels_of(n), n_nodes = (1 << n_levels) - 1;
    const int lt = min(kd_top_levels_of(n), n_levels), n_sub = 1 << lt;
    const int root = blockIdx.x;
    if (root >= n_sub) return;
    __shared__ KdShared<KD_TS> S;
    __shared__ unsigned long long s_a[KD_SUBCAP], s_kb[KD_SUBCAP];
    __shared__ int32_t s_ev[KD_SUBCAP];
    const int tid = threadIdx.x;
    if (tid == 0) S.gave_up = 0;
    __syncthreads();
    const int node0 = (1 << lt) - 1 + root;
    const int s0 = lt < n_levels ? T.start[node0] : 0, e0 = lt < n_levels ? T.end[node0] : 0;
    if (e0 - s0 > KD_SUBCAP) { if (tid == 0) S.gave_up = 1; }           // (cannot happen: ceil(n / 2^lt) <= KD_SUBCAP)
    else if (e0 > s0) kd_build_levels<KD_TS>(T, n_nodes, S, lt, n_levels, root, s_a, s_kb, s_ev, s0, sc, n);
    __syncthreads();
    if (tid == 0) {
        if (S.gave_up) atomicOr(&kd.state[8 + sc], 1);
        __threadfence();
        if (atomicAdd(&kd.state[12 + sc], 1) == n_sub - 1) {
            __threadfence();
            kd.state[4 + sc] = atomicOr(&kd.state[8 + sc], 0) ? 2 : 1;
        }
    }
}

struct KdHeapLds {
    unsigned long long patch[64];
};

__device__ inline int kd_min_rdist(const caelo_kd_scale &T, int node, const int q[3]) {
    int r = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int d_lo = (int)T.lo[3 * node + j] - q[j], d_hi = q[j] - (int)T.hi[3 * node + j];
        const int d = d_lo > 0 ? d_lo : (d_hi > 0 ? d_hi : 0);
        r += d * d;   // (a box further than 2^15 away overflows nothing that matters: coordinates < 2^14)
    }
    return r;
}

// The query's max-heap of 496 (distance, index) pairs lives in REGISTERS, 1-based: position p (root = 1, children 2p and 2p + 1) is
// lane p & 63 of register p >> 6, so that heap level L sits in compile-time registers (levels 0-5: register 0, level 6: register 1,
// level 7: registers 2-3, level 8: registers 4-7).  Every position of the sift-down is uniform and the sift-down is unrolled by
// level: a read is a v_readlane (plus scalar selects on the two deepest levels), a write a one-lane select -- a few tens of cycles
// per level where an LDS round trip plus the hop to a scalar register took ~400: a push cost 1.65 us with the heap in LDS (~750
// pushes per patch: 1.25 of the query's 1.35 ms).  (A run-time switch over the register compiled to branch chains and was SLOWER.)
template <int L>
__device__ __forceinline__ int kd_heap_get(const int (&a)[8], int pos) {   // pos uniform, on heap level L
    const int l = pos & 63;
    if constexpr (L <= 5) return __builtin_amdgcn_readlane(a[0], l);
    else if constexpr (L == 6) return __builtin_amdgcn_readlane(a[1], l);
    else if constexpr (L == 7) {
        const int x = __builtin_amdgcn_readlane(a[2], l), y = __builtin_amdgcn_readlane(a[3], l);
        return (pos & 64) ? y : x;
    } else {
        const int x0 = __builtin_amdgcn_readlane(a[4], l), x1 = __builtin_amdgcn_readlane(a[5], l);
        const int x2 = __builtin_amdgcn_readlane(a[6], l), x3 = __builtin_amdgcn_readlane(a[7], l);
        const int r = (pos >> 6) & 3;
        return r == 0 ? x0 : (r == 1 ? x1 : (r == 2 ? x2 : x3));
    }
}
template <int L>
__device__ __forceinline__ void kd_heap_set(int (&a)[8], int pos, int v, int lane) {   // pos, v uniform
    if constexpr (L <= 5) a[0] = lane == pos ? v : a[0];
    else if constexpr (L == 6) a[1] = lane + 64 == pos ? v : a[1];
    else if constexpr (L == 7) { a[2] = lane + 128 == pos ? v : a[2]; a[3] = lane + 192 == pos ? v : a[3]; }
    else { a[4] = lane + 256 == pos ? v : a[4]; a[5] = lane + 320 == pos ? v : a[5]; a[6] = lane + 384 == pos ? v : a[6]; a[7] = lane + 448 == pos ? v : a[7]; }
}
// heap push of (val, iv) at the root, the library's sift-down: the larger child moves up while it is larger than val (the FIRST child
// when the two are equal); position p is on level L
template <int L>
__device__ __forceinline__ void kd_heap_sift(int (&hd)[8], int (&hx)[8], int p, int val, int iv, int lane) {
    if constexpr (L < 8) {
        const int c1 = 2 * p, c2 = c1 + 1;
        if (c1 <= KD_K) {
            int sw = 0, dsw = 0;
            const int d1 = kd_heap_get<L + 1>(hd, c1);
            if (c2 > KD_K) { if (d1 > val) { sw = c1; dsw = d1; } }
            else {
                const int d2 = kd_heap_get<L + 1>(hd, c2);
                if (d1 >= d2) { if (val < d1) { sw = c1; dsw = d1; } }
                else { if (val < d2) { sw = c2; dsw = d2; } }
            }
            if (sw) {
                const int isw = kd_heap_get<L + 1>(hx, sw);
                kd_heap_set<L>(hd, p, dsw, lane); kd_heap_set<L>(hx, p, isw, lane);
                kd_heap_sift<L + 1>(hd, hx, sw, val, iv, lane);
                return;
            }
        }
    }
    kd_heap_set<L>(hd, p, val, lane); kd_heap_set<L>(hx, p, iv, lane);
}

// One wavefront per tie-split patch: the walk and the heap operations are uniform; a leaf's distances are computed by all lanes.
__global__ void __launch_bounds__(64) k_kd_query(const caelo_kd_set S_) {
    const caelo_kd &kd = S_.k[blockIdx.z];
    const float *__restrict__ pts = S_.pts[blockIdx.z];
    const int pts_ld = S_.pts_ld;
    unsigned long long *__restrict__ bits = S_.bits[blockIdx.z];
    uint8_t *__restrict__ flags = S_.flags[blockIdx.z];
    const int sc = blockIdx.y;
    const caelo_kd_scale T = kd.s[sc];
    const int cnt = min(kd.state[sc], (int)kd.k_cap);
    if ((int)blockIdx.x >= cnt || kd.state[4 + sc] != 1) return;
    __shared__ KdHeapLds L;
    const int n_nodes = (1 << kd_levels_of(kd.state[16 + sc])) - 1;
    const int lane = threadIdx.x;
    const int kp = T.queue[blockIdx.x];
    const double vs = sc == 0 ? 0.02 : (sc == 1 ? 0.02 * 8 : 0.02 * 32);                    // Voxel.py:31
    const int q[3] = {(int)(((double)pts[(size_t)pts_ld * kp] + 99.84) / vs), (int)(((double)pts[(size_t)pts_ld * kp + 1] + 99.84) / vs),
                      (int)(((double)pts[(size_t)pts_ld * kp + 2] + 14.72) / vs)};        // :185,:193 (float64 division, truncation)
    int hd[8], hx[8];   // the heap (see above)
#pragma unroll
    for (int r = 0; r < 8; ++r) { hd[r] = KD_INF; hx[r] = 0; }
    L.patch[lane] = 0ull;
    __syncthreads();
#ifdef KD_PROFILE
    long long tq0 = wall_clock64(), t_inner = 0, t_leafload = 0, t_push = 0;
    int n_inner = 0, n_leaf = 0, n_push = 0, n_skip = 0;
#endif
    int st_node[32], st_lb[32];
    int sp = 0;
    st_node[0] = 0; st_lb[0] = kd_min_rdist(T, 0, q);
    sp = 1;
    while (sp > 0) {                                // (uniform: every lane keeps the same stack)
        --sp;
        const int node = st_node[sp], lb = st_lb[sp];
#ifdef KD_PROFILE
        long long tp = wall_clock64();
        if (lb > __builtin_amdgcn_readlane(hd[0], 1)) { ++n_skip; continue; }
#else
        if (lb > __builtin_amdgcn_readlane(hd[0], 1)) continue;
#endif
        const int s = T.start[node], e = T.end[node];
        const bool leaf = 2 * node + 1 >= n_nodes || e - s < 2;
        if (leaf) {
            // A leaf's distances are computed by all lanes; the candidates are the lanes whose distance is below the heap's largest --
            // a ballot, taken again after every push (the largest shrinks), instead of one dependent test per point; the sift-down
            // runs on scalar registers.  Same pushes in the same order as the library's loop over the leaf.
            for (int c0 = s; c0 < e; c0 += 64) {
                const int i = c0 + lane;
                int p = 0, myval = KD_INF;
                if (i < e) {
                    p = T.idx[i];
                    const int16_t *v = T.vox + 3 * (int64_t)p;
                    const int dx = q[0] - v[0], dy = q[1] - v[1], dz = q[2] - v[2];
                    myval = dx * dx + dy * dy + dz * dz;
                }
#ifdef KD_PROFILE
                if (c0 == s) ++n_leaf;
                { const long long tn = wall_clock64() + (myval & 0); t_leafload += tn - tp; tp = tn; }
#endif
                int dist0 = __builtin_amdgcn_readlane(hd[0], 1);
                unsigned long long cand = __ballot(myval < dist0);
                while (cand) {
                    const int u = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(cand));
                    const int val = __builtin_amdgcn_readlane(myval, u), iv = __builtin_amdgcn_readlane(p, u);
                    kd_heap_sift<0>(hd, hx, 1, val, iv, lane);
                    dist0 = __builtin_amdgcn_readlane(hd[0], 1);
                    cand = __ballot(myval < dist0) & ((~0ull << u) << 1);
#ifdef KD_PROFILE
                    ++n_push;
#endif
                }
#ifdef KD_PROFILE
                { const long long tn = wall_clock64(); t_push += tn - tp; tp = tn; }
#endif
            }
        } else {
            const int i1 = 2 * node + 1, i2 = i1 + 1;
            const int l1 = kd_min_rdist(T, i1, q), l2 = kd_min_rdist(T, i2, q);
            // the nearer child first (<=: the left one on equality): it is pushed LAST
            if (l1 <= l2) { st_node[sp] = i2; st_lb[sp] = l2; st_node[sp + 1] = i1; st_lb[sp + 1] = l1; }
            else { st_node[sp] = i1; st_lb[sp] = l1; st_node[sp + 1] = i2; st_lb[sp + 1] = l2; }
            sp += 2;
#ifdef KD_PROFILE
            ++n_inner; t_inner += wall_clock64() - tp;
#endif
        }
    }
#ifdef KD_PROFILE
    if (lane == 0) printf("kd query scale %d: %.1f us; %d inner nodes %.1f us, %d leaves (load %.1f us), %d pushes %.1f us, %d popped nodes skipped\n", sc,
                          (wall_clock64() - tq0) * 0.01, n_inner, t_inner * 0.01, n_leaf, t_leafload * 0.01, n_push, t_push * 0.01, n_skip);
#endif
    __syncthreads();
    // ---- the 496 kept voxels -> the 16^3 window with the reference's wrap-around placement (Voxel.py:204-214)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r * 64 + lane < 1 || r * 64 + lane > KD_K || hd[r] == KD_INF) continue;   // (positions 1 .. 496)
        const int16_t *v = T.vox + 3 * (int64_t)hx[r];
        const int dx = v[0] - q[0], dy = v[1] - q[1], dz = v[2] - q[2];
        if (dx >= -8 && dx < 8 && dy >= -8 && dy < 8 && dz >= -8 && dz < 8) {
            const int lin = ((dx & 15) << 8) | ((dy & 15) << 4) | (dz & 15);
            atomicOr(&L.patch[lin >> 6], 1ull << (lin & 63));
        }
    }
    __syncthreads();
    const int64_t pw = (int64_t)kp * 3 + sc;
    bits[pw * 64 + lane] = L.patch[lane];
    if (lane == 0) flags[pw] = (uint8_t)((flags[pw] & ~2) | 4);
}

// ---- lists too short for the tree: np.argpartition's introselect on (distance << 10 | list index) words; the order relation looks at
// the distance alone (strict <, like the library's on the float64 row), a swap moves the pair
#define BQ_LESS(a, b) (((a) >> 10) < ((b) >> 10))
#define BQ_SWAP(i, j) do { const unsigned long long t_ = a[i]; a[i] = a[j]; a[j] = t_; } while (0)
__device__ inline void bq_dumb_select(unsigned long long *a, int num, int kth) {
    for (int i = 0; i <= kth; ++i) {
        int minidx = i;
        unsigned long long minval = a[i];
        for (int k = i + 1; k < num; ++k)
            if (BQ_LESS(a[k], minval)) { minidx = k; minval = a[k]; }
        BQ_SWAP(i, minidx);
    }
}
__device__ inline int bq_median5(unsigned long long *a) {
    if (BQ_LESS(a[1], a[0])) BQ_SWAP(1, 0);
    if (BQ_LESS(a[4], a[3])) BQ_SWAP(4, 3);
    if (BQ_LESS(a[3], a[0])) BQ_SWAP(3, 0);
    if (BQ_LESS(a[4], a[1])) BQ_SWAP(4, 1);
    if (BQ_LESS(a[2], a[1])) BQ_SWAP(2, 1);
    if (BQ_LESS(a[3], a[2])) return BQ_LESS(a[3], a[1]) ? 1 : 3;
    return 2;
}
// DEPTH: the median-of-medians fallback selects among num / 5 medians with the same routine -- 993 -> 198 -> 39 -> 7, and seven
// elements never fall back (hh - ll < 5): three nested instances, no recursion on the device
template <int DEPTH>
__device__ void bq_select(unsigned long long *a, const int num, const int kth) {
    int low = 0, high = num - 1;
    if (kth - low < 3) { bq_dumb_select(a, num, kth); return; }
    if (kth == num - 1) {   // (the library's NaN probe for floating point rows: a scan for the LAST maximum)
        int maxidx = low;
        unsigned long long maxval = a[low];
        for (int k = low + 1; k < num; ++k)
            if (!BQ_LESS(a[k], maxval)) { maxidx = k; maxval = a[k]; }
        BQ_SWAP(kth, maxidx);
        return;
    }
    int depth_limit = 2 * (31 - __builtin_clz((unsigned)num));
    while (low + 1 < high) {
        int ll = low + 1, hh = high;
        if (depth_limit > 0 || hh - ll < 5) {
            const int mid = low + (high - low) / 2;
            if (BQ_LESS(a[high], a[mid])) BQ_SWAP(high, mid);
            if (BQ_LESS(a[high], a[low])) BQ_SWAP(high, low);
            if (BQ_LESS(a[low], a[mid])) BQ_SWAP(low, mid);
            BQ_SWAP(mid, low + 1);
        } else {
            int mid = ll;
            if constexpr (DEPTH < 3) {
                unsigned long long *b = a + ll;
                const int nsub = hh - ll, nmed = nsub / 5;
                for (int i = 0, subleft = 0; i < nmed; ++i, subleft += 5) {
                    const int m = bq_median5(b + subleft);
                    const unsigned long long t_ = b[subleft + m]; b[subleft + m] = b[i]; b[i] = t_;
                }
                if (nmed > 2) bq_select<DEPTH + 1>(b, nmed, nmed / 2);
                mid = ll + nmed / 2;
            }
            BQ_SWAP(mid, low);
            --ll; ++hh;
        }
        --depth_limit;
        const unsigned long long pivot = a[low];
        for (;;) {
            do ++ll; while (BQ_LESS(a[ll], pivot));
            do --hh; while (BQ_LESS(pivot, a[hh]));
            if (hh < ll) break;
            BQ_SWAP(hh, ll);
        }
        BQ_SWAP(low, hh);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1 && BQ_LESS(a[high], a[low])) BQ_SWAP(high, low);
}

__global__ void __launch_bounds__(64) k_brute_query(const caelo_kd_set S_) {
    const caelo_kd &kd = S_.k[blockIdx.z];
    const int sc = blockIdx.y;
    const int cnt = min(kd.state[sc], (int)kd.k_cap);
    const int n = kd.state[16 + sc];
    if ((int)blockIdx.x >= cnt || n < KD_K || n >= KD_MIN_N) return;
    const float *__restrict__ pts = S_.pts[blockIdx.z];
    const int pts_ld = S_.pts_ld;
    unsigned long long *__restrict__ bits = S_.bits[blockIdx.z];
    uint8_t *__restrict__ flags = S_.flags[blockIdx.z];
    const caelo_kd_scale T = kd.s[sc];
    __shared__ unsigned long long a[1024];
    __shared__ unsigned long long patch[64];
    const int lane = threadIdx.x;
    const int kp = T.queue[blockIdx.x];
    const double vs = sc == 0 ? 0.02 : (sc == 1 ? 0.02 * 8 : 0.02 * 32);                    // Voxel.py:31
    const int q[3] = {(int)(((double)pts[(size_t)pts_ld * kp] + 99.84) / vs), (int)(((double)pts[(size_t)pts_ld * kp + 1] + 99.84) / vs),
                      (int)(((double)pts[(size_t)pts_ld * kp + 2] + 14.72) / vs)};        // :185,:193 (float64 division, truncation)
    for (int i = lane; i < n; i += 64) {
        const int16_t *v = T.vox + 3 * (int64_t)i;
        const long long dx = q[0] - v[0], dy = q[1] - v[1], dz = q[2] - v[2];
        a[i] = ((unsigned long long)(dx * dx + dy * dy + dz * dz) << 10) | (unsigned long long)i;
    }
    patch[lane] = 0ull;
    __syncthreads();
    if (lane == 0) bq_select<0>(a, n, KD_K - 1);
    __syncthreads();
    for (int i = lane; i < KD_K; i += 64) {
        const int16_t *v = T.vox + 3 * (int64_t)(a[i] & 1023ull);
        const int dx = v[0] - q[0], dy = v[1] - q[1], dz = v[2] - q[2];
        if (dx >= -8 && dx < 8 && dy >= -8 && dy < 8 && dz >= -8 && dz < 8) {
            const int lin = ((dx & 15) << 8) | ((dy & 15) << 4) | (dz & 15);
            atomicOr(&patch[lin >> 6], 1ull << (lin & 63));
        }
    }
    __syncthreads();
    const int64_t pw = (int64_t)kp * 3 + sc;
    bits[pw * 64 + lane] = patch[lane];
    if (lane == 0) flags[pw] = (uint8_t)((flags[pw] & ~2) | 4);
}

}  // namespace

void kd_destroy(caelo_voxmap *m) {
    if (m->kd) {
        if (m->kd->base) (void)hipFree(m->kd->base);
        delete m->kd;
        m->kd = nullptr;
    }
}

static int kd_alloc(caelo_voxmap *m) {
    if (m->kd) return CAELO_OK;
    caelo_kd *kd = new caelo_kd();
    kd->cap = m->max_points;
    kd->k_cap = CAELO_MAX_KEYPTS;
    const size_t per = (size_t)kd->cap * (6 + 4 + 8 + 8 + 4) + 256 * 5;
    const size_t nodes = (size_t)KD_MAX_NODES * (4 + 4 + 6 + 6) + 256 * 4;
    const size_t total = 3 * (per + nodes + (size_t)kd->k_cap * 4 + 256) + 256;
    if (hipMalloc((void **)&kd->base, total) != hipSuccess) {
        delete kd;
        caelo_set_error("kd_alloc: out of device memory");
        return CAELO_ERR_HIP;
    }
    char *p = kd->base;
    auto take = [&p](size_t bytes) { char *r = p; p += (bytes + 255) / 256 * 256; return r; };
    kd->state = (int32_t *)take(256);
    for (int i = 0; i < 3; ++i) {
        caelo_kd_scale &T = kd->s[i];
        T.keys = (unsigned long long *)take((size_t)kd->cap * 8);
        T.keys2 = (unsigned long long *)take((size_t)kd->cap * 8);
        T.ev = (int32_t *)take((size_t)kd->cap * 4);
        T.idx = (int32_t *)take((size_t)kd->cap * 4);
        T.vox = (int16_t *)take((size_t)kd->cap * 6);
        T.start = (int32_t *)take((size_t)KD_MAX_NODES * 4);
        T.end = (int32_t *)take((size_t)KD_MAX_NODES * 4);
        T.lo = (int16_t *)take((size_t)KD_MAX_NODES * 6);
        T.hi = (int16_t *)take((size_t)KD_MAX_NODES * 6);
        T.queue = (int32_t *)take((size_t)kd->k_cap * 4);
    }
    m->kd = kd;
    return CAELO_OK;
}

namespace {
__global__ void k_kd_set_n(int32_t *state, int n0, int n1, int n2) { state[16] = n0; state[17] = n1; state[18] = n2; }
}

// caelo_voxmap_from_lists: keep the three lists in the caller's order (device copies), forget any tree of older lists
int kd_store_lists(caelo_voxmap *m, const int16_t *const lists[3], const int64_t ns[3], hipStream_t s) {
    const int rc = kd_alloc(m);
    if (rc != CAELO_OK) return rc;
    caelo_kd *kd = m->kd;
    CAELO_HIP(hipMemsetAsync(kd->state, 0, 256, s));
    for (int i = 0; i < 3; ++i) {
        if ((1 << kd_levels_of(ns[i])) - 1 > KD_MAX_NODES || ns[i] > kd->cap) { caelo_set_error("kd_store_lists: list too long for the node table"); return CAELO_ERR_CAPACITY; }
        if (ns[i] > 0) CAELO_HIP(hipMemcpyAsync(kd->s[i].vox, lists[i], (size_t)ns[i] * 6, hipMemcpyDeviceToDevice, s));
    }
    k_kd_set_n<<<1, 1, 0, s>>>(kd->state, (int)ns[0], (int)ns[1], (int)ns[2]);
    CAELO_LAUNCH_CHECK();
    m->kd_lists = true;
    return CAELO_OK;
}

// caelo_voxmap_order: the lists are written on the device (export.hip) straight into the tree's storage -- vox_out[scale] [cap][3],
// n_out [3] (device words the build reads); every word of the state is wiped first
int kd_begin_device_lists(caelo_voxmap *m, int16_t *vox_out[3], int32_t **n_out, hipStream_t s) {
    const int rc = kd_alloc(m);
    if (rc != CAELO_OK) return rc;
    caelo_kd *kd = m->kd;
    CAELO_HIP(hipMemsetAsync(kd->state, 0, 256, s));
    for (int i = 0; i < 3; ++i) vox_out[i] = kd->s[i].vox;
    *n_out = kd->state + 16;
    m->kd_lists = true;
    return CAELO_OK;
}

// caelo_patches, after k_patches: the tie-split patches again, in the library's order
int kd_resolve(const caelo_voxmap *m, const float *pts, int pts_ld, int64_t k_max, const int32_t *n_key, uint64_t *bits, uint8_t *flags,
               hipStream_t s) {
    const caelo_voxmap *maps[1] = {m};
    const float *ptss[1] = {pts};
    const int32_t *nks[1] = {n_key};
    uint64_t *bitss[1] = {bits};
    uint8_t *flagss[1] = {flags};
    return kd_resolve_many(1, maps, ptss, pts_ld, k_max, nks, bitss, flagss, s);
}

// the same for n maps (each with its own lists) behind one launch of each kernel; maps without lists are skipped
int kd_resolve_many(int n, const caelo_voxmap *const *maps, const float *const *pts, int pts_ld, int64_t k_max, const int32_t *const *n_key,
                    uint64_t *const *bits, uint8_t *const *flags, hipStream_t s) {
    CAELO_REQUIRE(n >= 1 && n <= CAELO_FB_MAX, "kd_resolve_many: 1 .. 8 maps");
    caelo_kd_set S = {};
    int64_t cap = 0, k_cap = 0;
    for (int i = 0; i < n; ++i) {
        const caelo_voxmap *m = maps[i];
        if (!m->kd || !m->kd_lists) continue;
        S.k[S.n] = *m->kd;
        S.pts[S.n] = pts[i]; S.n_key[S.n] = n_key[i]; S.bits[S.n] = (unsigned long long *)bits[i]; S.flags[S.n] = flags[i];
        cap = m->kd->cap > cap ? m->kd->cap : cap;
        CAELO_REQUIRE(k_cap == 0 || k_cap == m->kd->k_cap, "kd_resolve_many: maps of one set share the queue capacity");
        k_cap = m->kd->k_cap;
        ++S.n;
    }
    if (S.n == 0) return CAELO_OK;
    S.pts_ld = pts_ld;
    for (int64_t k0 = 0; k0 < k_max; k0 += k_cap) {   // (the queues hold k_cap key points: longer point lists go chunk by chunk)
        for (int i = 0; i < S.n; ++i) CAELO_HIP(hipMemsetAsync(S.k[i].state, 0, 12, s));   // queue lengths (the built flags stay)
        k_kd_collect<<<dim3((unsigned)((k_cap * 3 + 255) / 256), 1, S.n), 256, 0, s>>>(S, k0, k_max);
        CAELO_LAUNCH_CHECK();
        k_kd_build_top<<<dim3(3, 1, S.n), KD_T, 0, s>>>(S);
        CAELO_LAUNCH_CHECK();
        k_kd_build_sub<<<dim3(1u << kd_top_levels_of(cap), 3, S.n), KD_TS, 0, s>>>(S);
        CAELO_LAUNCH_CHECK();
        k_kd_query<<<dim3((unsigned)k_cap, 3, S.n), 64, 0, s>>>(S);
        CAELO_LAUNCH_CHECK();
        k_brute_query<<<dim3((unsigned)k_cap, 3, S.n), 64, 0, s>>>(S);   // (lists of 496 .. 993 voxels; every other block leaves at once)
        CAELO_LAUNCH_CHECK();
    }
    return CAELO_OK;
}
