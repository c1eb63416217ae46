/*
 * caelo_oracle.c -- CPU restatement of the CAE-LO hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: a plain-C restatement of the reference's algorithm
 * (SRainGit/CAE-LO, /root/reference) for the path
 *     project -> response CNN -> keypoints -> voxelize -> patches -> encoder -> match.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (libcaelo.so, HIP) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - projection, keypoint rule, voxelization, patch gather, NN match: pinned bit-exactly
 *     against the reference's own Python run in this container (tools/make_goldens.py ->
 *     tests/golden/, checked by tests/test_oracle_golden.py).
 *   - the two CNNs (Keras/TensorFlow conv + dense): PARITY UNPINNED -- TensorFlow/Keras are
 *     not installable here; the restatement follows the documented Keras channels-last
 *     semantics read from the .h5 model_config.
 *
 * Every function cites the reference file:line it follows.  Build: oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EXPORT __attribute__((visibility("default")))

/* ---- constants: SphericalRing.py:28-58 ---------------------------------------------- */
#define N_LINES 64
#define IMG_H 69   /* nLines + SafeEdgeWidth4Top */
#define IMG_W 1800 /* int(2*pi / (0.2*pi/180)) */
#define NET_H 64
#define NET_W 1792 /* ImgW - CropWidth_SphericalRing */
#define RING_C 5

static double deg2rad(void) { return M_PI / 180.0; }

ORC_EXPORT int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_EXPORT void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- ProjectPC2SphericalRing: SphericalRing.py:72-94 --------------------------------- */
/* ring [69][1800][5] f32 zero-initialised here; counter [69][1800] i32.
 * returns 0, or -1 when a point lands on column 1800 (the reference raises IndexError). */
ORC_EXPORT int orc_project(const float *pc, int64_t n, float *ring, int32_t *counter) {
    const double az_res = 0.20 * deg2rad();                       /* :35,:48 */
    const double v_down = -24.8 * deg2rad(), v_up = 2.0 * deg2rad(); /* :49-50 */
    const double v_res = (v_up - v_down) / (N_LINES - 1);         /* :51 */
    const double v_off = -v_down / v_res;                         /* :52 */
    memset(ring, 0, sizeof(float) * IMG_H * IMG_W * RING_C);
    memset(counter, 0, sizeof(int32_t) * IMG_H * IMG_W);
    for (int64_t i = 0; i < n; ++i) {
        const float x = pc[4 * i], y = pc[4 * i + 1], z = pc[4 * i + 2];
        /* :77 LA.norm(PC[:,0:3],axis=1): f32 squares, sequential f32 sum, f32 sqrt */
        float s = x * x;
        s = s + y * y;
        s = s + z * z;
        const float r = sqrtf(s);
        if (r == 0.0f) continue; /* :78-80 (dropping r==0 points whenever present) */
        const int col = (int)((M_PI - atan2((double)y, (double)x)) / az_res); /* :86 */
        const float q = z / r;                                                /* :87 f32 quotient */
        const double beta = asin((double)q);
        const int row = IMG_H - (int)(beta / v_res + v_off); /* :88 */
        if (row < 0 || row >= IMG_H) continue;               /* :89 */
        if (col < 0 || col >= IMG_W) return -1;
        float *px = ring + ((int64_t)row * IMG_W + col) * RING_C;
        px[0] = x; px[1] = y; px[2] = z; px[3] = pc[4 * i + 3]; /* :91 */
        px[4] = r;                                              /* :92 */
        counter[row * IMG_W + col] += 1;                        /* :93 */
    }
    return 0;
}

/* ---- RespondLayer.predict: SphericalRingPCRespondLayer.h5 (Conv2D 3->32 3x3 same relu,
 *      Conv2D 32->8 1x1 relu), called at SphericalRing.py:405-408.  PARITY UNPINNED vs Keras.
 * Canonical summation order (shared bit-for-bit with the HIP kernel):
 *   h[c] = b1[c]; for ky,kx,ci ascending: h[c] = fmaf(in, W1[ky][kx][ci][c], h[c]); relu
 *   o[k] = b2[k]; for c ascending:        o[k] = fmaf(h[c], W2[c][k], o[k]);        relu
 * in  : ring [rows][ring_w][ring_c] (uses rows 0..63, cols 0..1791, channels 0..2)
 * out : [64][1792][8] */
ORC_EXPORT void orc_respond(const float *ring, int ring_w, int ring_c, const float *w1,
                            const float *b1, const float *w2, const float *b2, float *out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < NET_H; ++y) {
        for (int x = 0; x < NET_W; ++x) {
            float h[32];
            for (int c = 0; c < 32; ++c) h[c] = b1[c];
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= NET_H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    if (xx < 0 || xx >= NET_W) continue;
                    const float *px = ring + ((int64_t)yy * ring_w + xx) * ring_c;
                    for (int ci = 0; ci < 3; ++ci) {
                        const float v = px[ci];
                        const float *w = w1 + ((ky * 3 + kx) * 3 + ci) * 32;
                        for (int c = 0; c < 32; ++c) h[c] = fmaf(v, w[c], h[c]);
                    }
                }
            }
            for (int c = 0; c < 32; ++c) h[c] = h[c] > 0.0f ? h[c] : 0.0f;
            float *o = out + ((int64_t)y * NET_W + x) * 8;
            for (int k = 0; k < 8; ++k) {
                float a = b2[k];
                for (int c = 0; c < 32; ++c) a = fmaf(h[c], w2[c * 8 + k], a);
                o[k] = a > 0.0f ? a : 0.0f;
            }
        }
    }
}

/* ---- GetKeyPtsByAE: SphericalRing.py:113-291 ------------------------------------------
 * ring     : [rows][ring_w][ring_c]; dist_channels = ring_c (5 demo mode :414, 3 batch mode
 *            BatchPreprocess.py:97-98,131-136)
 * counter  : [rows][cnt_w] int32 (only >0 is used, :138)
 * resp     : [64][1792][8]
 * outputs  : key_pixels [1024][2] (row,col) ascending by (score, flat index); key_pts [1024][3];
 *            score_map (optional, may be NULL) [64][1792] f32 = masked min-diff (0 where not a
 *            candidate).  returns K (<=1024). */
typedef struct { float score; int32_t idx; } orc_cand_t;
static int cand_cmp(const void *a, const void *b) {
    const orc_cand_t *x = (const orc_cand_t *)a, *y = (const orc_cand_t *)b;
    if (x->score < y->score) return -1;
    if (x->score > y->score) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx); /* stable argsort :194 */
}

ORC_EXPORT int orc_keypoints(const float *ring, int ring_w, int ring_c, const int32_t *counter,
                             int cnt_w, const float *resp, int64_t *key_pixels, float *key_pts,
                             float *score_map) {
    orc_cand_t *cand = (orc_cand_t *)malloc(sizeof(orc_cand_t) * NET_H * NET_W);
    int ncand = 0;
    if (score_map) memset(score_map, 0, sizeof(float) * NET_H * NET_W);
    for (int y = 0; y < NET_H; ++y) {
        for (int x = 0; x < NET_W; ++x) {
            /* SelfMask :163-167 (incl. the column/row mix-up at :166-167), final crop :210-213 */
            if (y < 8 || y >= 56) continue;
            if (x < 8 || x >= NET_W - 8) continue;
            if (x >= 56 && x < 64) continue;
            if (!(counter[y * cnt_w + x] > 0)) continue;
            const float *rp = resp + ((int64_t)y * NET_W + x) * 8;
            int cnt = 0;
            float best = 0.0f;
            int have = 0;
            for (int oy = -2; oy <= 2; ++oy) {
                for (int ox = -2; ox <= 2; ++ox) {
                    if (oy == 0 && ox == 0) continue; /* :170 */
                    const int yy = y + oy, xx = x + ox;
                    const int occ = counter[yy * cnt_w + xx] > 0; /* :156-158,:173: an unoccupied neighbour gets + 1e10 ... */
                    const float *rq = resp + ((int64_t)yy * NET_W + xx) * 8;
                    float s[8];
                    for (int c = 0; c < 8; ++c) {
                        const float d = rq[c] - rp[c]; /* :153-154 neighbour - centre */
                        s[c] = d * d;
                    }
                    /* :159 LA.norm axis=-1, f32: NumPy pairwise block for n == 8 */
                    const float t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                    const float nd = sqrtf(t);
                    /* :179 cp.min over ALL 25 window entries: ... but a NaN norm stays NaN under + 1e10 and makes the minimum NaN whether
                     * the neighbour is occupied or not (so does a NaN in the pixel's own response: every difference is NaN) */
                    if (nd != nd) { best = nd; have = 1; }
                    if (!occ) continue;
                    if (!have || nd < best) { best = nd; have = 1; }
                    ++cnt;                                           /* :182 */
                }
            }
            if (cnt < 5) continue;                 /* :186 */
            if (!((double)best > 0.2)) continue;   /* :126,:199 (f64 compare) */
            /* :197-198 distance over all ring channels, f32 sequential sum */
            const float *px = ring + ((int64_t)y * ring_w + x) * ring_c;
            float d2 = px[0] * px[0];
            for (int c = 1; c < ring_c; ++c) d2 = d2 + px[c] * px[c];
            if (!(sqrtf(d2) >= 10.0f)) continue;
            cand[ncand].score = best;
            cand[ncand].idx = y * NET_W + x;
            ++ncand;
            if (score_map) score_map[y * NET_W + x] = best;
        }
    }
    qsort(cand, ncand, sizeof(orc_cand_t), cand_cmp);
    /* :216,:218  [-nFixedKeyPts-1 : -1]: top 1025 minus the single best */
    int start = ncand - 1025; if (start < 0) start = 0;
    int stop = ncand - 1;     if (stop < 0) stop = 0;
    int k = 0;
    for (int i = start; i < stop; ++i, ++k) {
        const int y = cand[i].idx / NET_W, x = cand[i].idx % NET_W;
        key_pixels[2 * k] = y; key_pixels[2 * k + 1] = x;
        const float *px = ring + ((int64_t)y * ring_w + x) * ring_c;
        key_pts[3 * k] = px[0]; key_pts[3 * k + 1] = px[1]; key_pts[3 * k + 2] = px[2];
    }
    free(cand);
    return k;
}

/* ---- Voxelization: Voxel.py:15-52 (constants), :89-97, :100-173 ------------------------ */
#define VOX_SIZE 0.02
#define BLOCK_REAL 1.28
#define BLOCK_SIZE 64
static const double VIS_L = 99.84, VIS_W = 99.84, VIS_H = 14.72; /* :50-52 (156/2*1.28, 23/2*1.28) */

typedef struct { uint64_t *keys; int32_t *vals; uint64_t mask; } orc_map_t;
static void map_init(orc_map_t *m, int64_t cap_pow2) {
    m->keys = (uint64_t *)malloc(sizeof(uint64_t) * cap_pow2);
    m->vals = (int32_t *)malloc(sizeof(int32_t) * cap_pow2);
    memset(m->keys, 0xff, sizeof(uint64_t) * cap_pow2);
    m->mask = (uint64_t)cap_pow2 - 1;
}
static void map_free(orc_map_t *m) { free(m->keys); free(m->vals); }
static inline uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
/* returns existing value or inserts val and returns -1 */
static inline int32_t map_get_or_put(orc_map_t *m, uint64_t key, int32_t val) {
    uint64_t h = mix64(key) & m->mask;
    for (;;) {
        if (m->keys[h] == key) return m->vals[h];
        if (m->keys[h] == ~0ULL) { m->keys[h] = key; m->vals[h] = val; return -1; }
        h = (h + 1) & m->mask;
    }
}
static inline uint64_t pack3(int64_t x, int64_t y, int64_t z) {
    return ((uint64_t)(x & 0xfffff) << 40) | ((uint64_t)(y & 0xfffff) << 20) | (uint64_t)(z & 0xfffff);
}

/* pc: [n][stride] f32 (first three columns used).  Outputs are int16 [n][3] arrays sized by the
 * caller for n rows; counts returned in n_out[3].  all0 = AllVoxels0 (global scale-0 index,
 * block-grouped first-touch order :161-165), all1/all2 first-touch order :153-158.
 * returns 0, or -1 where the reference would raise IndexError (voxel index outside its block). */
ORC_EXPORT int orc_voxelize(const float *pc, int64_t n, int stride, int16_t *all0, int16_t *all1,
                            int16_t *all2, int64_t *n_out) {
    int64_t cap = 1; while (cap < 2 * n + 16) cap <<= 1;
    orc_map_t mblk, m0, m1, m2;
    map_init(&mblk, cap); map_init(&m0, cap); map_init(&m1, cap); map_init(&m2, cap);
    int32_t *v0 = (int32_t *)malloc(sizeof(int32_t) * 3 * (n + 1));
    int32_t *v0blk = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    int64_t nb = 0, c0 = 0, c1 = 0, c2 = 0;
    int rc = 0;
    const float fl = (float)VIS_L, fw = (float)VIS_W, fh = (float)VIS_H;
    for (int64_t i = 0; i < n; ++i) {
        const float fx = pc[i * stride], fy = pc[i * stride + 1], fz = pc[i * stride + 2];
        if (fabsf(fx) > fl || fabsf(fy) > fw || fabsf(fz) > fh) continue; /* :89-97 (f32 compare) */
        const double x_ = (double)fx + VIS_L, y_ = (double)fy + VIS_W, z_ = (double)fz + VIS_H; /* :118-120 f64 */
        const int bx = (int)(x_ / BLOCK_REAL), by = (int)(y_ / BLOCK_REAL), bz = (int)(z_ / BLOCK_REAL); /* :122-124 */
        int32_t b = map_get_or_put(&mblk, pack3(bx, by, bz), (int32_t)nb); /* :126-132 */
        if (b < 0) b = (int32_t)nb++;
        const int vx = (int)((x_ - bx * BLOCK_REAL) / VOX_SIZE); /* :136-138 */
        const int vy = (int)((y_ - by * BLOCK_REAL) / VOX_SIZE);
        const int vz = (int)((z_ - bz * BLOCK_REAL) / VOX_SIZE);
        if (vx < 0 || vx >= BLOCK_SIZE || vy < 0 || vy >= BLOCK_SIZE || vz < 0 || vz >= BLOCK_SIZE) { rc = -1; break; }
        const int gx = vx + bx * BLOCK_SIZE, gy = vy + by * BLOCK_SIZE, gz = vz + bz * BLOCK_SIZE; /* :143 */
        if (map_get_or_put(&m0, pack3(gx, gy, gz), 1) >= 0) continue; /* :139-140 -- note: skips layers 1/2 too */
        v0[3 * c0] = gx; v0[3 * c0 + 1] = gy; v0[3 * c0 + 2] = gz; v0blk[c0] = b; ++c0;
        const int x1 = (int)(x_ / (VOX_SIZE * 8)), y1 = (int)(y_ / (VOX_SIZE * 8)), z1 = (int)(z_ / (VOX_SIZE * 8));   /* :147-149 */
        const int x2 = (int)(x_ / (VOX_SIZE * 32)), y2 = (int)(y_ / (VOX_SIZE * 32)), z2 = (int)(z_ / (VOX_SIZE * 32)); /* :150-152 */
        if (map_get_or_put(&m1, pack3(x1, y1, z1), 1) < 0) { all1[3 * c1] = (int16_t)x1; all1[3 * c1 + 1] = (int16_t)y1; all1[3 * c1 + 2] = (int16_t)z1; ++c1; }
        if (map_get_or_put(&m2, pack3(x2, y2, z2), 1) < 0) { all2[3 * c2] = (int16_t)x2; all2[3 * c2 + 1] = (int16_t)y2; all2[3 * c2 + 2] = (int16_t)z2; ++c2; }
    }
    if (rc == 0) {
        /* :161-165 concatenate per block in block first-touch order (stable counting sort) */
        int64_t *start = (int64_t *)calloc(nb + 1, sizeof(int64_t));
        for (int64_t j = 0; j < c0; ++j) start[v0blk[j] + 1]++;
        for (int64_t b = 0; b < nb; ++b) start[b + 1] += start[b];
        for (int64_t j = 0; j < c0; ++j) {
            const int64_t d = start[v0blk[j]]++;
            all0[3 * d] = (int16_t)v0[3 * j]; all0[3 * d + 1] = (int16_t)v0[3 * j + 1]; all0[3 * d + 2] = (int16_t)v0[3 * j + 2];
        }
        free(start);
    }
    n_out[0] = c0; n_out[1] = c1; n_out[2] = c2;
    free(v0); free(v0blk);
    map_free(&mblk); map_free(&m0); map_free(&m1); map_free(&m2);
    return rc;
}

/* ---- GetPatchesList: Voxel.py:177-216 ---------------------------------------------------
 * Bit-packed patches: patch[ix][iy][iz] (ix,iy,iz in 0..15, the *wrapped* index d mod 16 of
 * :213-214) lives at bit ((iy&3)*16 + iz) of 64-bit word (ix*4 + iy/4): 64 words per patch.
 * out bits  : [K][64] u64 for ONE scale.
 * out flags : [K] u8, bit0 = truncated (the 496-NN cap dropped >= 1 in-window voxel),
 *             bit1 = ambiguous (the cut fell inside a class of equidistant voxels so which ones
 *             survive depends on sklearn's kd-tree tie order; canonical rule used here and in
 *             the HIP kernel: ties at the cut distance are kept in ascending (x,y,z) order).
 * returns 0, or -1 when the scale holds < 496 voxels (sklearn raises ValueError :195-196). */
static int cmp_u64(const void *a, const void *b) {
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}
static int64_t lower_bound_u64(const uint64_t *a, int64_t n, uint64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

/* ---- the 496-nearest cut when it falls INSIDE a class of equidistant voxels (Voxel.py:182,195-196) ----------------------
 * NearestNeighbors(n_neighbors=496, algorithm='auto').fit(A).kneighbors(KeyVoxels): which members of the cut class are
 * returned is decided by scikit-learn's kd-tree (scikit-learn 0.24.2, pinned by import in tools/make_goldens.py, which
 * compares the tree's index array and every truncated patch with the library's own).  Restated from the library's
 * documented algorithm and validated against it (it is a third-party dependency, not part of /root/reference):
 *   fit     'auto' -> kd_tree iff n_samples // 2 > n_neighbors, i.e. n >= 994; otherwise brute force, whose tie order is
 *           NumPy's argpartition (introselect), restated further down (orc_brute_query).
 *   build   leaf_size 30; n_levels = int(log2(max(1, (n - 1) / 30)) + 1), n_nodes = 2^n_levels - 1; node i owns
 *           idx[start, end); a node with children 2i+1, 2i+2 splits on the dimension of largest spread (first of equals)
 *           at n / 2 by a quickselect whose partition is Lomuto's with the LAST element as pivot (strict <) -- the
 *           resulting order of idx inside every node is part of the contract.
 *   query   depth first from the root: a node whose lower bound (distance to its bounding box) exceeds the heap's largest
 *           distance is skipped; a leaf pushes its points in idx order; an inner node visits the child with the smaller
 *           lower bound first (<=: the left one on equality).  The heap holds the 496 best (max-heap on the squared
 *           distance, initially +inf); a candidate is rejected when its distance is >= the largest, otherwise it replaces
 *           the root and is sifted down (child choice: the first child when dist[c1] >= dist[c2]).
 * Coordinates are small integers, so every squared distance is exact: ties are exact ties. */
typedef struct {
    int64_t n;
    int n_nodes;
    int32_t *idx;             /* [n] */
    int32_t *start, *end;     /* [n_nodes] */
    uint8_t *leaf;            /* [n_nodes] */
    int16_t *lo, *hi;         /* [n_nodes][3] bounding boxes */
    const int16_t *vox;       /* [n][3], not owned */
} orc_kdtree_t;

static void kd_build_rec(orc_kdtree_t *t, int node, int64_t s, int64_t e) {
    const int16_t *X = t->vox;
    int32_t *a = t->idx + s;
    const int64_t m = e - s;
    int16_t lo[3] = {32767, 32767, 32767}, hi[3] = {-32768, -32768, -32768};
    for (int64_t i = 0; i < m; ++i)
        for (int j = 0; j < 3; ++j) {
            const int16_t v = X[3 * (int64_t)a[i] + j];
            if (v < lo[j]) lo[j] = v;
            if (v > hi[j]) hi[j] = v;
        }
    t->start[node] = (int32_t)s; t->end[node] = (int32_t)e;
    for (int j = 0; j < 3; ++j) { t->lo[3 * node + j] = lo[j]; t->hi[3 * node + j] = hi[j]; }
    if (2 * node + 1 >= t->n_nodes || m < 2) { t->leaf[node] = 1; return; }
    t->leaf[node] = 0;
    int jmax = 0, spread = 0;
    for (int j = 0; j < 3; ++j) if (hi[j] - lo[j] > spread) { spread = hi[j] - lo[j]; jmax = j; }
    const int64_t nmid = m / 2;
    int64_t left = 0, right = m - 1;
    for (;;) {
        int64_t mid = left;
        const int16_t d2 = X[3 * (int64_t)a[right] + jmax];
        for (int64_t i = left; i < right; ++i)
            if (X[3 * (int64_t)a[i] + jmax] < d2) { const int32_t tmp = a[i]; a[i] = a[mid]; a[mid] = tmp; ++mid; }
        { const int32_t tmp = a[mid]; a[mid] = a[right]; a[right] = tmp; }
        if (mid == nmid) break;
        if (mid < nmid) left = mid + 1; else right = mid - 1;
    }
    kd_build_rec(t, 2 * node + 1, s, s + nmid);
    kd_build_rec(t, 2 * node + 2, s + nmid, e);
}

static orc_kdtree_t *kd_build(const int16_t *vox, int64_t n) {
    orc_kdtree_t *t = (orc_kdtree_t *)calloc(1, sizeof(orc_kdtree_t));
    const double q = (double)(n - 1) / 30.0;
    const int n_levels = (int)(log2(q > 1.0 ? q : 1.0) + 1.0);
    t->n = n; t->n_nodes = (1 << n_levels) - 1; t->vox = vox;
    t->idx = (int32_t *)malloc(sizeof(int32_t) * n);
    for (int64_t i = 0; i < n; ++i) t->idx[i] = (int32_t)i;
    t->start = (int32_t *)malloc(sizeof(int32_t) * t->n_nodes); t->end = (int32_t *)malloc(sizeof(int32_t) * t->n_nodes);
    t->leaf = (uint8_t *)malloc(t->n_nodes);
    t->lo = (int16_t *)malloc(sizeof(int16_t) * 3 * t->n_nodes); t->hi = (int16_t *)malloc(sizeof(int16_t) * 3 * t->n_nodes);
    kd_build_rec(t, 0, 0, n);
    return t;
}
static void kd_free(orc_kdtree_t *t) { if (t) { free(t->idx); free(t->start); free(t->end); free(t->leaf); free(t->lo); free(t->hi); free(t); } }

#define KD_K 496
typedef struct { int64_t dist[KD_K]; int32_t ind[KD_K]; } kd_heap_t;   /* squared distances; INT64_MAX stands for +inf */
static void kd_push(kd_heap_t *h, int64_t val, int32_t i_val) {
    if (val >= h->dist[0]) return;
    int i = 0;
    for (;;) {
        const int c1 = 2 * i + 1, c2 = c1 + 1;
        int sw;
        if (c1 >= KD_K) break;
        else if (c2 >= KD_K) { if (h->dist[c1] > val) sw = c1; else break; }
        else if (h->dist[c1] >= h->dist[c2]) { if (val < h->dist[c1]) sw = c1; else break; }
        else { if (val < h->dist[c2]) sw = c2; else break; }
        h->dist[i] = h->dist[sw]; h->ind[i] = h->ind[sw];
        i = sw;
    }
    h->dist[i] = val; h->ind[i] = i_val;
}
static int64_t kd_min_rdist(const orc_kdtree_t *t, int node, const int q[3]) {
    int64_t r = 0;
    for (int j = 0; j < 3; ++j) {
        const int d_lo = t->lo[3 * node + j] - q[j], d_hi = q[j] - t->hi[3 * node + j];
        const int d = d_lo > 0 ? d_lo : (d_hi > 0 ? d_hi : 0);
        r += (int64_t)d * d;
    }
    return r;
}
static void kd_query_rec(const orc_kdtree_t *t, int node, const int q[3], kd_heap_t *h, int64_t lb) {
    if (lb > h->dist[0]) return;
    if (t->leaf[node]) {
        for (int32_t i = t->start[node]; i < t->end[node]; ++i) {
            const int32_t p = t->idx[i];
            int64_t d = 0;
            for (int j = 0; j < 3; ++j) { const int e = q[j] - t->vox[3 * (int64_t)p + j]; d += (int64_t)e * e; }
            kd_push(h, d, p);
        }
        return;
    }
    const int i1 = 2 * node + 1, i2 = i1 + 1;
    const int64_t l1 = kd_min_rdist(t, i1, q), l2 = kd_min_rdist(t, i2, q);
    if (l1 <= l2) { kd_query_rec(t, i1, q, h, l1); kd_query_rec(t, i2, q, h, l2); }
    else { kd_query_rec(t, i2, q, h, l2); kd_query_rec(t, i1, q, h, l1); }
}
/* the 496 neighbours of q as the library returns them (as a set): out[496] voxel list indices */
static void kd_query(const orc_kdtree_t *t, const int q[3], int32_t *out) {
    kd_heap_t h;
    for (int i = 0; i < KD_K; ++i) { h.dist[i] = INT64_MAX; h.ind[i] = 0; }
    kd_query_rec(t, 0, q, &h, kd_min_rdist(t, 0, q));
    memcpy(out, h.ind, sizeof(h.ind));
}
/* test hooks: the tree's index array (compared with KDTree.get_arrays()[1]) and one query */
ORC_EXPORT int orc_kdtree_idx(const int16_t *vox, int64_t n, int32_t *idx_out) {
    orc_kdtree_t *t = kd_build(vox, n);
    memcpy(idx_out, t->idx, sizeof(int32_t) * n);
    const int nn = t->n_nodes;
    kd_free(t);
    return nn;
}
ORC_EXPORT void orc_kdtree_query(const int16_t *vox, int64_t n, const int32_t *q, int64_t nq, int32_t *out) {
    orc_kdtree_t *t = kd_build(vox, n);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < nq; ++i) { const int qq[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]}; kd_query(t, qq, out + KD_K * i); }
    kd_free(t);
}

/* ---- the same cut on a list too short for the kd-tree (496 <= n < 994: `auto` = brute force) -------------------------------
 * scikit-learn 0.24.2 NearestNeighbors.kneighbors, brute branch: squared euclidean distances of the query to every list entry
 * (euclidean_distances(squared=True): XX + YY - 2 X.Y^T in float64 -- voxel indices are small integers, so every term and the
 * sum are exact: the row IS the integer squared distances, in list order), then `np.argpartition(dist, 495, axis=1)[:, :496]`.
 * The later sort by distance only reorders the 496; the patch is a SET.  Which members of the cut class are among the first
 * 496 is decided by NumPy's argpartition = introselect on the index array (NumPy 1.18 .. 1.26: npysort/selection: median of
 * three with the 3-lowest moved to low + 1, unguarded Hoare partition, median of medians of five when 2 * msb(n) partitions
 * made no end of it, selection sort for kth - low < 3).  A third-party dependency, not part of /root/reference: restated from
 * its algorithm and pinned by tools/make_goldens.py against the library itself (np.argpartition of NumPy 1.26.4 on tie-heavy
 * rows, and scikit-learn's brute kneighbors on the truncation fixtures); tests/golden/argpartition_rows.npz holds the vectors.
 * (NumPy >= 2.0 on AVX-512 hosts selects with another algorithm: the contract here is the reference's NumPy.) */
#define ISEL_SWAP(a, b) do { const int32_t t_ = ts[a]; ts[a] = ts[b]; ts[b] = t_; } while (0)
#define ISEL_V(i) v[ts[i]]
static void isel_dumb(const int64_t *v, int32_t *ts, int64_t num, int64_t kth) {
    for (int64_t i = 0; i <= kth; ++i) {
        int64_t minidx = i, minval = ISEL_V(i);
        for (int64_t k = i + 1; k < num; ++k)
            if (ISEL_V(k) < minval) { minidx = k; minval = ISEL_V(k); }
        ISEL_SWAP(i, minidx);
    }
}
static int64_t isel_median5(const int64_t *v, int32_t *ts) {
    if (ISEL_V(1) < ISEL_V(0)) ISEL_SWAP(1, 0);
    if (ISEL_V(4) < ISEL_V(3)) ISEL_SWAP(4, 3);
    if (ISEL_V(3) < ISEL_V(0)) ISEL_SWAP(3, 0);
    if (ISEL_V(4) < ISEL_V(1)) ISEL_SWAP(4, 1);
    if (ISEL_V(2) < ISEL_V(1)) ISEL_SWAP(2, 1);
    if (ISEL_V(3) < ISEL_V(2)) return ISEL_V(3) < ISEL_V(1) ? 1 : 3;
    return 2;
}
static void isel_select(const int64_t *v, int32_t *ts, int64_t num, int64_t kth);
static int64_t isel_mom_calls = 0;   /* test coverage only: how often the median-of-medians fallback ran */
ORC_EXPORT int64_t orc_argpartition_fallbacks(void) { return isel_mom_calls; }
static int64_t isel_median_of_median5(const int64_t *v, int32_t *ts, int64_t num) {
    const int64_t nmed = num / 5;
#pragma omp atomic
    ++isel_mom_calls;
    for (int64_t i = 0, subleft = 0; i < nmed; ++i, subleft += 5) {
        const int64_t m = isel_median5(v, ts + subleft);
        ISEL_SWAP(subleft + m, i);
    }
    if (nmed > 2) isel_select(v, ts, nmed, nmed / 2);
    return nmed / 2;
}
/* one call on a fresh pivot stack (argpartition gives every row its own): the stack is written, never read -- left out */
static void isel_select(const int64_t *v, int32_t *ts, int64_t num, int64_t kth) {
    int64_t low = 0, high = num - 1;
    if (kth - low < 3) { isel_dumb(v, ts + low, high - low + 1, kth - low); return; }
    /* (the "kth == num - 1" shortcut exists for floating point rows only to find NaNs; it is a full scan for the maximum) */
    if (kth == num - 1) {
        int64_t maxidx = low, maxval = ISEL_V(low);
        for (int64_t k = low + 1; k < num; ++k)
            if (!(ISEL_V(k) < maxval)) { maxidx = k; maxval = ISEL_V(k); }
        ISEL_SWAP(kth, maxidx);
        return;
    }
    int depth_limit = 0;
    for (uint64_t n_ = (uint64_t)num; n_ >>= 1;) ++depth_limit;
    depth_limit *= 2;
    for (; low + 1 < high;) {
        int64_t ll = low + 1, hh = high;
        if (depth_limit > 0 || hh - ll < 5) {
            const int64_t mid = low + (high - low) / 2;
            if (ISEL_V(high) < ISEL_V(mid)) ISEL_SWAP(high, mid);
            if (ISEL_V(high) < ISEL_V(low)) ISEL_SWAP(high, low);
            if (ISEL_V(low) < ISEL_V(mid)) ISEL_SWAP(low, mid);   /* the median to low */
            ISEL_SWAP(mid, low + 1);                              /* the 3-lowest to low + 1 */
        } else {
            const int64_t mid = ll + isel_median_of_median5(v, ts + ll, hh - ll);
            ISEL_SWAP(mid, low);
            --ll; ++hh;
        }
        --depth_limit;
        const int64_t pivot = ISEL_V(low);
        for (;;) {
            do ++ll; while (ISEL_V(ll) < pivot);
            do --hh; while (pivot < ISEL_V(hh));
            if (hh < ll) break;
            ISEL_SWAP(hh, ll);
        }
        ISEL_SWAP(low, hh);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1 && ISEL_V(high) < ISEL_V(low)) ISEL_SWAP(high, low);
}
/* np.argpartition(v, kth) of one row of integers (as float64 in the library: the same order): idx_out [num] */
ORC_EXPORT void orc_argpartition(const int64_t *v, int64_t num, int64_t kth, int32_t *idx_out) {
    for (int64_t i = 0; i < num; ++i) idx_out[i] = (int32_t)i;
    isel_select(v, idx_out, num, kth);
}
/* the brute branch for queries q [nq][3]: out [nq][496] = argpartition(dist row, 495)[:496] (unsorted: a set) */
ORC_EXPORT void orc_brute_query(const int16_t *vox, int64_t n, const int32_t *q, int64_t nq, int32_t *out) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < nq; ++i) {
        int64_t *d = (int64_t *)malloc(sizeof(int64_t) * n);
        int32_t *ts = (int32_t *)malloc(sizeof(int32_t) * n);
        for (int64_t j = 0; j < n; ++j) {
            const int64_t dx = vox[3 * j] - q[3 * i], dy = vox[3 * j + 1] - q[3 * i + 1], dz = vox[3 * j + 2] - q[3 * i + 2];
            d[j] = dx * dx + dy * dy + dz * dz;
        }
        orc_argpartition(d, n, KD_K - 1, ts);
        memcpy(out + KD_K * i, ts, sizeof(int32_t) * KD_K);
        free(d); free(ts);
    }
}

/* flags per patch: 1 = truncated (an in-window voxel lost to the 496-nearest cut); 2 = the cut splits a class of equidistant
 * voxels that has in-window members and the patch is still on the canonical rule (ascending (x, y, z)) -- never left set by this
 * function; 4 = such a split resolved in the library's own order (scikit-learn's kd-tree from 994 voxels on, NumPy's argpartition
 * below): equal to the reference. */
ORC_EXPORT int orc_patches(const float *pts, int64_t k, const int16_t *vox, int64_t nvox, int scale,
                           uint64_t *bits, uint8_t *flags) {
    if (nvox < 496) return -1;
    const double vs = scale == 0 ? VOX_SIZE : (scale == 1 ? VOX_SIZE * 8 : VOX_SIZE * 32); /* :31 */
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * nvox);
    for (int64_t i = 0; i < nvox; ++i) keys[i] = pack3(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]);
    qsort(keys, nvox, sizeof(uint64_t), cmp_u64);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t p = 0; p < k; ++p) {
        /* :185,:193 KeyVoxels = int32((Pts + Visible*) / VoxelSizes[s]) in f64 */
        const int kx = (int)(((double)pts[3 * p] + VIS_L) / vs);
        const int ky = (int)(((double)pts[3 * p + 1] + VIS_W) / vs);
        const int kz = (int)(((double)pts[3 * p + 2] + VIS_H) / vs);
        /* all occupied voxels with |d| <= 13 per axis and d2 <= 192 (= 3*8^2, the farthest
         * in-window offset); histogram by squared distance */
        int hist[193]; memset(hist, 0, sizeof(hist));
        int win_d2[4096]; int win_lin[4096]; uint64_t win_key[4096]; int nwin = 0;
        for (int dx = -13; dx <= 13; ++dx) {
            for (int dy = -13; dy <= 13; ++dy) {
                const int x = kx + dx, y = ky + dy;
                if (x < 0 || y < 0) continue;
                int zlo = kz - 13; if (zlo < 0) zlo = 0;
                const uint64_t klo = pack3(x, y, zlo), khi = pack3(x, y, kz + 13);
                for (int64_t i = lower_bound_u64(keys, nvox, klo); i < nvox && keys[i] <= khi; ++i) {
                    const int dz = (int)(keys[i] & 0xfffff) - kz;
                    const int d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 > 192) continue;
                    hist[d2]++;
                    if (dx >= -8 && dx < 8 && dy >= -8 && dy < 8 && dz >= -8 && dz < 8) { /* :204-210 */
                        win_d2[nwin] = d2;
                        win_lin[nwin] = ((dx & 15) << 8) | ((dy & 15) << 4) | (dz & 15); /* :213-214 wrap */
                        win_key[nwin] = keys[i];
                        ++nwin;
                    }
                }
            }
        }
        /* 496 nearest (:182,:195-196): find the cut class */
        int cum = 0, cut = 193, room = 0;
        for (int d2 = 0; d2 <= 192; ++d2) {
            if (cum + hist[d2] > 496) { cut = d2; room = 496 - cum; break; }
            cum += hist[d2];
        }
        uint64_t *w = bits + p * 64; memset(w, 0, 64 * sizeof(uint64_t));
        uint8_t fl = 0;
        /* in-window members of the cut class, ascending key */
        int ncut = 0; uint64_t cutkeys[4096]; int cutlin[4096];
        for (int i = 0; i < nwin; ++i) {
            if (win_d2[i] < cut) { w[win_lin[i] >> 6] |= 1ULL << (win_lin[i] & 63); }
            else if (win_d2[i] == cut) { cutkeys[ncut] = win_key[i]; cutlin[ncut] = win_lin[i]; ++ncut; }
            else fl |= 1;
        }
        if (ncut > 0) {
            /* canonical tie rule: keep the `room` smallest keys of the WHOLE class; we only know
             * the in-window ones exactly, the rest of the class must be enumerated too */
            /* enumerate class members (all voxels at d2 == cut) to rank the in-window ones */
            int kept = 0;
            if (room > 0) {
                /* count class members with key smaller than each in-window member */
                for (int a = 0; a < ncut; ++a) {
                    int rank = 0;
                    for (int dx = -13; dx <= 13; ++dx) for (int dy = -13; dy <= 13; ++dy) {
                        const int rem = cut - dx * dx - dy * dy; if (rem < 0) continue;
                        const int dz = (int)(sqrt((double)rem) + 0.5); if (dz * dz != rem) continue;
                        for (int sgn = -1; sgn <= 1; sgn += 2) {
                            if (dz == 0 && sgn == 1) continue;
                            const int x = kx + dx, y = ky + dy, z = kz + sgn * dz;
                            if (x < 0 || y < 0 || z < 0) continue;
                            const uint64_t key = pack3(x, y, z);
                            const int64_t i = lower_bound_u64(keys, nvox, key);
                            if (i < nvox && keys[i] == key && key < cutkeys[a]) ++rank;
                        }
                    }
                    if (rank < room) { w[cutlin[a] >> 6] |= 1ULL << (cutlin[a] & 63); ++kept; }
                }
            }
            if (kept < ncut) fl |= 1;
            fl |= 2; /* the cut class contains in-window voxels and is split: tie-order dependent */
            if (room == 0) fl &= ~2, fl |= 1; /* whole class dropped: not ambiguous, just truncated */
        }
        flags[p] = fl;
    }
    free(keys);
    /* ---- tie-ambiguous patches in the library's own order (n >= 994: 'auto' picks the kd-tree; below: brute force) */
    {
        int any = 0;
        for (int64_t p = 0; p < k; ++p) any |= flags[p] & 2;
        if (any) {
            const int tree = nvox / 2 > KD_K;
            orc_kdtree_t *t = tree ? kd_build(vox, nvox) : NULL;
#pragma omp parallel for schedule(dynamic, 1)
            for (int64_t p = 0; p < k; ++p) {
                if (!(flags[p] & 2)) continue;
                const int q[3] = {(int)(((double)pts[3 * p] + VIS_L) / vs), (int)(((double)pts[3 * p + 1] + VIS_W) / vs),
                                  (int)(((double)pts[3 * p + 2] + VIS_H) / vs)};
                int32_t nb[KD_K];
                if (tree) kd_query(t, q, nb);
                else { const int32_t q32[3] = {q[0], q[1], q[2]}; orc_brute_query(vox, nvox, q32, 1, nb); }
                uint64_t *w = bits + p * 64; memset(w, 0, 64 * sizeof(uint64_t));
                for (int i = 0; i < KD_K; ++i) {
                    const int dx = vox[3 * (int64_t)nb[i]] - q[0], dy = vox[3 * (int64_t)nb[i] + 1] - q[1], dz = vox[3 * (int64_t)nb[i] + 2] - q[2];
                    if (dx >= -8 && dx < 8 && dy >= -8 && dy < 8 && dz >= -8 && dz < 8) {
                        const int lin = ((dx & 15) << 8) | ((dy & 15) << 4) | (dz & 15);
                        w[lin >> 6] |= 1ULL << (lin & 63);
                    }
                }
                flags[p] = (uint8_t)((flags[p] & ~2) | 4);
            }
            if (t) kd_free(t);
        }
    }
    return 0;
}

/* ---- PatchEncoder.predict: EncoderModel4VoxelPatch.h5 via GetFeaturesFromPatches Match.py:130-135
 * Conv3D(1->8)tanh, MaxPool2, Conv3D(8->16)tanh, MaxPool2, Conv3D(16->32)tanh, Flatten(x,y,z,c),
 * Dense(200)tanh, Dense(20)tanh.  Keras channels-last, 'same' zero padding, cross-correlation.
 * PARITY UNPINNED vs Keras (see header).  bits: [n][64] u64 packed patches; out: [n][out_stride]
 * written at columns [col0, col0+20). */
typedef struct {
    const float *w1, *b1; /* [27][1][8],  [8]  */
    const float *w2, *b2; /* [27][8][16], [16] */
    const float *w3, *b3; /* [27][16][32],[32] */
    const float *wd1, *bd1; /* [2048][200], [200] */
    const float *wd2, *bd2; /* [200][20], [20] */
} orc_enc_weights_t;

static void conv3d_same(const float *in, int D, int cin, const float *w, const float *b, int cout, float *out) {
    for (int x = 0; x < D; ++x) for (int y = 0; y < D; ++y) for (int z = 0; z < D; ++z) {
        float acc[32];
        for (int o = 0; o < cout; ++o) acc[o] = b[o];
        for (int kx = 0; kx < 3; ++kx) { const int xx = x + kx - 1; if (xx < 0 || xx >= D) continue;
        for (int ky = 0; ky < 3; ++ky) { const int yy = y + ky - 1; if (yy < 0 || yy >= D) continue;
        for (int kz = 0; kz < 3; ++kz) { const int zz = z + kz - 1; if (zz < 0 || zz >= D) continue;
            const float *pi = in + (((int64_t)xx * D + yy) * D + zz) * cin;
            const float *pw = w + (int64_t)((kx * 3 + ky) * 3 + kz) * cin * cout;
            for (int c = 0; c < cin; ++c) {
                const float v = pi[c];
                if (v == 0.0f) continue;
                const float *pwc = pw + (int64_t)c * cout;
                for (int o = 0; o < cout; ++o) acc[o] += v * pwc[o];
            }
        }}}
        float *po = out + (((int64_t)x * D + y) * D + z) * cout;
        for (int o = 0; o < cout; ++o) po[o] = tanhf(acc[o]);
    }
}
static void maxpool2(const float *in, int D, int c, float *out) {
    const int H = D / 2;
    for (int x = 0; x < H; ++x) for (int y = 0; y < H; ++y) for (int z = 0; z < H; ++z) for (int k = 0; k < c; ++k) {
        float m = -INFINITY;
        for (int a = 0; a < 2; ++a) for (int bb = 0; bb < 2; ++bb) for (int cc = 0; cc < 2; ++cc) {
            const float v = in[((((int64_t)(2 * x + a)) * D + (2 * y + bb)) * D + (2 * z + cc)) * c + k];
            if (v > m) m = v;
        }
        out[(((int64_t)x * H + y) * H + z) * c + k] = m;
    }
}

static void encode_layers(const uint64_t *bits, int64_t n, const orc_enc_weights_t *W, float *out, int out_stride, int col0,
                          float *p2_out, float *f3_out, float *h_out);
ORC_EXPORT void orc_encode(const uint64_t *bits, int64_t n, const orc_enc_weights_t *W, float *out,
                           int out_stride, int col0) {
    encode_layers(bits, n, W, out, out_stride, col0, NULL, NULL, NULL);
}
/* The same network with its intermediate activations written out (test aid: the per-layer error budget of the HIP
 * encoder): p2 [n][4][4][4][16] after the second pooling, f3 [n][2048] after conv3's tanh (flatten order), h [n][200]
 * after Dense(200)'s tanh.  Any of the three may be NULL. */
ORC_EXPORT void orc_encode_layers(const uint64_t *bits, int64_t n, const orc_enc_weights_t *W, float *out, int out_stride,
                                  float *p2_out, float *f3_out, float *h_out) {
    encode_layers(bits, n, W, out, out_stride, 0, p2_out, f3_out, h_out);
}
static void encode_layers(const uint64_t *bits, int64_t n, const orc_enc_weights_t *W, float *out, int out_stride, int col0,
                          float *p2_out, float *f3_out, float *h_out) {
#pragma omp parallel
    {
        float *p0 = (float *)malloc(sizeof(float) * 4096);
        float *a1 = (float *)malloc(sizeof(float) * 4096 * 8);
        float *q1 = (float *)malloc(sizeof(float) * 512 * 8);
        float *a2 = (float *)malloc(sizeof(float) * 512 * 16);
        float *q2 = (float *)malloc(sizeof(float) * 64 * 16);
        float *a3 = (float *)malloc(sizeof(float) * 64 * 32);
        float h[200];
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < n; ++p) {
            const uint64_t *w = bits + p * 64;
            for (int lin = 0; lin < 4096; ++lin) p0[lin] = (float)((w[lin >> 6] >> (lin & 63)) & 1);
            conv3d_same(p0, 16, 1, W->w1, W->b1, 8, a1);
            maxpool2(a1, 16, 8, q1);
            conv3d_same(q1, 8, 8, W->w2, W->b2, 16, a2);
            maxpool2(a2, 8, 16, q2);
            conv3d_same(q2, 4, 16, W->w3, W->b3, 32, a3); /* flatten order x,y,z,c == memory order */
            for (int j = 0; j < 200; ++j) h[j] = W->bd1[j];
            for (int i = 0; i < 2048; ++i) {
                const float v = a3[i];
                const float *row = W->wd1 + (int64_t)i * 200;
                for (int j = 0; j < 200; ++j) h[j] += v * row[j];
            }
            for (int j = 0; j < 200; ++j) h[j] = tanhf(h[j]);
            if (p2_out) memcpy(p2_out + p * 1024, q2, sizeof(float) * 1024);
            if (f3_out) memcpy(f3_out + p * 2048, a3, sizeof(float) * 2048);
            if (h_out) memcpy(h_out + p * 200, h, sizeof(float) * 200);
            float *o = out + p * out_stride + col0;
            for (int j = 0; j < 20; ++j) {
                float acc = W->bd2[j];
                for (int i = 0; i < 200; ++i) acc += h[i] * W->wd2[i * 20 + j];
                o[j] = tanhf(acc);
            }
        }
        free(p0); free(a1); free(q1); free(a2); free(q2); free(a3);
    }
}

/* ---- BASELINE.json configs[4]: the "2x voxel-patch resolution" stress case -----
 * NOT a reference code path: the reference has PatchSize=16 only (Voxel.py:31-33).  Config 5 keeps
 * GetPatchesList's rule (Voxel.py:177-216) with PatchSize=32: window [-16,16)^3 around the key voxel,
 * wrap-around placement d mod 32 (:213-214), and the 496-nearest cap DISABLED (a 32^3 window holds up
 * to 32768 voxels; n_neighbors=496 would truncate nearly every dense patch, and the cap's tie order is
 * sklearn-defined) -- every occupied voxel inside the window is set.
 * bits: [K][512] u64 for ONE scale; patch[ix][iy][iz] at bit (lin & 63) of word (lin >> 6),
 * lin = (ix*32 + iy)*32 + iz. */
ORC_EXPORT int orc_patches32(const float *pts, int64_t k, const int16_t *vox, int64_t nvox, int scale,
                             uint64_t *bits) {
    const double vs = scale == 0 ? VOX_SIZE : (scale == 1 ? VOX_SIZE * 8 : VOX_SIZE * 32);
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (nvox + 1));
    for (int64_t i = 0; i < nvox; ++i) keys[i] = pack3(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]);
    qsort(keys, nvox, sizeof(uint64_t), cmp_u64);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t p = 0; p < k; ++p) {
        const int kx = (int)(((double)pts[3 * p] + VIS_L) / vs);
        const int ky = (int)(((double)pts[3 * p + 1] + VIS_W) / vs);
        const int kz = (int)(((double)pts[3 * p + 2] + VIS_H) / vs);
        uint64_t *w = bits + p * 512; memset(w, 0, 512 * sizeof(uint64_t));
        for (int dx = -16; dx < 16; ++dx) for (int dy = -16; dy < 16; ++dy) {
            const int x = kx + dx, y = ky + dy;
            if (x < 0 || y < 0) continue;
            int zlo = kz - 16; if (zlo < 0) zlo = 0;
            if (kz + 15 < 0) continue;
            const uint64_t klo = pack3(x, y, zlo), khi = pack3(x, y, kz + 15);
            for (int64_t i = lower_bound_u64(keys, nvox, klo); i < nvox && keys[i] <= khi; ++i) {
                const int dz = (int)(keys[i] & 0xfffff) - kz;
                const int lin = ((dx & 31) << 10) | ((dy & 31) << 5) | (dz & 31);
                w[lin >> 6] |= 1ULL << (lin & 63);
            }
        }
    }
    free(keys);
    return 0;
}

/* Config-5 encoder: the same layer stack as orc_encode on a 32^3 patch -- Conv3D(1->8) tanh, MaxPool2
 * (16^3), Conv3D(8->16) tanh, MaxPool2 (8^3), Conv3D(16->32) tanh, Flatten (8*8*8*32 = 16384),
 * Dense(200) tanh, Dense(20) tanh.  Conv kernels, dense_2 and all biases come from the .h5; W->wd1 is
 * the seeded stand-in [16384][200] (no trained dense_1 exists at this size). */
ORC_EXPORT void orc_encode32(const uint64_t *bits, int64_t n, const orc_enc_weights_t *W, float *out,
                             int out_stride, int col0) {
#pragma omp parallel
    {
        float *p0 = (float *)malloc(sizeof(float) * 32768);
        float *a1 = (float *)malloc(sizeof(float) * 32768 * 8);
        float *q1 = (float *)malloc(sizeof(float) * 4096 * 8);
        float *a2 = (float *)malloc(sizeof(float) * 4096 * 16);
        float *q2 = (float *)malloc(sizeof(float) * 512 * 16);
        float *a3 = (float *)malloc(sizeof(float) * 512 * 32);
        float h[200];
#pragma omp for schedule(dynamic, 1)
        for (int64_t p = 0; p < n; ++p) {
            const uint64_t *w = bits + p * 512;
            for (int lin = 0; lin < 32768; ++lin) p0[lin] = (float)((w[lin >> 6] >> (lin & 63)) & 1);
            conv3d_same(p0, 32, 1, W->w1, W->b1, 8, a1);
            maxpool2(a1, 32, 8, q1);
            conv3d_same(q1, 16, 8, W->w2, W->b2, 16, a2);
            maxpool2(a2, 16, 16, q2);
            conv3d_same(q2, 8, 16, W->w3, W->b3, 32, a3);
            for (int j = 0; j < 200; ++j) h[j] = W->bd1[j];
            for (int i = 0; i < 16384; ++i) {
                const float v = a3[i];
                const float *row = W->wd1 + (int64_t)i * 200;
                for (int j = 0; j < 200; ++j) h[j] += v * row[j];
            }
            for (int j = 0; j < 200; ++j) h[j] = tanhf(h[j]);
            float *o = out + p * out_stride + col0;
            for (int j = 0; j < 20; ++j) {
                float acc = W->bd2[j];
                for (int i = 0; i < 200; ++i) acc += h[i] * W->wd2[i * 20 + j];
                o[j] = tanhf(acc);
            }
        }
        free(p0); free(a1); free(q1); free(a2); free(q2); free(a3);
    }
}

/* ---- NN match: Match.py:257-258  cdist(Codes0,Codes1) f64 + argmin(axis=0), first min wins.
 * SciPy's euclidean kernel: sequential f64 sum of squared differences, sqrt. */
ORC_EXPORT void orc_match(const float *f0, int64_t k0, const float *f1, int64_t k1, int dim, int64_t *pair_idx,
                          double *min_dist) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < k1; ++j) {
        double best = INFINITY; int64_t bi = 0;
        for (int64_t i = 0; i < k0; ++i) {
            double s = 0.0;
            for (int c = 0; c < dim; ++c) {
                const double d = (double)f0[i * dim + c] - (double)f1[j * dim + c];
                s += d * d;
            }
            const double dd = sqrt(s);
            if (dd < best) { best = dd; bi = i; }
        }
        pair_idx[j] = bi;
        if (min_dist) min_dist[j] = best;
    }
}

/* ---- RANSAC residual count for one hypothesis: Match.py:191-194 (f32) ------------------- */
ORC_EXPORT int orc_count_inliers(const float *p0, const float *p1, int64_t n, const float *R, const float *T,
                                 float thr, uint8_t *mask) {
    int cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        float s = 0.0f;
        for (int r = 0; r < 3; ++r) {
            float v = R[3 * r] * p1[3 * i];
            v += R[3 * r + 1] * p1[3 * i + 1];
            v += R[3 * r + 2] * p1[3 * i + 2];
            v += T[r];
            const float d = p0[3 * i + r] - v;
            s += d * d;
        }
        const int in = sqrtf(s) < thr;
        if (mask) mask[i] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}
