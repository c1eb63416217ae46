/*
 * caelo.h -- C ABI of libcaelo.so, the MI355X-native (gfx950, HIP) CAE-LO feature-and-matching
 * engine.  The reference (SRainGit/CAE-LO) has no FFI: its boundary is a set of module-level
 * Python functions on NumPy arrays.  Each entry point below replaces the
 * device work behind one of those functions; caelo/api.py (ctypes) keeps the Python names,
 * argument order and return tuples.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer unless the name ends in _host; the caller owns all
 *     buffers (torch allocates them); no hidden allocation outside caelo_create /
 *     caelo_voxmap_create / caelo_set_*_weights;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 on success, negative caelo_status otherwise; caelo_last_error() gives text;
 *   - data-dependent conditions the reference reports as Python exceptions are written to a
 *     device-side int32 status word (CAELO_ST_* bit flags) so no call forces a host sync.
 */
#ifndef CAELO_H
#define CAELO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAELO_ABI_VERSION 5   /* 5: caelo_host_unbind_blas / _blas_probe / _bound_violations, caelo_host_random_sample, caelo_seqloader_*, caelo_patches_many, caelo_pipeline_run_uploading; 4: caelo_voxmap_export never waits for the device (a count above capacity is the overflow report), caelo_voxmap_order; 3: certificates, caelo_host_* */

/* geometry fixed by the reference: SphericalRing.py:28-58, Voxel.py:15-52 */
#define CAELO_RING_H 69
#define CAELO_RING_W 1800
#define CAELO_RING_C 5
#define CAELO_NET_H 64
#define CAELO_NET_W 1792
#define CAELO_RESP_C 8
#define CAELO_MAX_KEYPTS 1024
#define CAELO_PATCH_WORDS 64 /* 16^3 bits = 64 x u64, bit ((iy&3)*16+iz) of word (ix*4+iy/4) */
#define CAELO_DESC_DIM 20    /* per scale; 3 scales -> 60 */
#define CAELO_RANSAC_MAX_TRIALS 500
#define CAELO_RANSAC_LEVELS 3

typedef enum {
    CAELO_OK = 0,
    CAELO_ERR_ARG = -1,     /* bad argument (shape, null pointer, missing weights) */
    CAELO_ERR_HIP = -2,     /* a HIP runtime call failed */
    CAELO_ERR_CAPACITY = -3 /* voxel map too small for the cloud */
} caelo_status;

/* device-side status bits (int32 word, OR-ed by kernels) */
#define CAELO_ST_COL_OOB 1        /* a point projects to column 1800: IndexError at SphericalRing.py:91 */
#define CAELO_ST_VOXEL_OOB 2      /* voxel index outside its block: IndexError at Voxel.py:139 */
#define CAELO_ST_MAP_FULL 4       /* voxel hash table overflow (capacity bug, never data) */
#define CAELO_ST_FEW_VOXELS 8     /* a scale holds < 496 voxels: sklearn ValueError at Voxel.py:195-196 */
#define CAELO_ST_FEW_KEYPTS 16    /* K <= 50: assert at SphericalRing.py:286 */
#define CAELO_ST_NONFINITE 32     /* a NaN coordinate: int(nan) raises ValueError at SphericalRing.py:86-88 / Voxel.py:122-124 (an
                                   * infinite one is a point like any other there: dropped by range / row test or kept) */
#define CAELO_EXTRACT_EXACT_VOXELS 1 /* caelo_extract mode bit: the two-pass first-touch voxelization of caelo_voxelize
                                        (Voxel.py:139-141) instead of the one-pass build; both give the reference's voxel
                                        sets on every input, points on voxel faces included (tests compare them) */
#define CAELO_EXTRACT_NO_DEDUP 2     /* caelo_extract mode bit: encode every patch, also bit-identical copies of another one
                                        (by default equal patches of a frame are encoded once: same descriptors, bit for bit) */

typedef struct caelo_ctx caelo_ctx;
typedef struct caelo_voxmap caelo_voxmap;

int caelo_abi_version(void);
/* How this binary was compiled (no reference counterpart; stamped by csrc/Makefile).  caelo/_ffi.py refuses to load a
 * library whose word has CAELO_BUILD_PACKED_F32 set: packed-f32 VALU instructions mis-execute in lanes 48-63 on MI355X
 * when three hardware queues are busy, so such a binary is silently wrong about once in 1 000 frames. */
#define CAELO_BUILD_PACKED_F32 1 /* built WITH v_pk_*_f32 (make PACKED_F32=1: the fault demonstrator, never the product) */
#define CAELO_BUILD_PROF 2       /* built with the per-phase cycle counters of the encoder (make PROF=1) */
#define CAELO_BUILD_STAMPED 256  /* the Makefile passed its flag word (a hand-rolled hipcc line that did not is refused too) */
int caelo_build_flags(void);
const char *caelo_last_error(void);
int caelo_create(caelo_ctx **ctx, int device);
void caelo_destroy(caelo_ctx *ctx);
/* Hardware self-check of the pose kernels (no reference counterpart): every lane of a wavefront derives the same RANSAC
 * hypothesis, so lanes that disagree mean a mis-executed instruction.  Synchronises the device and returns how many
 * wavefronts saw that since caelo_create.  0 on healthy hardware; tests and bench.py assert it. */
int caelo_lane_faults(caelo_ctx *ctx, int64_t *count_host);

/* Weights (HOST pointers, Keras layouts as stored in the .h5).  Replaces
 * keras.models.load_model(...) at Match.py:313,324 / Dirs.py:29-30. */
int caelo_set_respond_weights(caelo_ctx *ctx, const float *w1_host /*[3][3][3][32]*/, const float *b1_host /*[32]*/,
                              const float *w2_host /*[32][8]*/, const float *b2_host /*[8]*/);
int caelo_set_encoder_weights(caelo_ctx *ctx, const float *w1_host /*[27][1][8]*/, const float *b1_host,
                              const float *w2_host /*[27][8][16]*/, const float *b2_host,
                              const float *w3_host /*[27][16][32]*/, const float *b3_host,
                              const float *wd1_host /*[2048][200]*/, const float *bd1_host,
                              const float *wd2_host /*[200][20]*/, const float *bd2_host);
/* on != 0: this context's encoder runs conv1 / conv2 with the exact-f32 kernel of round 2 instead of the default (f32 products
 * from 2-way f16 splits on the f16 matrix pipe) -- the precision reference of the tests and of tools/enc_layer_errors.py; slower.
 * An explicit, per-context choice: the library reads NO environment variable that changes arithmetic (no reference counterpart:
 * PatchEncoder.predict, Match.py:131-133, has one arithmetic). */
int caelo_set_encoder_reference(caelo_ctx *ctx, int on);

/* ProjectPC2SphericalRing  (SphericalRing.py:72-94)
 * pc [n][4] f32 -> ring [69][1800][5] f32, counter [69][1800] i32.  workspace: winner [69*1800] i32. */
int caelo_project(caelo_ctx *ctx, const float *pc, int64_t n, float *ring, int32_t *counter, int32_t *winner_ws,
                  int32_t *status, void *stream);

/* RespondLayer.predict  (SphericalRing.py:405-408; SphericalRingPCRespondLayer.h5)
 * in [rows>=64][in_w][in_c] (channels 0..2 of rows 0..63, cols 0..1791) -> resp [64][1792][8] */
int caelo_respond(caelo_ctx *ctx, const float *in, int in_w, int in_c, float *resp, void *stream);

/* GetKeyPtsByAE  (SphericalRing.py:113-291).  ring_c = 5: demo mode (:414); 3: batch mode
 * (BatchPreprocess.py:97-98,131-136).  workspace: ws of caelo_keypoints_ws_bytes() bytes, 16-byte aligned.
 * outputs: key_pixels [1024][2] i64 (row,col), key_pts [1024][3] f32, n_key [1] i32. */
int64_t caelo_keypoints_ws_bytes(void);
int caelo_keypoints(caelo_ctx *ctx, const float *ring, int ring_w, int ring_c, const int32_t *counter, int cnt_w,
                    const float *resp, void *ws, int64_t *key_pixels, float *key_pts, int32_t *n_key,
                    int32_t *status, void *stream);

/* debug aid: out_host[40] = 16 phase timestamps (100 MHz ticks) of the last keypoint-selection kernel, 16 of the
 * last encoder stage-1 kernel (workgroup 0, first patch), 8 of the last patch-gather kernel */
int caelo_debug_read(unsigned long long *out_host);

/* Voxelization  (Voxel.py:100-173) into a device voxel map (3 scales of 8^3-voxel bricks). */
int caelo_voxmap_create(caelo_ctx *ctx, int64_t max_points, caelo_voxmap **map);
void caelo_voxmap_destroy(caelo_voxmap *map);
int caelo_voxelize(caelo_ctx *ctx, caelo_voxmap *map, const float *pc, int64_t n, int stride, int32_t *status,
                   void *stream);
/* The one-pass build caelo_extract uses (scales 1/2 derived from the scale-0 bricks, points within an ulp of a voxel
 * face resolved through their voxel's first point): same voxel SETS as caelo_voxelize, no first-touch order, so
 * caelo_voxmap_export does not apply; caelo_patches and caelo_voxmap_dump do. */
int caelo_voxelize_fast(caelo_ctx *ctx, caelo_voxmap *map, const float *pc, int64_t n, int stride, int32_t *status,
                        void *stream);
/* Occupied 8x8x8-voxel bricks of one scale, unordered: keys [capacity] u64 (brick x << 40 | y << 20 | z),
 * bits [capacity][8] u64 (word = x & 7, bit = (y & 7) * 8 + (z & 7)), count [1] i32 (device, may exceed capacity:
 * then only `capacity` bricks were written).  Diagnostic / test access to the device voxel map. */
int caelo_voxmap_dump(caelo_ctx *ctx, const caelo_voxmap *map, int scale, uint64_t *keys, uint64_t *bits, int64_t capacity,
                      int32_t *count, void *stream);
/* AllVoxels0/1/2 in the reference's order (first touch; scale 0 block-grouped, Voxel.py:161-165).
 * out [capacity][3] i16 per scale, counts [3] i64 (device).  Valid after caelo_voxelize.  Asynchronous on `stream`: a count
 * above `capacity` means only the first `capacity` voxels of that scale were written (the caller checks after reading counts). */
int caelo_voxmap_export(caelo_ctx *ctx, caelo_voxmap *map, int16_t *all0, int16_t *all1, int16_t *all2,
                        int64_t capacity, int64_t *counts, void *stream);
/* Records the first-touch order of the map's voxel lists (scales in `scale_mask`, bit s = scale s) on the device, so that
 * caelo_patches on THIS map redoes its tie-split patches in scikit-learn's kd-tree order (csrc/kdorder.hip) -- what
 * caelo_voxmap_export + caelo_voxmap_from_lists + caelo_patches give, without the lists leaving the map and without a host
 * round trip for their lengths.  Valid after caelo_voxelize (which records first touch); asynchronous on `stream`.  A tie-split
 * patch of a scale outside the mask keeps the canonical rule and flag bit 2. */
int caelo_voxmap_order(caelo_ctx *ctx, caelo_voxmap *map, int scale_mask, void *stream);
/* Build the map from reference-format lists instead (GetPatchesList called with arrays). */
int caelo_voxmap_from_lists(caelo_ctx *ctx, caelo_voxmap *map, const int16_t *all0, int64_t n0, const int16_t *all1,
                            int64_t n1, const int16_t *all2, int64_t n2, int32_t *status, void *stream);

/* GetPatchesList  (Voxel.py:177-216): pts [k][3] f32 (k read from n_key when non-null, else k_max)
 * -> bits [k_max][3][64] u64, flags [k_max][3] u8 (bit0 truncated by the 496-NN cap; bit1 the cut falls inside a class of
 * equidistant voxels and the patch is on the canonical rule -- the map has no ordered lists, or the kd build gave up; bit2 such a
 * patch redone in the library's own order: scikit-learn's kd-tree from 994 voxels on, np.argpartition's below (csrc/kdorder.hip)). */
int caelo_patches(caelo_ctx *ctx, const caelo_voxmap *map, const float *pts, int64_t k_max, const int32_t *n_key,
                  uint64_t *bits, uint8_t *flags, int32_t *status, void *stream);
/* The same for n <= 8 maps / key point sets (arrays of n HOST pointers to the device buffers of caelo_patches): the gathers one after the
 * other, then the kd-tree redo of the tie-split patches of ALL of them behind one launch of each kd kernel -- the chain of a redo (a
 * millisecond or two of dependent quickselect passes, Voxel.py:195-196 in scikit-learn's order) is paid once per set, not per frame. */
int caelo_patches_many(caelo_ctx *ctx, int n, const caelo_voxmap *const *maps, const float *const *pts, int64_t k_max,
                       const int32_t *const *n_key, uint64_t *const *bits, uint8_t *const *flags, int32_t *const *status, void *stream);
/* dense <-> packed conversion for callers that want the reference's [K,16,16,16,1] f32 arrays */
int caelo_unpack_patches(caelo_ctx *ctx, const uint64_t *bits, int64_t n_patches, float *dense, void *stream);
int caelo_pack_patches(caelo_ctx *ctx, const float *dense, int64_t n_patches, uint64_t *bits, void *stream);

/* PatchEncoder.predict / GetFeaturesFromPatches  (Match.py:130-135; EncoderModel4VoxelPatch.h5)
 * bits [n_patches][64] u64 -> out[(p / group) * out_stride + (p % group) * 20 + j].
 * group = 1, out_stride = 20: plain predict; group = 3, out_stride = 60 on [K][3][64]: Features [K][60].
 * workspace: ws of caelo_encode_ws_bytes(n_patches) bytes, zero-filled ONCE by its owner before the first call
 * (its header holds work counters that every call returns to zero), one ws per stream.
 * caelo_encode_ws_layout (test aid): byte offsets in ws of what the four kernels leave behind -- out[0] P2 [np][1024] f32 (after
 * pool2), out[1] F3 [np][2048] (after conv3), out[2] Dense(200) partial sums [slices][np][208]; out[3] = np (rows padded to whole
 * row tiles), out[4] = k slices in use, out[5] = k slices the buffer is sized for. */
int64_t caelo_encode_ws_bytes(int64_t n_patches);
int caelo_encode_ws_layout(int64_t n_patches, int64_t out[6]);
int caelo_encode(caelo_ctx *ctx, const uint64_t *bits, int64_t n_patches, int group, float *out, int out_stride,
                 void *ws, void *stream);

/* ---- BASELINE.json configs[4]: 32^3 patches (a stress case of this repo, NOT a reference code path) ----------
 * The reference has PatchSize = 16 only (Voxel.py:31-33).  These three entry points keep GetPatchesList's rule
 * (Voxel.py:177-216: key voxel in f64, wrap-around placement) with the window [-16,16)^3 and the 496-NN cap
 * disabled, and the encoder's layer stack (Match.py:130-135) on 32^3 inputs: Flatten is 16384 wide, so dense_1
 * is a caller-supplied [16384][200] matrix (host pointers, like caelo_set_encoder_weights; conv kernels, biases
 * and dense_2 are the ones already set).
 * bits [k_max][3][512] u64: voxel (ix,iy,iz) at bit (lin & 63) of word (lin >> 6), lin = (ix*32 + iy)*32 + iz.
 * caelo_encode32: bits [n_patches][512] -> out[(p / group) * out_stride + (p % group) * 20 + j]; ws of
 * caelo_encode32_ws_bytes(n_patches) bytes (no initialisation needed). */
int caelo_patches32(caelo_ctx *ctx, const caelo_voxmap *map, const float *pts, int pts_ld, int64_t k_max,
                    const int32_t *n_key, uint64_t *bits, void *stream);
int caelo_set_encoder32_dense(caelo_ctx *ctx, const float *wd1, const float *bd1);
int64_t caelo_encode32_ws_bytes(int64_t n_patches);
int caelo_encode32(caelo_ctx *ctx, const uint64_t *bits, int64_t n_patches, int group, float *out, int out_stride,
                   void *ws, void *stream);

/* caelo_encode32 with a HIP event between its launches; synchronises; ms_host[4] = conv1+pool, conv2+pool, conv3,
 * Dense(200)+head in ms (measurement aid for bench.py --config 5) */
int caelo_encode32_profile(caelo_ctx *ctx, const uint64_t *bits, int64_t n_patches, int group, float *out, int out_stride,
                           void *ws, void *stream, float *ms_host);

/* caelo_encode with a HIP event between its four kernels (stage1 = conv1+pool1+conv2+pool2, conv3,
 * dense1, head) on the launch stream; synchronises, writes the durations in ms to ms_host[0..3] and the number of
 * MFMA instructions stage 1 actually executed, in millions, to ms_host[4] (it skips all-background rows) and the FLOPs of
 * one such instruction to ms_host[5] (16384: v_mfma_f32_16x16x32_f16; 2048: the f32-input kernel).  ms_host holds 6 floats. */
int caelo_encode_profile(caelo_ctx *ctx, const uint64_t *bits, int64_t n_patches, int group, float *out,
                         int out_stride, void *ws, void *stream, float *ms_host);

/* NN match  (Match.py:257-258): pair_idx[j] = argmin_i ||f0[i]-f1[j]|| (f64, first minimum).
 * f0 [k0][ld0], f1 [k1][ld1] (leading dimensions in floats, >= dim, dim <= 64); k0/k1 read from the
 * n0/n1 device words when non-null.
 * ws: caelo_match_ws_bytes(max(k0_max, k1_max)) bytes: two statistics counters the calls only add to (columns re-scanned
 * exactly, columns decided between two rows; zero-fill the first 256 bytes once if you read them) and the scratch images of
 * the two frames (every call rewrites them).  Nothing crosses workgroups.  The distances are screened on the f16 matrix pipe
 * inside a rigorous error window and only the rows that can be the minimum are evaluated in float64 like scipy's cdist: the
 * result is the float64 argmin, bit for bit. */
int64_t caelo_match_ws_bytes(int64_t k_max);
int caelo_match(caelo_ctx *ctx, const float *f0, int ld0, int64_t k0_max, const int32_t *n0, const float *f1, int ld1,
                int64_t k1_max, const int32_t *n1, int dim, int64_t *pair_idx, void *ws, void *stream);

/* Measurement aid (no reference counterpart): the pipeline's launch shape of the NN match -- n_pairs (<= 8) pairs of consecutive
 * frame rows (rows [n_pairs + 1] device pointers to [1024][64] f32, descriptor in columns 0:60; n_key [n_pairs + 1] device counts
 * or NULL) behind one k_match_prep + one k_match_screen launch, `repeats` times between HIP events on `stream`.
 * pair_idx [n_pairs][1024] out; ws: n_pairs * caelo_match_ws_bytes(1024) bytes; ms_host[2] = both kernels / the prep alone, averaged. */
int caelo_match_profile(caelo_ctx *ctx, const float *const *rows, int n_pairs, const int32_t *const *n_key, int64_t *pair_idx,
                        void *ws, int repeats, void *stream, float *ms_host);

/* SolveRT  (Match.py:138-158): p0 ~ R p1 + T over n point pairs.  R [9], T [3] f32 (device);
 * credible [1] i32 (optional): the reference's isCredible, -1 when det(R) < 0 was met. */
int caelo_solve_rt(caelo_ctx *ctx, const float *p0, const float *p1, int64_t n, float *R, float *T,
                   int32_t *credible, void *stream);

/* RANSAC4RT + SolveRelativePose tail  (Match.py:162-218, :260-283).
 * pc0 [k0][ld0], pc1 [k1][ld1] (xyz in the first 3 columns), pair_idx [k1]; rand [3*500][4] f64 = the uniform doubles the
 * reference's np.random.random((4,)) would return, in consumption order.
 * result (device, caelo_pose_result) + inlier mask [k1_max] u8 (4-byte aligned for dword stores).  workspace ws:
 * caelo_ransac_ws_bytes() bytes (the hypotheses' inlier counts between the two kernels of a call; no
 * initialisation), one ws per call in flight. */
typedef struct {
    float R[9];           /* final refit rotation, row-major */
    float T[3];
    float R_ransac[9];    /* best hypothesis before the refit (RANSAC4RT's R_star) */
    float T_ransac[3];
    float threshold;      /* residualThreshold returned (0.4 / 0.8 / 1.6) */
    int32_t success;      /* isSuccess */
    int32_t iterations;   /* cntIters of the last level */
    int32_t n_inliers;
    int32_t best_trial;   /* index into rand of the winning hypothesis, -1 if none */
    int32_t n_pairs;
} caelo_pose_result;
/* The certificate a RANSAC call leaves for the host half (optional, device memory, 16-byte aligned): for each of the 500
 * hypotheses of the first threshold level an UPPER BOUND `hi` on the inlier count the reference's own arithmetic can give it --
 * NumPy forms the 4-point covariance, R, T and the residuals in float32 through its BLAS (Match.py:141-157,:190-193), so a
 * count depends on that BLAS's rounding; the kernels fit in float64 and count the pairs whose residual lies below the
 * threshold PLUS a bound on that difference -- the four sample indices of each hypothesis, and the matched pairs
 * (P0[pair_idx[i]], P1[i]).  caelo_host_certify replays Match.py:181-214 over the bounds and re-evaluates the deciding
 * hypotheses through NumPy's BLAS / LAPACK entry points: inlier sets, R_star / T_star and the refit are then the reference's
 * bits on this host. */
#define CAELO_CERT_MAX_PAIRS 1024
#define CAELO_CERT_MAGIC 0x43455254
#define CAELO_CERT_NO_BOUNDS 1    /* flags: more than CAELO_CERT_MAX_PAIRS pairs -- no bounds, no pairs stored */
typedef struct {
    int32_t magic;        /* CAELO_CERT_MAGIC once k_ransac_finish has written the record */
    int32_t n_pairs;      /* N */
    int32_t flags;
    int32_t levels_up;    /* 1: hi_up / idx_up hold the bounds of the 0.8 m and 1.6 m levels (written by the pipeline's kernels for a pair
                           * whose first level reached leastInliers with no hypothesis: Match.py:207-214) */
    int32_t reserved[12];
    int32_t hi[512];      /* [500] used */
    int32_t idx[512][4];  /* sample indices int32(u * N) (Match.py:182-184) of the first level's hypotheses */
    float p0[CAELO_CERT_MAX_PAIRS][3]; /* Pairs0 = PC0[pairIdx] (Match.py:260) */
    float p1[CAELO_CERT_MAX_PAIRS][3]; /* Pairs1 */
    int32_t hi_up[2][512];      /* the same for the hypotheses of the 0.8 m and the 1.6 m level (draws 500 .. 999, 1000 .. 1499) */
    int32_t idx_up[2][512][4];
} caelo_ransac_cert;
int64_t caelo_cert_bytes(void);
int64_t caelo_ransac_ws_bytes(void);
int caelo_ransac(caelo_ctx *ctx, const float *pc0, int ld0, const float *pc1, int ld1, const int64_t *pair_idx,
                 int64_t k1_max, const int32_t *n1, const double *rand, caelo_pose_result *result,
                 uint8_t *inlier_mask, void *ws, caelo_ransac_cert *cert, void *stream);

/* Host half of the exact RANSAC (csrc/certify.hip; every pointer below is HOST memory, no device is touched).
 * caelo_host_bind_blas: the cblas_sgemm / cblas_sgemv / dgesdd (Fortran) entry points of the BLAS the process's NumPy is
 *   linked against (caelo/hostblas.py finds them), ilp64 = 1 when their integers are 64 bits wide.  Process-wide.
 * caelo_host_solve_rt: SolveRT (Match.py:138-158) by NumPy's own call sequence -- float32 row sums, cblas_sgemm for the
 *   covariance, dgesdd on its float64 copy, U / Vh cast to float32, cblas_sgemm for R, cblas_sgemv for T.
 * caelo_host_ransac: RANSAC4RT + the refit of SolveRelativePose (Match.py:162-218,:269-283) on pairs [n][3]; with `hi`
 *   (the first level's bounds) only the deciding hypotheses are evaluated, without it all of them, like the reference.
 * caelo_host_certify: k certificates (copied from the device) -> k results + masks [k][mask_ld]; rand_host [k] (nullable, as
 *   may be its entries) = the pairs' draws, needed only when a pair escalates beyond 0.4 m; evals [k] (nullable) = hypotheses
 *   evaluated; status [k] (nullable): 0 exact, 1 the draws were needed and missing, 2 no bounds in the record (more than
 *   1024 pairs: use caelo_host_ransac), 3 no record.  `threads` host threads share the k records. */
int caelo_host_bind_blas(void *cblas_sgemm, void *cblas_sgemv, void *dgesdd, int ilp64);
int caelo_host_blas_bound(void);
int caelo_host_unbind_blas(void);   /* forget the bound entry points (a library that failed the binder's bit-for-bit check) */
/* one BLAS / LAPACK call of the host half, issued exactly as the host half issues it (flags, leading dimensions): what a binder
 * compares with np.dot / np.linalg.svd.  op 0: out[9] = a^T b, a, b [n][3]; 1: out[18] = U | Vh of svd(a [3][3]); 2: out[9] = a^T b^T,
 * [3][3] each; 3: out[3] = a b, a [3][3], b [3]; 4: out[3][n] = a b^T, a [3][3], b [n][3] */
int caelo_host_blas_probe(int op, const float *a_host, const float *b_host, int64_t n, float *out_host);
/* exact hypothesis counts the host half found ABOVE the device's upper bound since the process started.  The bound's constant is
 * calibrated, not proven: the host half checks every count it evaluates, and on a violation decides the pair by the reference's
 * loop without bounds (still exact) and counts it here.  0 on every run so far. */
int64_t caelo_host_bound_violations(void);
int caelo_host_solve_rt(const float *p0_host, const float *p1_host, int64_t n, float *R_host, float *T_host, int32_t *credible_host);
int caelo_host_ransac(const float *pairs0_host, const float *pairs1_host, int64_t n, const double *rand_host, const int32_t *hi_host,
                      caelo_pose_result *result_host, uint8_t *mask_host, int32_t *evals_host);
int caelo_host_certify(const void *certs_host, int64_t k, const double *const *rand_host, caelo_pose_result *results_host,
                       uint8_t *masks_host, int64_t mask_ld, int32_t *evals_host, int32_t *status_host, int threads);

/* Fused per-scan hot path: ProjectPC2SphericalRing -> RespondLayer.predict -> GetKeyPtsByAE ->
 * Voxelization -> GetPatchesList -> GetFeaturesFromPatches in one call, one stream, no host sync.
 * pc [n][4] f32.  dist_channels 5 = demo calling mode (SphericalRing.py:414), 3 = batch mode
 * (BatchPreprocess.py:97-98,131-136).  Outputs (device): key_pts rows [1024][kp_ld], features rows
 * [1024][feat_ld] (60 used), valid (optional) [1024] with stride valid_ld = 1.0 for rows < K,
 * key_pixels [1024][2], n_key [1], flags [1024][3] (caelo_patches), status int32[4] 16-byte aligned
 * (word 0 = CAELO_ST_* bits, cleared by the call).  ws: caelo_extract_ws_bytes() bytes, 256-byte aligned, zero-filled
 * once by its owner before the first call (it embeds an encoder workspace), one ws per stream. */
int64_t caelo_extract_ws_bytes(void);
int caelo_extract(caelo_ctx *ctx, caelo_voxmap *map, const float *pc, int64_t n, int dist_channels, int mode, float *key_pts,
                  int kp_ld, float *features, int feat_ld, float *valid, int valid_ld, int64_t *key_pixels,
                  int32_t *n_key, uint8_t *flags, int32_t *status, void *ws, void *stream);

/* ExtendKeyPtsInShpericalRing  (SphericalRing.py:294-317; BatchPreprocess.py:139): the occupied ring pixels of the
 * 13 x 13 window of every keypixel, keypixels in order, each window row-major, a pixel only for the FIRST keypixel
 * whose window covers it.  Like the reference it ZEROES those windows in the caller's counter.
 * ring [>=rows][ring_w][ring_c] f32, counter [>=rows][cnt_w] i32 (in/out); rows x cols = the extent both cover
 * (69 x 1800 in demo mode, 64 x 1792 in batch mode); key_pixels [k_max][2] i64 (row, col), n_key optional device count.
 * ext_pts [k_max * 169][3] f32 out, n_ext [1] i32 out.  ws: caelo_extend_ws_bytes(rows, cols) bytes. */
int64_t caelo_extend_ws_bytes(int rows, int cols);
int caelo_extend_keypts(caelo_ctx *ctx, const float *ring, int ring_w, int ring_c, int32_t *counter, int cnt_w, int rows,
                        int cols, const int64_t *key_pixels, int k_max, const int32_t *n_key, float *ext_pts,
                        int32_t *n_ext, void *ws, void *stream);

/* One iteration of the reference's point-to-point ICP (MyICP.py:26-72; GetPtsInliners :75-85): nearest neighbour in
 * pc0 [n0][3] of every point of pc1 [n1][3] (exact float64 Euclidean distance, first minimum), the pairs closer than
 * `threshold`, SolveRT on them -> rt [12] (R row-major | T, device) and pc1 <- R pc1 + T IN PLACE.  n_inliers [1]
 * (device) = number of pairs; with fewer than min_inliers nothing is fitted or moved (the caller stops: :38-40).
 * Threshold decay and the Euler-angle stop rule stay with the host loop (caelo.api.ICP).  ws: caelo_icp_ws_bytes(n1). */
int64_t caelo_icp_ws_bytes(int64_t n1);
int caelo_icp_step(caelo_ctx *ctx, const float *pc0, int64_t n0, float *pc1, int64_t n1, double threshold, int min_inliers,
                   float *rt, int32_t *n_inliers, void *ws, void *stream);

/* The whole loop of the reference's re-registration on the device:
 *   ICP                    MyICP.py:26-72    use_planar = 0, min_pairs = 100, fail_only_first = 0, threshold0 = 0.5, decay0 = 0.9,
 *                                            small_shift = 0.05, ep = 0.001, max_iter = 50, min_iter = 19
 *   ICP_Pt2PtAndPt2Plane   MyICP.py:127-201  use_planar = 1, min_pairs = 200, fail_only_first = 1 (too few pairs is a failure only in
 *                                            the first iteration), threshold1 / decay1 for the planar gate (caller RefinePoses.py:291-295)
 * pc0 [n0][3], pc1 [n1][3] f32 (pc1 is MOVED in place by every iteration); planar0 [m0][6], planar1 [m1][6] f32 = xyz | normal
 * (planar1's xyz moves, its normals do not: MyICP.py:177).  With use_planar an empty planar set is an error, like sklearn's
 * ValueError at MyICP.py:94 -- GetKeyPtsByAE always returns an empty PlanarPts (SphericalRing.py:219,285).
 * Per iteration: nearest neighbours (exact float64 distance, first minimum) of both sets, the gates, ONE SolveRT over all pairs,
 * R_star / T_star accumulated in float64, the Euler-angle stop rule and the threshold decay -- no host synchronisation; `result`
 * (device) is complete when the stream has passed the call.  ws: caelo_icp_loop_ws_bytes(n1, m1) bytes.
 * Regime: the reference's use -- up to 50 iterations on a few thousand extended key points.  Every iteration is an exact
 * brute-force nearest-neighbour pass (O(n0 n1) float64 distances) and a one-workgroup update, and ALL max_iter (<= 1000)
 * iterations are enqueued up front (those after convergence return at once): tens of thousands of points or hundreds of
 * iterations work but are not what this entry point is built for.  Nearest-neighbour ties resolve to the lowest index; sklearn's
 * kd-tree (MyICP.py:34-38) makes no promise on exact ties, so clouds with duplicated points may pair differently there. */
typedef struct {
    double threshold0, threshold1, decay0, decay1, small_shift, ep;
    int32_t max_iter, min_iter, min_pairs, fail_only_first, use_planar, reserved;
} caelo_icp_params;
typedef struct {
    double R_star[9], T_star[3];
    double threshold0, threshold1;   /* after the last decay */
    int32_t iterations;              /* iterations that moved pc1 */
    int32_t success;                 /* isSuccess */
    int32_t n_inliers_pts, n_inliers_planar; /* pairs of the last evaluated iteration */
} caelo_icp_result;
int64_t caelo_icp_loop_ws_bytes(int64_t n1, int64_t m1);
int caelo_icp(caelo_ctx *ctx, const float *pc0, int64_t n0, float *pc1, int64_t n1, const float *planar0, int64_t m0, float *planar1,
              int64_t m1, const caelo_icp_params *params, caelo_icp_result *result, void *ws, void *stream);

/* ---- frame pipeline: batches of frames behind single launches, three stages on three HIP streams ------------------
 * Replaces the reference's per-frame driver loops (BatchPreprocess.py:88-140 extract loop, Match.py:296-353 /
 * PoseEstimation.py pair loop) for throughput.  `batch` consecutive frames share ONE launch of every kernel of the
 * front half of caelo_extract (ring image, response, keypoints, voxel map, patch gather), one encoder launch set and
 * one caelo_match / caelo_ransac launch per threshold level; front(b+1), encoder(b) and pairs(b-1) overlap on three
 * streams.  Per-frame results are identical to the single-call entry points (same kernels, same per-frame arithmetic).
 *   create(ctx, batch in [1,8], n_buffers in [2,4], max_points)   n_buffers = batches of patches in flight between
 *                                                                   the front and the encoder
 *   begin(stream)   the stages wait for work already queued on `stream` (the producers of the jobs' inputs)
 *   submit(job)     copy the job; every `batch` jobs (or when the mode changes) the batch is launched:
 *                     caelo_extract(pc -> rows [1024][64] = descriptor 0:60 | xyz 60:63 | valid 63, ...)
 *                     pair == CAELO_PAIR_CHAIN:    caelo_match + caelo_ransac against the previously submitted frame
 *                     pair == CAELO_PAIR_EXPLICIT: ... against prev_rows / prev_n_key (e.g. rows gathered from a peer)
 *   flush(stream)   launch a partial batch, make `stream` wait for all three stages
 * Buffers named by a job must stay alive until `stream` has passed the flush.  Calls come from ONE host thread.
 * A failed launch is returned by the submit / flush that issued it (caelo_last_error() holds the text). */
typedef struct caelo_pipeline caelo_pipeline;
#define CAELO_PAIR_NONE 0
#define CAELO_PAIR_CHAIN 1
#define CAELO_PAIR_EXPLICIT 2
typedef struct caelo_frame_job {
    const float *pc;            /* [n][4] f32 */
    int64_t n;
    int32_t dist_channels;      /* 5 | 3, see caelo_extract */
    int32_t mode;               /* CAELO_EXTRACT_* bits */
    float *rows;                /* [1024][64] f32 out */
    int64_t *key_pixels;        /* [1024][2] out */
    int32_t *n_key;             /* [1] out */
    uint8_t *flags;             /* [1024][3] out */
    int32_t *status;            /* int32[4], 16-byte aligned, out */
    int32_t pair;               /* CAELO_PAIR_* */
    int32_t reserved;
    const float *prev_rows;     /* CAELO_PAIR_EXPLICIT: [1024][64] rows of frame 0 of the pair */
    const int32_t *prev_n_key;  /* CAELO_PAIR_EXPLICIT: its keypoint count (NULL = 1024) */
    const double *rand;         /* [1500][4] f64 uniform draws (caelo_ransac) */
    caelo_pose_result *result;  /* out */
    uint8_t *inlier_mask;       /* [1024] out */
    int64_t *pair_idx;          /* [1024] out */
    caelo_ransac_cert *cert;    /* out, nullable: the pair's certificate for caelo_host_certify */
    /* The host half inside the pipeline (needs caelo_host_bind_blas): when result_host is given, the kernels write the pair's
     * certificate straight into pinned host memory of the pipeline (`cert` is not used; CAELO_CERT_ZEROCOPY=0: into `cert`, copied by
     * a copy command), the issuing thread finds the pair stage finished two batches later (no device-side wait), certifier threads
     * run the host half while later batches are on the GPU and write the EXACT result -- the reference's
     * inlier set, R_star / T_star, refit, bit for bit -- to these HOST buffers; caelo_pipeline_flush returns when all are written.
     * The kernels then stop at the certificate: `result` and `inlier_mask` (device) are NOT written for such a pair -- the winner,
     * mask and refit k_ransac_finish would compute are exactly what the host half replaces. */
    caelo_pose_result *result_host; /* out, nullable */
    uint8_t *mask_host;             /* [1024] out (required with result_host) */
    const double *rand_host;        /* nullable: host copy of `rand` (only read when the pair escalates beyond 0.4 m; fetched from `rand` otherwise) */
    int32_t *info_host;             /* nullable: [2] = hypotheses evaluated on the host, status (0 exact, 2 no bounds, 3 no record) */
} caelo_frame_job;
int caelo_pipeline_create(caelo_ctx *ctx, int batch, int n_buffers, int64_t max_points, caelo_pipeline **out);
/* host half inside the pipeline: out_host[4] = pairs certified, hypotheses evaluated for them, nanoseconds of the certifier thread, nanoseconds the issuing thread spent handing certificates over -- since the last call */
int caelo_pipeline_cert_stats(caelo_pipeline *pipe, int64_t *out_host);
void caelo_pipeline_destroy(caelo_pipeline *pipe);
int caelo_pipeline_batch(const caelo_pipeline *pipe);
int caelo_pipeline_begin(caelo_pipeline *pipe, void *stream);
int caelo_pipeline_submit(caelo_pipeline *pipe, const caelo_frame_job *job);
/* n jobs in one call (the same as n caelo_pipeline_submit calls in order; stops at the first error): a host language with a
 * costly foreign-call path (ctypes: ~10 us per call) hands a whole run over at once -- the odometry loop of PoseEstimation.py:241-267. */
int caelo_pipeline_submit_many(caelo_pipeline *pipe, const caelo_frame_job *jobs, int64_t n);
int caelo_pipeline_flush(caelo_pipeline *pipe, void *stream);
/* Scans that arrive while the pipeline runs -- the overlap of the reference's producer process, which prepares frame i + 1 while
 * frame i is matched (PoseEstimation.py:214-245).  caelo_pipeline_wait_stream: the front stage of every batch submitted from
 * now on starts after the work `stream` holds at this moment (the uploads of those scans).  caelo_pipeline_release_scans:
 * `stream` waits until the front stages of the batches issued so far are done -- nothing else reads a scan -- before it may
 * overwrite their buffers.  Both between caelo_pipeline_begin and caelo_pipeline_flush. */
int caelo_pipeline_wait_stream(caelo_pipeline *p, void *stream);
int caelo_pipeline_release_scans(caelo_pipeline *p, void *stream);
/* Results that leave while the pipeline runs -- the sharded sequence's descriptor all-gather (PoseEstimation.py:241-251 needs the
 * previous frame's features wherever that frame was extracted): `stream` waits until the rows (descriptors | key points | valid)
 * of every frame of the batches ISSUED so far are written, and of nothing later -- a collective enqueued on `stream` then moves
 * finished rows while the next batches are still being extracted.  Between caelo_pipeline_begin and caelo_pipeline_flush. */
int caelo_pipeline_wait_encoded(caelo_pipeline *p, void *stream);
/* The same hand-over paced by the host: blocks the calling thread until the rows of every batch issued so far except the last
 * `lag` (0 <= lag < buffers) are written; work enqueued afterwards on any stream of this device may read them without a
 * device-side wait.  With lag >= 1 the pipeline keeps `lag` batches queued behind the one being waited for. */
int caelo_pipeline_sync_encoded(caelo_pipeline *p, int lag);
/* Pacing of the issuing thread: after issuing a batch, caelo_pipeline_submit waits until the batch `lag` before it is through the
 * encoder (default 1, or CAELO_PIPE_PACE; -1: never waits, the thread runs as far ahead as the queues take).  Waits that sit
 * unsatisfied in the hardware queues cost throughput, and a thread running far ahead leaves many: 16.5 k -> 17.3 k frames/s on
 * a 20-batch run.  A caller that paces itself (caelo_pipeline_sync_encoded between its own work) turns this off. */
int caelo_pipeline_set_pace(caelo_pipeline *p, int lag);
int caelo_pipeline_get_pace(const caelo_pipeline *p);   /* the pacing in effect (the library's default until set) */
/* n host -> device copies on `stream` behind one call (the scans of a batch that live in pinned host memory, the producer side of
 * PoseEstimation.py:214-245): dst[i] <- src[i], bytes[i] each, asynchronous like hipMemcpyAsync.  The pipeline does not take part --
 * the caller orders the copies against it (caelo_pipeline_wait_stream / an event of its own / caelo_pipeline_sync_encoded). */
int caelo_upload_many(void *const *dst, const void *const *src, const size_t *bytes, int n, void *stream);
/* A whole run of the upload mode behind ONE call: jobs [k] (in nb batches of the pipeline's batch size, the remainder last) whose
 * scans arrive by ONE copy command per batch on `copy_stream`, `ahead` batches ahead of the batch being issued; the calling thread waits
 * for a batch's arrival, submits it, queues the next copy and paces itself one batch behind the encoder -- natively: the interpreter's
 * ~60 us between that wait and the next batch's front launches were 20 % of the rate.  Includes caelo_pipeline_begin and the flush.
 * Without a loader: batch b is copied from src[b] to dst[b], bytes[b] (host arrays of nb entries).  With one (caelo_seqloader, below):
 * batch b0 + b comes from its ring slot (ring_host, slot_bytes) into dst[(b0 + b) % n_slots] (n_slots >= ahead + 2 device slots of
 * slot_bytes), and the point counts of its jobs (job.n) are filled in from the loader, which gets its slot back as soon as the copy is
 * through.  times_ns_host (nullable) [4]: waiting for the loader / for arrivals, submitting, copy issue + pacing. */
struct caelo_seqloader;
int caelo_pipeline_run_uploading(caelo_pipeline *pipe, caelo_frame_job *jobs, int64_t k, int64_t nb, struct caelo_seqloader *loader, int64_t b0,
                                 void *const *dst, const void *const *src, const size_t *bytes, int n_slots, const void *ring_host,
                                 int64_t slot_bytes, int ahead, void *copy_stream, void *stream, int64_t *times_ns_host);

/* ---- a sequence from files (PoseEstimation.py:173-245: the generator process that prepares frame i + 1 while frame i is matched) ----
 * caelo_host_random_sample: numpy.random.RandomState(seed).random_sample(n) bit for bit (MT19937, init_genrand seeding, 53-bit doubles):
 * the stream RANSAC4RT draws its samples from (Match.py:182-184), CAELO_SEQ_DRAWS doubles per pair (3 levels x 500 trials x 4).
 * caelo_seqloader: `threads` native threads read the n KITTI .bin files of `paths` ([points][4] f32) batch after batch into the caller's
 * (pinned) ring of `ring_batches` slots.  A slot (caelo_seqloader_slot_bytes) = [batch][cap_points][4] f32 scans, then
 * [batch][CAELO_SEQ_DRAWS] f64 draws -- file i's are those of pair (frame - 1, frame) = RandomState(seed_base + first_frame + i - 1) -- so
 * that one copy command moves a batch to a device slot of the same layout; draws_keep_host (nullable) [keep_batches][batch][CAELO_SEQ_DRAWS]
 * keeps batch b's draws at b % keep_batches for the host half (caelo_frame_job::rand_host).  _wait blocks until batch b (files b * batch ..)
 * sits in slot b % ring_batches and reports the point counts; _release hands the slot back (in order).  A file that is missing, not a
 * multiple of 16 bytes or larger than a slot fails the wait with an error naming it.  Host code only. */
#define CAELO_SEQ_DRAWS (CAELO_RANSAC_LEVELS * CAELO_RANSAC_MAX_TRIALS * 4)
typedef struct caelo_seqloader caelo_seqloader;
int caelo_host_random_sample(uint32_t seed, int64_t n, double *out_host);
int64_t caelo_seqloader_slot_bytes(int batch, int64_t cap_points);
int caelo_seqloader_create(const char *const *paths, int64_t n, int64_t first_frame, int batch, int ring_batches, int64_t cap_points,
                           void *ring_host, double *draws_keep_host, int keep_batches, int64_t seed_base, int threads, caelo_seqloader **out);
int caelo_seqloader_wait(caelo_seqloader *loader, int64_t b, int32_t *slot_host, int64_t *n_points_host);
int caelo_seqloader_release(caelo_seqloader *loader, int64_t b);
int caelo_seqloader_stats(caelo_seqloader *loader, int64_t *out_host);   /* [3]: ns reading, drawing, waiting for a slot (summed over threads) */
void caelo_seqloader_destroy(caelo_seqloader *loader);
/* host-side counters since the last call (then reset): out_host[6] = jobs, ns the calling thread spent issuing their
 * launches, batches launched, batch size, hand-off buffers, HIP streams used */
/* Optional hint before caelo_pipeline_begin: the run will submit n_frames jobs.  If that is not a multiple of the batch size, the
 * frames are spread evenly over ceil(n_frames / batch) batches (20 on batch 8: 6 + 7 + 7) instead of full batches and a
 * remainder -- nothing overlaps the first batch's front stage nor the last batch's encoder + pair stages, so neither should be
 * the odd one out.  The hint holds for one run (cleared by caelo_pipeline_flush); results do not depend on it. */
int caelo_pipeline_expect(caelo_pipeline *pipe, int64_t n_frames);
int caelo_pipeline_stats(caelo_pipeline *pipe, int64_t *out_host);

#ifdef __cplusplus
}
#endif
#endif /* CAELO_H */
