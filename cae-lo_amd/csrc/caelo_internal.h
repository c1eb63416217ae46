// caelo_internal.h -- shared declarations for the HIP translation units of libcaelo.so (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/caelo.h"

#define CAELO_API extern "C" __attribute__((visibility("default")))

// ---- error plumbing ---------------------------------------------------------------------------
void caelo_set_error(const char *fmt, ...);

#define CAELO_HIP(expr)                                                                     \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            caelo_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return CAELO_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

#define CAELO_LAUNCH_CHECK()                                                                \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess) {                                                             \
            caelo_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return CAELO_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

#define CAELO_REQUIRE(cond, msg)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            caelo_set_error("%s: %s", __func__, msg);    \
            return CAELO_ERR_ARG;                        \
        }                                                \
    } while (0)

// ---- context -----------------------------------------------------------------------------------
struct caelo_ctx {
    int device;
    // response layer (SphericalRingPCRespondLayer.h5)
    float *resp_w;  // [27*32 + 32 + 32*8 + 8] = w1 | b1 | w2 | b2
    bool has_resp;
    // encoder (EncoderModel4VoxelPatch.h5), device copies in kernel-friendly layouts
    float *enc_w1;   // [27][8]
    float *enc_b1;   // [8]
    float *enc_w2;   // [27][8][16] (Keras order)
    float *enc_b2;   // [16]
    float *enc_c0;   // [512][16] conv2 response of the all-background patch incl. bias, then bg[8]
    float *enc_w3;   // [27][16][32]
    float *enc_b3;   // [32]
    void *enc_w1f;   // conv1 as k_enc_stage1x's B operand (w1 scattered over the 4^3 receptive field): [k-step 2][n-tile 4][f16 term 2][lane 64] x 16 B
    void *enc_w2x;   // conv2 as k_enc_stage1x's B operand: [tap row 9][fragment 3][lane 64] x 16 B (f16 terms, enc_stage1x.inc)
    void *enc_w3x;   // W3 as the conv3 kernel's B operand: [ntile 2][tap pair 14][bf16 split 3][lane 64] x 16 B
    void *enc_wd1x;  // dense_1 as the dense-1 kernel's B operand: [k-step 64][bf16 split 3][n-tile 13][lane 64] x 16 B
    float *enc_bd1;  // [208]
    float *enc_wd2;  // [200][20]
    void *enc_wd2q;  // the same as k_enc_head_mfma's B operand: [hidden group 13][n-tile 2][lane 64] x float4, zero padded
    float *enc_bd2;  // [20]
    void *enc32_wd1x;  // dense_1 [16384][200] of the 32^3 stress case (config5.hip) in the same operand layout, null until set
    float *enc32_bd1;  // [208]
    bool has_enc;
    bool enc_reference;   // caelo_set_encoder_reference: stage 1 = the exact-f32 k_enc_stage1 (precision reference)
    int32_t *faults;  // device counter of the pair kernels' lane-agreement checks (match.hip); 0 on healthy hardware
};

// ---- voxel map ---------------------------------------------------------------------------------
// Three scales of 8x8x8-voxel bricks in open-addressing hash tables.  A brick payload is 8 u64
// words: word (x&7), bit ((y&7)*8 + (z&7)).  Scale 0 additionally has a voxel-level table that
// records the first point index touching each voxel (Voxel.py:139-141 first-touch semantics).
#define CAELO_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

struct caelo_brick_table {
    unsigned long long *keys;  // [slots]
    unsigned long long *bits;  // [slots][8]
    uint32_t mask;             // slots - 1
};

struct caelo_kd;   // kdorder.hip: the voxel lists in the reference's order + scikit-learn's kd-tree over them
struct caelo_voxmap {
    int64_t max_points;
    caelo_kd *kd;      // allocated by the first caelo_voxmap_from_lists / caelo_voxmap_order
    bool order_tracked;  // the first-touch tables are valid: the map was filled by caelo_voxelize (what caelo_voxmap_export / _order read)
    bool kd_lists;     // kd holds the map's voxel lists in the reference's order (caelo_voxmap_from_lists, caelo_voxmap_order)
    caelo_brick_table brick[3];
    // voxel-level first-touch tables (value = smallest inserting point index, 0xFFFFFFFF = none)
    unsigned long long *vkeys[3];
    uint32_t *vfirst[3];
    uint32_t vmask[3];
    int32_t *counts;  // [16] device ints: [0..2] unique voxels per scale, [4],[5] lengths of list0/list1 (list0 has holes), [6] length of
                      // sp_list
    uint32_t *list0, *list1;  // slots of the occupied scale-0 / scale-1 bricks, in insertion order (fast path)
    // "Suspect" voxels of the fused build (voxel.hip, k_vox_points): scale-0 voxels holding a point whose own
    // int(x_/0.16), int(x_/0.64) differ from its scale-0 index >> 3, >> 5 (x_ within an ulp of a voxel face -- every
    // metrically quantised cloud has some).  sp_* : voxel key -> smallest point index in it; sb_*: brick key -> number of
    // suspect voxels in that brick (stored as count - 1 mod 2^32, so that "empty" is the 0xFF fill); sp_list: one entry
    // per inconsistent point (the slots it touched), which is also how the next build wipes the tables.
    unsigned long long *sp_keys, *sb_keys;
    uint32_t *sp_first, *sb_cnt;
    uint4 *sp_list;     // [max_points]: x = point index, y = sp slot, z = sb slot (0xFFFFFFFF: none), w unused
    uint32_t sp_mask;   // slots - 1 (>= 2 x max_points slots: never full)
    size_t sp_off, sp_bytes;  // the four tables, one 0xFF region
    // One allocation, two clear regions (MI355X: two large fills instead of thirteen small ones):
    //   ff region (cleared to 0xFF): brick keys x3 | vkeys0 | vfirst0 || vkeys1 | vkeys2 | vfirst1 | vfirst2
    //   zero region (cleared to 0):  brick bits x3 | counts
    // Without first-touch order tracking only the part of the ff region before `||` is cleared.
    char *base;
    size_t ff_bytes_keys, ff_bytes_min, ff_bytes_all, zero_off, zero_bytes, total_bytes;
    // host-side: the map's only contents are the bricks of the last fused build, all listed in list0 / list1 (+ scale 2):
    // the next fused build may wipe exactly those entries instead of the whole 25 MB (vox_clear_for_fast_build)
    bool lists_valid;
    // export scratch
    void *scratch;
    int64_t scratch_bytes;
};

void kd_destroy(caelo_voxmap *m);
int kd_store_lists(caelo_voxmap *m, const int16_t *const lists[3], const int64_t ns[3], hipStream_t s);
int kd_begin_device_lists(caelo_voxmap *m, int16_t *vox_out[3], int32_t **n_out, hipStream_t s);
int kd_resolve(const caelo_voxmap *m, const float *pts, int pts_ld, int64_t k_max, const int32_t *n_key, uint64_t *bits, uint8_t *flags,
               hipStream_t s);
int kd_resolve_many(int n, const caelo_voxmap *const *maps, const float *const *pts, int pts_ld, int64_t k_max, const int32_t *const *n_key,
                    uint64_t *const *bits, uint8_t *const *flags, hipStream_t s);

struct SuspectTables {
    unsigned long long *sp_keys, *sb_keys;
    uint32_t *sp_first, *sb_cnt;
    uint4 *list;
    uint32_t mask;
};
inline SuspectTables suspect_tables(const caelo_voxmap *m) {
    return SuspectTables{m->sp_keys, m->sb_keys, m->sp_first, m->sb_cnt, m->sp_list, m->sp_mask};
}

// ---- a set of frames behind ONE launch ------------------------------------------------------------------------------
// The front half of a frame (ring image, response, keypoints, voxel hash, patch gather) is ~15 short, latency-bound
// kernels that leave most of the 256 CUs idle.  Every such kernel therefore takes a caelo_frame_set -- up to
// CAELO_FB_MAX frames, blockIdx.z = frame -- so that a batch of frames costs the launches (and the tails, and the
// single-workgroup stretches) of one.  The single-frame C-ABI entry points pass a set of one.
#ifndef CAELO_FB_MAX
#define CAELO_FB_MAX 8
#endif
struct DedupScratch;
struct caelo_frame_dev {
    const float *pc;              // [n][pc_stride] f32
    int64_t n;
    int32_t pc_stride, dist_c;
    float *ring;                  // [69][1800][5] (or the caller's image for the staged entry points)
    int32_t *counter, *winner;
    float *resp;                  // [64][1792][8]
    unsigned long long *cand;
    int32_t *cand_count;
    int64_t *key_pixels;
    float *key_pts, *valid;
    int32_t *n_key;
    uint8_t *flags;
    int32_t *status;
    int32_t kp_ld, valid_ld;
    caelo_brick_table brick[3];
    int32_t *counts;
    uint32_t *list0, *list1;
    SuspectTables sp;
    unsigned long long *vkeys[3];  // exact path: voxel-level first-touch tables
    uint32_t *vfirst[3];
    uint32_t vmask0, vmask12;
    unsigned long long *bits;     // bit-packed patches [1024][3][64] (+ dedup tables behind them)
    DedupScratch *dd;
};
struct caelo_frame_set {
    caelo_frame_dev f[CAELO_FB_MAX];
    int32_t n;
};
static_assert(sizeof(caelo_frame_set) <= 3800, "caelo_frame_set must fit the kernel argument segment");
void frame_dev_set_map(caelo_frame_dev &d, const caelo_voxmap *m);

// the pair half (NN match + RANSAC + refit) of up to CAELO_FB_MAX frame pairs behind one launch each, blockIdx.z = pair
struct caelo_pair_dev {
    const float *f0, *f1;        // descriptors [k][ld]
    const int32_t *n0, *n1;      // device key point counts (null = k_max)
    const float *pc0, *pc1;      // key points [k][pld] (xyz first)
    int64_t *pair_idx;
    void *ws_match, *ws_ransac;
    const double *rand;
    caelo_pose_result *result;
    uint8_t *mask;
    caelo_ransac_cert *cert;     // nullable: the certificate for the host half (certify.hip)
    int32_t cert_only;           // the host half produces this pair's result (result / mask are not written by the kernels)
};
struct caelo_pair_set {
    caelo_pair_dev p[CAELO_FB_MAX];
    int32_t n;
    int32_t *faults;  // caelo_ctx::faults (may be null)
};
int match_set(const caelo_pair_set &ps, int ld0, int64_t k0_max, int ld1, int64_t k1_max, int dim, hipStream_t s);
int ransac_set(const caelo_pair_set &ps, int pld0, int pld1, int64_t k1_max, hipStream_t s);
// the host half of the exact RANSAC on one certificate record (certify.hip): 0 exact, 1 the draws are needed (an escalation, rnd null),
// 2 no bounds in the record, 3 no record, -1 failure
int certify_record(const caelo_ransac_cert &c, const double *rnd, caelo_pose_result *res, uint8_t *mask, int64_t mask_len, int32_t *evals);

// a (pointer, bytes, byte value) triple for the multi-buffer clear kernel (frame.hip)
struct caelo_clear_item {
    void *ptr;
    size_t bytes;  // multiple of 16
    uint32_t pattern;
};
#define CAELO_CLEAR_MAX 12
struct caelo_clear_list {
    caelo_clear_item item[CAELO_CLEAR_MAX];
    int n;
};
int caelo_clear_many(const caelo_clear_list &list, hipStream_t s);
int caelo_clear_many_set(const caelo_clear_list *lists, int n_frames, hipStream_t s);  // one launch, blockIdx.y = frame

// internal launchers shared by the staged entry points and the fused caelo_extract (no clears inside)
int ring_project_launch(const float *pc, int64_t n, float *ring, int32_t *counter, int32_t *winner, int32_t *status,
                        hipStream_t s);
int ring_respond_launch(caelo_ctx *c, const float *in, int in_w, int in_c, float *resp, hipStream_t s);
int ring_keypoints_launch(const float *ring, int ring_w, int ring_c, int dist_c, const int32_t *counter, int cnt_w,
                          const float *resp, unsigned long long *cand, int32_t *cand_count,
                          int64_t *key_pixels, float *key_pts, int kp_ld, float *valid, int valid_ld, int32_t *n_key,
                          int32_t *status, hipStream_t s);
int ring_project_set(const caelo_frame_set &fs, hipStream_t s);
int ring_respond_set(caelo_ctx *c, const caelo_frame_set &fs, int in_w, int in_c, hipStream_t s, int row0, int rows);
int ring_keypoints_set(const caelo_frame_set &fs, int ring_w, int ring_c, int cnt_w, hipStream_t s);
void vox_clear_items(caelo_voxmap *m, int level, caelo_clear_list &list);  // 0 brick keys only, 1 + scale-0 first-touch table, 2 everything
// clear for a fused build: if the map holds nothing but the previous fused build, its listed bricks are wiped by a
// kernel launched here (before the caller's clear list runs) and only the small scale-2 table + counters join the list
int vox_clear_for_fast_build(caelo_voxmap *m, caelo_clear_list &list, hipStream_t s);
int vox_clear_for_fast_build_set(caelo_voxmap *const *maps, int n, caelo_clear_list *lists, hipStream_t s);
int vox_build_fast_set(caelo_voxmap *const *maps, const caelo_frame_set &fs, hipStream_t s);
int vox_build_set(caelo_voxmap *const *maps, const caelo_frame_set &fs, bool track_order, hipStream_t s);
int vox_patches_set(const caelo_frame_set &fs, int64_t k_max, bool check_counts, hipStream_t s);
int dedup_set(const caelo_frame_set &fs, bool enabled, hipStream_t s);
int vox_build_launch(caelo_voxmap *m, const float *pc, int64_t n, int stride, bool track_order, int32_t *status,
                     hipStream_t s);
int vox_build_fast_launch(caelo_voxmap *m, const float *pc, int64_t n, int stride, int32_t *status, hipStream_t s);
int vox_patches_launch(const caelo_voxmap *m, const float *pts, int pts_ld, int64_t k_max, const int32_t *n_key,
                       uint64_t *bits, uint8_t *flags, int32_t *status, bool check_counts, hipStream_t s,
                       void *dedup_scratch = nullptr);  // non-null: every patch is entered into the frame's dedup table
// ---- a frame's patch buffer: bit-packed patches [3072][64] u64, then the de-duplication tables (dedup.hip) ----------
#define CAELO_FRAME_PATCHES (CAELO_MAX_KEYPTS * 3)
struct caelo_dedup_tables {
    int32_t count;  // distinct patches of the frame
    int32_t pad[63];
    int32_t list[CAELO_FRAME_PATCHES];     // their patch indices (key point * 3 + scale), coarsest scale first
    int32_t slot_of[CAELO_FRAME_PATCHES];  // a distinct patch -> its ROW in the launch set (frame * 3072 + position in the frame's list); a copy -> -(representative + 1), representative = frame * 3072 + patch
};
#define CAELO_FRAME_BITS_BYTES ((size_t)CAELO_FRAME_PATCHES * 64 * 8)
#define CAELO_FRAME_BUF_BYTES (CAELO_FRAME_BITS_BYTES + sizeof(caelo_dedup_tables))
__host__ __device__ inline caelo_dedup_tables *caelo_frame_tables(const uint64_t *frame_bits) {
    return (caelo_dedup_tables *)((char *)frame_bits + CAELO_FRAME_BITS_BYTES);
}
// Equal patches are looked for across ALL frames of a launch set (round 3: a batch of 8 consecutive scans shares ~13 % more of
// its patches than its frames do one by one): the frames of a set enter their patches into ONE table, frame 0's, under the
// index (frame * 3072 + patch); a set of one frame uses the first DD_SLOTS_FRAME slots only (that is all it has to wipe).
#define DD_SLOTS_FRAME 8192    // >= 2.6 x the 3072 patches of a frame
#define DD_SLOTS_SET 65536     // >= 2.6 x the patches of CAELO_FB_MAX frames
#define DD_EMPTY 0xFFFFFFFFFFFFFFFFull
struct DedupScratch {
    unsigned long long table[DD_SLOTS_SET];  // (hash40 << 24) | smallest (frame * 3072 + patch), DD_EMPTY = free (cleared per set)
    int32_t pslot[CAELO_FRAME_PATCHES];      // this frame's patches: their slots in the set's table
    int32_t rep[CAELO_FRAME_PATCHES];        // ... and their representatives (frame * 3072 + patch; itself: a distinct patch)
};
__host__ __device__ inline uint32_t dedup_slot_mask(int n_frames) { return (n_frames > 1 ? DD_SLOTS_SET : DD_SLOTS_FRAME) - 1; }
int64_t dedup_scratch_bytes();
void dedup_clear_item(void *scratch, int n_frames, caelo_clear_list &list);
unsigned long long dedup_hash_mask();  // 40 bits unless CAELO_DEDUP_HASH_BITS shrinks it (collision tests)
bool dedup_enabled(int mode);          // mode bit CAELO_EXTRACT_NO_DEDUP / CAELO_NO_DEDUP=1 switch it off
int dedup_launch(uint64_t *frame_bits, void *scratch, bool enabled, hipStream_t s);  // after k_patches filled the table

// patches of up to CAELO_ENC_MAX_FRAMES frames encoded by one launch set (the fixed costs of the four encoder
// kernels are ~47 us per launch set): frame f = patch / per_frame gets its descriptors in base[f]
#define CAELO_ENC_MAX_FRAMES CAELO_FB_MAX
struct caelo_enc_out {
    float *base[CAELO_ENC_MAX_FRAMES];
    int64_t per_frame;
};
// input side of a launch set.  dedup = 0: n_patches patches, contiguous at bits.  dedup = 1: n_frames frame buffers
// (bits + f * frame_stride u64 words: [per_frame][64] patches followed by the frame's caelo_dedup_tables); only the
// distinct patches of each frame are encoded, into rows f * per_frame + (position in the frame's list), and
// k_enc_head hands patch p of frame f the result of row f * per_frame + slot_of[p].
// encoder workspace header: [0] work counter of the per-workgroup stage-1 / conv-2 kernels, [8..15] executed-MFMA count of the
// last counting launch, [1024..2047] eight per-XCD queue counters of k_enc_stage1x (a 128-byte line each), [2048..3071] the same for
// k_enc_conv3 (each kernel zeroes the other's).  Zero between calls.
#define CAELO_ENC_WS_HEADER 4096
#define CAELO_ENC_WS_MFMA 3072     // bytes 3072 .. 4095 of the header: stage 1's MFMA counters (profiling launches only)
struct caelo_enc_in {
    const unsigned long long *bits;
    int64_t frame_stride;
    int32_t per_frame, n_frames, dedup;
    int32_t yield;  // the launch shares the GPU with other streams (frame pipeline): persistent grids leave CU slots free --
                    // bit 0: stage 1 a fifth of its slots, bit 1: conv3 half of its, bit 2: conv3 a quarter
};
__device__ inline const caelo_dedup_tables *enc_tables(const caelo_enc_in &in, int f) {
    return (const caelo_dedup_tables *)(in.bits + (size_t)f * in.frame_stride + (size_t)in.per_frame * 64);
}
int encode_batch_impl(caelo_ctx *c, const uint64_t *bits, int64_t n_patches, int group, const caelo_enc_out &outs,
                      int out_stride, void *ws, hipStream_t s, hipEvent_t *ev, const caelo_enc_in *in = nullptr);
int64_t enc_dense_pad(int64_t n);  // rows padded to whole dense-1 tiles
int enc_upload_dense1(const float *wd1, const float *bd1, int K, void **wx_dev, float **bd_dev);
int64_t enc_dense32_part_bytes(int64_t np);
int enc_dense32_head_launch(caelo_ctx *c, const float *f3, int64_t n_patches, int64_t np, float *part, int group, float *out,
                            int out_stride, hipStream_t s);
int encode_impl(caelo_ctx *c, const uint64_t *bits, int64_t n_patches, int group, float *out, int out_stride, void *ws,
                hipStream_t s, hipEvent_t *ev);

// caelo_extract in two halves (frame.hip), for the frame pipeline
struct caelo_extract_args {
    caelo_ctx *ctx;
    caelo_voxmap *map;
    const float *pc;
    int64_t n;
    int dist_channels, mode;
    float *key_pts;
    int kp_ld;
    float *features;
    int feat_ld;
    float *valid;
    int valid_ld;
    int64_t *key_pixels;
    int32_t *n_key;
    uint8_t *flags;
    int32_t *status;
    void *ws;
    uint64_t *bits;  // bit-packed patches [1024][3][64]; null = inside ws (the pipeline points it into a batch buffer)
};
int extract_check(const caelo_extract_args &a);
int extract_front_launch(const caelo_extract_args &a, hipStream_t s);
int extract_front_set(const caelo_extract_args *args, int n, hipStream_t s, hipStream_t s_vox = nullptr, hipEvent_t ev_fork = nullptr,
                      hipEvent_t ev_join = nullptr);  // s_vox: build the voxel maps beside the key-point chain  // n frames, the launches of one   // everything up to the bit-packed patches
int extract_encode_launch(const caelo_extract_args &a, hipStream_t s);  // the four encoder kernels

#define CAELO_KP_HIST_BINS 2048

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. stalls the
// whole workgroup on every global load / store / atomic still in flight (microseconds for a contended
// atomic); kernels that hand data between waves through LDS and keep global traffic asynchronous use this.
// Frame <-> XCD: the workgroups of a launch are dealt to the 8 XCDs round robin (workgroup b runs on XCD b % 8).  A launch over
// (blocks per frame) x (n frames) as ONE row of blocks-per-frame * n workgroups, frame = b % n when n is 8, 4, 2 or 1: the
// workgroups of a frame then share one XCD (or 2, 4, 8 of them) and that XCD's L2 holds ONE frame's hash tables instead of a slice
// of all eight (k_patches 50 -> 44 us per 8 frames; the voxel build did not gain: k_vox_points 39 -> 38, k_vox_coarse 30 -> 48 us).  `block`: the workgroup's index within its frame, `per_frame`: their number.
__device__ inline void caelo_frame_block(int n_frames, unsigned &frame, unsigned &block, unsigned &per_frame) {
    per_frame = gridDim.x / (unsigned)n_frames;
    if ((n_frames & (n_frames - 1)) == 0) {
        frame = blockIdx.x & (unsigned)(n_frames - 1);
        block = blockIdx.x / (unsigned)n_frames;
    } else {
        frame = blockIdx.x / per_frame;
        block = blockIdx.x - frame * per_frame;
    }
}

__device__ inline void caelo_lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__host__ __device__ inline unsigned long long caelo_pack3(int x, int y, int z) {
    return ((unsigned long long)(unsigned)(x & 0xFFFFF) << 40) | ((unsigned long long)(unsigned)(y & 0xFFFFF) << 20) |
           (unsigned long long)(unsigned)(z & 0xFFFFF);
}

__device__ inline uint32_t caelo_hash64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (uint32_t)k;
}

// One wavefront holds patch p, lane l its word l: hash the 64 words and claim / join the hash's entry of the frame's
// de-duplication table (dedup.hip); the entry keeps the smallest patch index of the group.
__device__ inline unsigned long long caelo_dd_mix(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
// S: the frame's scratch (pslot); T: the set's table (frame 0's); gp = frame * 3072 + p
__device__ inline void caelo_dedup_insert(unsigned long long word, int lane, int p, int gp, DedupScratch *S, DedupScratch *T, uint32_t slot_mask,
                                          unsigned long long hash_mask) {
    unsigned long long h = caelo_dd_mix(word + 0x9E3779B97F4A7C15ull * (unsigned)(lane + 1));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);  // order independent across lanes, position dependent per word
    if (lane != 0) return;
    h = caelo_dd_mix(h) & hash_mask & 0xFFFFFFFFFEull;  // 40 bits, never all ones
    const unsigned long long mine = (h << 24) | (unsigned)gp;
    uint32_t slot = (uint32_t)caelo_dd_mix(h) & slot_mask;
    for (;;) {
        // compare-and-swap first, no look: one memory round trip for the wave to wait out instead of two (the table is mostly
        // empty at its load factor, and a wave of k_patches lives ~5 round trips in all)
        const unsigned long long cur = atomicCAS(&T->table[slot], DD_EMPTY, mine);
        if (cur == DD_EMPTY) break;
        if ((cur >> 24) == h) {
            // hundreds of patches share a popular pattern: only a smaller index than the one seen needs the atomic (a
            // stale larger value only costs an atomic that changes nothing)
            if (mine < cur) atomicMin(&T->table[slot], mine);
            break;
        }
        slot = (slot + 1) & slot_mask;
    }
    S->pslot[p] = (int32_t)slot;
}

// read-only lookup: returns slot or -1
__device__ inline int caelo_brick_find(const caelo_brick_table &t, unsigned long long key) {
    uint32_t h = caelo_hash64(key) & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        const unsigned long long k = t.keys[h];
        if (k == key) return (int)h;
        if (k == CAELO_EMPTY_KEY) return -1;
        h = (h + 1) & t.mask;
    }
    return -1;
}

static inline hipStream_t caelo_stream(void *s) { return (hipStream_t)s; }

// ---- one global atomic per WORKGROUP ------------------------------------------------------------------
// Device-scope atomics execute at the memory side on MI355X (8 XCDs, non-coherent L2s): ~10 ns each when
// they hit the same address, i.e. 2000 per-wavefront appends to one counter cost 20+ us whatever else the
// kernel does.  These helpers funnel a workgroup's appends / sums through LDS first.  Every thread of the
// workgroup must call them (two barriers inside); s_tmp is a caller-provided __shared__ int[2].
#ifdef __HIPCC__
__device__ inline int caelo_block_reserve(int32_t *gcounter, bool pred, int *s_tmp) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_tmp[0] = 0;
    __syncthreads();
    const unsigned long long m = __ballot(pred);
    int wbase = 0;
    if (m && lane == __ffsll((long long)m) - 1) wbase = atomicAdd(&s_tmp[0], __popcll(m));
    if (m) wbase = __shfl(wbase, __ffsll((long long)m) - 1);
    __syncthreads();
    if (threadIdx.x == 0) s_tmp[1] = s_tmp[0] ? atomicAdd(gcounter, s_tmp[0]) : 0;
    __syncthreads();
    return s_tmp[1] + wbase + __popcll(m & ((1ull << lane) - 1ull));
}
// The same with LDS-only barriers: the caller's own global loads / atomics stay in flight across it (a __syncthreads() waits
// for them: vmcnt(0)).  s_tmp: __shared__ int[2], not otherwise in use.
__device__ inline int caelo_block_reserve_async(int32_t *gcounter, bool pred, int *s_tmp) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_tmp[0] = 0;
    caelo_lds_barrier();
    const unsigned long long m = __ballot(pred);
    int wbase = 0;
    if (m && lane == __ffsll((long long)m) - 1) wbase = atomicAdd(&s_tmp[0], __popcll(m));
    if (m) wbase = __shfl(wbase, __ffsll((long long)m) - 1);
    caelo_lds_barrier();
    if (threadIdx.x == 0) s_tmp[1] = s_tmp[0] ? atomicAdd(gcounter, s_tmp[0]) : 0;
    caelo_lds_barrier();
    return s_tmp[1] + wbase + __popcll(m & ((1ull << lane) - 1ull));
}
__device__ inline void caelo_block_add(int32_t *gcounter, int value, int *s_tmp) {
    if (threadIdx.x == 0) s_tmp[0] = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) value += __shfl_xor(value, o);
    if ((threadIdx.x & 63) == 0 && value) atomicAdd(&s_tmp[0], value);
    __syncthreads();
    if (threadIdx.x == 0 && s_tmp[0]) atomicAdd(gcounter, s_tmp[0]);
}
#endif
