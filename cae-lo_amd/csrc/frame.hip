// frame.hip -- the fused per-scan hot path behind ONE C-ABI call, and the multi-buffer clear.
//
// caelo_extract = project -> response CNN -> keypoints -> voxelize -> patch gather -> 3x encoder
// (SphericalRing.py:72-94,:405-416; Voxel.py:100-216; Match.py:130-135) on one stream with no host
// synchronisation and no allocation: every intermediate (ring image, response image, candidate
// list, packed patches, encoder activations) lives in a caller-provided workspace, so a frame costs
// ~15 kernel launches and two ctypes calls instead of ~30 launches and a dozen Python round trips.
#include "caelo_internal.h"

// ------------------------------------------------------------------------------------------------
// clear several buffers with one launch (each region a multiple of 16 bytes, 16-byte aligned)
// ------------------------------------------------------------------------------------------------
struct ClearArgs {
    uint4 *ptr[CAELO_CLEAR_MAX];
    unsigned long long end[CAELO_CLEAR_MAX];  // cumulative uint4 counts
    uint32_t pattern[CAELO_CLEAR_MAX];
    int n;
};

struct ClearSet {   // blockIdx.y = frame
    uint4 *ptr[CAELO_FB_MAX][CAELO_CLEAR_MAX];
    unsigned int end[CAELO_FB_MAX][CAELO_CLEAR_MAX];  // cumulative uint4 counts (< 2^32 x 16 B per frame)
    uint32_t pattern[CAELO_FB_MAX][CAELO_CLEAR_MAX];
    int n[CAELO_FB_MAX];
};
static_assert(sizeof(ClearSet) <= 3800, "ClearSet must fit the kernel argument segment");

__global__ void __launch_bounds__(256) k_clear_set(const ClearSet a) {
    const int f = blockIdx.y;
    const int n = a.n[f];
    if (n == 0) return;
    const unsigned int total = a.end[f][n - 1];
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int r = 0;
        while (i >= a.end[f][r]) ++r;
        const unsigned int local = i - (r ? a.end[f][r - 1] : 0u);
        const uint32_t p = a.pattern[f][r];
        a.ptr[f][r][local] = make_uint4(p, p, p, p);
    }
}

__global__ void __launch_bounds__(256) k_clear_many(ClearArgs a) {
    const unsigned long long total = a.end[a.n - 1];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        int r = 0;
        while (i >= a.end[r]) ++r;
        const unsigned long long local = i - (r ? a.end[r - 1] : 0ull);
        const uint32_t p = a.pattern[r];
        a.ptr[r][local] = make_uint4(p, p, p, p);
    }
}

int caelo_clear_many(const caelo_clear_list &list, hipStream_t s) {
    if (list.n == 0) return CAELO_OK;
    ClearArgs a;
    unsigned long long cum = 0;
    a.n = 0;
    for (int i = 0; i < list.n; ++i) {
        const caelo_clear_item &it = list.item[i];
        if (it.bytes == 0) continue;
        if (((uintptr_t)it.ptr & 15u) || (it.bytes & 15u)) {
            // unaligned tail: fall back to the runtime fill for this item
            CAELO_HIP(hipMemsetAsync(it.ptr, (int)(it.pattern & 0xFF), it.bytes, s));
            continue;
        }
        cum += it.bytes / 16;
        a.ptr[a.n] = (uint4 *)it.ptr;
        a.end[a.n] = cum;
        a.pattern[a.n] = it.pattern;
        ++a.n;
    }
    if (a.n == 0) return CAELO_OK;
    unsigned long long blocks = (cum + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride: 16 workgroups per CU
    k_clear_many<<<(unsigned)blocks, 256, 0, s>>>(a);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

int caelo_clear_many_set(const caelo_clear_list *lists, int n_frames, hipStream_t s) {
    if (n_frames == 1) return caelo_clear_many(lists[0], s);
    ClearSet a;
    unsigned long long most = 0;
    for (int f = 0; f < n_frames; ++f) {
        unsigned long long cum = 0;
        a.n[f] = 0;
        for (int i = 0; i < lists[f].n; ++i) {
            const caelo_clear_item &it = lists[f].item[i];
            if (it.bytes == 0) continue;
            if (((uintptr_t)it.ptr & 15u) || (it.bytes & 15u)) {
                CAELO_HIP(hipMemsetAsync(it.ptr, (int)(it.pattern & 0xFF), it.bytes, s));
                continue;
            }
            cum += it.bytes / 16;
            a.ptr[f][a.n[f]] = (uint4 *)it.ptr;
            a.end[f][a.n[f]] = (unsigned int)cum;
            a.pattern[f][a.n[f]] = it.pattern;
            ++a.n[f];
        }
        if (cum >= 0xFFFFFFFFull) { caelo_set_error("caelo_clear_many_set: region too large"); return CAELO_ERR_ARG; }
        most = cum > most ? cum : most;
    }
    if (most == 0) return CAELO_OK;
    unsigned long long blocks = (most + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // grid-stride
    k_clear_set<<<dim3((unsigned)blocks, n_frames), 256, 0, s>>>(a);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// fused extract
// ------------------------------------------------------------------------------------------------
struct ExtractLayout {
    size_t ring, winner, resp, cand, cand_count, bits, dd, enc, total;
};

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static ExtractLayout extract_layout() {
    ExtractLayout L;
    size_t off = 0;
    const size_t npix = (size_t)CAELO_RING_H * CAELO_RING_W;
    // winner | cand_count are cleared together: keep them adjacent
    L.winner = off; off += align256(npix * 4);
    L.cand_count = off; off += 256;
    L.ring = off; off += align256(npix * CAELO_RING_C * 4);
    L.resp = off; off += align256((size_t)CAELO_NET_H * CAELO_NET_W * 8 * 4);
    L.cand = off; off += align256((size_t)CAELO_NET_H * CAELO_NET_W * 8);
    L.bits = off; off += align256(CAELO_FRAME_BUF_BYTES);  // bit-packed patches + de-duplication tables
    L.dd = off; off += align256((size_t)dedup_scratch_bytes());
    L.enc = off; off += align256((size_t)caelo_encode_ws_bytes(CAELO_MAX_KEYPTS * 3));
    L.total = off;
    return L;
}

CAELO_API int64_t caelo_extract_ws_bytes(void) { return (int64_t)extract_layout().total; }

// The fused path in two halves so that the frame pipeline can put an event edge between them:
// front = clear + ring image + response + keypoints + voxel map + patch gather (latency-bound kernels),
// encode = the four MFMA-bound encoder kernels.
int extract_check(const caelo_extract_args &a) {
    CAELO_REQUIRE(a.ctx && a.map && a.pc && a.key_pts && a.features && a.key_pixels && a.n_key && a.flags && a.status && a.ws,
                  "null argument");
    CAELO_REQUIRE(a.ctx->has_resp && a.ctx->has_enc, "weights not set");
    CAELO_REQUIRE(a.n > 3, "PC.shape[0] > 3 (SphericalRing.py:73)");
    CAELO_REQUIRE(a.dist_channels == 5 || a.dist_channels == 3, "dist_channels must be 5 (demo mode) or 3 (batch mode)");
    CAELO_REQUIRE(a.kp_ld >= 3 && a.feat_ld >= 60, "bad leading dimension");
    CAELO_REQUIRE(((uintptr_t)a.status & 15u) == 0, "status must be a 16-byte aligned int32[4]");
    if (a.n > a.map->max_points) {
        caelo_set_error("caelo_extract: %lld points exceed the map capacity %lld", (long long)a.n, (long long)a.map->max_points);
        return CAELO_ERR_CAPACITY;
    }
    return CAELO_OK;
}

int extract_front_launch(const caelo_extract_args &a, hipStream_t s) { return extract_front_set(&a, 1, s); }

// The front halves of n frames (n <= CAELO_FB_MAX, one mode for all) with the launches of one: every kernel takes the
// frame set and runs frame blockIdx.z.  Each frame brings its own voxel map and workspace.
int extract_front_set(const caelo_extract_args *args, int n, hipStream_t s, hipStream_t s_vox, hipEvent_t ev_fork, hipEvent_t ev_join) {
    CAELO_REQUIRE(n >= 1 && n <= CAELO_FB_MAX, "bad frame count");
    const ExtractLayout L = extract_layout();
    const bool exact_vox = (args[0].mode & CAELO_EXTRACT_EXACT_VOXELS) != 0;
    const bool dd = dedup_enabled(args[0].mode);
    caelo_frame_set fs = {};
    fs.n = n;
    caelo_clear_list cl[CAELO_FB_MAX];
    caelo_voxmap *maps[CAELO_FB_MAX];
    for (int i = 0; i < n; ++i) {
        const caelo_extract_args &a = args[i];
        CAELO_REQUIRE(a.mode == args[0].mode, "the frames of a set share one mode");
        char *ws = (char *)a.ws;
        caelo_frame_dev &d = fs.f[i];
        maps[i] = a.map;
        frame_dev_set_map(d, a.map);
        d.pc = a.pc; d.n = a.n; d.pc_stride = 4; d.dist_c = a.dist_channels;
        d.ring = (float *)(ws + L.ring); d.counter = nullptr; d.winner = (int32_t *)(ws + L.winner);  // (occupied = has a winner)
        d.resp = (float *)(ws + L.resp); d.cand = (unsigned long long *)(ws + L.cand); d.cand_count = (int32_t *)(ws + L.cand_count);
        d.key_pixels = a.key_pixels; d.key_pts = a.key_pts; d.kp_ld = a.kp_ld; d.valid = a.valid; d.valid_ld = a.valid_ld;
        d.n_key = a.n_key; d.flags = a.flags; d.status = a.status;
        d.bits = (unsigned long long *)(a.bits ? a.bits : (uint64_t *)(ws + L.bits));
        d.dd = dd ? (DedupScratch *)(ws + L.dd) : nullptr;
        // ---- one clear for everything the frame accumulates into
        cl[i].n = 0;
        cl[i].item[cl[i].n++] = {d.winner, L.cand_count - L.winner, 0xFFFFFFFFu};
        cl[i].item[cl[i].n++] = {d.cand_count, L.ring - L.cand_count, 0u};
        cl[i].item[cl[i].n++] = {a.status, 16, 0u};  // status is int32[4], 16-byte aligned (word 0 carries the bits)
        if (exact_vox) vox_clear_items(a.map, 1, cl[i]);
        if (i == 0) dedup_clear_item(ws + L.dd, n, cl[i]);  // the set's table lives in frame 0's scratch
    }
    int rc = CAELO_OK;
    if (!exact_vox && (rc = vox_clear_for_fast_build_set(maps, n, cl, s))) return rc;  // wipes the previous frames' bricks only
    if ((rc = caelo_clear_many_set(cl, n, s))) return rc;
    // The voxel map only needs the points: with a second stream (the frame pipeline passes one when the runtime has hardware
    // queues to spare) it is built beside the ring image -> response -> key points chain; both meet before the patch gather.
    hipStream_t sv = s_vox ? s_vox : s;
    if (s_vox) {
        CAELO_HIP(hipEventRecord(ev_fork, s));
        CAELO_HIP(hipStreamWaitEvent(s_vox, ev_fork, 0));
    }
    // ---- voxel map
    if (exact_vox) rc = vox_build_set(maps, fs, false, sv);
    else rc = vox_build_fast_set(maps, fs, sv);
    if (rc) return rc;
    if (s_vox) CAELO_HIP(hipEventRecord(ev_join, s_vox));
    // ---- ring image, response, keypoints
    if ((rc = ring_project_set(fs, s))) return rc;
    // (the key point rule reads response rows 8..55 and their 5 x 5 neighbourhoods: rows 6..57 of 64)
    if ((rc = ring_respond_set(args[0].ctx, fs, CAELO_RING_W, CAELO_RING_C, s, 6, 52))) return rc;
    if ((rc = ring_keypoints_set(fs, CAELO_RING_W, CAELO_RING_C, CAELO_RING_W, s))) return rc;
    if (s_vox) CAELO_HIP(hipStreamWaitEvent(s, ev_join, 0));
    // ---- patches
    // equal patches are encoded once (dedup.hip): k_patches enters every patch into the hash table, the tables land
    // behind the frame's bits
    if ((rc = vox_patches_set(fs, CAELO_MAX_KEYPTS, true, s))) return rc;
    return dedup_set(fs, dd, s);
}

int extract_encode_launch(const caelo_extract_args &a, hipStream_t s) {
    const ExtractLayout L = extract_layout();
    char *ws = (char *)a.ws;
    const uint64_t *bits = a.bits ? a.bits : (const uint64_t *)(ws + L.bits);
    caelo_enc_out outs;
    outs.base[0] = a.features;
    outs.per_frame = CAELO_FRAME_PATCHES;
    const caelo_enc_in in = {(const unsigned long long *)bits, 0, CAELO_FRAME_PATCHES, 1, 1, 0};
    return encode_batch_impl(a.ctx, bits, CAELO_FRAME_PATCHES, 3, outs, a.feat_ld, ws + L.enc, s, nullptr, &in);
}

CAELO_API int caelo_extract(caelo_ctx *c, caelo_voxmap *m, const float *pc, int64_t n, int dist_channels, int mode,
                            float *key_pts, int kp_ld, float *features, int feat_ld, float *valid, int valid_ld,
                            int64_t *key_pixels, int32_t *n_key, uint8_t *flags, int32_t *status, void *wsv,
                            void *stream) {
    const caelo_extract_args a = {c, m, pc, n, dist_channels, mode, key_pts, kp_ld, features, feat_ld, valid, valid_ld,
                                  key_pixels, n_key, flags, status, wsv, nullptr};
    int rc = extract_check(a);
    if (rc) return rc;
    hipStream_t s = caelo_stream(stream);
    if ((rc = extract_front_launch(a, s))) return rc;
    return extract_encode_launch(a, s);
}
