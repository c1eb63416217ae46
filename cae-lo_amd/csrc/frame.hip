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

// ------------------------------------------------------------------------------------------------
// fused extract
// ------------------------------------------------------------------------------------------------
struct ExtractLayout {
    size_t ring, counter, winner, resp, cand, hist, cand_count, bits, enc, total;
};

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static ExtractLayout extract_layout() {
    ExtractLayout L;
    size_t off = 0;
    const size_t npix = (size_t)CAELO_RING_H * CAELO_RING_W;
    // winner | counter | hist | cand_count are cleared together: keep them adjacent
    L.winner = off; off += align256(npix * 4);
    L.counter = off; off += align256(npix * 4);
    L.hist = off; off += (size_t)CAELO_KP_HIST_BINS * 4;
    L.cand_count = off; off += 256;
    L.ring = off; off += align256(npix * CAELO_RING_C * 4);
    L.resp = off; off += align256((size_t)CAELO_NET_H * CAELO_NET_W * 8 * 4);
    L.cand = off; off += align256((size_t)CAELO_NET_H * CAELO_NET_W * 8);
    L.bits = off; off += align256((size_t)CAELO_MAX_KEYPTS * 3 * 64 * 8);
    L.enc = off; off += align256((size_t)caelo_encode_ws_bytes(CAELO_MAX_KEYPTS * 3));
    L.total = off;
    return L;
}

CAELO_API int64_t caelo_extract_ws_bytes(void) { return (int64_t)extract_layout().total; }

CAELO_API int caelo_extract(caelo_ctx *c, caelo_voxmap *m, const float *pc, int64_t n, int dist_channels, int mode,
                            float *key_pts, int kp_ld, float *features, int feat_ld, float *valid, int valid_ld,
                            int64_t *key_pixels, int32_t *n_key, uint8_t *flags, int32_t *status, void *wsv,
                            void *stream) {
    CAELO_REQUIRE(c && m && pc && key_pts && features && key_pixels && n_key && flags && status && wsv, "null argument");
    CAELO_REQUIRE(c->has_resp && c->has_enc, "weights not set");
    CAELO_REQUIRE(n > 3, "PC.shape[0] > 3 (SphericalRing.py:73)");
    CAELO_REQUIRE(dist_channels == 5 || dist_channels == 3, "dist_channels must be 5 (demo mode) or 3 (batch mode)");
    CAELO_REQUIRE(kp_ld >= 3 && feat_ld >= 60, "bad leading dimension");
    CAELO_REQUIRE(((uintptr_t)status & 15u) == 0, "status must be a 16-byte aligned int32[4]");
    if (n > m->max_points) {
        caelo_set_error("caelo_extract: %lld points exceed the map capacity %lld", (long long)n, (long long)m->max_points);
        return CAELO_ERR_CAPACITY;
    }
    hipStream_t s = caelo_stream(stream);
    const ExtractLayout L = extract_layout();
    char *ws = (char *)wsv;
    float *ring = (float *)(ws + L.ring);
    int32_t *counter = (int32_t *)(ws + L.counter);
    int32_t *winner = (int32_t *)(ws + L.winner);
    float *resp = (float *)(ws + L.resp);
    unsigned long long *cand = (unsigned long long *)(ws + L.cand);
    uint32_t *hist = (uint32_t *)(ws + L.hist);
    int32_t *cand_count = (int32_t *)(ws + L.cand_count);
    uint64_t *bits = (uint64_t *)(ws + L.bits);
    // ---- one clear for everything the frame accumulates into
    caelo_clear_list cl;
    cl.n = 0;
    cl.item[cl.n++] = {winner, L.counter - L.winner, 0xFFFFFFFFu};
    cl.item[cl.n++] = {counter, L.ring - L.counter, 0u};  // counter | hist | cand_count
    cl.item[cl.n++] = {status, 16, 0u};  // status is int32[4], 16-byte aligned (word 0 carries the bits)
    const bool exact_vox = (mode & CAELO_EXTRACT_EXACT_VOXELS) != 0;
    vox_clear_items(m, exact_vox ? 1 : 0, cl);
    int rc = caelo_clear_many(cl, s);
    if (rc) return rc;
    // ---- ring image, response, keypoints
    if ((rc = ring_project_launch(pc, n, ring, counter, winner, status, s))) return rc;
    if ((rc = ring_respond_launch(c, ring, CAELO_RING_W, CAELO_RING_C, resp, s))) return rc;
    if ((rc = ring_keypoints_launch(ring, CAELO_RING_W, CAELO_RING_C, dist_channels, counter, CAELO_RING_W, resp, cand,
                                    hist, cand_count, key_pixels, key_pts, kp_ld, valid, valid_ld, n_key, status, s)))
        return rc;
    // ---- voxel map, patches, descriptors
    if (exact_vox) rc = vox_build_launch(m, pc, n, 4, false, status, s);
    else rc = vox_build_fast_launch(m, pc, n, 4, status, s);
    if (rc) return rc;
    if ((rc = vox_patches_launch(m, key_pts, kp_ld, CAELO_MAX_KEYPTS, n_key, bits, flags, status, true, s))) return rc;
    return encode_impl(c, bits, CAELO_MAX_KEYPTS * 3, 3, features, feat_ld, ws + L.enc, s, nullptr);
}
