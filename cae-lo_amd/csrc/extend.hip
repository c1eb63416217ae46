// extend.hip -- ExtendKeyPtsInShpericalRing (SphericalRing.py:294-317; caller BatchPreprocess.py:139): the points of
// the 13 x 13 ring neighbourhood of every keypixel, for the pose refinement that follows the odometry.
//
// The reference walks the keypixels in order, gathers the occupied pixels of the window in row-major order and then
// ZEROES the window in the caller's GridCounter, so a pixel belongs to the FIRST keypixel whose window covers it.
// Restated data-parallel:  owner[pixel] = min k over the windows that cover it (atomicMin);  keypixel k emits the
// occupied pixels it owns, row-major; offsets are an exclusive scan of the per-keypixel counts; every owned pixel
// of the counter is cleared afterwards (the reference's in-place side effect, kept).
#include "caelo_internal.h"

#define EXT_R 6                      // nNeighborRadius (:295)
#define EXT_W (2 * EXT_R + 1)        // 13
#define EXT_CELLS (EXT_W * EXT_W)    // 169

__global__ void __launch_bounds__(256) k_ext_claim(const int64_t *__restrict__ kpix, const int32_t *__restrict__ n_key,
                                                   int k_max, int rows, int cols, int32_t *__restrict__ owner) {
    const int K = n_key ? min(max(*n_key, 0), k_max) : k_max;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = idx / EXT_CELLS, c = idx - k * EXT_CELLS;
    if (k >= K) return;
    const int r = (int)kpix[2 * k] + c / EXT_W - EXT_R, q = (int)kpix[2 * k + 1] + c % EXT_W - EXT_R;
    if (r >= 0 && r < rows && q >= 0 && q < cols) atomicMin(&owner[r * cols + q], k);
}

// one workgroup, thread k = keypixel k: count, block scan, emit in window row-major order, clear the counter
__global__ void __launch_bounds__(CAELO_MAX_KEYPTS) k_ext_emit(const float *__restrict__ ring, int ring_w, int ring_c,
                                                                int32_t *__restrict__ counter, int cnt_w,
                                                                const int64_t *__restrict__ kpix,
                                                                const int32_t *__restrict__ n_key, int k_max, int rows,
                                                                int cols, const int32_t *__restrict__ owner,
                                                                float *__restrict__ ext, int32_t *__restrict__ n_ext) {
    __shared__ int s_scan[CAELO_MAX_KEYPTS];
    const int K = n_key ? min(max(*n_key, 0), k_max) : k_max;
    const int k = threadIdx.x;
    int r0 = 0, q0 = 0, cnt = 0;
    if (k < K) {
        r0 = (int)kpix[2 * k] - EXT_R;
        q0 = (int)kpix[2 * k + 1] - EXT_R;
        for (int c = 0; c < EXT_CELLS; ++c) {
            const int r = r0 + c / EXT_W, q = q0 + c % EXT_W;
            if (r >= 0 && r < rows && q >= 0 && q < cols && owner[r * cols + q] == k && counter[r * cnt_w + q] > 0) ++cnt;
        }
    }
    s_scan[k] = cnt;
    __syncthreads();
    for (int off = 1; off < CAELO_MAX_KEYPTS; off <<= 1) {  // Hillis-Steele inclusive scan
        const int v = k >= off ? s_scan[k - off] : 0;
        __syncthreads();
        s_scan[k] += v;
        __syncthreads();
    }
    int pos = s_scan[k] - cnt;
    if (k == CAELO_MAX_KEYPTS - 1) *n_ext = s_scan[k];
    if (k >= K) return;
    for (int c = 0; c < EXT_CELLS; ++c) {
        const int r = r0 + c / EXT_W, q = q0 + c % EXT_W;
        if (r < 0 || r >= rows || q < 0 || q >= cols || owner[r * cols + q] != k) continue;
        int32_t *cc = &counter[r * cnt_w + q];
        if (*cc > 0) {
            const float *p = ring + ((size_t)r * ring_w + q) * ring_c;
            ext[3 * (size_t)pos] = p[0]; ext[3 * (size_t)pos + 1] = p[1]; ext[3 * (size_t)pos + 2] = p[2];
            ++pos;
        }
        *cc = 0;  // oneMask[:] = 0 (:307): the caller's counter loses every window
    }
}

CAELO_API int64_t caelo_extend_ws_bytes(int rows, int cols) { return ((int64_t)rows * cols * 4 + 255) / 256 * 256; }

CAELO_API int caelo_extend_keypts(caelo_ctx *c, const float *ring, int ring_w, int ring_c, int32_t *counter, int cnt_w,
                                  int rows, int cols, const int64_t *key_pixels, int k_max, const int32_t *n_key,
                                  float *ext_pts, int32_t *n_ext, void *ws, void *stream) {
    CAELO_REQUIRE(c && ring && counter && key_pixels && ext_pts && n_ext && ws, "null argument");
    CAELO_REQUIRE(k_max > 0 && k_max <= CAELO_MAX_KEYPTS && ring_c >= 3, "bad shape");
    CAELO_REQUIRE(rows > 0 && cols > 0 && cols <= ring_w && cols <= cnt_w, "rows x cols must lie inside the ring and the counter");
    hipStream_t s = caelo_stream(stream);
    int32_t *owner = (int32_t *)ws;
    caelo_clear_list cl;
    cl.n = 0;
    cl.item[cl.n++] = {owner, (size_t)caelo_extend_ws_bytes(rows, cols), 0x7F7F7F7Fu};
    int rc = caelo_clear_many(cl, s);
    if (rc) return rc;
    k_ext_claim<<<(unsigned)((k_max * EXT_CELLS + 255) / 256), 256, 0, s>>>(key_pixels, n_key, k_max, rows, cols, owner);
    CAELO_LAUNCH_CHECK();
    k_ext_emit<<<1, CAELO_MAX_KEYPTS, 0, s>>>(ring, ring_w, ring_c, counter, cnt_w, key_pixels, n_key, k_max, rows, cols, owner,
                                              ext_pts, n_ext);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
