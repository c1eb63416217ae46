// ring.hip -- spherical-ring front end: projection, 2D-CAE response layer, keypoint rule.
//
// Reference behaviour restated here (never its code):
//   ProjectPC2SphericalRing  SphericalRing.py:72-94
//   RespondLayer.predict     SphericalRingPCRespondLayer.h5 (Conv2D 3->32 3x3 relu, Conv2D 32->8 1x1 relu)
//   GetKeyPtsByAE            SphericalRing.py:113-291
// All three are HBM/latency-bound at these sizes (2 MB of points, 3.7 MB response image):
// one thread per point / pixel, coalesced rows, no intermediate tensors (the reference's CuPy path
// materialises a 92 MB [64,1792,25,8] difference tensor, SphericalRing.py:144).
#include <stdarg.h>

#include "caelo_internal.h"

// ------------------------------------------------------------------------------------------------
// error string + context
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void caelo_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

CAELO_API const char *caelo_last_error(void) { return g_err; }
CAELO_API int caelo_abi_version(void) { return CAELO_ABI_VERSION; }

CAELO_API int caelo_create(caelo_ctx **out, int device) {
    CAELO_REQUIRE(out != nullptr, "null ctx pointer");
    int ndev = 0;
    CAELO_HIP(hipGetDeviceCount(&ndev));
    CAELO_REQUIRE(device >= 0 && device < ndev, "no such HIP device (libcaelo needs a GPU; there is no CPU fallback)");
    CAELO_HIP(hipSetDevice(device));
    caelo_ctx *c = new caelo_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    *out = c;
    return CAELO_OK;
}

CAELO_API void caelo_destroy(caelo_ctx *c) {
    if (!c) return;
    float *ptrs[] = {c->resp_w, c->enc_c0, c->enc_w1, c->enc_b1, c->enc_w2, c->enc_b2, c->enc_w3,
                     c->enc_b3, c->enc_wd1, c->enc_bd1, c->enc_wd2, c->enc_bd2};
    for (float *p : ptrs)
        if (p) (void)hipFree(p);
    delete c;
}

CAELO_API int caelo_set_respond_weights(caelo_ctx *c, const float *w1, const float *b1, const float *w2,
                                        const float *b2) {
    CAELO_REQUIRE(c && w1 && b1 && w2 && b2, "null argument");
    const size_t n = 27 * 32 + 32 + 32 * 8 + 8;
    if (!c->resp_w) CAELO_HIP(hipMalloc(&c->resp_w, n * sizeof(float)));
    float host[27 * 32 + 32 + 32 * 8 + 8];
    memcpy(host, w1, 27 * 32 * 4);
    memcpy(host + 864, b1, 32 * 4);
    memcpy(host + 896, w2, 256 * 4);
    memcpy(host + 1152, b2, 8 * 4);
    CAELO_HIP(hipMemcpy(c->resp_w, host, n * sizeof(float), hipMemcpyHostToDevice));
    c->has_resp = true;
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// K1: projection.  Last point in file order wins a pixel (SphericalRing.py:81-93): atomicMax of the
// point index per pixel, then one gather pass writes the 5-channel ring.
// ------------------------------------------------------------------------------------------------
struct ProjConst {
    double pi, az_res, v_res, v_off;
};

__global__ void __launch_bounds__(256) k_project_points(const float4 *__restrict__ pc, int64_t n, int32_t *winner,
                                                        int32_t *counter, int32_t *status, ProjConst k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pc[i];
    // :77 LA.norm axis=1 in f32: squares, sequential sum, sqrt
    float s = __fmul_rn(p.x, p.x);
    s = __fadd_rn(s, __fmul_rn(p.y, p.y));
    s = __fadd_rn(s, __fmul_rn(p.z, p.z));
    const float r = sqrtf(s);
    if (r == 0.0f) return;                                                            // :78-80
    const int col = (int)((k.pi - atan2((double)p.y, (double)p.x)) / k.az_res);      // :86
    const float q = __fdiv_rn(p.z, r);                                                // :87 f32 quotient
    const int row = CAELO_RING_H - (int)(asin((double)q) / k.v_res + k.v_off);        // :88
    if (row < 0 || row >= CAELO_RING_H) return;                                       // :89
    if (col < 0 || col >= CAELO_RING_W) {
        atomicOr(status, CAELO_ST_COL_OOB);
        return;
    }
    const int pix = row * CAELO_RING_W + col;
    atomicMax(&winner[pix], (int32_t)i);
    atomicAdd(&counter[pix], 1);                                                      // :93
}

__global__ void __launch_bounds__(256) k_ring_fill(const float4 *__restrict__ pc, const int32_t *__restrict__ winner,
                                                   float *__restrict__ ring) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= CAELO_RING_H * CAELO_RING_W) return;
    const int w = winner[pix];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (w >= 0) {
        const float4 p = pc[w];
        float s = __fmul_rn(p.x, p.x);
        s = __fadd_rn(s, __fmul_rn(p.y, p.y));
        s = __fadd_rn(s, __fmul_rn(p.z, p.z));
        v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = sqrtf(s);         // :91-92
    }
    float *o = ring + (int64_t)pix * CAELO_RING_C;
#pragma unroll
    for (int c = 0; c < 5; ++c) o[c] = v[c];
}

int ring_project_launch(const float *pc, int64_t n, float *ring, int32_t *counter, int32_t *winner_ws, int32_t *status,
                        hipStream_t s) {
    const int npix = CAELO_RING_H * CAELO_RING_W;
    ProjConst k;
    k.pi = 3.14159265358979323846;
    const double d2r = k.pi / 180.0;                        // SphericalRing.py:28
    k.az_res = 0.20 * d2r;                                  // :35,:48
    const double vdown = -24.8 * d2r, vup = 2.0 * d2r;      // :49-50
    k.v_res = (vup - vdown) / (64 - 1);                     // :51
    k.v_off = -vdown / k.v_res;                             // :52
    k_project_points<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const float4 *)pc, n, winner_ws, counter, status, k);
    CAELO_LAUNCH_CHECK();
    k_ring_fill<<<(npix + 255) / 256, 256, 0, s>>>((const float4 *)pc, winner_ws, ring);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_project(caelo_ctx *c, const float *pc, int64_t n, float *ring, int32_t *counter,
                            int32_t *winner_ws, int32_t *status, void *stream) {
    CAELO_REQUIRE(c && pc && ring && counter && winner_ws && status, "null argument");
    CAELO_REQUIRE(n > 3, "PC.shape[0] > 3 (SphericalRing.py:73)");
    hipStream_t s = caelo_stream(stream);
    const size_t npix = CAELO_RING_H * CAELO_RING_W;
    caelo_clear_list cl;
    cl.n = 0;
    cl.item[cl.n++] = {winner_ws, npix * sizeof(int32_t), 0xFFFFFFFFu};
    cl.item[cl.n++] = {counter, npix * sizeof(int32_t), 0u};
    int rc = caelo_clear_many(cl, s);
    if (rc) return rc;
    return ring_project_launch(pc, n, ring, counter, winner_ws, status, s);
}

// ------------------------------------------------------------------------------------------------
// K2: response layer, fused conv3x3(3->32)+relu+conv1x1(32->8)+relu, one thread per pixel.
// Summation order is the canonical one documented in oracle/caelo_oracle.c (orc_respond) so the
// response image -- and therefore the keypoint indices -- are bit-identical to the oracle's.
// Weights are indexed uniformly across the wave -> scalar loads.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_respond(const float *__restrict__ in, int in_w, int in_c,
                                                 const float *__restrict__ wts, float *__restrict__ resp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= CAELO_NET_W) return;
    const float *w1 = wts, *b1 = wts + 864, *w2 = wts + 896, *b2 = wts + 1152;
    float h[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) h[c] = b1[c];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= CAELO_NET_H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if (xx < 0 || xx >= CAELO_NET_W) continue;
            const float *px = in + ((int64_t)yy * in_w + xx) * in_c;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = px[ci];
                const float *w = w1 + ((ky * 3 + kx) * 3 + ci) * 32;
#pragma unroll
                for (int c = 0; c < 32; ++c) h[c] = fmaf(v, w[c], h[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) h[c] = h[c] > 0.0f ? h[c] : 0.0f;
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float a = b2[k];
#pragma unroll
        for (int c = 0; c < 32; ++c) a = fmaf(h[c], w2[c * 8 + k], a);
        o[k] = a > 0.0f ? a : 0.0f;
    }
    float4 *dst = (float4 *)(resp + ((int64_t)y * CAELO_NET_W + x) * 8);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

int ring_respond_launch(caelo_ctx *c, const float *in, int in_w, int in_c, float *resp, hipStream_t s) {
    dim3 grid((CAELO_NET_W + 255) / 256, CAELO_NET_H);
    k_respond<<<grid, 256, 0, s>>>(in, in_w, in_c, c->resp_w, resp);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_respond(caelo_ctx *c, const float *in, int in_w, int in_c, float *resp, void *stream) {
    CAELO_REQUIRE(c && in && resp, "null argument");
    CAELO_REQUIRE(c->has_resp, "response-layer weights not set (caelo_set_respond_weights)");
    CAELO_REQUIRE(in_w >= CAELO_NET_W && in_c >= 3, "input must hold >= 1792 columns and >= 3 channels");
    return ring_respond_launch(c, in, in_w, in_c, resp, caelo_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// K3: keypoint score.  For every pixel that can be a keypoint (SphericalRing.py:163-167,:186,:197-199,
// :210-213) compute the minimum L2 distance of its response vector to the occupied neighbours of
// the 5x5 window; append key = (float bits of score << 32 | flat index) to a compact list.
// f32 norm in NumPy's 8-lane pairwise order (SURVEY 8a-3'), no FMA contraction.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_kp_score(const float *__restrict__ ring, int ring_w, int ring_c, int dist_c,
                                                  const int32_t *__restrict__ counter, int cnt_w,
                                                  const float *__restrict__ resp, unsigned long long *__restrict__ cand,
                                                  uint32_t *__restrict__ hist, int32_t *cand_count) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y + 8;  // rows 8..55 only
    if (x < 8 || x >= CAELO_NET_W - 8) return;
    if (x >= 56 && x < 64) return;  // the row/column mix-up of :166-167, reproduced
    if (!(counter[y * cnt_w + x] > 0)) return;
    const float4 *rp4 = (const float4 *)(resp + ((int64_t)y * CAELO_NET_W + x) * 8);
    const float4 pa = rp4[0], pb = rp4[1];
    int cnt = 0;
    float best = 0.0f;
    bool have = false;
#pragma unroll
    for (int oy = -2; oy <= 2; ++oy) {
#pragma unroll
        for (int ox = -2; ox <= 2; ++ox) {
            if (oy == 0 && ox == 0) continue;
            const int yy = y + oy, xx = x + ox;
            if (!(counter[yy * cnt_w + xx] > 0)) continue;
            const float4 *rq4 = (const float4 *)(resp + ((int64_t)yy * CAELO_NET_W + xx) * 8);
            const float4 qa = rq4[0], qb = rq4[1];
            float d;
            d = __fsub_rn(qa.x, pa.x); const float s0 = __fmul_rn(d, d);
            d = __fsub_rn(qa.y, pa.y); const float s1 = __fmul_rn(d, d);
            d = __fsub_rn(qa.z, pa.z); const float s2 = __fmul_rn(d, d);
            d = __fsub_rn(qa.w, pa.w); const float s3 = __fmul_rn(d, d);
            d = __fsub_rn(qb.x, pb.x); const float s4 = __fmul_rn(d, d);
            d = __fsub_rn(qb.y, pb.y); const float s5 = __fmul_rn(d, d);
            d = __fsub_rn(qb.z, pb.z); const float s6 = __fmul_rn(d, d);
            d = __fsub_rn(qb.w, pb.w); const float s7 = __fmul_rn(d, d);
            const float t = __fadd_rn(__fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3)),
                                      __fadd_rn(__fadd_rn(s4, s5), __fadd_rn(s6, s7)));
            const float nd = sqrtf(t);
            if (!have || nd < best) { best = nd; have = true; }
            ++cnt;
        }
    }
    if (cnt < 5) return;                   // :186
    if (!((double)best > 0.2)) return;     // :126,:199
    const float *px = ring + ((int64_t)y * ring_w + x) * ring_c;
    float d2 = __fmul_rn(px[0], px[0]);
    for (int c = 1; c < dist_c; ++c) d2 = __fadd_rn(d2, __fmul_rn(px[c], px[c]));  // :197
    if (!(sqrtf(d2) >= 10.0f)) return;                                        // :198 VisibleBottom
    const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(y * CAELO_NET_W + x);
    const int pos = atomicAdd(cand_count, 1);
    cand[pos] = key;
    atomicAdd(&hist[__float_as_uint(best) >> 16], 1u);  // 16-bit score histogram for the top-k cut
}

// ------------------------------------------------------------------------------------------------
// K4: stable top-(1025) select.  Keys are unique, so "ascending by (score, flat index)" (the stable
// argsort of :194) is plain ascending key order.  One 1024-thread workgroup:
//   1. suffix-scan the 65536-bin histogram of the score's top 16 bits (built by k_kp_score) to find
//      the bin holding the 1025th largest key;
//   2. gather every key from that bin upwards (1025 + a few) into LDS;  bitonic sort;
//   3. emit sorted[-1025:-1] (:216,:218).
// If more than 2048 keys share the cut bin and above (pathological ties) an 8-bit MSD radix select
// over the full keys finds the exact threshold instead.
// ------------------------------------------------------------------------------------------------
#define SEL_THREADS 1024
#define SEL_N 2048

__device__ unsigned long long radix_select_threshold(const unsigned long long *cand, int M, int keep, unsigned int *hist,
                                                     unsigned long long *s_prefix, int *s_want) {
    const int tid = threadIdx.x;
    if (tid == 0) { *s_prefix = 0ull; *s_want = keep; }
    __syncthreads();
    for (int byte = 7; byte >= 0; --byte) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = *s_prefix;
        const int hi_shift = (byte + 1) * 8;
        for (int i = tid; i < M; i += SEL_THREADS) {
            const unsigned long long k = cand[i];
            const bool match = (byte == 7) ? true : ((k >> hi_shift) == (prefix >> hi_shift));
            if (match) atomicAdd(&hist[(unsigned)(k >> (byte * 8)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int want = *s_want;
            int b = 255;
            for (; b > 0; --b) {
                const int cnt = (int)hist[b];
                if (cnt >= want) break;
                want -= cnt;
            }
            *s_want = want;
            *s_prefix = prefix | ((unsigned long long)b << (byte * 8));
        }
        __syncthreads();
    }
    return *s_prefix;
}

__global__ void __launch_bounds__(SEL_THREADS) k_kp_select(const unsigned long long *__restrict__ cand,
                                                           const uint32_t *__restrict__ ghist, const int32_t *cand_count,
                                                           const float *__restrict__ ring, int ring_w, int ring_c,
                                                           int64_t *__restrict__ key_pixels, float *__restrict__ key_pts,
                                                           int kp_ld, float *__restrict__ valid, int valid_ld,
                                                           int32_t *n_key, int32_t *status) {
    __shared__ __attribute__((aligned(16))) unsigned long long sel[SEL_N];
    __shared__ unsigned int hist[256];
    __shared__ unsigned int part[SEL_THREADS];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_want, s_nsel, s_cutbin;
    const int tid = threadIdx.x;
    const int M = *cand_count;
    const int keep = M < 1025 ? M : 1025;
    unsigned long long thresh = 0ull;
    if (M > 1025) {
        // ---- bin of the keep-th largest key: suffix sums over 64 bins per thread
        unsigned int loc = 0;
        for (int b = 0; b < 64; ++b) loc += ghist[tid * 64 + b];
        part[tid] = loc;
        __syncthreads();
        // inclusive suffix scan (Hillis-Steele over 1024 entries)
        for (int off = 1; off < SEL_THREADS; off <<= 1) {
            const unsigned int add = (tid + off < SEL_THREADS) ? part[tid + off] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        const unsigned int above = (tid + 1 < SEL_THREADS) ? part[tid + 1] : 0u;  // keys in higher thread ranges
        if (above < (unsigned)keep && part[tid] >= (unsigned)keep) {
            unsigned int cum = above;
            int b = 63;
            for (; b > 0; --b) {
                cum += ghist[tid * 64 + b];
                if (cum >= (unsigned)keep) break;
            }
            if (b == 0) cum += ghist[tid * 64];
            s_cutbin = tid * 64 + b;
            s_nsel = (int)cum;  // keys with top16 >= cut bin
        }
        __syncthreads();
        if (s_nsel <= SEL_N) {
            thresh = (unsigned long long)s_cutbin << 48;
        } else {
            thresh = radix_select_threshold(cand, M, keep, hist, &s_prefix, &s_want);
        }
        __syncthreads();
    }
    if (tid == 0) s_nsel = 0;
    for (int i = tid; i < SEL_N; i += SEL_THREADS) sel[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < M; i += SEL_THREADS) {
        const unsigned long long k = cand[i];
        if (k >= thresh) {
            const int p = atomicAdd(&s_nsel, 1);
            if (p < SEL_N) sel[p] = k;
        }
    }
    __syncthreads();
    const int nsel = s_nsel < SEL_N ? s_nsel : SEL_N;  // >= keep real keys
    // rank sort: keys are unique, so rank = number of smaller keys; one pass over LDS (broadcast reads),
    // no barriers inside (a 2048-element bitonic network needs 66 of them)
    __shared__ unsigned long long sorted[SEL_N];
    for (int i = tid; i < nsel; i += SEL_THREADS) {
        const unsigned long long mine = sel[i];
        int rank = 0;
        const ulonglong2 *s2 = (const ulonglong2 *)sel;  // entries >= nsel hold ~0 (never smaller)
#pragma unroll 8
        for (int j = 0; j < (nsel + 1) / 2; ++j) {
            const ulonglong2 v = s2[j];
            rank += (v.x < mine ? 1 : 0) + (v.y < mine ? 1 : 0);
        }
        sorted[rank] = mine;
    }
    __syncthreads();
    const int K = keep > 0 ? keep - 1 : 0;  // drop the single best (:216,:218)
    const int first = nsel - keep;           // the keep largest real keys are sel[first .. nsel)
    for (int i = tid; i < CAELO_MAX_KEYPTS; i += SEL_THREADS) {
        if (i < K) {
            const unsigned idx = (unsigned)(sorted[first + i] & 0xFFFFFFFFull);
            const int y = idx / CAELO_NET_W, x = idx % CAELO_NET_W;
            key_pixels[2 * i] = y;
            key_pixels[2 * i + 1] = x;
            const float *px = ring + ((int64_t)y * ring_w + x) * ring_c;
            key_pts[(size_t)kp_ld * i] = px[0];
            key_pts[(size_t)kp_ld * i + 1] = px[1];
            key_pts[(size_t)kp_ld * i + 2] = px[2];
        }
        if (valid) valid[(size_t)valid_ld * i] = i < K ? 1.0f : 0.0f;
    }
    if (tid == 0) {
        *n_key = K;
        if (K <= 50) atomicOr(status, CAELO_ST_FEW_KEYPTS);  // :286
    }
}

int ring_keypoints_launch(const float *ring, int ring_w, int ring_c, int dist_c, const int32_t *counter, int cnt_w,
                          const float *resp, unsigned long long *cand, uint32_t *hist, int32_t *cand_count,
                          int64_t *key_pixels, float *key_pts, int kp_ld, float *valid, int valid_ld, int32_t *n_key,
                          int32_t *status, hipStream_t s) {
    dim3 grid((CAELO_NET_W + 255) / 256, 48);
    k_kp_score<<<grid, 256, 0, s>>>(ring, ring_w, ring_c, dist_c, counter, cnt_w, resp, cand, hist, cand_count);
    CAELO_LAUNCH_CHECK();
    k_kp_select<<<1, SEL_THREADS, 0, s>>>(cand, hist, cand_count, ring, ring_w, ring_c, key_pixels, key_pts, kp_ld, valid,
                                          valid_ld, n_key, status);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int64_t caelo_keypoints_ws_bytes(void) {
    return (int64_t)CAELO_NET_H * CAELO_NET_W * 8 + (int64_t)CAELO_KP_HIST_BINS * 4 + 64;
}

CAELO_API int caelo_keypoints(caelo_ctx *c, const float *ring, int ring_w, int ring_c, const int32_t *counter,
                              int cnt_w, const float *resp, void *ws, int64_t *key_pixels, float *key_pts,
                              int32_t *n_key, int32_t *status, void *stream) {
    CAELO_REQUIRE(c && ring && counter && resp && ws && key_pixels && key_pts && n_key && status, "null argument");
    CAELO_REQUIRE(ring_w >= CAELO_NET_W && cnt_w >= CAELO_NET_W && ring_c >= 3 && ring_c <= 5, "bad ring shape");
    hipStream_t s = caelo_stream(stream);
    unsigned long long *cand = (unsigned long long *)ws;
    uint32_t *hist = (uint32_t *)(cand + CAELO_NET_H * CAELO_NET_W);
    int32_t *cand_count = (int32_t *)(hist + CAELO_KP_HIST_BINS);
    caelo_clear_list cl;
    cl.n = 0;
    cl.item[cl.n++] = {hist, (size_t)CAELO_KP_HIST_BINS * 4 + 64, 0u};
    int rc = caelo_clear_many(cl, s);
    if (rc) return rc;
    return ring_keypoints_launch(ring, ring_w, ring_c, ring_c, counter, cnt_w, resp, cand, hist, cand_count, key_pixels,
                                 key_pts, 3, nullptr, 0, n_key, status, s);
}
