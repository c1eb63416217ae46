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
#ifndef CAELO_BUILD_WORD  // csrc/Makefile always defines it; 0 = "not built by the Makefile", which caelo/_ffi.py refuses
#define CAELO_BUILD_WORD 0
#endif
CAELO_API int caelo_build_flags(void) { return CAELO_BUILD_WORD; }

CAELO_API int caelo_create(caelo_ctx **out, int device) {
    CAELO_REQUIRE(out != nullptr, "null ctx pointer");
    int ndev = 0;
    CAELO_HIP(hipGetDeviceCount(&ndev));
    CAELO_REQUIRE(device >= 0 && device < ndev, "no such HIP device (libcaelo needs a GPU; there is no CPU fallback)");
    CAELO_HIP(hipSetDevice(device));
    caelo_ctx *c = new caelo_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    CAELO_HIP(hipMalloc((void **)&c->faults, sizeof(int32_t)));
    CAELO_HIP(hipMemset(c->faults, 0, sizeof(int32_t)));
    *out = c;
    return CAELO_OK;
}

CAELO_API void caelo_destroy(caelo_ctx *c) {
    if (!c) return;
    float *ptrs[] = {c->resp_w, c->enc_c0, c->enc_w1, c->enc_b1, c->enc_w2, c->enc_b2, c->enc_w3,
                     c->enc_b3, c->enc_bd1, c->enc_wd2, c->enc_bd2, c->enc32_bd1};
    for (float *p : ptrs)
        if (p) (void)hipFree(p);
    for (void *p : {c->enc_w1f, c->enc_w2x, c->enc_w3x, c->enc_wd1x, c->enc32_wd1x, (void *)c->faults})
        if (p) (void)hipFree(p);
    delete c;
}

CAELO_API int caelo_lane_faults(caelo_ctx *c, int64_t *count_host) {
    CAELO_REQUIRE(c && count_host, "null argument");
    int32_t v = 0;
    CAELO_HIP(hipDeviceSynchronize());
    CAELO_HIP(hipMemcpy(&v, c->faults, sizeof(v), hipMemcpyDeviceToHost));
    *count_host = v;
    return CAELO_OK;
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

// every kernel below: blockIdx.z = frame of the set (caelo_internal.h)
__global__ void __launch_bounds__(256) k_project_points(const caelo_frame_set fs, ProjConst k) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F.n) return;
    int32_t *const winner = F.winner, *const counter = F.counter, *const status = F.status;
    const float4 p = ((const float4 *)F.pc)[i];
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

__global__ void __launch_bounds__(256) k_ring_fill(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const float4 *__restrict__ pc = (const float4 *)F.pc;
    const int32_t *__restrict__ winner = F.winner;
    float *__restrict__ ring = F.ring;
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

static int64_t set_max_points(const caelo_frame_set &fs) {
    int64_t n = 0;
    for (int i = 0; i < fs.n; ++i) n = fs.f[i].n > n ? fs.f[i].n : n;
    return n;
}

int ring_project_launch(const float *pc, int64_t n, float *ring, int32_t *counter, int32_t *winner_ws, int32_t *status,
                        hipStream_t s) {
    caelo_frame_set fs = {};
    fs.n = 1;
    fs.f[0].pc = pc; fs.f[0].n = n; fs.f[0].ring = ring; fs.f[0].counter = counter; fs.f[0].winner = winner_ws; fs.f[0].status = status;
    return ring_project_set(fs, s);
}

int ring_project_set(const caelo_frame_set &fs, hipStream_t s) {
    const int npix = CAELO_RING_H * CAELO_RING_W;
    const int64_t n = set_max_points(fs);
    ProjConst k;
    k.pi = 3.14159265358979323846;
    const double d2r = k.pi / 180.0;                        // SphericalRing.py:28
    k.az_res = 0.20 * d2r;                                  // :35,:48
    const double vdown = -24.8 * d2r, vup = 2.0 * d2r;      // :49-50
    k.v_res = (vup - vdown) / (64 - 1);                     // :51
    k.v_off = -vdown / k.v_res;                             // :52
    k_project_points<<<dim3((unsigned)((n + 255) / 256), 1, fs.n), 256, 0, s>>>(fs, k);
    CAELO_LAUNCH_CHECK();
    k_ring_fill<<<dim3((npix + 255) / 256, 1, fs.n), 256, 0, s>>>(fs);
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
__global__ void __launch_bounds__(256) k_respond(const caelo_frame_set fs, int in_w, int in_c, const float *__restrict__ wts) {
    const float *__restrict__ in = fs.f[blockIdx.z].ring;
    float *__restrict__ resp = fs.f[blockIdx.z].resp;
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
    caelo_frame_set fs = {};
    fs.n = 1;
    fs.f[0].ring = const_cast<float *>(in); fs.f[0].resp = resp;
    return ring_respond_set(c, fs, in_w, in_c, s);
}

int ring_respond_set(caelo_ctx *c, const caelo_frame_set &fs, int in_w, int in_c, hipStream_t s) {
    dim3 grid((CAELO_NET_W + 255) / 256, CAELO_NET_H, fs.n);
    k_respond<<<grid, 256, 0, s>>>(fs, in_w, in_c, c->resp_w);
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
// compact score histogram: bin = clamp((score bits >> 16) - 0x3E00, 0, 2047).  Scores in (0.2, 6.5e4)
// spread over ~2000 bins of 2^-7 relative width; anything larger shares the last bin.
#define KP_BIN_BASE 0x3E00u
__device__ __host__ inline unsigned int kp_bin(unsigned int score_bits) {
    const unsigned int t = score_bits >> 16;
    return t <= KP_BIN_BASE ? 0u : (t - KP_BIN_BASE > (unsigned)(CAELO_KP_HIST_BINS - 1) ? (unsigned)(CAELO_KP_HIST_BINS - 1) : t - KP_BIN_BASE);
}

// Tile = 4 rows x 64 columns of pixels per workgroup; the 8 x 68 halo of response vectors and occupancy
// flags is staged in LDS with coalesced loads first (the per-neighbour "occupied?" test used to put 24
// dependent global round trips in front of every pixel).
#define KS_ROWS 4
#define KS_COLS 64
#define KS_HR (KS_ROWS + 4)
#define KS_HC (KS_COLS + 4)

__global__ void __launch_bounds__(256) k_kp_score(const caelo_frame_set fs, int ring_w, int ring_c, int cnt_w) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const float *__restrict__ ring = F.ring;
    const int dist_c = F.dist_c;
    const int32_t *__restrict__ counter = F.counter;
    const float *__restrict__ resp = F.resp;
    unsigned long long *__restrict__ cand = F.cand;
    int32_t *cand_count = F.cand_count;
    __shared__ float4 sR[KS_HR * KS_HC * 2];
    __shared__ unsigned char sOcc[KS_HR * KS_HC];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * KS_COLS, y0 = 8 + blockIdx.y * KS_ROWS;  // rows 8..55 only
    for (int i = tid; i < KS_HR * KS_HC; i += 256) {
        const int hy = i / KS_HC, hx = i % KS_HC;
        const int yy = y0 - 2 + hy, xx = x0 - 2 + hx;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        unsigned char occ = 0;
        if (xx >= 0 && xx < CAELO_NET_W && yy >= 0 && yy < CAELO_NET_H) {
            const float4 *rq4 = (const float4 *)(resp + ((int64_t)yy * CAELO_NET_W + xx) * 8);
            a = rq4[0];
            b = rq4[1];
            occ = counter[yy * cnt_w + xx] > 0;
        }
        sR[2 * i] = a;
        sR[2 * i + 1] = b;
        sOcc[i] = occ;
    }
    __syncthreads();
    const int lx = tid & (KS_COLS - 1), ly = tid / KS_COLS;
    const int x = x0 + lx, y = y0 + ly;
    float best = 0.0f;
    bool is_cand = false;
    const int c = (ly + 2) * KS_HC + lx + 2;
    if (x >= 8 && x < CAELO_NET_W - 8 && !(x >= 56 && x < 64) && sOcc[c]) {  // :163-167 (incl. the 56..63 quirk), :210-213
        const float4 pa = sR[2 * c], pb = sR[2 * c + 1];
        int cnt = 0;
        bool have = false;
#pragma unroll
        for (int oy = -2; oy <= 2; ++oy) {
#pragma unroll
            for (int ox = -2; ox <= 2; ++ox) {
                if (oy == 0 && ox == 0) continue;
                const int q = c + oy * KS_HC + ox;
                if (!sOcc[q]) continue;
                const float4 qa = sR[2 * q], qb = sR[2 * q + 1];
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
        if (cnt >= 5 && (double)best > 0.2) {  // :186, :126,:199
            const float *px = ring + ((int64_t)y * ring_w + x) * ring_c;
            float d2 = __fmul_rn(px[0], px[0]);
            for (int ch = 1; ch < dist_c; ++ch) d2 = __fadd_rn(d2, __fmul_rn(px[ch], px[ch]));  // :197
            is_cand = sqrtf(d2) >= 10.0f;                                                       // :198 VisibleBottom
        }
    }
    // one returning global atomic per WORKGROUP reserves the slots of all its candidates
    __shared__ int s_tmp[2];
    const int pos = caelo_block_reserve(cand_count, is_cand, s_tmp);
    if (is_cand) {
        cand[pos] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(y * CAELO_NET_W + x);
    }
}

// ------------------------------------------------------------------------------------------------
// K4: stable top-(1025) select.  Keys are unique, so "ascending by (score, flat index)" (the stable
// argsort of :194) is plain ascending key order.  One 1024-thread workgroup:
//   1. histogram the candidates' scores (2048 bins of the top 16 score bits, see kp_bin) in LDS and suffix-scan
//      it to find the bin holding the 1025th largest key;
//   2. gather every key from that bin upwards (1025 + a few) into LDS;  bitonic sort;
//   3. emit sorted[-1025:-1] (:216,:218).
// If more than 2048 keys share the cut bin and above (pathological ties) an 8-bit MSD radix select
// over the full keys finds the exact threshold instead.
// ------------------------------------------------------------------------------------------------
#define SEL_THREADS 1024
#define SEL_N 2048

// phase timestamps of the last k_kp_select launch (wall_clock64, 100 MHz), read by caelo_debug_read
__device__ unsigned long long g_sel_stamp[16];
#define SEL_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.z == 0) g_sel_stamp[i] = wall_clock64(); } while (0)

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

__global__ void __launch_bounds__(SEL_THREADS) k_kp_select(const caelo_frame_set fs, int ring_w, int ring_c) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const unsigned long long *__restrict__ cand = F.cand;
    const int32_t *cand_count = F.cand_count;
    const float *__restrict__ ring = F.ring;
    int64_t *__restrict__ key_pixels = F.key_pixels;
    float *__restrict__ key_pts = F.key_pts;
    const int kp_ld = F.kp_ld, valid_ld = F.valid_ld;
    float *__restrict__ valid = F.valid;
    int32_t *n_key = F.n_key, *status = F.status;
    __shared__ __attribute__((aligned(16))) unsigned long long sel[SEL_N];
    __shared__ unsigned int hist[256];
    __shared__ unsigned int part[SEL_THREADS];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_want, s_nsel, s_cutbin;
    const int tid = threadIdx.x;
    SEL_STAMP(0);
    const int M = *cand_count;
    const int keep = M < 1025 ? M : 1025;
    unsigned long long thresh = 0ull;
    if (M > 1025) {
        // ---- score histogram in LDS (2048 bins; a global one funnels 32k device atomics into one
        //      memory channel: 15 us), then the bin of the keep-th largest key: thread t owns bins 2t, 2t+1
        unsigned int *lh = (unsigned int *)sel;  // 8 KB of the 16 KB key buffer, free until the gather
        lh[2 * tid] = 0u;
        lh[2 * tid + 1] = 0u;
        __syncthreads();
        for (int i0 = 0; i0 < M; i0 += SEL_THREADS * 8) {
            unsigned long long kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * SEL_THREADS + tid;
                kk[u] = i < M ? cand[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (kk[u] != 0ull) atomicAdd(&lh[kp_bin((unsigned int)(kk[u] >> 32))], 1u);
        }
        __syncthreads();
        const uint2 hb = make_uint2(lh[2 * tid], lh[2 * tid + 1]);
        part[tid] = hb.x + hb.y;
        __syncthreads();
        SEL_STAMP(1);
        for (int off = 1; off < SEL_THREADS; off <<= 1) {  // inclusive suffix scan (Hillis-Steele)
            const unsigned int add = (tid + off < SEL_THREADS) ? part[tid + off] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        SEL_STAMP(2);
        const unsigned int above = (tid + 1 < SEL_THREADS) ? part[tid + 1] : 0u;  // keys in higher bins
        if (above < (unsigned)keep && part[tid] >= (unsigned)keep) {
            // the cut is bin 2t+1 if that alone reaches `keep`, else bin 2t
            if (above + hb.y >= (unsigned)keep) { s_cutbin = 2 * tid + 1; s_nsel = (int)(above + hb.y); }
            else { s_cutbin = 2 * tid; s_nsel = (int)part[tid]; }
        }
        __syncthreads();
        if (s_nsel <= SEL_N) {
            thresh = s_cutbin == 0 ? 0ull : ((unsigned long long)(KP_BIN_BASE + s_cutbin) << 48);
        } else {
            thresh = radix_select_threshold(cand, M, keep, hist, &s_prefix, &s_want);
        }
        __syncthreads();
    }
    SEL_STAMP(3);
    if (tid == 0) s_nsel = 0;
    for (int i = tid; i < SEL_N; i += SEL_THREADS) sel[i] = ~0ull;
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += SEL_THREADS * 8) {
        unsigned long long kk[8];  // 8 independent loads in flight, then the (rare) LDS appends
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * SEL_THREADS + tid;
            kk[u] = i < M ? cand[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (kk[u] >= thresh && kk[u] != 0ull) {
                const int p = atomicAdd(&s_nsel, 1);
                if (p < SEL_N) sel[p] = kk[u];
            }
        }
    }
    __syncthreads();
    SEL_STAMP(4);
    const int nsel = s_nsel < SEL_N ? s_nsel : SEL_N;  // >= keep real keys, padding (~0) sorts to the end
    // bitonic sort ascending, 2048 elements, 1024 threads.  For strides j <= 64 a wavefront's 64 threads
    // only touch their own 128 consecutive elements: no workgroup barrier needed (56 of the 66 steps).
    for (int k = 2; k <= SEL_N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));
            const int p = i | j;
            const unsigned long long a = sel[i], b = sel[p];
            const bool up = ((i & k) == 0);
            if ((a > b) == up) { sel[i] = b; sel[p] = a; }
            if (j > 64 || (j == 1 && k >= 128)) __syncthreads();  // next step crosses wavefronts
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        }
    }
    __syncthreads();
    SEL_STAMP(5);
    const int K = keep > 0 ? keep - 1 : 0;  // drop the single best (:216,:218)
    const int first = nsel - keep;           // the keep largest real keys are sel[first .. nsel)
    for (int i = tid; i < CAELO_MAX_KEYPTS; i += SEL_THREADS) {
        if (i < K) {
            const unsigned idx = (unsigned)(sel[first + i] & 0xFFFFFFFFull);
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
    SEL_STAMP(6);
    if (tid == 0) {
        *n_key = K;
        if (K <= 50) atomicOr(status, CAELO_ST_FEW_KEYPTS);  // :286
    }
}

int ring_keypoints_launch(const float *ring, int ring_w, int ring_c, int dist_c, const int32_t *counter, int cnt_w,
                          const float *resp, unsigned long long *cand, int32_t *cand_count,
                          int64_t *key_pixels, float *key_pts, int kp_ld, float *valid, int valid_ld, int32_t *n_key,
                          int32_t *status, hipStream_t s) {
    caelo_frame_set fs = {};
    fs.n = 1;
    caelo_frame_dev &d = fs.f[0];
    d.ring = const_cast<float *>(ring); d.dist_c = dist_c; d.counter = const_cast<int32_t *>(counter); d.resp = const_cast<float *>(resp);
    d.cand = cand; d.cand_count = cand_count; d.key_pixels = key_pixels; d.key_pts = key_pts; d.kp_ld = kp_ld;
    d.valid = valid; d.valid_ld = valid_ld; d.n_key = n_key; d.status = status;
    return ring_keypoints_set(fs, ring_w, ring_c, cnt_w, s);
}

int ring_keypoints_set(const caelo_frame_set &fs, int ring_w, int ring_c, int cnt_w, hipStream_t s) {
    dim3 grid(CAELO_NET_W / KS_COLS, 48 / KS_ROWS, fs.n);
    k_kp_score<<<grid, 256, 0, s>>>(fs, ring_w, ring_c, cnt_w);
    CAELO_LAUNCH_CHECK();
    k_kp_select<<<dim3(1, 1, fs.n), SEL_THREADS, 0, s>>>(fs, ring_w, ring_c);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// debug aid: copies the 16 phase timestamps of the last keypoint selection (100 MHz ticks) to the host
int enc_debug_copy(unsigned long long *out_host);
int patch_debug_copy(unsigned long long *out_host);
CAELO_API int caelo_debug_read(unsigned long long *out_host) {  // out_host[40]: keypoint select | encoder stage 1 | patches
    CAELO_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_sel_stamp), sizeof(unsigned long long) * 16));
    int rc = enc_debug_copy(out_host + 16);
    return rc ? rc : patch_debug_copy(out_host + 32);
}

CAELO_API int64_t caelo_keypoints_ws_bytes(void) {
    return (int64_t)CAELO_NET_H * CAELO_NET_W * 8 + 64;
}

CAELO_API int caelo_keypoints(caelo_ctx *c, const float *ring, int ring_w, int ring_c, const int32_t *counter,
                              int cnt_w, const float *resp, void *ws, int64_t *key_pixels, float *key_pts,
                              int32_t *n_key, int32_t *status, void *stream) {
    CAELO_REQUIRE(c && ring && counter && resp && ws && key_pixels && key_pts && n_key && status, "null argument");
    CAELO_REQUIRE(ring_w >= CAELO_NET_W && cnt_w >= CAELO_NET_W && ring_c >= 3 && ring_c <= 5, "bad ring shape");
    hipStream_t s = caelo_stream(stream);
    unsigned long long *cand = (unsigned long long *)ws;
    int32_t *cand_count = (int32_t *)(cand + CAELO_NET_H * CAELO_NET_W);
    caelo_clear_list cl;
    cl.n = 0;
    cl.item[cl.n++] = {cand_count, 64, 0u};
    int rc = caelo_clear_many(cl, s);
    if (rc) return rc;
    return ring_keypoints_launch(ring, ring_w, ring_c, ring_c, counter, cnt_w, resp, cand, cand_count, key_pixels,
                                 key_pts, 3, nullptr, 0, n_key, status, s);
}
