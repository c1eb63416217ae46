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
    for (void *p : {c->enc_w1f, c->enc_w2x, c->enc_w3x, c->enc_wd1x, c->enc32_wd1x, c->enc_wd2q, (void *)c->faults})
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

// k_respond_mfma's per-lane operand fragments live behind the raw weights in ctx->resp_w
#define RESP_FRAGS 34                 // a1[2][7], c1[2][4], a2[8], c2[4]
#define RESP_FRAG_OFF 1280            // floats into ctx->resp_w (behind the 1160 raw weights)
static void respond_fragments(const float *w1, const float *b1, const float *w2, const float *b2, float *out);

CAELO_API int caelo_set_respond_weights(caelo_ctx *c, const float *w1, const float *b1, const float *w2,
                                        const float *b2) {
    CAELO_REQUIRE(c && w1 && b1 && w2 && b2, "null argument");
    const size_t n = RESP_FRAG_OFF + RESP_FRAGS * 64;  // the raw weights (1160 floats), then k_respond_mfma's per-lane fragments
    if (!c->resp_w) CAELO_HIP(hipMalloc(&c->resp_w, n * sizeof(float)));
    float host[RESP_FRAG_OFF + RESP_FRAGS * 64] = {0};
    memcpy(host, w1, 27 * 32 * 4);
    memcpy(host + 864, b1, 32 * 4);
    memcpy(host + 896, w2, 256 * 4);
    memcpy(host + 1152, b2, 8 * 4);
    respond_fragments(w1, b1, w2, b2, host + RESP_FRAG_OFF);
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
    const double colf = (k.pi - atan2((double)p.y, (double)p.x)) / k.az_res;         // :86
    const float q = __fdiv_rn(p.z, r);                                                // :87 f32 quotient
    const double rowf = asin((double)q) / k.v_res + k.v_off;                          // :88
    if (colf != colf || rowf != rowf) {   // int(nan): ValueError in the reference (a NaN coordinate, or inf / inf in z / r)
        atomicOr(status, CAELO_ST_NONFINITE);
        return;
    }
    const int col = (int)colf;
    const int row = CAELO_RING_H - (int)rowf;
    if (row < 0 || row >= CAELO_RING_H) return;                                       // :89
    if (col < 0 || col >= CAELO_RING_W) {
        atomicOr(status, CAELO_ST_COL_OOB);
        return;
    }
    const int pix = row * CAELO_RING_W + col;
    atomicMax(&winner[pix], (int32_t)i);
    if (counter) atomicAdd(&counter[pix], 1);                                         // :93 (the fused path keeps no counts:
                                                                                      //  a pixel is occupied iff it has a winner)
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
// K2: response layer, fused conv3x3(3->32)+relu+conv1x1(32->8)+relu.
// Summation order is the canonical one documented in oracle/caelo_oracle.c (orc_respond) so the
// response image -- and therefore the keypoint indices -- are bit-identical to the oracle's.
// (Rounds 1-2 ran it one thread per pixel on the VALU, 37 us per 8 frames; the matrix-pipe form below is bit-identical.)
// ------------------------------------------------------------------------------------------------

// The same layer on the f32 matrix pipe.  v_mfma_f32_16x16x4_f32 adds its four products to the accumulator as a chain of fused
// multiply-adds in ascending k (tools/micro/mfma_f32_order.hip: 0 of 1 M results differ from fmaf(a3,b3,fmaf(a2,b2,fmaf(a1,b1,
// fmaf(a0,b0,c)))), wide exponent ranges included) -- so a GEMM whose k runs in the oracle's canonical order is bit-identical
// to k_respond.  What it buys is modest: the f32 matrix instruction and the VALU share a datapath (same micro-benchmark: eight
// MFMAs + 64 v_fma_f32 per step take the SUM of their times with four waves per SIMD), so the layer's 1.26 M MFMAs (18.4 us
// per 8 frames at 32 cycles each) and everything else the waves execute add up: 37 us (k_respond) -> 31 us.
// Transposed GEMMs, D[channel][pixel], so that layer 1's accumulators ARE layer 2's B operands (no shuffle, no LDS):
//   layer 1  A = W1^T tile [16 channels][4 k], B = taps [4 k][16 pixels], k = (ky, kx, ci) ascending + one zero pad, C = b1;
//            accumulator row 4g + r of tile t is made channel 16t + 4r + g (a row permutation of W1^T), so that
//   layer 2  step s = 4t + r takes register r of tile t as B: lane group g then holds channel 4s + g -- k ascending again;
//            A = W2^T [8 outputs (+8 idle rows)][4 k], C = b2.
// A tap outside the image enters as 0: fmaf(0, w, acc) = acc, the skipped tap of k_respond.
// One WAVE per workgroup: it walks the 32-pixel blocks blockIdx.x, + gridDim.x, ... of its row, staging the three input rows
// of a block (+ one pixel either side, channels 0..2, zero outside the image) in its own LDS with coalesced loads -- gathering
// the taps from memory is 28 scattered loads per lane, 42 us per 8 frames, slower than the VALU kernel -- and the next block's
// rows are in flight while this block's MFMAs run.  No workgroup barrier: the waves of a SIMD drift apart, and one wave's loads /
// address arithmetic / relu run under the other's MFMAs (four waves behind one barrier: matrix pipe 54 % busy).
// The per-lane operand fragments come ready-made from caelo_set_respond_weights (RESP_FRAGS floats per lane).
#define RESP_PT 2                      // 16-pixel tiles per block
#define RESP_BPX (16 * RESP_PT)        // pixels per block
#define RESP_ROW ((RESP_BPX + 2) * 3)  // floats of a staged row: the block + one pixel either side, 3 channels
#define RESP_STAGE (3 * RESP_ROW)      // floats of a staged block
typedef float resp_f4 __attribute__((ext_vector_type(4)));

// host: the fragment image, [RESP_FRAGS][64 lanes]
static void respond_fragments(const float *w1, const float *b1, const float *w2, const float *b2, float *out) {
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, n = lane & 15;
        int f = 0;
        for (int t = 0; t < 2; ++t)
            for (int s = 0; s < 7; ++s) {
                const int k = 4 * s + g;
                out[(f++) * 64 + lane] = k < 27 ? w1[k * 32 + 16 * t + 4 * (n & 3) + (n >> 2)] : 0.0f;
            }
        for (int t = 0; t < 2; ++t)
            for (int r = 0; r < 4; ++r) out[(f++) * 64 + lane] = b1[16 * t + 4 * r + g];
        for (int s = 0; s < 8; ++s) out[(f++) * 64 + lane] = n < 8 ? w2[(4 * s + g) * 8 + n] : 0.0f;
        for (int r = 0; r < 4; ++r) out[(f++) * 64 + lane] = g < 2 ? b2[4 * g + r] : 0.0f;
    }
}

__global__ void __launch_bounds__(64) k_respond_mfma(const caelo_frame_set fs, int in_w, int in_c, const float *__restrict__ wts, int row0) {
    const float *__restrict__ in = fs.f[blockIdx.z].ring;
    float *__restrict__ resp = fs.f[blockIdx.z].resp;
    const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
    const int y = blockIdx.y + row0;
    constexpr int NBLK = CAELO_NET_W / RESP_BPX, NLD = (RESP_STAGE + 63) / 64;
    __shared__ float s_in[2][((RESP_STAGE + 63) / 64) * 64];  // (padded: every lane stores NLD elements)
    // staging: element i = lane + 64 q of [row 3][pixel RESP_BPX + 2][channel 3]; everything but the block's column is loop-invariant
    int goff[NLD];
    unsigned int edge = 0u;  // bit q: the element is the left halo pixel; bit 16 + q: the right one
    float v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int i = lane + 64 * q;
        const int row = i / RESP_ROW, j = i - row * RESP_ROW, px = j / 3, ci = j - px * 3;
        const int yy = y + row - 1;
        const bool ok = i < RESP_STAGE && (unsigned)yy < (unsigned)CAELO_NET_H;
        goff[q] = ok ? (yy * in_w + px - 1) * in_c + ci : -1;
        edge |= (px == 0 ? 1u : 0u) << q | (px == RESP_BPX + 1 ? 1u : 0u) << (16 + q);
    }
    // branch-free through a buffer descriptor: an element outside the image gets an offset past the end of the buffer and the
    // range check returns 0.  (Written as `ok ? in[off] : 0`, hipcc sinks every load under its condition: a branch and an
    // s_waitcnt vmcnt(0) per element -- ten serial memory round trips per block.)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, CAELO_NET_H * in_w * in_c * 4, 0x00020000);
#define RESP_LOAD(BLK)                                                                                            \
    _Pragma("unroll") for (int q = 0; q < NLD; ++q) {                                                              \
        const unsigned int hide = ((BLK) == 0 ? edge : 0u) | ((BLK) == NBLK - 1 ? edge >> 16 : 0u);               \
        const bool ok_ = goff[q] >= 0 && !((hide >> q) & 1u);                                                     \
        const int ob_ = ok_ ? (goff[q] + (BLK) * RESP_BPX * in_c) * 4 : 0x7FFFFFF0;                               \
        v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ob_, 0, 0));                  \
    }
#define RESP_STORE(BUF)                                                                                           \
    _Pragma("unroll") for (int q = 0; q < NLD; ++q) s_in[BUF][lane + 64 * q] = v[q];
    int blk = blockIdx.x, buf = 0;
    RESP_LOAD(blk)  // (in flight under the fragment loads below)
    const float *fr = wts + RESP_FRAG_OFF + lane;
    float a1[2][7], a2[8];
    resp_f4 c1[2], c2;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 7; ++s) a1[t][s] = fr[(t * 7 + s) * 64];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) c1[t][r] = fr[(14 + t * 4 + r) * 64];
#pragma unroll
    for (int s = 0; s < 8; ++s) a2[s] = fr[(22 + s) * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) c2[r] = fr[(30 + r) * 64];
    int loff[7];  // this lane's tap of step s: staged row ky, pixel n + kx (relative to the block's column - 1), channel ci
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + g;
        const int kc = k < 27 ? k : 26;  // (the pad's weight is 0: any finite tap does)
        const int ky = kc / 9, kx = (kc / 3) % 3, ci = kc % 3;
        loff[s] = ky * RESP_ROW + (n + kx) * 3 + ci;
    }
    // every fragment has landed before the loop: left to itself hipcc waits for the last of them INSIDE the loop with vmcnt(0),
    // i.e. for the rows just requested for the next block as well -- a memory round trip per block
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int s = 0; s < 7; ++s) __asm__ volatile("" : "+v"(a1[t][s]));
        __asm__ volatile("" : "+v"(c1[t]));
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) __asm__ volatile("" : "+v"(a2[s]));
    __asm__ volatile("" : "+v"(c2));
    RESP_STORE(0)
    for (; blk < NBLK; blk += gridDim.x, buf ^= 1) {
        const bool more = blk + (int)gridDim.x < NBLK;
        if (more) RESP_LOAD(blk + gridDim.x)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the staged block is complete (one wave: program order + this)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float b[RESP_PT][7];
#pragma unroll
        for (int pt = 0; pt < RESP_PT; ++pt)
#pragma unroll
            for (int s = 0; s < 7; ++s) b[pt][s] = s_in[buf][loff[s] + 48 * pt];
        __builtin_amdgcn_sched_barrier(0);  // all taps requested before the first MFMA (hipcc otherwise reads, waits, multiplies)
        // independent accumulator chains (pixel tiles x channel tiles in layer 1, pixel tiles in layer 2) issued round robin
        resp_f4 h[RESP_PT][2], o[RESP_PT];
#pragma unroll
        for (int pt = 0; pt < RESP_PT; ++pt) { h[pt][0] = c1[0]; h[pt][1] = c1[1]; o[pt] = c2; }
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int pt = 0; pt < RESP_PT; ++pt)
#pragma unroll
                for (int t = 0; t < 2; ++t) h[pt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][s], b[pt][s], h[pt][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int pt = 0; pt < RESP_PT; ++pt) {
                    const float hv = h[pt][t][r] > 0.0f ? h[pt][t][r] : 0.0f;
                    o[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * t + r], hv, o[pt], 0, 0, 0);
                }
        if (g < 2) {
#pragma unroll
            for (int pt = 0; pt < RESP_PT; ++pt) {
                const int x = blk * RESP_BPX + 16 * pt + n;
                *(float4 *)(resp + ((int64_t)y * CAELO_NET_W + x) * 8 + 4 * g) =
                    make_float4(o[pt][0] > 0.0f ? o[pt][0] : 0.0f, o[pt][1] > 0.0f ? o[pt][1] : 0.0f, o[pt][2] > 0.0f ? o[pt][2] : 0.0f,
                                o[pt][3] > 0.0f ? o[pt][3] : 0.0f);
            }
        }
        if (more) RESP_STORE(buf ^ 1)  // (this wave read that buffer an iteration ago: program order)
    }
#undef RESP_LOAD
#undef RESP_STORE
}

int ring_respond_launch(caelo_ctx *c, const float *in, int in_w, int in_c, float *resp, hipStream_t s) {
    caelo_frame_set fs = {};
    fs.n = 1;
    fs.f[0].ring = const_cast<float *>(in); fs.f[0].resp = resp;
    return ring_respond_set(c, fs, in_w, in_c, s, 0, CAELO_NET_H);
}

// rows row0 .. row0 + rows - 1 of the response image (the fused path only needs the rows the key point rule reads: 8..55 and two
// either side; the staged entry point computes all 64)
int ring_respond_set(caelo_ctx *c, const caelo_frame_set &fs, int in_w, int in_c, hipStream_t s, int row0, int rows) {
    static_assert(CAELO_NET_W % RESP_BPX == 0, "k_respond_mfma has no partial blocks");
    {
        // Waves per row.  The dispatcher spreads a grid evenly over the 1024 SIMDs (tools/micro/wave_placement.hip), so the grid
        // should come close to a multiple of 1024 waves with equal work each: 8 frames x 64 rows x 8 = 4096 waves of 7 blocks (2048
        // waves of 64-pixel blocks left 109 SIMDs with three waves and 109 with one: the matrix pipe of the former set the time).
        // For the fused path's 52 rows, measured: 4 / 7 / 8 / 14 / 28 waves per row = 40.5 / 32.7 / 30.3 / 28.0 / 29.4 us
        const unsigned gx = fs.n >= 4 ? (rows == CAELO_NET_H ? 8u : 14u) : (fs.n >= 2 ? 28u : 56u);
        k_respond_mfma<<<dim3(gx, rows, fs.n), 64, 0, s>>>(fs, in_w, in_c, c->resp_w, row0);
    }
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
    const int32_t *__restrict__ counter = F.counter, *__restrict__ winner = F.winner;  // counter == null: occupied = has a winner
    const float *__restrict__ resp = F.resp;
    unsigned long long *__restrict__ cand = F.cand;
    int32_t *cand_count = F.cand_count;
    __shared__ float4 sR[KS_HR * KS_HC * 2];
    __shared__ unsigned int sBits[KS_HR][4];   // occupancy of a halo row as bits (68 columns: three words)
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * KS_COLS, y0 = 8 + blockIdx.y * KS_ROWS;  // rows 8..55 only
    if (tid < KS_HR * 4) (&sBits[0][0])[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < KS_HR * KS_HC; i += 256) {
        const int hy = i / KS_HC, hx = i % KS_HC;
        const int yy = y0 - 2 + hy, xx = x0 - 2 + hx;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        bool occ = false;
        if (xx >= 0 && xx < CAELO_NET_W && yy >= 0 && yy < CAELO_NET_H) {
            const float4 *rq4 = (const float4 *)(resp + ((int64_t)yy * CAELO_NET_W + xx) * 8);
            a = rq4[0];
            b = rq4[1];
            occ = counter ? counter[yy * cnt_w + xx] > 0 : winner[yy * cnt_w + xx] >= 0;
        }
        sR[2 * i] = a;
        sR[2 * i + 1] = b;
        if (occ) atomicOr(&sBits[hy][hx >> 5], 1u << (hx & 31));
    }
    __syncthreads();
    const int lx = tid & (KS_COLS - 1), ly = tid / KS_COLS;
    const int x = x0 + lx, y = y0 + ly;
    float best = 0.0f;
    bool is_cand = false;
    const int c = (ly + 2) * KS_HC + lx + 2;
    // the 5 x 5 window's occupancy as 25 bits (row oy at bits 5 (oy + 2) ..): one look at LDS instead of a test in front of every
    // neighbour -- round 3 read a flag byte, branched, and only then fetched the neighbour's vector: 24 dependent round trips per pixel
    unsigned int win = 0u;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const unsigned int *w = sBits[ly + r];
        const int k = lx >> 5, sh = lx & 31;
        const unsigned long long two = (unsigned long long)w[k] | ((unsigned long long)w[k + 1] << 32);
        win |= ((unsigned int)(two >> sh) & 31u) << (5 * r);
    }
    if (x >= 8 && x < CAELO_NET_W - 8 && !(x >= 56 && x < 64) && ((win >> 12) & 1u)) {  // :163-167 (incl. the 56..63 quirk), :210-213
        const float4 pa = sR[2 * c], pb = sR[2 * c + 1];
        const int cnt = __popc(win & ~(1u << 12));
        best = __builtin_inff();
#pragma unroll
        for (int oy = -2; oy <= 2; ++oy) {
            // one row of the window at a time: its five vectors are requested together (ten 16-byte reads in flight), unconditionally;
            // an unoccupied neighbour's sum is computed and dropped
            float4 qa[5], qb[5];
#pragma unroll
            for (int ox = -2; ox <= 2; ++ox) {
                const int q = c + oy * KS_HC + ox;
                qa[ox + 2] = sR[2 * q];
                qb[ox + 2] = sR[2 * q + 1];
            }
#pragma unroll
            for (int ox = -2; ox <= 2; ++ox) {
                if (oy == 0 && ox == 0) continue;
                const float4 va = qa[ox + 2], vb = qb[ox + 2];
                float d;
                d = __fsub_rn(va.x, pa.x); const float s0 = __fmul_rn(d, d);
                d = __fsub_rn(va.y, pa.y); const float s1 = __fmul_rn(d, d);
                d = __fsub_rn(va.z, pa.z); const float s2 = __fmul_rn(d, d);
                d = __fsub_rn(va.w, pa.w); const float s3 = __fmul_rn(d, d);
                d = __fsub_rn(vb.x, pb.x); const float s4 = __fmul_rn(d, d);
                d = __fsub_rn(vb.y, pb.y); const float s5 = __fmul_rn(d, d);
                d = __fsub_rn(vb.z, pb.z); const float s6 = __fmul_rn(d, d);
                d = __fsub_rn(vb.w, pb.w); const float s7 = __fmul_rn(d, d);
                const float t = __fadd_rn(__fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3)),
                                          __fadd_rn(__fadd_rn(s4, s5), __fadd_rn(s6, s7)));
                // the square root is monotone (correctly rounded): min over sqrt(t) = sqrt(min t), taken once below
                const bool o = (win >> (5 * (oy + 2) + ox + 2)) & 1u;
                // (NaN propagates like cp.min over the 25 norms of SphericalRing.py:159,179: a NaN in the response of ANY window pixel,
                // occupied or not -- the reference adds 1e10 to the unoccupied ones, which leaves a NaN a NaN -- makes the score NaN
                // and `score > 0.2` false.  Once NaN, `t < NaN` keeps it.)
                best = ((o && t < best) || t != t) ? t : best;
            }
        }
        best = cnt > 0 ? sqrtf(best) : 0.0f;
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

// ---- the multi-workgroup form (k_kp_hist -> k_kp_gather -> k_kp_emit, KP_WGS workgroups per frame each) ----------------------
// One 1024-thread workgroup reads the ~50 k candidate keys of a frame twice at the bandwidth of one CU and then sorts 2048 keys
// on one CU: 34 us for a kernel that occupies 8 of 256 CUs.  Spread over KP_WGS workgroups per frame:
//   k_kp_hist    every workgroup histograms its share of the keys in LDS (the same 2048 bins) and writes its partial histogram;
//   k_kp_gather  every workgroup adds the partial histograms up, finds the cut bin (all of them get the same one), and appends
//                the keys of its share from that bin upwards to the frame's selection list (one global atomic per workgroup);
//   k_kp_emit    every workgroup stages the ~1 100 selected keys in LDS and RANKS its share of them (keys are unique: rank =
//                number of smaller keys = position in the stable argsort of SphericalRing.py:194); rank decides the output row.
// Scratch: the tail of the candidate buffer (candidates only come from rows 8..55: entries past 48 x 1792 are never written).
// A frame the bins cannot split (more than SEL_N keys from the cut bin upwards -- pathological ties) is done by workgroup 0 of
// k_kp_emit alone, with the single-workgroup code below; M <= 1025 needs no threshold at all.
#define KP_WGS 16
#define KP_SCRATCH_OFF (48 * CAELO_NET_W)                                    // u64 entries into `cand`
#define KP_SEL_OFF (KP_SCRATCH_OFF + KP_WGS * CAELO_KP_HIST_BINS / 2)        // partial histograms: KP_WGS x 2048 u32
static_assert(KP_SEL_OFF + SEL_N <= CAELO_NET_H * CAELO_NET_W, "selection scratch must fit behind the candidates");
static_assert(CAELO_KP_HIST_BINS == 2 * SEL_THREADS, "a thread owns two bins");
#define KP_CC_NSEL 1       // cand_count[1]: keys on the selection list
#define KP_CC_STATE 2      // cand_count[2]: 0 = not decided, else
#define KP_STATE_LISTED 1  //   the selection list holds every key from the cut bin upwards
#define KP_STATE_FALLBACK 2

__global__ void __launch_bounds__(SEL_THREADS) k_kp_hist(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const int M = *F.cand_count, tid = threadIdx.x;
    if (M <= 1025) return;
    __shared__ unsigned int lh[CAELO_KP_HIST_BINS];
    lh[2 * tid] = 0u;
    lh[2 * tid + 1] = 0u;
    __syncthreads();
    const int per = (M + KP_WGS - 1) / KP_WGS, i0 = blockIdx.x * per, i1 = min(M, i0 + per);
    for (int i = i0 + tid; i < i1; i += SEL_THREADS) atomicAdd(&lh[kp_bin((unsigned int)(F.cand[i] >> 32))], 1u);
    __syncthreads();
    uint2 *out = (uint2 *)(F.cand + KP_SCRATCH_OFF) + (size_t)blockIdx.x * SEL_THREADS;
    out[tid] = make_uint2(lh[2 * tid], lh[2 * tid + 1]);
}

// inclusive suffix sum over the 1024 threads of the workgroup (wave shuffles + one LDS hop)
__device__ inline unsigned int kp_suffix_sum(unsigned int v, unsigned int *s_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int up = __shfl_down(v, o);
        if (lane + o < 64) v += up;
    }
    if (lane == 0) s_wave[wave] = v;  // the wave's total
    __syncthreads();
    unsigned int above = 0u;
    for (int w = wave + 1; w < SEL_THREADS / 64; ++w) above += s_wave[w];
    return v + above;
}

__global__ void __launch_bounds__(SEL_THREADS) k_kp_gather(const caelo_frame_set fs) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const int M = *F.cand_count, tid = threadIdx.x;
    if (M <= 1025) return;
    __shared__ unsigned int s_wave[SEL_THREADS / 64];
    __shared__ int s_cut[2], s_n, s_base;
    __shared__ unsigned long long s_keys[SEL_N];
    const uint2 *ph = (const uint2 *)(F.cand + KP_SCRATCH_OFF);
    uint2 hb = make_uint2(0u, 0u);
#pragma unroll 8
    for (int w = 0; w < KP_WGS; ++w) {
        const uint2 t = ph[(size_t)w * SEL_THREADS + tid];
        hb.x += t.x;
        hb.y += t.y;
    }
    if (tid == 0) s_n = 0;
    const unsigned int incl = kp_suffix_sum(hb.x + hb.y, s_wave);  // keys in bins >= 2 tid
    const unsigned int above = incl - hb.x - hb.y;                 // keys in bins > 2 tid + 1
    const unsigned int keep = 1025u;
    if (above < keep && incl >= keep) {  // exactly one thread: the cut is bin 2t+1 if that alone reaches `keep`, else bin 2t
        if (above + hb.y >= keep) { s_cut[0] = 2 * tid + 1; s_cut[1] = (int)(above + hb.y); }
        else { s_cut[0] = 2 * tid; s_cut[1] = (int)incl; }
    }
    __syncthreads();
    const int cutbin = s_cut[0], nsel = s_cut[1];
    if (nsel > SEL_N) {
        if (blockIdx.x == 0 && tid == 0) F.cand_count[KP_CC_STATE] = KP_STATE_FALLBACK;
        return;
    }
    const unsigned long long thresh = cutbin == 0 ? 0ull : ((unsigned long long)(KP_BIN_BASE + cutbin) << 48);
    const int per = (M + KP_WGS - 1) / KP_WGS, i0 = blockIdx.x * per, i1 = min(M, i0 + per);
    for (int i = i0 + tid; i < i1; i += SEL_THREADS) {
        const unsigned long long k = F.cand[i];
        if (k >= thresh) s_keys[atomicAdd(&s_n, 1)] = k;  // (at most nsel <= SEL_N keys in the whole frame)
    }
    __syncthreads();
    const int n = s_n;
    if (tid == 0) {
        s_base = n ? atomicAdd(&F.cand_count[KP_CC_NSEL], n) : 0;
        if (blockIdx.x == 0) F.cand_count[KP_CC_STATE] = KP_STATE_LISTED;
    }
    __syncthreads();
    unsigned long long *sel = F.cand + KP_SEL_OFF;
    for (int i = tid; i < n; i += SEL_THREADS) sel[s_base + i] = s_keys[i];
}

// The single-workgroup form: everything in one workgroup of 1024 threads (histogram, cut bin -- or an 8-bit radix select over
// the full keys when the bins cannot split the candidates --, gather, bitonic sort, output).  k_kp_emit's workgroup 0 calls it
// for the frames the three kernels above leave over; round 2 ran it alone as k_kp_select (34 us per 8 frames).
__device__ void kp_select_one_workgroup(const caelo_frame_dev &F, int ring_w, int ring_c) {
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

__global__ void __launch_bounds__(SEL_THREADS) k_kp_emit(const caelo_frame_set fs, int ring_w, int ring_c) {
    const caelo_frame_dev &F = fs.f[blockIdx.z];
    const int M = *F.cand_count, tid = threadIdx.x;
    const int state = F.cand_count[KP_CC_STATE];
    if (M > 1025 && state != KP_STATE_LISTED) {  // (workgroup-uniform) the bins could not split this frame's candidates
        if (blockIdx.x == 0) kp_select_one_workgroup(F, ring_w, ring_c);
        return;
    }
    // M <= 1025: every candidate is selected, the candidate list is the selection list
    const unsigned long long *sel = M > 1025 ? F.cand + KP_SEL_OFF : F.cand;
    const int nsel = M > 1025 ? F.cand_count[KP_CC_NSEL] : M;
    const int keep = M < 1025 ? M : 1025;
    const int K = keep > 0 ? keep - 1 : 0;  // drop the single best (:216,:218)
    const int first = nsel - keep;           // ranks first .. nsel - 2 are rows 0 .. K - 1
    __shared__ unsigned long long s_keys[SEL_N];
    for (int i = tid; i < nsel; i += SEL_THREADS) s_keys[i] = sel[i];
    __syncthreads();
    // thread (k = tid >> 4, part = tid & 15): key number blockIdx.x * 64 + k against the keys q = part (mod 16)
    const int e = blockIdx.x * 64 + (tid >> 4), part = tid & 15;
    const unsigned long long mine = e < nsel ? s_keys[e] : 0ull;
    int rank = 0;
    if (e < nsel)
        for (int q = part; q < nsel; q += 16) rank += s_keys[q] < mine ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) rank += __shfl_xor(rank, o);
    const int i = rank - first;
    if (e < nsel && part == 0 && i >= 0 && i < K) {
        const unsigned idx = (unsigned)(mine & 0xFFFFFFFFull);
        const int y = idx / CAELO_NET_W, x = idx % CAELO_NET_W;
        F.key_pixels[2 * i] = y;
        F.key_pixels[2 * i + 1] = x;
        const float *px = F.ring + ((int64_t)y * ring_w + x) * ring_c;
        F.key_pts[(size_t)F.kp_ld * i] = px[0];
        F.key_pts[(size_t)F.kp_ld * i + 1] = px[1];
        F.key_pts[(size_t)F.kp_ld * i + 2] = px[2];
    }
    if (blockIdx.x == 0) {
        if (F.valid)
            for (int r = tid; r < CAELO_MAX_KEYPTS; r += SEL_THREADS) F.valid[(size_t)F.valid_ld * r] = r < K ? 1.0f : 0.0f;
        if (tid == 0) {
            *F.n_key = K;
            if (K <= 50) atomicOr(F.status, CAELO_ST_FEW_KEYPTS);  // :286
        }
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
    k_kp_hist<<<dim3(KP_WGS, 1, fs.n), SEL_THREADS, 0, s>>>(fs);
    CAELO_LAUNCH_CHECK();
    k_kp_gather<<<dim3(KP_WGS, 1, fs.n), SEL_THREADS, 0, s>>>(fs);
    CAELO_LAUNCH_CHECK();
    k_kp_emit<<<dim3(SEL_N / 64, 1, fs.n), SEL_THREADS, 0, s>>>(fs, ring_w, ring_c);   // (falls back to the one-workgroup select on > 2048 tied keys)
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
