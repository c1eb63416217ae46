// config5.hip -- BASELINE.json configs[4]: "synthetic 128-beam dense scan, 2x voxel-patch resolution".
//
// NOT a reference code path.  The reference has PatchSize = 16 only (Voxel.py:31-33); this stress case keeps
// GetPatchesList's rule (Voxel.py:177-216) and the encoder's layer stack (EncoderModel4VoxelPatch.h5) at twice the
// patch resolution, as SURVEY.md section 8d defines it:
//   * window [-16,16)^3 around the key voxel, wrap-around placement d mod 32 (:213-214);
//   * the 496-nearest cap (:182,:195-196) DISABLED: a 32^3 window holds up to 32768 voxels, n_neighbors=496 would
//     truncate nearly every dense patch and its tie order is sklearn-defined -- every occupied voxel in the
//     window is set;
//   * Conv3D(1->8) tanh, MaxPool2 (16^3), Conv3D(8->16) tanh, MaxPool2 (8^3), Conv3D(16->32) tanh, Flatten
//     (16384), Dense(200) tanh, Dense(20) tanh; conv kernels, biases and dense_2 from the .h5, dense_1 a caller
//     supplied [16384][200] matrix (no trained weights exist at this size).
// Parity is against oracle/caelo_oracle.c (orc_patches32, orc_encode32) only.
//
// Data layout: a patch is 512 u64 words, voxel (ix,iy,iz) at bit (lin & 63) of word (lin >> 6),
// lin = (ix*32 + iy)*32 + iz; activations are channels-last f32 ([x][y][z][c]) like the 16^3 path.
//
// Kernels: the three conv stages are plain LDS-tiled VALU kernels (one x slab of one patch per workgroup) -- this
// configuration is a parity-test case, not a bench line, and only Dense(200) reuses the tuned MFMA kernel
// (k_enc_dense1<16384>, encoder.hip).  See DESIGN.md section 4.5 for measured times.
#include <math.h>

#include "caelo_internal.h"

#define VOX_SIZE 0.02
#define VIS_L 99.84
#define VIS_W 99.84
#define VIS_H 14.72

__device__ inline float c5_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // exp(2x), same form as enc_tanh (encoder.hip)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ---- patches: one workgroup per (key point, scale); the 5x5x5 bricks the window can touch staged in LDS --------
__global__ void __launch_bounds__(256) k5_patches(const float *__restrict__ pts, int pts_ld, int64_t k_max,
                                                  const int32_t *__restrict__ n_key, caelo_brick_table t0,
                                                  caelo_brick_table t1, caelo_brick_table t2,
                                                  unsigned long long *__restrict__ bits) {
    __shared__ int slot[125];
    __shared__ unsigned long long win[125 * 8];
    const int tid = threadIdx.x;
    const int64_t pw = blockIdx.x;
    const int64_t kp = pw / 3;
    const int scale = (int)(pw % 3);
    uint32_t *out = (uint32_t *)(bits + pw * 512);
    const int K = n_key ? *n_key : (int)k_max;
    if (kp >= K) {
        for (int i = tid; i < 1024; i += 256) out[i] = 0u;
        return;
    }
    const caelo_brick_table tab = scale == 0 ? t0 : (scale == 1 ? t1 : t2);
    const double vs = scale == 0 ? VOX_SIZE : (scale == 1 ? VOX_SIZE * 8 : VOX_SIZE * 32);  // Voxel.py:31
    // Voxel.py:185,:193  KeyVoxels = int32((Pts + Visible*) / VoxelSizes[s])  (f64)
    const int kx = (int)(((double)pts[(size_t)pts_ld * kp] + VIS_L) / vs);
    const int ky = (int)(((double)pts[(size_t)pts_ld * kp + 1] + VIS_W) / vs);
    const int kz = (int)(((double)pts[(size_t)pts_ld * kp + 2] + VIS_H) / vs);
    const int bx0 = (kx - 16) >> 3, by0 = (ky - 16) >> 3, bz0 = (kz - 16) >> 3;
    if (tid < 125) {
        const int bx = bx0 + tid / 25, by = by0 + (tid / 5) % 5, bz = bz0 + tid % 5;
        slot[tid] = (bx >= 0 && by >= 0 && bz >= 0) ? caelo_brick_find(tab, caelo_pack3(bx, by, bz)) : -1;
    }
    __syncthreads();
    for (int i = tid; i < 1000; i += 256) {
        const int sl = slot[i >> 3];
        win[i] = sl >= 0 ? tab.bits[(size_t)sl * 8 + (i & 7)] : 0ull;
    }
    __syncthreads();
    const int zsh = (kz - 16) & 7;
    for (int r = tid; r < 1024; r += 256) {  // one (ix, iy) row of 32 z bits per iteration
        const int ix = r >> 5, iy = r & 31;
        const int x = kx + (ix < 16 ? ix : ix - 32), y = ky + (iy < 16 ? iy : iy - 32);  // :213-214 wrap-around
        const int bxl = (x >> 3) - bx0, byl = (y >> 3) - by0;
        const int ysh = (y & 7) << 3;
        unsigned long long strip = 0ull;  // 40 z bits starting at the first z brick
#pragma unroll
        for (int j = 0; j < 5; ++j)
            strip |= ((win[((bxl * 5 + byl) * 5 + j) * 8 + (x & 7)] >> ysh) & 0xFFull) << (8 * j);
        const uint32_t s32 = (uint32_t)(strip >> zsh);  // bit d <-> dz = d - 16
        out[r] = (s32 >> 16) | (s32 << 16);             // iz = dz mod 32
    }
}

// ---- conv1 + pool1 from bits: workgroup = (patch, pooled x plane), thread = pooled (y, z) cell, 8 channels ------
__global__ void __launch_bounds__(256) k5_conv1pool(const unsigned long long *__restrict__ bits, const float *__restrict__ w1,
                                                    const float *__restrict__ b1, float *__restrict__ p1) {
    __shared__ float in[4][34][34];
    __shared__ float w[27 * 8 + 8];
    const int tid = threadIdx.x;
    const int64_t patch = blockIdx.x >> 4;
    const int px = blockIdx.x & 15;
    const uint32_t *src = (const uint32_t *)(bits + patch * 512);
    for (int i = tid; i < 224; i += 256) w[i] = i < 216 ? w1[i] : b1[i - 216];
    for (int i = tid; i < 4 * 34 * 34; i += 256) {
        const int pl = i / (34 * 34), rem = i % (34 * 34);
        const int x = 2 * px - 1 + pl, y = rem / 34 - 1, z = rem % 34 - 1;
        float v = 0.0f;
        if (x >= 0 && x < 32 && y >= 0 && y < 32 && z >= 0 && z < 32) v = (float)((src[x * 32 + y] >> z) & 1u);
        in[pl][rem / 34][rem % 34] = v;
    }
    __syncthreads();
    const int py = tid >> 4, pz = tid & 15;
    float m[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) m[o] = -INFINITY;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int c = 0; c < 2; ++c) {
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = w[216 + o];
        for (int ka = 0; ka < 3; ++ka) for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                const float v = in[a + ka][2 * py + b + kb][2 * pz + c + kc];
                const float *wt = &w[((ka * 3 + kb) * 3 + kc) * 8];
#pragma unroll
                for (int o = 0; o < 8; ++o) acc[o] += v * wt[o];
            }
#pragma unroll
        for (int o = 0; o < 8; ++o) m[o] = fmaxf(m[o], acc[o]);
    }
    float4 *dst = (float4 *)(p1 + ((((size_t)patch * 16 + px) * 16 + py) * 16 + pz) * 8);
    dst[0] = make_float4(c5_tanh(m[0]), c5_tanh(m[1]), c5_tanh(m[2]), c5_tanh(m[3]));  // tanh is monotone: tanh(max) = max(tanh)
    dst[1] = make_float4(c5_tanh(m[4]), c5_tanh(m[5]), c5_tanh(m[6]), c5_tanh(m[7]));
}

// ---- conv (+ optional 2x pool) with an 8 x 8 (y, z) output plane: workgroup = (patch, output x plane), -------------
// thread = (cell, quarter of the output channels); the 3 (4 with pooling) input x planes it reads sit in LDS
// with a zero halo.  in [n][D][D][D][CIN], w [27][CIN][COUT] (Keras order), out [n][8][8][8][COUT].
template <int D, int CIN, int COUT, bool POOL>
__global__ void __launch_bounds__(256) k5_conv(const float *__restrict__ in, const float *__restrict__ w,
                                               const float *__restrict__ b, float *__restrict__ out) {
    static_assert((POOL ? D / 2 : D) == 8 && CIN % 4 == 0 && COUT % 16 == 0, "8x8 output plane, float4 channel groups");
    constexpr int PL = POOL ? 4 : 3, DP = D + 2, CPT = COUT / 4;
    __shared__ __attribute__((aligned(16))) float tile[PL * DP * DP * CIN];
    const int tid = threadIdx.x;
    const int64_t patch = blockIdx.x >> 3;
    const int xo = blockIdx.x & 7;
    const int x_first = (POOL ? 2 * xo : xo) - 1;
    const float *src = in + (size_t)patch * D * D * D * CIN;
    for (int i = tid; i < PL * DP * DP * CIN / 4; i += 256) {
        const int c4 = i % (CIN / 4), cell = i / (CIN / 4);
        const int pl = cell / (DP * DP), y = (cell / DP) % DP - 1, z = cell % DP - 1, x = x_first + pl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D) v = ((const float4 *)(src + (((size_t)x * D + y) * D + z) * CIN))[c4];
        ((float4 *)tile)[i] = v;
    }
    __syncthreads();
    const int cy = tid >> 5, cz = (tid >> 2) & 7, cg = tid & 3;
    const float *wq = w + cg * CPT;
    float m[CPT];
#pragma unroll
    for (int o = 0; o < CPT; ++o) m[o] = -INFINITY;
    for (int sub = 0; sub < (POOL ? 8 : 1); ++sub) {
        const int a = POOL ? sub >> 2 : 0, y = POOL ? 2 * cy + ((sub >> 1) & 1) : cy, z = POOL ? 2 * cz + (sub & 1) : cz;
        float acc[CPT];
#pragma unroll
        for (int o = 0; o < CPT; ++o) acc[o] = b[cg * CPT + o];
        for (int ka = 0; ka < 3; ++ka) for (int kb = 0; kb < 3; ++kb) for (int kc = 0; kc < 3; ++kc) {
            const float *ti = &tile[(((a + ka) * DP + (y + kb)) * DP + (z + kc)) * CIN];
            const float *wt = wq + (size_t)((ka * 3 + kb) * 3 + kc) * CIN * COUT;
#pragma unroll
            for (int c4 = 0; c4 < CIN / 4; ++c4) {
                const float4 v = *(const float4 *)(ti + 4 * c4);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int o4 = 0; o4 < CPT / 4; ++o4) {
                        const float4 wv = *(const float4 *)(wt + (size_t)(4 * c4 + ci) * COUT + 4 * o4);
                        acc[4 * o4 + 0] += vv[ci] * wv.x;
                        acc[4 * o4 + 1] += vv[ci] * wv.y;
                        acc[4 * o4 + 2] += vv[ci] * wv.z;
                        acc[4 * o4 + 3] += vv[ci] * wv.w;
                    }
            }
        }
#pragma unroll
        for (int o = 0; o < CPT; ++o) m[o] = fmaxf(m[o], acc[o]);
    }
    float *dst = out + ((((size_t)patch * 8 + xo) * 8 + cy) * 8 + cz) * COUT + cg * CPT;
#pragma unroll
    for (int o4 = 0; o4 < CPT / 4; ++o4)
        ((float4 *)dst)[o4] = make_float4(c5_tanh(m[4 * o4]), c5_tanh(m[4 * o4 + 1]), c5_tanh(m[4 * o4 + 2]), c5_tanh(m[4 * o4 + 3]));
}

// ---- C ABI -------------------------------------------------------------------------------------------------------
CAELO_API int caelo_patches32(caelo_ctx *c, const caelo_voxmap *m, const float *pts, int pts_ld, int64_t k_max,
                              const int32_t *n_key, uint64_t *bits, void *stream) {
    CAELO_REQUIRE(c && m && pts && bits, "null argument");
    CAELO_REQUIRE(k_max > 0 && pts_ld >= 3 && k_max * 3 < (1ll << 31), "bad shape");
    k5_patches<<<(unsigned)(k_max * 3), 256, 0, caelo_stream(stream)>>>(pts, pts_ld, k_max, n_key, m->brick[0], m->brick[1],
                                                                       m->brick[2], (unsigned long long *)bits);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_set_encoder32_dense(caelo_ctx *c, const float *wd1, const float *bd1) {
    CAELO_REQUIRE(c && wd1 && bd1, "null argument");
    const size_t K = 16384, N = 200, NP = 208;
    float *pad = (float *)calloc(K * NP + NP, sizeof(float));
    if (!pad) { caelo_set_error("out of host memory"); return CAELO_ERR_ARG; }
    for (size_t k = 0; k < K; ++k) memcpy(pad + k * NP, wd1 + k * N, N * sizeof(float));
    memcpy(pad + K * NP, bd1, N * sizeof(float));
    if (!c->enc32_wd1) CAELO_HIP(hipMalloc(&c->enc32_wd1, (K * NP + NP) * sizeof(float)));
    CAELO_HIP(hipMemcpy(c->enc32_wd1, pad, (K * NP + NP) * sizeof(float), hipMemcpyHostToDevice));
    free(pad);
    return CAELO_OK;
}

static inline int64_t c5_pad(int64_t n) { return (n + 47) / 48 * 48; }  // whole dense-1 row tiles (D1_BM, encoder.hip)

// ws = P1 [n][16^3][8] | P2 [n][8^3][16] | F3 [np][16384] | dense-1 split-K partial sums (sized by encoder.hip)
CAELO_API int64_t caelo_encode32_ws_bytes(int64_t n_patches) {
    if (n_patches <= 0) return 0;
    const int64_t np = c5_pad(n_patches);
    return (n_patches * (32768 + 8192) + np * 16384) * (int64_t)sizeof(float) + enc_dense32_part_bytes(np);
}

CAELO_API int caelo_encode32(caelo_ctx *c, const uint64_t *bits, int64_t n_patches, int group, float *out, int out_stride,
                             void *ws, void *stream) {
    CAELO_REQUIRE(c && bits && out && ws, "null argument");
    CAELO_REQUIRE(c->has_enc, "encoder weights not set (caelo_set_encoder_weights)");
    CAELO_REQUIRE(c->enc32_wd1, "32^3 dense_1 not set (caelo_set_encoder32_dense)");
    CAELO_REQUIRE(n_patches > 0 && n_patches < (1ll << 27) && group >= 1 && out_stride >= group * 20, "bad shape");
    hipStream_t s = caelo_stream(stream);
    const int64_t np = c5_pad(n_patches);
    float *p1 = (float *)ws;
    float *p2 = p1 + n_patches * 32768;
    float *f3 = p2 + n_patches * 8192;
    float *part = f3 + np * 16384;
    if (np > n_patches) CAELO_HIP(hipMemsetAsync(f3 + n_patches * 16384, 0, (size_t)(np - n_patches) * 16384 * sizeof(float), s));
    k5_conv1pool<<<(unsigned)(n_patches * 16), 256, 0, s>>>((const unsigned long long *)bits, c->enc_w1, c->enc_b1, p1);
    CAELO_LAUNCH_CHECK();
    k5_conv<16, 8, 16, true><<<(unsigned)(n_patches * 8), 256, 0, s>>>(p1, c->enc_w2, c->enc_b2, p2);
    CAELO_LAUNCH_CHECK();
    k5_conv<8, 16, 32, false><<<(unsigned)(n_patches * 8), 256, 0, s>>>(p2, c->enc_w3, c->enc_b3, f3);
    CAELO_LAUNCH_CHECK();
    return enc_dense32_head_launch(c, f3, n_patches, np, part, group, out, out_stride, s);
}
